"""N > 1 path on CPU: two gloo ranks each render their contiguous ray block (with the CPU oracle standing in
for the HIP renderer -- tests may use it) and all-gather the 36-float rows; the assembled frame must equal
the single-process render bit for bit, for even and ragged splits."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO, pkg


def _worker(rank, world, port, num_rays, path):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import importlib
        D = importlib.import_module("sahs-deformable-nerf_amd.distributed")
        data = np.load(path)
        from oracle import oracle

        def render(lo, hi):   # draws keyed by GLOBAL ray index (sahs_ray_uniforms restated): no random tensor is shipped or sliced
            o = oracle.render_rays(data["flat"], data["rays"][lo:hi], 8, 8, data["drv"], data["p36"], bg=data["bg"][lo:hi],
                                   t_rand=oracle.ray_uniforms(42, 0, lo, hi - lo, 8), u=oracle.ray_uniforms(42, 1, lo, hi - lo, 8))
            return tuple(torch.from_numpy(o[k]) for k in ["rgb_c", "disp_c", "acc_c", "rgb_f", "disp_f", "acc_f", "w_bg", "depth_f"])

        full = D.render_sharded(render, num_rays)
        ref = np.load(path.replace(".npz", "_ref.npy"))
        got = D.pack_outputs(full).numpy()
        assert got.shape == ref.shape and np.array_equal(got, ref), "rank %d: gathered frame differs" % rank
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("num_rays", [24, 23])
def test_two_rank_ray_sharding(tmp_path, flat_weights, num_rays):
    from oracle import oracle
    D = pkg("distributed")
    assert [D.shard_bounds(23, 2, r) for r in range(2)] == [(0, 11), (11, 23)]
    rng = np.random.default_rng(num_rays)
    flat = flat_weights(density_bias=8.0, density_gain=30.0)
    rays = np.zeros((num_rays, 8), np.float32)
    rays[:, 0:3] = [0, 0, 0.8]
    rays[:, 3:6] = rng.normal(0, 0.15, (num_rays, 3)) + np.array([0, 0, -1.0])
    rays[:, 6], rays[:, 7] = 0.48, 1.08
    audio = rng.standard_normal((16, 29)).astype(np.float32)
    pose = np.concatenate([np.eye(3), [[0.0], [0.0], [0.8]]], axis=1).astype(np.float32)
    d = dict(flat=flat, rays=rays, drv=oracle.audionet(flat, audio), p36=oracle.pose_encoding(pose),
             bg=rng.uniform(0, 1, (num_rays, 15)).astype(np.float32))
    path = str(tmp_path / "shard.npz")
    np.savez(path, **d)
    o = oracle.render_rays(flat, rays, 8, 8, d["drv"], d["p36"], bg=d["bg"], t_rand=oracle.ray_uniforms(42, 0, 0, num_rays, 8),
                           u=oracle.ray_uniforms(42, 1, 0, num_rays, 8))
    ref = D.pack_outputs(tuple(torch.from_numpy(o[k]) for k in ["rgb_c", "disp_c", "acc_c", "rgb_f", "disp_f", "acc_f", "w_bg", "depth_f"]))
    assert ref.shape == (num_rays, D.OUT_COLUMNS)
    np.save(path.replace(".npz", "_ref.npy"), ref.numpy())
    port = 29500 + (os.getpid() % 2000)
    mp.start_processes(_worker, args=(2, port, num_rays, path), nprocs=2, join=True, start_method="spawn")


def _bench_worker(rank, world, port, num_rays, path):
    """bench.py's own control flow (frame_step: shard bounds -> render -> all-gather of (n,36) rows; timed_steps: barrier-bracketed
    timing with the MAX over ranks; run_headline: the JSON record) over gloo, the CPU oracle standing in for the HIP renderer."""
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import importlib
        import json
        bench = importlib.import_module("bench")
        D = importlib.import_module("sahs-deformable-nerf_amd.distributed")
        data = np.load(path)
        from oracle import oracle

        class OracleRenderer:
            calls = []

            def render(self, lo, hi):   # draws keyed by GLOBAL ray index, as HipRenderer.render's ops.ray_uniforms(seed, stream, lo + s, ...)
                self.calls.append((lo, hi))
                o = oracle.render_rays(data["flat"], data["rays"][lo:hi], 8, 8, data["drv"], data["p36"], bg=data["bg"][lo:hi],
                                       t_rand=oracle.ray_uniforms(42, 0, lo, hi - lo, 8), u=oracle.ray_uniforms(42, 1, lo, hi - lo, 8))
                return D.pack_outputs(tuple(torch.from_numpy(o[k]) for k in ["rgb_c", "disp_c", "acc_c", "rgb_f", "disp_f", "acc_f", "w_bg", "depth_f"]))

        r = OracleRenderer()
        rec, out = bench.run_headline(r, num_rays, world, rank, steps=2, warmup=1, dist=dist, gather=D.all_gather_rows, sync=lambda: None,
                                      dtype="f32", config={"workload": "gloo rehearsal"}, roofline_fn=lambda dt: None)
        assert r.calls == [D.shard_bounds(num_rays, world, rank)] * 3          # 1 warmup + 2 timed steps, this rank's block only
        ref = np.load(path.replace(".npz", "_ref.npy"))
        assert np.array_equal(out.numpy(), ref), "rank %d: gathered frame differs from the single-process render" % rank
        line = json.loads(json.dumps(rec))
        assert line["n_gpus"] == world == line["rccl_ranks"] == dist.get_world_size()
        assert line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "strong" and line["vs_baseline"] is None
        assert abs(line["value"] - num_rays * 2 / (line["ms_per_step"] * 2e-3)) < 1e-6 * line["value"]
        # the reported time is the MAX over ranks: every rank holds the same number
        t = torch.tensor([line["ms_per_step"]], dtype=torch.float64)
        both = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(both, t)
        assert float(both[0]) == float(both[1])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("num_rays", [24, 23])
def test_bench_control_flow_over_gloo(tmp_path, flat_weights, num_rays):
    from oracle import oracle
    D = pkg("distributed")
    rng = np.random.default_rng(100 + num_rays)
    flat = flat_weights(density_bias=8.0, density_gain=30.0)
    rays = np.zeros((num_rays, 8), np.float32)
    rays[:, 0:3] = [0, 0, 0.8]
    rays[:, 3:6] = rng.normal(0, 0.15, (num_rays, 3)) + np.array([0, 0, -1.0])
    rays[:, 6], rays[:, 7] = 0.48, 1.08
    audio = rng.standard_normal((16, 29)).astype(np.float32)
    pose = np.concatenate([np.eye(3), [[0.0], [0.0], [0.8]]], axis=1).astype(np.float32)
    d = dict(flat=flat, rays=rays, drv=oracle.audionet(flat, audio), p36=oracle.pose_encoding(pose),
             bg=rng.uniform(0, 1, (num_rays, 15)).astype(np.float32))
    path = str(tmp_path / "bench.npz")
    np.savez(path, **d)
    o = oracle.render_rays(flat, rays, 8, 8, d["drv"], d["p36"], bg=d["bg"], t_rand=oracle.ray_uniforms(42, 0, 0, num_rays, 8),
                           u=oracle.ray_uniforms(42, 1, 0, num_rays, 8))
    ref = D.pack_outputs(tuple(torch.from_numpy(o[k]) for k in ["rgb_c", "disp_c", "acc_c", "rgb_f", "disp_f", "acc_f", "w_bg", "depth_f"]))
    np.save(path.replace(".npz", "_ref.npy"), ref.numpy())
    port = 33500 + (os.getpid() % 2000)
    mp.start_processes(_bench_worker, args=(2, port, num_rays, path), nprocs=2, join=True, start_method="spawn")


def _grad_worker(rank, world, port):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import importlib
        D = importlib.import_module("sahs-deformable-nerf_amd.distributed")
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
        frozen = torch.nn.Parameter(torch.zeros(4), requires_grad=False)
        x = torch.arange(40, dtype=torch.float32).reshape(8, 5) / 10
        sl = D.shard_batch(8)
        assert (sl.start, sl.stop) == (rank * 4, rank * 4 + 4)
        net(x[sl]).square().mean().backward()
        net[1].bias.grad = None                       # a parameter without gradient on this rank counts as zeros
        D.all_reduce_gradients(list(net.parameters()) + [frozen])
        ref = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
        ref.load_state_dict(net.state_dict())
        ref(x).square().mean().backward()             # equal shards: mean of shard-mean gradients == full-batch gradient
        for (n, p), q in zip(net.named_parameters(), ref.parameters()):
            want = torch.zeros_like(q.grad) if n == "1.bias" else q.grad
            assert torch.allclose(p.grad, want, rtol=1e-5, atol=1e-6), n
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_all_reduce():
    port = 31500 + (os.getpid() % 2000)
    mp.start_processes(_grad_worker, args=(2, port), nprocs=2, join=True, start_method="spawn")
