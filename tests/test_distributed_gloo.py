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

from conftest import REPO, free_port, pkg


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
    port = free_port()
    mp.start_processes(_worker, args=(2, port, num_rays, path), nprocs=2, join=True, start_method="spawn")


@pytest.mark.parametrize("size", [4, 5])
def test_bench_main_self_launch_over_gloo(size):
    """`python bench.py --gpus 2` exactly as the driver invokes it -- no launcher, WORLD_SIZE unset -- rehearsed on CPU: main() starts two
    fresh ranks itself (before any GPU call), they rendezvous over gloo, run the timed-step control flow (barriers, MAX over ranks) around
    the product's sharding function with the oracle as renderer (tests/gloo_bench_renderer.py, which also asserts that each rank rendered
    only its block and that the gathered frame equals the single-process render bit for bit: 16 rays even, 25 rays ragged), and rank 0's
    ONE JSON line comes back through the parent, whose exit status is the children's."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SAHS_BENCH_BACKEND="gloo", SAHS_BENCH_RENDERER=os.path.join(REPO, "tests", "gloo_bench_renderer.py") + ":make", OMP_NUM_THREADS="2")
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--size", str(size)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    # ONE line of ours on stdout, from rank 0 (the gloo library itself prints "[Gloo] Rank ..." connection notes there; RCCL does not)
    lines = [l for l in p.stdout.splitlines() if l.strip() and "[Gloo]" not in l and "peer ranks" not in l]
    assert len(lines) == 1 and lines[0].startswith("{"), p.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 == line["rccl_ranks"] and line["collective_backend"] == "gloo" and "rehearsal" in line
    assert line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "strong" and line["vs_baseline"] is None
    assert abs(line["value"] - size * size * 2 / (line["ms_per_step"] * 2e-3)) < 1e-6 * line["value"]
    # a failing rank fails the run: the parent relays the children's status
    env["SAHS_BENCH_RENDERER"] = os.path.join(REPO, "tests", "gloo_bench_renderer.py") + ":does_not_exist"
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--size", "4"],
                       env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in p.stdout.splitlines() if l.lstrip().startswith("{")]
    assert p.returncode != 0 and len(lines) == 1 and "error" in json.loads(lines[0]) and json.loads(lines[0])["n_gpus"] == 2, p.stdout


def test_bench_main_wall_clock_guard_ends_a_hung_run():
    """A run whose ranks never finish (here: a renderer that sleeps; on hardware: a wedged rendezvous or RCCL collective) is ended by the
    parent's wall-clock guard: the children's process group is terminated, ONE JSON line {"error", "n_gpus"} is printed, the exit status is
    non-zero -- instead of being killed at the driver's limit with nothing written."""
    import json
    import subprocess
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SAHS_BENCH_BACKEND="gloo", SAHS_BENCH_RENDERER=os.path.join(REPO, "tests", "gloo_bench_renderer.py") + ":make_sleepy",
               OMP_NUM_THREADS="2", SAHS_BENCH_GUARD_S="25")
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--size", "4"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 124 and time.time() - t0 < 120, (p.returncode, p.stderr[-2000:])
    lines = [l for l in p.stdout.splitlines() if l.lstrip().startswith("{")]
    assert len(lines) == 1, p.stdout
    err = json.loads(lines[0])
    assert err["n_gpus"] == 2 and "guard" in err["error"]


def test_bench_refuses_a_mismatched_launcher():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "--nproc-per-node must equal --gpus" in p.stderr


def _inplace_worker(rank, world, port, num_rays):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import importlib
        D = importlib.import_module("sahs-deformable-nerf_amd.distributed")
        calls = []

        def block(lo, hi, rows):      # a rank writes ONLY its rows; everything else arrives by the all-gather
            calls.append((lo, hi))
            assert rows.shape == (hi - lo, 36) and rows.data_ptr() != 0
            rows.copy_(torch.arange(lo, hi, dtype=torch.float32)[:, None] * 100 + torch.arange(36, dtype=torch.float32)[None, :])

        full = D.render_rows_sharded(block, num_rays, torch.device("cpu"))
        want = torch.arange(num_rays, dtype=torch.float32)[:, None] * 100 + torch.arange(36, dtype=torch.float32)[None, :]
        lo, hi = D.shard_bounds(num_rays, world, rank)
        assert calls == ([(lo, hi)] if hi > lo else []) and torch.equal(full, want), rank
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("num_rays,world", [(24, 2), (23, 2), (36, 3), (2, 3)])
def test_render_rows_sharded_in_place(num_rays, world):
    """The product's sharding function: even splits all-gather IN PLACE (send buffer = the rank's slice of the receive buffer), ragged
    ones through the padded gather; also a frame with fewer rays than ranks (an empty block)."""
    port = free_port()
    mp.start_processes(_inplace_worker, args=(world, port, num_rays), nprocs=world, join=True, start_method="spawn")


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
    port = free_port()
    mp.start_processes(_grad_worker, args=(2, port), nprocs=2, join=True, start_method="spawn")
