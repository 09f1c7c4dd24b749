"""The product's ray-sharded frame loop on real kernels with TWO ranks: both processes drive the one GPU of the test box (HIP kernels for the
rendering, gloo for the collective -- RCCL refuses two ranks on one device; the 8-GPU run over RCCL is the driver's).  Each rank calls
evaluation.render_frames(shard=True) -> run_one_iter_of_nerf(_shard=...) -> distributed.render_rows_sharded: its block of the frame's rays
rendered in place, one in-place all-gather; the frames must equal the single-process frames bit for bit, for an even and a ragged split."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO, VARIANT_KW, free_port

pytestmark = pytest.mark.gpu


def _frames(sahs, size):
    rng = np.random.default_rng(size)
    pose = np.concatenate([np.eye(3), [[0.0], [0.0], [0.8]]], axis=1).astype(np.float32)
    return [dict(pose=pose, audio=rng.standard_normal((16, 29)).astype(np.float32), name="f_%d.png" % i) for i in range(2)]


def _render(sahs, W, size, shard, savedir=None):
    dev = torch.device("cuda:0")
    cfg = sahs.default_config()
    cfg.nerf.validation.chunksize = 64          # several chunks per rank, a ragged last one
    model = sahs.AudioFaceModel(cfg).to(dev).load_flat(W.flatten_state_dict(W.hash_state_dict(**VARIANT_KW["hdr"])))
    bg = torch.from_numpy(np.random.default_rng(1).uniform(0, 1, (size * size, 15)).astype(np.float32)).to(dev)
    intr = np.array([1200.0 * size / 512, 1200.0 * size / 512, 0.5, 0.5], np.float32)
    out = sahs.evaluation.render_frames(model, cfg, _frames(sahs, size), (size, size, intr), background=bg, savedir=savedir, shard=shard, log=lambda *a: None)
    return [torch.cat([o["rgb"].reshape(size * size, -1), o["disp"].reshape(-1, 1), o["depth"].reshape(-1, 1), o["w_bg"].reshape(-1, 1)], 1).cpu() for o in out]


def _worker(rank, world, port, size, path):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import importlib
    sahs = importlib.import_module("sahs-deformable-nerf_amd")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        got = _render(sahs, sahs.weights, size, True, savedir=os.path.join(path, "out"))
        ref = torch.load(os.path.join(path, "ref.pt"))
        for i, (a, b) in enumerate(zip(got, ref)):
            assert torch.equal(a, b), "rank %d frame %d: the sharded frame differs from the single-process frame (max %.3e)" % (
                rank, i, float((a - b).abs().max()))
        dist.barrier()
        if rank == 0:      # rank 0 alone writes the images
            assert sorted(os.listdir(os.path.join(path, "out"))) == ["f_0.png", "f_1.png", "masks", "normals"]
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("size", [16, 15])       # 256 rays: 128 + 128; 225 rays: 112 + 113
def test_two_ranks_render_one_frame_together(tmp_path, size):
    import importlib
    sahs = importlib.import_module("sahs-deformable-nerf_amd")
    ref = _render(sahs, sahs.weights, size, True)           # no process group: the single-process frame under the same keyed draws
    assert all(bool(torch.isfinite(r).all()) for r in ref) and not torch.equal(ref[0], ref[1])
    torch.save(ref, str(tmp_path / "ref.pt"))
    port = free_port()
    mp.start_processes(_worker, args=(2, port, size, str(tmp_path)), nprocs=2, join=True, start_method="spawn")


def _train_worker(rank, world, port, path):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import importlib
    sahs = importlib.import_module("sahs-deformable-nerf_amd")
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        W, Tr = sahs.weights, sahs.training
        cfg = sahs.default_config()
        cfg.nerf.train.num_random_rays = 256
        model = sahs.AudioFaceModel(cfg).to(dev).load_flat(W.flatten_state_dict(W.hash_state_dict(**VARIANT_KW["hdr"]))).train()
        opt = torch.optim.Adam(model.parameters(), lr=float(cfg.optimizer.lr))
        g = torch.Generator(device=dev).manual_seed(5)
        H = Wd = 32
        image = torch.rand(H, Wd, 3, device=dev, generator=g)
        mask = torch.zeros(H, Wd, 12, device=dev).scatter_(2, torch.randint(0, 12, (H, Wd, 1), device=dev, generator=g), 1.0)
        bg = torch.cat([torch.rand(H, Wd, 3, device=dev, generator=g), torch.ones(H, Wd, 1, device=dev), torch.zeros(H, Wd, 11, device=dev)], 2)
        audio = torch.randn(16, 29, device=dev, generator=g)
        pose = torch.from_numpy(np.concatenate([np.eye(3), [[0.0], [0.0], [0.8]]], 1).astype(np.float32)).to(dev)
        intr = np.array([1200.0 * H / 512, 1200.0 * H / 512, 0.5, 0.5], np.float32)
        prob = torch.ones(12, device=dev) / 12
        torch.manual_seed(100 + rank)          # the ranks' own noise streams differ (as they would on separate GPUs) ...
        out = None
        for step in range(2):                  # ... the batch draw does not: one generator state on every rank
            out = Tr.train_step(model, opt, cfg, step, image, mask, pose, intr, audio, bg, prob, generator=torch.Generator(device=dev).manual_seed(9 + step))
            prob = out["sample_prob"]
        flat = model.flat_params().cpu()
        both = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        assert torch.equal(both[0], both[1]), "the replicas diverged: max %.3e" % float((both[0] - both[1]).abs().max())
        start = torch.from_numpy(W.flatten_state_dict(W.hash_state_dict(**VARIANT_KW["hdr"])))
        assert bool(torch.isfinite(flat).all()) and float((flat - start).abs().max()) > 0.0 and np.isfinite(out["loss"])
        assert abs(float(prob.sum()) - 1.0) < 1e-5
        if rank == 0:
            torch.save(dict(loss=out["loss"], prob=prob.cpu()), os.path.join(path, "r0.pt"))
        dist.barrier()
        if rank == 1:
            r0 = torch.load(os.path.join(path, "r0.pt"))
            assert r0["loss"] == out["loss"] and torch.equal(r0["prob"], prob.cpu())       # the reduced statistics are the same number on both
    finally:
        dist.destroy_process_group()


def test_two_ranks_train_data_parallel(tmp_path):
    """training.train_step on two ranks (HIP forward/backward on each rank's slice of the 256-ray batch, gradients and the sampling
    feedback all-reduced over gloo): after two steps the replicas hold bit-identical parameters, loss and sample_prob."""
    port = free_port()
    mp.start_processes(_train_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")


def _rccl_worker(rank, world, port):
    """ONE rank over the real collective backend ("nccl" = RCCL): every collective CALL the product and bench.py make, in the form they make
    it -- the in-place all-gather whose send buffer is a row slice of its receive buffer, the flat gradient all-reduce, the MAX-reduce of
    the step time, the barrier with device ids -- so that an argument the library rejects (aliasing, views, device ids) shows up on the
    one-GPU box and not first in the driver's 8-GPU run.  (With one rank the data movement itself is trivial; two RCCL ranks cannot share
    a device.)"""
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import importlib
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        assert dist.get_backend() == "nccl"
        R, C = 4096, 36
        want = torch.arange(R * C, dtype=torch.float32, device=dev).reshape(R, C)
        full = want.clone()
        dist.all_gather_into_tensor(full, full[0:R])                  # distributed.all_gather_rows_inplace: send = the rank's own rows of `full`
        pad = torch.empty(R, C, device=dev)
        dist.all_gather_into_tensor(pad, want.contiguous())           # distributed.all_gather_rows (the ragged / out-of-place form)
        flat = torch.ones(2775633, device=dev)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)                   # distributed.all_reduce_gradients: one 11.1 MB bucket
        t = torch.tensor([1.25], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)                      # bench.py: MAX over ranks of the timed region
        dist.barrier(device_ids=[0])
        torch.cuda.synchronize()
        assert torch.equal(full, want) and torch.equal(pad, want) and float(flat.sum()) == 2775633.0 and float(t) == 1.25
        # and the product's own function inside an initialised RCCL group (one rank: one block, the buffer comes back complete)
        D = importlib.import_module("sahs-deformable-nerf_amd.distributed")
        got = D.render_rows_sharded(lambda lo, hi, rows: rows.copy_(want[lo:hi]), R, dev)
        assert torch.equal(got, want)
    finally:
        dist.destroy_process_group()


def test_rccl_accepts_the_products_collective_calls():
    mp.start_processes(_rccl_worker, args=(1, free_port()), nprocs=1, join=True, start_method="spawn")


def test_bench_main_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` exactly as the driver invokes it, on the one-GPU test box: main() starts two fresh ranks, both drive
    cuda:0 (SAHS_BENCH_ONE_GPU=1) and talk over gloo, and everything else is the N > 1 run's own code -- ProductRenderer(shard=True) ->
    run_one_iter_of_nerf(_shard=True), the launch probe, the roofline of a rank's launches, barrier-bracketed timing, MAX over ranks,
    rank 0's JSON line.  (RCCL itself: test_rccl_accepts_the_products_collective_calls.)"""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SAHS_BENCH_BACKEND="gloo", SAHS_BENCH_ONE_GPU="1")
    for size in (64, 63):      # 4096 rays: even split, in-place all-gather; 3969 rays: ragged split, padded gather
        p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--size", str(size)],
                           env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, p.stdout
        d = json.loads(lines[0])
        assert d["n_gpus"] == 2 == d["rccl_ranks"] and d["collective_backend"] == "gloo" and "rehearsal" in d and d["scaling"] == "strong"
        assert d["config"]["rays_per_step"] == size * size and "rays x2" in d["config"]["parallelism"]
        assert abs(d["value"] - size * size / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
        r = d["roofline"]
        # rank 0's launches: its half of the rays, 3 field launches per chunk and step
        assert r["bound"] == "mfma" and r["launches"] >= 2 and 0.0 < r["frac"] < 1.0 and r["chain"]["launches"] >= 6, r
        assert "bf16" not in d and "cpu_baseline" not in d          # N > 1: the headline only
