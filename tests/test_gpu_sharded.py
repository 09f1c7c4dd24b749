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

from conftest import REPO, VARIANT_KW

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
    port = 37500 + (os.getpid() % 2000)
    mp.start_processes(_worker, args=(2, port, size, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
