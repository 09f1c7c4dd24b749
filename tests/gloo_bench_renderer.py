"""Renderer injected into bench.py's main() by tests/test_distributed_gloo.py (SAHS_BENCH_RENDERER=<this file>:make, SAHS_BENCH_BACKEND=gloo):
the CPU oracle stands in for the HIP kernels (tests may use it) behind the PRODUCT's own sharding function,
distributed.render_rows_sharded -- shard bounds, in-place rows, one all-gather -- so that `python bench.py --gpus 2` is rehearsed end to end
without a GPU: self-launch, rendezvous, barrier-bracketed timing, MAX over ranks, rank 0's JSON line.  Frame: size x size rays."""
import importlib
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
KEYS = ["rgb_c", "disp_c", "acc_c", "rgb_f", "disp_f", "acc_f", "w_bg", "depth_f"]


class OracleRenderer:
    def __init__(self, args, world, rank):
        from oracle import oracle
        self.oracle, self.world, self.rank = oracle, world, rank
        self.D = importlib.import_module("sahs-deformable-nerf_amd.distributed")
        Wm = importlib.import_module("sahs-deformable-nerf_amd.weights")
        self.num_rays = n = args.size * args.size
        rng = np.random.default_rng(100 + n)
        self.flat = Wm.flatten_state_dict(Wm.hash_state_dict(0, 8.0, 30.0))
        rays = np.zeros((n, 8), np.float32)
        rays[:, 0:3] = [0, 0, 0.8]
        rays[:, 3:6] = rng.normal(0, 0.15, (n, 3)) + np.array([0, 0, -1.0])
        rays[:, 6], rays[:, 7] = 0.48, 1.08
        self.rays = rays
        audio = rng.standard_normal((16, 29)).astype(np.float32)
        pose = np.concatenate([np.eye(3), [[0.0], [0.0], [0.8]]], axis=1).astype(np.float32)
        self.drv, self.p36 = oracle.audionet(self.flat, audio), oracle.pose_encoding(pose)
        self.bg = rng.uniform(0, 1, (n, 15)).astype(np.float32)
        self.blocks = []

    def describe(self):
        return "%d rays x (8 + 16) samples by the CPU oracle behind distributed.render_rows_sharded" % self.num_rays

    def _render(self, lo, hi):      # draws keyed by GLOBAL ray index (sahs_ray_uniforms restated by the oracle)
        o = self.oracle.render_rays(self.flat, self.rays[lo:hi], 8, 8, self.drv, self.p36, bg=self.bg[lo:hi],
                                    t_rand=self.oracle.ray_uniforms(42, 0, lo, hi - lo, 8), u=self.oracle.ray_uniforms(42, 1, lo, hi - lo, 8))
        return self.D.pack_outputs(tuple(torch.from_numpy(o[k]) for k in KEYS))

    def frame(self):
        def block(lo, hi, rows):
            self.blocks.append((lo, hi))
            rows.copy_(self._render(lo, hi))
        return self.D.render_rows_sharded(block, self.num_rays, torch.device("cpu"))

    def check(self, out):
        assert tuple(out.shape) == (self.num_rays, 36)
        assert set(self.blocks) == {self.D.shard_bounds(self.num_rays, self.world, self.rank)}, self.blocks     # this rank's block only, every step
        ref = self._render(0, self.num_rays)
        assert torch.equal(out, ref), "rank %d: the gathered frame differs from the single-process render" % self.rank


def make(args, world, rank):
    return OracleRenderer(args, world, rank)


class SleepyRenderer(OracleRenderer):
    """A rank that never comes back from its frame (a wedged collective, as far as the launcher can tell): bench.py's wall-clock guard must
    end the run with an error line instead of waiting for the driver's limit (tests/test_distributed_gloo.py)."""

    def frame(self):
        import time
        time.sleep(3600)


def make_sleepy(args, world, rank):
    return SleepyRenderer(args, world, rank)
