#!/usr/bin/env python3
"""Should the deformation nets of precision bf16x3 run on the split-operand bf16 pipe too?  Measured BEFORE building the
kernel, on the CPU, with the eager restatement (oracle/torch_eager.py): a linear layer computed as  x_hi W_hi + x_lo W_hi + x_hi W_lo  (hi =
bf16(v), lo = bf16(v - hi), fp32 accumulation) for (a) the radiance nets only = today's bf16x3, (b) the deformation nets as well, each against
the all-fp32 frame under tests/test_gpu_bf16.py's criterion (HDR weights, 48 x 48 frame, same draws; 4e-5 + 4e-4 |a| per output, every
coarse output of every ray).   python tools/experiments/emulate_x3_deform.py"""
import importlib
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from oracle import torch_eager as TE  # noqa: E402


def split(v):
    hi = v.to(torch.bfloat16).to(torch.float32)
    return hi, (v - hi).to(torch.bfloat16).to(torch.float32)


class X3Field(TE.EagerField):
    x3_prefixes = ()

    def lin(self, name, x):
        if not name.startswith(self.x3_prefixes) or not self.x3_prefixes:
            return super().lin(name, x)
        wh, wl = split(self.sd[name + ".weight"])
        xh, xl = split(x)
        return (F.linear(xh, wh) + F.linear(xl, wh)) + F.linear(xh, wl) + self.sd[name + ".bias"]


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    W = importlib.import_module("sahs-deformable-nerf_amd.weights")
    sd = {k: torch.from_numpy(v) for k, v in W.hash_state_dict(0, 2.0, 30.0, hdr=True).items()}
    H = Wd = 48
    rng = np.random.default_rng(3)
    audio = torch.from_numpy(rng.standard_normal((16, 29)).astype(np.float32))
    pose = torch.from_numpy(np.concatenate([np.eye(3), [[0.0], [0.0], [0.8]]], 1).astype(np.float32))
    intr = np.array([1200.0 * Wd / 512, 1200.0 * Wd / 512, 0.5, 0.5], np.float32)
    bg = torch.from_numpy(np.concatenate([rng.uniform(0, 1, (H * Wd, 3)), np.ones((H * Wd, 1)), np.zeros((H * Wd, 11))], 1).astype(np.float32))
    ro, rd = TE.get_ray_bundle(H, Wd, intr, pose)
    R = H * Wd
    rand = [dict(t_rand=torch.rand(R, 64), u=torch.rand(R, 64))]
    outs = {}
    for tag, pref in (("fp32", ()), ("x3 radiance", ("nerf_mlps.",)), ("x3 radiance + deformation", ("nerf_mlps.", "warp_field_mlp.", "hyper_sheep_mlp."))):
        f = X3Field(sd)
        f.x3_prefixes = pref
        with torch.no_grad():
            outs[tag] = TE.run_one_iter(f, ro, rd, 0.48, 1.08, audio, pose, bg=bg, rand=rand, perturb=True)
    names = ["rgb_c", "disp_c", "acc_c", "rgb_f", "disp_f", "acc_f", "w_bg", "depth_f"]
    for tag in list(outs)[1:]:
        print(tag)
        for nm, a, b in zip(names, outs["fp32"], outs[tag]):
            a, b = a.reshape(R, -1), b.reshape(R, -1)
            bad = ((a - b).abs() > 4e-5 + 4e-4 * a.abs()).any(dim=1)
            print("   %-8s max |d| %.2e   rays beyond 4x the fp32 tolerance: %.4f" % (nm, float((a - b).abs().max()), float(bad.float().mean())))


if __name__ == "__main__":
    main()
