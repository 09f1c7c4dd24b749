"""bf16x3 (fp32 deformation nets + radiance nets with bf16 hi/lo operands, three MFMAs per product) against the fp32 and the plain bf16
kernels on the same inputs: python tools/cmp_bf16x3.py [N S ...]   (first size small: a new kernel meets the GPU on a tiny case first)"""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("sahs-deformable-nerf_amd")
ops, W = pkg.ops, pkg.weights
dev = torch.device("cuda:0")
flat = torch.from_numpy(W.flatten_state_dict(W.hash_state_dict(0, 2.0, 30.0, hdr=True))).to(dev)
rng = np.random.default_rng(0)
frame = ops.fold_conditioning(flat, torch.from_numpy(rng.standard_normal((16, 29)).astype(np.float32)).to(dev),
                              torch.from_numpy(np.concatenate([np.eye(3), [[0.0], [0.0], [0.8]]], 1).astype(np.float32)).to(dev))
packs = {k: ops.pack_weights(flat, ops.PRECISIONS[k]) for k in ("fp32", "bf16", "bf16x3")}
sizes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)] or [(2, 64), (37, 128), (1000, 128)]
for N, S in sizes:
    g = torch.Generator(device=dev).manual_seed(N * 1000 + S)
    rays = torch.zeros(N, 8, device=dev)
    rays[:, 2] = 0.8
    rays[:, 3:6] = torch.randn(N, 3, device=dev, generator=g) * 0.15 + torch.tensor([0, 0, -1.0], device=dev)
    z = torch.sort(torch.rand(N, S, device=dev, generator=g) * 0.6 + 0.48, dim=1).values
    out = {}
    for k in ("fp32", "bf16", "bf16x3"):
        xw = torch.zeros(N, S, 8, device=dev)
        out[k] = ops.field_forward_split(packs[k], frame, 1, ops.FIELD_ALL, rays, xw, z=z, precision=ops.PRECISIONS[k])
        torch.cuda.synchronize()
    f = out["fp32"]
    sc_c, sc_s = float(f[..., :15].abs().max()), float(f[..., 15].abs().max())
    for k in ("bf16", "bf16x3"):
        d = (out[k] - f).abs()
        print("N %5d S %3d %-7s: colour/seg max %.3e rms %.3e (scale %.2f) | sigma max %.3e rms %.3e (scale %.2f) | finite %s"
              % (N, S, k, float(d[..., :15].max()), float((d[..., :15] ** 2).mean().sqrt()), sc_c, float(d[..., 15].max()),
                 float((d[..., 15] ** 2).mean().sqrt()), sc_s, bool(torch.isfinite(out[k]).all())), flush=True)
