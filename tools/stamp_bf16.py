#!/usr/bin/env python3
"""Where do the bf16 field kernel's waves spend a sample tile?  Diagnostic build with s_memtime stamps (SAHS_STAMP) at every chunk
start / end-of-work / after-barrier, for waves 0 and 4 (SIMD partners) of workgroup 0 on its 4th tile.
  python tools/stamp_bf16.py build     (here)      python tools/stamp_bf16.py run   (GPU box)"""
import importlib
import importlib.util
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(REPO, "sahs-deformable-nerf_amd", "build", "variants", "libsahs_stamp.so")

if sys.argv[1] == "build":
    spec = importlib.util.spec_from_file_location("sahs_build", os.path.join(REPO, "sahs-deformable-nerf_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    print(mod.build(defines=["SAHS_STAMP"] + sys.argv[2:], out=LIB))
    sys.exit(0)

import torch
sys.path.insert(0, REPO)
pkg = importlib.import_module("sahs-deformable-nerf_amd")
pkg._lib._lib, pkg._lib.LIB_PATH = None, LIB
dev = torch.device("cuda:0")
W = pkg.weights
flat = torch.from_numpy(W.flatten_state_dict(W.hash_state_dict(0, 8.0, 30.0))).to(dev)
prec = pkg.ops.PRECISIONS["bf16"]
packed = pkg.ops.pack_weights(flat, prec)
rng = np.random.default_rng(0)
frame = pkg.ops.fold_conditioning(flat, torch.from_numpy(rng.standard_normal((16, 29)).astype(np.float32)).to(dev),
                                  torch.from_numpy(np.concatenate([np.eye(3), [[0.0], [0.0], [0.8]]], 1).astype(np.float32)).to(dev))
N, S = 131072, 128
rays = torch.zeros(N, 8, device=dev)
rays[:, 2] = 0.8
rays[:, 3:6] = torch.randn(N, 3, device=dev) * 0.15 + torch.tensor([0, 0, -1.0], device=dev)
z = torch.sort(torch.rand(N, S, device=dev) * 0.6 + 0.48, dim=1).values
for _ in range(3):
    pkg.ops.field_forward(packed, frame, 1, rays, z, precision=prec)
torch.cuda.synchronize()
dbg = torch.zeros(N * S * 88, dtype=torch.float32, device=dev)
raw = torch.empty(N, S, 16, device=dev)
pkg._lib.check(pkg._lib.lib().sahs_field_forward(pkg.ops._p(packed), pkg.ops._p(frame), 1, N, S, pkg.ops._p(rays), 8, pkg.ops._p(z), pkg.ops._p(raw),
                                                 pkg.ops._p(dbg), prec, pkg.ops._stream()), "stamp run")
torch.cuda.synchronize()
st = dbg[:4096].cpu().numpy().view(np.int64)
for wname, base in (("wave0", 0), ("wave4", 512)):
    s = st[base:base + 512]
    n = int(np.count_nonzero(s))
    s = s[:n]
    t0 = s[0]
    tri = s[1:1 + 3 * ((n - 1) // 3)].reshape(-1, 3)
    work = tri[:, 1] - tri[:, 0]
    wait = tri[:, 2] - tri[:, 1]
    gaps = np.concatenate([[tri[0, 0] - t0], tri[1:, 0] - tri[:-1, 2]])
    total = tri[-1, 2] - t0
    print("%s: %d chunks, tile %d ticks: in-chunk work %d (%.1f %%), barrier wait %d (%.1f %%), between chunks (PE, grid, bias, DMA setup) %d (%.1f %%)"
          % (wname, len(tri), total, work.sum(), 100 * work.sum() / total, wait.sum(), 100 * wait.sum() / total, gaps.sum(), 100 * gaps.sum() / total))
    print("  chunk: work / wait / gap-before  (ticks)")
    print("  " + "  ".join("%d/%d/%d" % (w, b, g) for w, b, g in zip(work, wait, gaps)))
