#!/usr/bin/env python3
"""Where does the one-wave-per-SIMD bf16 field kernel spend a sample tile?  Diagnostic build (-DSAHS_STAMP_W) with s_memtime stamps
between the sections of the network, wave 0 of workgroup 0 on its 4th tile; each section is priced against its own MFMA count
(32 cycles per v_mfma_f32_32x32x16_bf16 on the wave's SIMD).
  python tools/stamp_bf16w.py build [NAME [DEFINE ...]]    (here)      python tools/stamp_bf16w.py run [NAME ...]   (GPU box)
NAME labels a variant of the diagnostic build (extra -D defines: cycle counts are immune to the clock the chip happens to hold, so
ablations that change the data -- and with it the power and the clock -- can be compared in cycles)."""
import importlib
import importlib.util
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VDIR = os.path.join(REPO, "sahs-deformable-nerf_amd", "build", "variants")
lib_of = lambda name: os.path.join(VDIR, "libsahs_stampw_%s.so" % name)
# (label, MFMAs this wave issues in the section): layer = NT32 tiles x 2 KB32 k-steps x 2 halves
SECTIONS = [("ray points + PE(x)", 0), ("warp net (7 layers)", 400), ("hyper net (7 layers) + stash", 120), ("PE(x'), PE(w)", 0), ("T0 (95->256)", 96),
            ("T1", 256), ("T2", 256), ("PE(x'), PE(w) again", 0), ("T3 (skip, 351->256)", 352), ("T4, T5 (loop pass 1)", 512), ("T6, T7 (loop pass 2, same code)", 512), ("feat", 256), ("sigma head", 32),
            ("PE(dir) + grid lookup", 0), ("colour branch", 368), ("seg branch", 336), ("store", 0)]

if sys.argv[1] == "build":
    spec = importlib.util.spec_from_file_location("sahs_build", os.path.join(REPO, "sahs-deformable-nerf_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    os.makedirs(VDIR, exist_ok=True)
    print(mod.build(defines=["SAHS_DIAG", "SAHS_STAMP_W"] + sys.argv[3:], out=lib_of(sys.argv[2] if len(sys.argv) > 2 else "base")))
    sys.exit(0)

import torch
sys.path.insert(0, REPO)
pkg = importlib.import_module("sahs-deformable-nerf_amd")
dev = torch.device("cuda:0")
W = pkg.weights
flat = torch.from_numpy(W.flatten_state_dict(W.hash_state_dict(0, 2.0, 30.0, hdr=True))).to(dev)
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
for name in (sys.argv[2:] or ["base"]):
    pkg._lib._lib, pkg._lib.LIB_PATH = None, lib_of(name)
    print("== variant", name)
    for _ in range(3):
        pkg.ops.field_forward(packed, frame, 1, rays, z, precision=prec)
    torch.cuda.synchronize()
    dbg = torch.zeros(N * S * 88, dtype=torch.float32, device=dev)
    raw = torch.empty(N, S, 16, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    pkg._lib.check(pkg._lib.lib().sahs_field_forward(pkg.ops._p(packed), pkg.ops._p(frame), 1, N, S, pkg.ops._p(rays), 8, pkg.ops._p(z), pkg.ops._p(raw),
                                                     pkg.ops._p(dbg), prec, pkg.ops._stream()), "stamp run")
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    st = dbg[:256].cpu().numpy().view(np.int64)
    n = int(np.count_nonzero(st))
    st = st[:n]
    assert n == len(SECTIONS) + 1, n
    d = np.diff(st)
    total = st[-1] - st[0]
    tiles_per_cu = N * S / 256 / 256
    print("launch %.2f ms = %.1f us per tile; stamped tile %d ticks of s_memtime (%.1f ticks/us)" % (ms, ms * 1e3 / tiles_per_cu, total, total / (ms * 1e3 / tiles_per_cu)))
    mf_total = sum(m for _, m in SECTIONS)
    print("%-32s %9s %7s %7s %9s" % ("section", "ticks", "share", "MFMAs", "ticks/MFMA"))
    for (sec, mf), t in zip(SECTIONS, d):
        print("%-32s %9d %6.1f%% %7d %9s" % (sec, t, 100.0 * t / total, mf, ("%.1f" % (t / mf)) if mf else "-"))
    print("%-32s %9d %6.1f%% %7d %9.1f" % ("tile", total, 100.0, mf_total, total / mf_total))
