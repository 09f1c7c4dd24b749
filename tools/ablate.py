#!/usr/bin/env python3
"""Timing-only ablations of the field kernels (results of ablated builds are WRONG by construction).
Build here:  python tools/ablate.py build      (variants -> sahs-deformable-nerf_amd/build/variants/)
Run on GPU:  python tools/ablate.py run [bf16|fp32]
"""
import importlib.util
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VDIR = os.path.join(REPO, "sahs-deformable-nerf_amd", "build", "variants")
# every timing-only switch in the sources sits behind SAHS_DIAG (sahs-deformable-nerf_amd/build.py refuses it for the shipped library)
VARIANTS = {"base": ["SAHS_DIAG"],
            # the split-operand forward / backward pipeline (csrc/bf16x3_pipe.hpp): without the weight DMA, the chunk barriers, the A-fragment reads
            "x3nodma": ["SAHS_DIAG", "SAHS_X3_NODMA"], "x3nobar": ["SAHS_DIAG", "SAHS_X3_NOBARRIER"], "x3noaread": ["SAHS_DIAG", "SAHS_X3_NOAREAD"],
            "x3saveplain": ["SAHS_DIAG", "SAHS_X3_SAVE_PLAIN"],      # the saving forward's activation stores with the default cache policy
            "tnnodma": ["SAHS_DIAG", "SAHS_TN_NODMA"],      # the weight-gradient job kernels without their operand fetch (what the K loop costs by itself)
            "tnfnodma": ["SAHS_DIAG", "SAHS_TNF_NODMA"], "tnfnomfma": ["SAHS_DIAG", "SAHS_TNF_NOMFMA"], "tnfnoatomic": ["SAHS_DIAG", "SAHS_TNF_NOATOMIC"],      # the fp32 weight-gradient job kernels: without the operand fetch / without the wide kernel's MFMAs
            "x3noreadback": ["SAHS_DIAG", "SAHS_X3_NOREADBACK"],      # saving forward + backward chains: the line stores without the LDS read-back they wait for
            "x3floor": ["SAHS_DIAG", "SAHS_X3_NODMA", "SAHS_X3_NOBARRIER", "SAHS_X3_NOAREAD"],
            # the backward chain kernels (csrc/field_bwd_chain.hip): what they wait for (tools/ab_chain.sh runs them under rocprofv3)
            "cnostore": ["SAHS_DIAG", "SAHS_BWC_NOSTORE"], "cnogstore": ["SAHS_DIAG", "SAHS_BWC_NOGSTORE"], "ctilemajor": ["SAHS_DIAG", "SAHS_BWC_TILEMAJOR"], "cplainstore": ["SAHS_DIAG", "SAHS_BWC_PLAINSTORE"],
            "cnomask": ["SAHS_DIAG", "SAHS_BWC_NOMASK"], "cnostore_nomask": ["SAHS_DIAG", "SAHS_BWC_NOSTORE", "SAHS_BWC_NOMASK"]}


def build():
    spec = importlib.util.spec_from_file_location("sahs_build", os.path.join(REPO, "sahs-deformable-nerf_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    os.makedirs(VDIR, exist_ok=True)
    only = sys.argv[2].split(",") if len(sys.argv) > 2 else list(VARIANTS)
    for name in only:
        print(mod.build(defines=VARIANTS[name], out=os.path.join(VDIR, "libsahs_%s.so" % name)))


def time_one(precision, libpath=None, N=131072, S=128):
    import numpy as np
    import torch
    sys.path.insert(0, REPO)
    pkg = importlib.import_module("sahs-deformable-nerf_amd")
    if libpath is not None:      # swap the bound library in-process (one torch import for all variants)
        pkg._lib._lib, pkg._lib.LIB_PATH = None, libpath
    dev = torch.device("cuda:0")
    W = pkg.weights
    flat = torch.from_numpy(W.flatten_state_dict(W.hash_state_dict(0, 8.0, 30.0))).to(dev)
    prec = pkg.ops.PRECISIONS[precision]
    packed = pkg.ops.pack_weights(flat, prec)
    rng = np.random.default_rng(0)
    frame = pkg.ops.fold_conditioning(flat, torch.from_numpy(rng.standard_normal((16, 29)).astype(np.float32)).to(dev),
                                      torch.from_numpy(np.concatenate([np.eye(3), [[0.0], [0.0], [0.8]]], 1).astype(np.float32)).to(dev))
    rays = torch.zeros(N, 8, device=dev)
    rays[:, 2] = 0.8
    rays[:, 3:6] = torch.randn(N, 3, device=dev) * 0.15 + torch.tensor([0, 0, -1.0], device=dev)
    z = torch.sort(torch.rand(N, S, device=dev) * 0.6 + 0.48, dim=1).values
    raw = torch.empty(N, S, 16, device=dev)
    if precision == "bf16x3":       # the radiance launch of the split chain alone (x', w from a deformation launch)
        xw = torch.zeros(N, S, 8, device=dev)
        pkg.ops.field_forward_split(packed, frame, 1, pkg.ops.FIELD_DEFORM, rays, xw, z=z, precision=prec)
        src = torch.arange(S, device=dev, dtype=torch.int32).repeat(N, 1).contiguous()
        run = lambda: pkg.ops.field_forward_split(packed, frame, 1, pkg.ops.FIELD_RADIANCE, rays, xw, src=src, out=raw, precision=prec)
    else:
        run = lambda: pkg.ops.field_forward(packed, frame, 1, rays, z, precision=prec, out=raw)
    for _ in range(2):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    return {"ms": round(ms, 3), "tflops": round(N * S * 1855744 / ms / 1e9, 1)}


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        precision = sys.argv[2] if len(sys.argv) > 2 else "bf16"
        names = sys.argv[3].split(",") if len(sys.argv) > 3 else list(VARIANTS)
        for rep in range(2):
            for name in names:
                print("%-18s %s" % (name, time_one(precision, os.path.join(VDIR, "libsahs_%s.so" % name))), flush=True)
