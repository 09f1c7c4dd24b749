"""One part of the field backward alone, on the training step's sizes (2048 rays x 128 fine samples by default): saving forward once, then
`--reps` backward walks of the radiance part of level 1 (fused or per-layer).  Meant to run under `rocprofv3 --kernel-trace --stats` (per-
kernel times) or bare (prints the wall time per walk from HIP events).  SAHS_NERF_LIB selects an ablation build (tools/ablate.py)."""
import argparse
import importlib
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=2048)
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--part", default="rad", choices=["rad", "def"])
    ap.add_argument("--per-layer", action="store_true")
    a = ap.parse_args()
    pkg = importlib.import_module("sahs-deformable-nerf_amd")
    ops, W = pkg.ops, pkg.weights
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)
    flat = torch.from_numpy(W.flatten_state_dict(W.hash_state_dict(0, 2.0, 30.0, hdr=True))).to(dev)
    packed = ops.pack_weights(flat)
    pose = torch.from_numpy(np.concatenate([np.eye(3), [[0.0], [0.0], [0.8]]], 1).astype(np.float32)).to(dev)
    frame = ops.fold_conditioning(flat, torch.randn(16, 29, device=dev, generator=g), pose)
    N, S = a.rays, a.samples
    rays = torch.zeros(N, 8, device=dev)
    rays[:, 2] = 0.8
    rays[:, 3:6] = torch.randn(N, 3, device=dev, generator=g) * 0.15 + torch.tensor([0, 0, -1.0], device=dev)
    z = torch.sort(torch.rand(N, S, device=dev, generator=g) * 0.6 + 0.48, dim=1).values
    xw = torch.empty(N, S, 8, device=dev)
    ops.fused_backward(not a.per_layer)
    gf, gc = torch.zeros_like(flat), torch.zeros(128, device=dev)
    if a.part == "rad":
        bits_d = ops.alloc_sign_bits(N * S, ops.FIELD_DEFORM, "audio", dev)
        ops.field_forward_split_save(packed, frame, 1, ops.FIELD_DEFORM, rays, xw, z=z, bits=bits_d)
        src = torch.arange(S, device=dev, dtype=torch.int32).repeat(N, 1).contiguous()
        bits = ops.alloc_sign_bits(N * S, ops.FIELD_RADIANCE, "audio", dev)
        _, act = ops.field_forward_split_save(packed, frame, 1, ops.FIELD_RADIANCE, rays, xw, src=src, bits=bits)
        d_raw = torch.randn(N * S, 16, device=dev, generator=g)
        run = lambda: ops.field_backward_split(flat, frame, 1, ops.FIELD_RADIANCE, act, gf, gc, d_raw=d_raw, bits=bits)
    else:
        bits = ops.alloc_sign_bits(N * S, ops.FIELD_DEFORM, "audio", dev)
        _, act = ops.field_forward_split_save(packed, frame, 1, ops.FIELD_DEFORM, rays, xw, z=z, bits=bits)
        seam = torch.randn(N * S, 8, device=dev, generator=g)
        run = lambda: ops.field_backward_split(flat, frame, 1, ops.FIELD_DEFORM, act, gf, gc, xw_grad_in=seam, bits=bits)
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    print(json.dumps(dict(part=a.part, fused=ops.fused_backward(), samples=N * S, ms_per_walk=e0.elapsed_time(e1) / a.reps,
                          lib=os.environ.get("SAHS_NERF_LIB", "default"))))


if __name__ == "__main__":
    main()
