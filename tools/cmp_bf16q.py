"""bf16 on the 16x16x32 MFMA shape (precision "bf16q", field_bf16q.hip) against the 32x32x16 kernel ("bf16") and fp32 on the same inputs:
whole-network evaluation and the three-launch split chain.  python tools/cmp_bf16q.py [N S ...]  (first size small)"""
import importlib
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
# the A/B kernels (csrc/ab/) are not in the shipped library: this tool runs against a development build that has them
AB_LIB = os.path.join(REPO, "sahs-deformable-nerf_amd", "build", "libsahs_ab.so")
if not os.path.exists(AB_LIB):
    importlib.import_module("sahs-deformable-nerf_amd.build").build(defines=["SAHS_AB_KERNELS"], out=AB_LIB)
os.environ["SAHS_NERF_LIB"] = AB_LIB
pkg = importlib.import_module("sahs-deformable-nerf_amd")
ops, W = pkg.ops, pkg.weights
ops.PRECISIONS = dict(ops.PRECISIONS, **ops.AB_PRECISIONS)
dev = torch.device("cuda:0")
flat = torch.from_numpy(W.flatten_state_dict(W.hash_state_dict(0, 2.0, 30.0, hdr=True))).to(dev)
rng = np.random.default_rng(0)
frame = ops.fold_conditioning(flat, torch.from_numpy(rng.standard_normal((16, 29)).astype(np.float32)).to(dev),
                              torch.from_numpy(np.concatenate([np.eye(3), [[0.0], [0.0], [0.8]]], 1).astype(np.float32)).to(dev))
packs = {k: ops.pack_weights(flat, ops.PRECISIONS[k]) for k in ("fp32", "bf16", "bf16q")}
sizes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)] or [(2, 64), (37, 128), (1000, 128)]
for N, S in sizes:
    g = torch.Generator(device=dev).manual_seed(N * 1000 + S)
    rays = torch.zeros(N, 8, device=dev)
    rays[:, 2] = 0.8
    rays[:, 3:6] = torch.randn(N, 3, device=dev, generator=g) * 0.15 + torch.tensor([0, 0, -1.0], device=dev)
    z = torch.sort(torch.rand(N, S, device=dev, generator=g) * 0.6 + 0.48, dim=1).values
    out, xws = {}, {}
    for k in ("fp32", "bf16", "bf16q"):
        for level in (0, 1):
            xw = torch.zeros(N, S, 8, device=dev)
            out[(k, level)] = ops.field_forward_split(packs[k], frame, level, ops.FIELD_ALL, rays, xw, z=z, precision=ops.PRECISIONS[k])
            xws[(k, level)] = xw
            torch.cuda.synchronize()
    for level in (0, 1):
        f, w, q = out[("fp32", level)], out[("bf16", level)], out[("bf16q", level)]
        sc = float(f.abs().max())
        print("N %5d S %3d level %d: |q - w| raw max %.3e, xw max %.3e | |w - f32| %.3e, |q - f32| %.3e (scale %.1f) | finite %s"
              % (N, S, level, float((q - w).abs().max()), float((xws[("bf16q", level)] - xws[("bf16", level)]).abs().max()),
                 float((w - f).abs().max()), float((q - f).abs().max()), sc, bool(torch.isfinite(q).all())), flush=True)
    # plain whole-network entry point and the radiance-only launch through a permutation
    q0 = ops.field_forward(packs["bf16q"], frame, 1, rays, z, precision=ops.PRECISIONS["bf16q"])
    src = torch.arange(S - 1, -1, -1, device=dev, dtype=torch.int32).repeat(N, 1).contiguous()
    qr = ops.field_forward_split(packs["bf16q"], frame, 1, ops.FIELD_RADIANCE, rays, xws[("bf16q", 1)], src=src, precision=ops.PRECISIONS["bf16q"])
    print("      plain == split(ALL): %s ; radiance through a reversing permutation == flipped: %s"
          % (bool(torch.equal(q0, out[("bf16q", 1)])), bool(torch.equal(qr, out[("bf16q", 1)].flip(1)))), flush=True)
