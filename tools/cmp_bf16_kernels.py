"""The one-wave-per-SIMD bf16 kernel (precision "bf16") against the round-1 two-waves-per-SIMD kernel ("bf16_2w") and the fp32 kernel on
the same inputs: sizes from one partial tile to many tiles; prints the largest differences.  python tools/cmp_bf16_kernels.py [N S ...]"""
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
packs = {k: ops.pack_weights(flat, ops.PRECISIONS[k]) for k in ("fp32", "bf16", "bf16_2w")}
sizes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)] or [(3, 64), (4, 64), (37, 128), (1000, 128)]
ok = True
for N, S in sizes:
    g = torch.Generator(device=dev).manual_seed(N * 1000 + S)
    rays = torch.zeros(N, 8, device=dev)
    rays[:, 2] = 0.8
    rays[:, 3:6] = torch.randn(N, 3, device=dev, generator=g) * 0.15 + torch.tensor([0, 0, -1.0], device=dev)
    z = torch.sort(torch.rand(N, S, device=dev, generator=g) * 0.6 + 0.48, dim=1).values
    out = {}
    for k in ("fp32", "bf16_2w", "bf16"):
        for level in (0, 1):
            r = ops.field_forward(packs[k], frame, level, rays, z, precision=ops.PRECISIONS[k], debug=True)
            torch.cuda.synchronize()
            out[(k, level)] = r
    for level in (0, 1):
        a, b, f = out[("bf16", level)], out[("bf16_2w", level)], out[("fp32", level)]
        d_new_old = [float((x - y).abs().max()) for x, y in zip(a, b)]
        d_new_f32 = [float((x - y).abs().max()) for x, y in zip(a, f)]
        d_old_f32 = [float((x - y).abs().max()) for x, y in zip(b, f)]
        fin = all(bool(torch.isfinite(x).all()) for x in a)
        print("N %5d S %3d level %d: max |new-old| raw %.3e dx %.3e w %.3e grid %.3e | new-f32 raw %.3e | old-f32 raw %.3e | finite %s"
              % (N, S, level, d_new_old[0], d_new_old[1], d_new_old[2], d_new_old[3], d_new_f32[0], d_old_f32[0], fin), flush=True)
        ok = ok and fin and d_new_f32[0] <= 2.0 * d_old_f32[0] + 1e-3
print("OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
