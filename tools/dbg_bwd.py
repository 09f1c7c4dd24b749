"""Diagnostic: per-tensor error of the HIP field backward against float64 autograd (and fp32 autograd's own error), audio model."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import torch_eager as TE
sahs = importlib.import_module("sahs-deformable-nerf_amd")
ops, W = sahs.ops, sahs.weights
dev = torch.device("cuda:0")
arch = sys.argv[1] if len(sys.argv) > 1 else "audio"
level = int(sys.argv[2]) if len(sys.argv) > 2 else 1
N = int(sys.argv[3]) if len(sys.argv) > 3 else 24
S = 64 if level == 0 else 128
gen = torch.Generator(device=dev).manual_seed(11 + level)
kw = {} if arch == "audio" else dict(model=arch)
sd_np = W.hash_state_dict(0, 8.0, 30.0, **kw)
flat = torch.from_numpy(W.flatten_state_dict(sd_np, **kw)).to(dev)
packed = ops.pack_weights(flat, arch=arch)
driving = torch.randn(16, 29, device=dev, generator=gen) if arch == "audio" else torch.randn(76, device=dev, generator=gen) * 0.5
near, far, cam = (0.48, 1.08, 0.8) if arch == "audio" else (0.2, 0.8, 0.5)
pose = torch.from_numpy(np.concatenate([np.eye(3), [[0.0], [0.0], [cam]]], 1).astype(np.float32)).to(dev)
rays = torch.zeros(N, 8, device=dev); rays[:, 2] = cam
rays[:, 3:6] = torch.randn(N, 3, device=dev, generator=gen) * 0.15 + torch.tensor([0, 0, -1.0], device=dev)
z = torch.sort(torch.rand(N, S, device=dev, generator=gen) * (far - near) + near, dim=1).values
d_raw = torch.randn(N * S, 16, device=dev, generator=gen)
frame = ops.fold_conditioning(flat, driving, pose, arch=arch)
raw, act = ops.field_forward_save(packed, frame, level, rays, z, arch)
gf = torch.zeros_like(flat); gc = torch.zeros(128, device=dev)
ops.field_backward(flat, frame, level, act, d_raw, gf, gc, arch)
x6 = torch.cat([rays[:, None, 0:3] + rays[:, None, 3:6] * z[..., None], rays[:, None, 3:6].expand(N, S, 3)], -1).reshape(-1, 6)
lvl = "coarse" if level == 0 else "fine"
def autograd(dt):
    sd = {k: torch.from_numpy(v).to(dev).to(dt).requires_grad_(True) for k, v in sd_np.items()}
    r = TE.EagerField(sd, arch=arch).forward(lvl, x6.to(dt), driving.to(dt), pose.to(dt))
    (r * d_raw.to(dt)).sum().backward()
    return r.detach(), sd
r32, g32 = autograd(torch.float32); r64, g64 = autograd(torch.float64)
print("raw: hip-vs-f64 %.3e  f32-vs-f64 %.3e" % (float((raw.reshape(-1, 16).double() - r64).abs().max()), float((r32.double() - r64).abs().max())))
rows = []
for k, (o, shape) in W.canonical_offsets(arch).items():
    if ("nerf_mlps." in k and lvl not in k) or k.startswith("audNet"): continue
    ref = g64[k].grad; got = gf[o:o + ref.numel()].view_as(ref).double()
    sc = float(ref.abs().max()) + 1e-30
    rows.append((float((got - ref).abs().max()) / sc, float((g32[k].grad.double() - ref).abs().max()) / sc, k, sc))
for e, e32, k, sc in sorted(rows, reverse=True)[:12]:
    print("%-44s hip %.3e   fp32 autograd %.3e   scale %.3e" % (k, e, e32, sc))
