"""Torch-eager restatement of the hot path (TEST INFRASTRUCTURE ONLY, like the rest of oracle/).

Own code, written op for op the way the reference evaluates the path (same chunking, the same
`.repeat` of the per-frame conditioning to every point, the same `cat`s), so that it serves as
  (a) the differentiable oracle for gradients (autograd through plain torch ops), and
  (b) the "reference single-GPU PyTorch path" baseline on the GPU box, where the reference's own
      files cannot be present.
It is pinned to the real reference by tests/test_torch_eager_vs_golden.py (forward seams, five
end-to-end configurations, and the golden gradients).  Citations: /root/reference/nerf-pytorch/nerf/.
The one deliberate deviation: sigma's `+= 1e-6` on the last sample is applied out of place, because the
reference's in-place form breaks autograd under torch >= 1.10; the maths is the same.
"""
import torch
import torch.nn.functional as F


def positional_encoding(x, L, include_input=True):
    """nerf_helpers.py:305-349 (log sampling)."""
    out = [x] if include_input else []
    for k in range(L):
        f = 2.0 ** k
        out.append(torch.sin(x * f))
        out.append(torch.cos(x * f))
    return torch.cat(out, dim=-1)


def get_ray_bundle(H, W, intr, c2w):
    """nerf_helpers.py:178-233."""
    ii, jj = torch.meshgrid(torch.arange(W, dtype=c2w.dtype, device=c2w.device), torch.arange(H, dtype=c2w.dtype, device=c2w.device), indexing="xy")
    d = torch.stack([(ii - W * intr[2]) / intr[0], -(jj - H * intr[3]) / intr[1], -torch.ones_like(ii)], dim=-1)
    rd = torch.sum(d[..., None, :] * c2w[:3, :3], dim=-1)
    return c2w[:3, -1].expand(rd.shape), rd


class EagerField:
    """AudioFaceModel.forward (models.py:514-528) / NeRFaceModel.forward (models.py:366-378) over a state_dict of tensors.
    arch: "audio" (config/audio), "nerface" (config/expression/person_2|3.yml), "nerface_static" (person_1.yml)."""

    ARCH = {"audio": dict(L=10, L_amb=4, amb_inc=True, layers=8, deform=True, audionet=True, trunk_pose=True),
            "nerface": dict(L=15, L_amb=15, amb_inc=False, layers=4, deform=True, audionet=False, trunk_pose=False),
            "nerface_static": dict(L=10, L_amb=0, amb_inc=False, layers=4, deform=False, audionet=False, trunk_pose=False)}

    def __init__(self, sd, num_coarse=64, num_fine=64, arch="audio", masks=None):
        self.sd = sd
        self.num_coarse, self.num_fine = num_coarse, num_fine
        self.a = self.ARCH[arch]
        # masks: optional {"warp.i" | "hyper.i" | "trunk.i" | "dir.i" | "seg.i": bool (P, width)} -- the branch every (leaky) ReLU
        # takes, imposed from outside.  Used by the gradient tests to compare derivatives on the SAME side of every kink: two
        # correct fp32 forwards that differ by 1e-5 put a few pre-activations on different sides of zero, and one such sample
        # moves a bias-gradient entry by percents.
        self.masks = masks

    def act(self, name, x, slope):
        if self.masks is None:
            return F.leaky_relu(x, slope) if slope > 0 else torch.relu(x)
        return x * torch.where(self.masks[name], torch.ones((), dtype=x.dtype, device=x.device), torch.full((), slope, dtype=x.dtype, device=x.device))

    def lin(self, name, x):
        return F.linear(x, self.sd[name + ".weight"], self.sd[name + ".bias"])

    def audionet(self, audio):
        """modules.py:68-73."""
        x = audio.unsqueeze(0)[:, 0:16, :].permute(0, 2, 1)
        for i in (0, 2, 4, 6):
            x = F.leaky_relu(F.conv1d(x, self.sd["audNet_head.encoder_conv.%d.weight" % i], self.sd["audNet_head.encoder_conv.%d.bias" % i],
                                      stride=2, padding=1), 0.02)
        x = x.squeeze(-1)
        x = F.leaky_relu(self.lin("audNet_head.encoder_fc1.0", x), 0.02)
        return self.lin("audNet_head.encoder_fc1.2", x).squeeze()

    @staticmethod
    def pose_encoding(pose):
        """models.py:482-504 + encode_pose_fn :203-207."""
        R = pose[:3, :3]
        e = torch.stack([torch.atan2(R[2, 2], R[1, 2]), torch.asin(-R[0, 2]), torch.atan2(R[0, 0], -R[0, 1])])
        return positional_encoding(torch.cat([e, pose[:3, 3]])[None], 3, include_input=False)

    def deform(self, prefix, list_name, final, n_layers, skip, initial):
        x = initial
        tag = "warp" if prefix.startswith("warp") else "hyper"
        for i in range(n_layers):
            x = self.act("%s.%d" % (tag, i), self.lin("%s.%s.%d" % (prefix, list_name, i), torch.cat((x, initial), -1) if i == skip else x), 0.0)
        return self.lin("%s.%s" % (prefix, final), x)

    def grid(self, level, xyz):
        """models.py:346-365."""
        n = self.num_coarse + (self.num_fine if level == "fine" else 0)
        g = self.sd["spatial_embeddings"]
        c = xyz.to(g.dtype).reshape((-1, n, 3))      # models.py:355 `.float()`; float64 when the whole field is evaluated in double
        B = c.shape[0]
        s = F.grid_sample(g.expand(B, -1, -1, -1, -1), c.reshape(B, 1, 1, -1, 3), mode="bilinear", padding_mode="zeros", align_corners=True)
        N, C, H, W, D = s.shape
        return s.permute(0, 4, 3, 2, 1).reshape(N * H * W * D, C)

    def forward(self, level, x, audio, pose):
        a = self.a
        P = x.shape[0]
        xyz, dirs = x[..., :3], x[..., 3:6]
        driving = (self.audionet(audio) if a["audionet"] else audio).repeat(P, 1)      # models.py:517-518 / :368
        pose36 = self.pose_encoding(pose).repeat(P, 1)                                 # :519-521 / :369-370
        L = a["L"]
        if a["deform"]:
            initial = torch.cat((positional_encoding(xyz, L), driving, pose36), dim=1)
            warped = xyz + torch.tanh(self.deform("warp_field_mlp", "layers_xyz", "fc_final", 6, 4, initial))     # :304-305
            initial = torch.cat((positional_encoding(xyz, L), driving, pose36), dim=1)                           # PE recomputed, :310
            amb = self.deform("hyper_sheep_mlp", "layers_ambient", "fc_ambient", 6, 4, initial)
            enc = torch.cat((positional_encoding(warped, L), positional_encoding(amb, a["L_amb"], a["amb_inc"])), dim=1)
        else:
            warped = xyz                                                                                          # :316-327
            enc = positional_encoding(warped, L)
        feats = self.grid(level, warped)
        p = "nerf_mlps.%s." % level
        init = torch.cat((enc, pose36 if a["trunk_pose"] else driving), dim=1)                                    # modules.py:255-267
        h = init
        for i in range(a["layers"]):
            h = self.act("trunk.%d" % i, self.lin(p + "layers_xyz.%d" % i, torch.cat((h, init), -1) if i == 3 else h), 0.01)
        feat = self.lin(p + "fc_feat", h)
        alpha = self.lin(p + "fc_alpha", feat)
        c = torch.cat((feat, positional_encoding(dirs, 4), feats), -1)
        for i in range(4):
            c = self.act("dir.%d" % i, self.lin(p + "layers_dir.%d" % i, c), 0.01)
        rgb = self.lin(p + "fc_rgb", c)
        s = feat
        for i in range(4):
            s = self.act("seg.%d" % i, self.lin(p + "layers_seg.%d" % i, s), 0.01)
        return torch.cat((rgb, self.lin(p + "fc_seg", s), alpha), dim=-1)


def run_network(field, level, pts, rays, chunksize, audio, pose):
    """train_utils.py:9-50 (point rows carry the 12-class mask when the ray table does)."""
    flat = pts.reshape((-1, 3))
    dirs = rays[..., None, 3:6].expand(pts.shape).reshape((-1, 3))
    rows = torch.cat((flat, dirs), dim=-1)
    if rays.shape[-1] > 8:
        rows = torch.cat((rows, rays[..., None, 8:].expand(pts.shape[0], pts.shape[1], rays.shape[-1] - 8).reshape((-1, rays.shape[-1] - 8))), dim=-1)
    out = torch.cat([field.forward(level, rows[i:i + chunksize], audio, pose) for i in range(0, rows.shape[0], chunksize)], dim=0)
    return out.reshape(list(pts.shape[:-1]) + [16])


def volume_render(raw, z, rd, noise=None, white_background=False, bg_mode=True):
    """volume_rendering_utils.py:7-78; raw's last sample already holds the prior when bg_mode."""
    dists = torch.cat((z[..., 1:] - z[..., :-1], torch.full_like(z[..., :1], 1e10)), dim=-1) * rd[..., None, :].norm(p=2, dim=-1)
    if bg_mode:
        col = torch.cat((torch.sigmoid(raw[:, :-1, :3]), torch.softmax(raw[:, :-1, 3:-1], dim=-1)), dim=-1)
        col = torch.cat((col, raw[:, -1, :-1].unsqueeze(1)), dim=1)
    else:
        col = torch.sigmoid(raw[..., :-1])
    sigma = torch.relu(raw[..., -1] + (noise if noise is not None else 0.0))
    sigma = torch.cat((sigma[:, :-1], sigma[:, -1:] + 1e-6), dim=1)
    alpha = 1.0 - torch.exp(-sigma * dists)
    T = torch.cumprod(1.0 - alpha + 1e-10, -1)
    T = torch.cat((torch.ones_like(T[:, :1]), T[:, :-1]), dim=-1)
    w = alpha * T
    rgb = (w[..., None] * col).sum(dim=-2)
    depth = (w * z).sum(dim=-1)
    acc = w.sum(dim=-1)
    disp = 1.0 / torch.max(1e-10 * torch.ones_like(depth), depth / acc)
    if white_background:
        rgb = rgb + (1.0 - acc[..., None])
    return rgb, disp, acc, w, depth


def sample_pdf_2(bins, weights, num_samples, u=None):
    """nerf_helpers.py:454-497; u=None -> det=True."""
    weights = weights + 1e-5
    pdf = weights / torch.sum(weights, dim=-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[..., :1]), torch.cumsum(pdf, dim=-1)], dim=-1)
    if u is None:
        u = torch.linspace(0.0, 1.0, steps=num_samples, dtype=weights.dtype, device=weights.device).expand(list(cdf.shape[:-1]) + [num_samples])
    u = u.contiguous()
    inds = torch.searchsorted(cdf.detach().contiguous(), u, right=True)
    below, above = torch.clamp(inds - 1, min=0), torch.clamp(inds, max=cdf.shape[-1] - 1)
    cb, ca = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    bb, ba = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    den = ca - cb
    den = torch.where(den < 1e-5, torch.ones_like(den), den)
    return bb + (u - cb) / den * (ba - bb)


def render_rays(field, rays, audio, pose, num_coarse, num_fine, chunksize, bg=None, t_rand=None, noise_c=None, u=None, noise_f=None,
                lindisp=False, white_background=False, perturb=True):
    """predict_and_render_radiance, train_utils.py:72-206. Random tensors None => drawn here in the reference's order."""
    N = rays.shape[0]
    ro, rd = rays[..., :3], rays[..., 3:6]
    near, far = rays[..., 6:7], rays[..., 7:8]
    t = torch.linspace(0.0, 1.0, num_coarse, dtype=ro.dtype, device=ro.device)
    z = near * (1.0 - t) + far * t if not lindisp else 1.0 / (1.0 / near * (1.0 - t) + 1.0 / far * t)
    z = z.expand([N, num_coarse])
    if perturb:
        mids = 0.5 * (z[..., 1:] + z[..., :-1])
        upper, lower = torch.cat((mids, z[..., -1:]), dim=-1), torch.cat((z[..., :1], mids), dim=-1)
        if t_rand is None:
            t_rand = torch.rand(z.shape, dtype=ro.dtype, device=ro.device)
        z = lower + (upper - lower) * t_rand
    pts = ro[..., None, :] + rd[..., None, :] * z[..., :, None]
    raw = run_network(field, "coarse", pts, rays, chunksize, audio, pose)
    if bg is not None:
        raw = torch.cat((raw[:, :-1], torch.cat((bg, raw[:, -1, -1:]), dim=-1).unsqueeze(1)), dim=1)   # train_utils.py:135-136, out of place
    rgb_c, disp_c, acc_c, w, _ = volume_render(raw, z, rd, noise_c, white_background, bg is not None)
    z_mid = 0.5 * (z[..., 1:] + z[..., :-1])
    if perturb and u is None:
        u = torch.rand((N, num_fine), dtype=ro.dtype, device=ro.device)
    zs = sample_pdf_2(z_mid, w[..., 1:-1], num_fine, u=u if perturb else None).detach()
    zf, _ = torch.sort(torch.cat((z, zs), dim=-1), dim=-1)
    pts = ro[..., None, :] + rd[..., None, :] * zf[..., :, None]
    raw = run_network(field, "fine", pts, rays, chunksize, audio, pose)
    if bg is not None:
        raw = torch.cat((raw[:, :-1], torch.cat((bg, raw[:, -1, -1:]), dim=-1).unsqueeze(1)), dim=1)
    rgb_f, disp_f, acc_f, wf, depth_f = volume_render(raw, zf, rd, noise_f, white_background, bg is not None)
    return rgb_c, disp_c, acc_c, rgb_f, disp_f, acc_f, wf[:, -1], depth_f


def run_one_iter(field, ro, rd, near, far, audio, pose, num_coarse=64, num_fine=64, chunksize=131072, bg=None, rand=None, perturb=True,
                 noise_std=0.0, mask=None):
    """run_one_iter_of_nerf (train_utils.py:209-321), flat outputs. rand: per ray chunk dict(t_rand, noise_c, u, noise_f) or None."""
    ro, rd = ro.reshape((-1, 3)), rd.reshape((-1, 3))
    parts = [ro, rd, near * torch.ones_like(rd[..., :1]), far * torch.ones_like(rd[..., :1])]
    if mask is not None:
        parts.append(mask.reshape((-1, mask.shape[-1])))
    rays = torch.cat(parts, dim=-1)
    outs = []
    for ci, s in enumerate(range(0, rays.shape[0], chunksize)):
        r = (rand[ci] if rand is not None else None) or {}
        N = min(chunksize, rays.shape[0] - s)
        nz = lambda k, S: (r[k] if k in r else (torch.randn((N, S), device=rays.device) * noise_std if noise_std > 0 else None))
        t_rand = r.get("t_rand")
        if perturb and t_rand is None:
            t_rand = torch.rand((N, num_coarse), device=rays.device)
        noise_c = nz("noise_c", num_coarse)
        u = r.get("u")
        if perturb and u is None:
            u = torch.rand((N, num_fine), device=rays.device)
        noise_f = nz("noise_f", num_coarse + num_fine)
        outs.append(render_rays(field, rays[s:s + chunksize], audio, pose, num_coarse, num_fine, chunksize,
                                bg=None if bg is None else bg[s:s + chunksize], t_rand=t_rand, noise_c=noise_c, u=u, noise_f=noise_f, perturb=perturb))
    return tuple(torch.cat(o, dim=0) for o in zip(*outs))
