#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Runs only in the build container: it imports the unmodified reference from
/root/reference/nerf-pytorch/nerf (CPU PyTorch) through the harness shim of SURVEY.md
section 8(c) -- an empty ``nerf`` package object whose ``__path__`` points at the reference (so its
``__init__`` with the cv2-dependent loaders never runs) plus empty ``pytorch3d`` stubs (only an
off-path helper uses them) -- and refuses to run when the reference is absent.  The GPU box
never sees the reference; it only sees the .npz files this script wrote (inputs + expected
outputs = data, no reference text).

Weights are the hash-filled state_dict of ``sahs-deformable-nerf_amd/weights.py`` and are NOT
stored: consumers regenerate them from (seed, density_bias, density_gain) kept in each file.

usage:  python tests/golden/make_golden.py            (writes tests/golden/*.npz)
"""
import importlib
import os
import sys
import types

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/nerf-pytorch"
sys.path.insert(0, REPO)


def import_reference():
    if not os.path.isdir(REF + "/nerf"):
        raise SystemExit("reference not present at %s: golden vectors can only be regenerated in the build container" % REF)
    pkg = types.ModuleType("nerf")
    pkg.__path__ = [REF + "/nerf"]
    sys.modules["nerf"] = pkg
    for n in ("pytorch3d", "pytorch3d.transforms"):
        sys.modules[n] = types.ModuleType(n)
    names = ("cfgnode", "nerf_helpers", "volume_rendering_utils", "train_utils", "modules", "models")
    return types.SimpleNamespace(**{n: importlib.import_module("nerf." + n) for n in names})


class RandCapture:
    """Record every torch.rand / torch.randn result in draw order."""

    def __enter__(self):
        self.log = []
        self._rand, self._randn = torch.rand, torch.randn

        def rand(*a, **k):
            t = self._rand(*a, **k)
            self.log.append(("rand", t.detach().clone().numpy()))
            return t

        def randn(*a, **k):
            t = self._randn(*a, **k)
            self.log.append(("randn", t.detach().clone().numpy()))
            return t

        torch.rand, torch.randn = rand, randn
        return self

    def __exit__(self, *exc):
        torch.rand, torch.randn = self._rand, self._randn


class RandFeed:
    """Replay a captured draw log (in order) to torch.rand / torch.randn, cast to the dtype the caller asks for: the float64
    yardstick runs consume exactly the draws of the fp32 run they are compared with."""

    def __init__(self, log):
        self.log = list(log)

    def __enter__(self):
        self._rand, self._randn = torch.rand, torch.randn

        def make(kind):
            def f(*a, **k):
                knd, arr = self.log.pop(0)
                assert knd == kind, (knd, kind)
                return torch.from_numpy(arr).to(k.get("dtype", torch.float32))
            return f

        torch.rand, torch.randn = make("rand"), make("randn")
        return self

    def __exit__(self, *exc):
        torch.rand, torch.randn = self._rand, self._randn
        assert not self.log, "yardstick run consumed fewer draws than the fp32 run"


class Float64Yardstick:
    """Context for running the reference MODEL IN FLOAT64 (``model.double()``, double inputs) as the accuracy yardstick: how far
    the reference's own fp32 result is from the exact value of the same formulas.  The one fp32 cast on the path,
    ``coordinates.float()`` in sample_from_3dgrid (models.py:355), is made the identity on double tensors for the duration
    (harness-level patch of Tensor.float, no reference edits), so the yardstick is float64 end to end."""

    def __enter__(self):
        self._float = torch.Tensor.float
        orig = self._float
        torch.Tensor.float = lambda t, *a, **k: t if t.dtype == torch.float64 else orig(t, *a, **k)
        return self

    def __exit__(self, *exc):
        torch.Tensor.float = self._float


def rot(ax, ay, az):
    cx, sx, cy, sy, cz, sz = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az)
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return (rz @ ry @ rx).astype(np.float32)


def main():
    torch.set_num_threads(8)
    ref = import_reference()
    W = importlib.import_module("sahs-deformable-nerf_amd.weights")
    cfg = ref.cfgnode.CfgNode(yaml.load(open(REF + "/config/audio/person_2_auto.yml"), Loader=yaml.SafeLoader))
    rng = np.random.default_rng(20241008)

    def build_model(**kw):
        m = ref.models.AudioFaceModel(cfg)
        sd = W.hash_state_dict(**kw)
        ref_sd = m.state_dict()
        assert list(ref_sd.keys()) == [k for k, _ in W.canonical_spec()], "state_dict order differs from canonical_spec"
        for k, s in W.canonical_spec():
            assert tuple(ref_sd[k].shape) == tuple(s), (k, ref_sd[k].shape, s)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        return m.eval()

    VARIANTS = {"default": dict(seed=0, density_bias=0.0, density_gain=1.0, hdr=False),
                "boosted": dict(seed=0, density_bias=8.0, density_gain=30.0, hdr=False),
                "hdr": dict(seed=0, density_bias=2.0, density_gain=30.0, hdr=True)}   # high dynamic range (weights.hash_state_dict)
    models = {k: build_model(**v) for k, v in VARIANTS.items()}
    models64 = {k: build_model(**v).double() for k, v in VARIANTS.items()}              # float64 yardsticks
    out = {}

    # ---- rays: get_ray_bundle (nerf_helpers.py:178-233) ----
    H, Wd = 6, 8
    intr = np.array([1200.0, 1100.0, 0.45, 0.55], np.float32)
    c2w = np.concatenate([rot(0.1, -0.2, 0.05), np.array([[0.02], [-0.03], [0.8]], np.float32)], axis=1)
    ro, rd = ref.nerf_helpers.get_ray_bundle(H, Wd, intr, torch.from_numpy(c2w))
    out["rays"] = dict(H=H, W=Wd, intrinsics=intr, c2w=c2w, ro=ro.numpy(), rd=rd.numpy())

    # ---- conditioning: AudioNet (modules.py:43-73), pose encoding (models.py:482-504,203-207) ----
    audio = rng.standard_normal((16, 29)).astype(np.float32)
    pose = np.concatenate([rot(0.3, 0.2, -0.4), np.array([[0.01], [0.02], [0.8]], np.float32)], axis=1)
    m = models["default"]
    with torch.no_grad():
        driving = m.audNet_head(torch.from_numpy(audio).unsqueeze(0))
        p6 = ref.models.pose_to_euler_trans(torch.from_numpy(pose).unsqueeze(0), "cpu")
        pose36 = m.encode_pose_fn(p6)
    out["cond"] = dict(audio=audio, pose=pose, driving=driving.numpy(), pose6=p6.numpy()[0], pose36=pose36.numpy()[0])

    # ---- positional_encoding (nerf_helpers.py:305-349) ----
    xs = (rng.standard_normal((5, 3)) * 0.4).astype(np.float32)
    ws = (rng.standard_normal((5, 2)) * 0.7).astype(np.float32)
    out["pe"] = dict(x=xs, w=ws,
                     pe_xyz=ref.nerf_helpers.positional_encoding(torch.from_numpy(xs), 10, True, True).numpy(),
                     pe_dir=ref.nerf_helpers.positional_encoding(torch.from_numpy(xs), 4, True, True).numpy(),
                     pe_amb=ref.nerf_helpers.positional_encoding(torch.from_numpy(ws), 4, True, True).numpy(),
                     pe_pose=ref.nerf_helpers.positional_encoding(torch.from_numpy(xs), 3, False, True).numpy())

    # ---- field: AudioFaceModel.forward (models.py:514-528) with seam captures ----
    P = 256  # multiple of 64 and 128 (models.py:353-358 reshapes to (-1, n_coords, 3))
    xyz = (rng.uniform(-0.35, 0.35, (P, 3))).astype(np.float32)
    xyz[200:232] *= 3.2   # some points outside [-1,1]: zero padding of grid_sample
    xyz[232:240, 0] = 1.0  # exactly on the boundary
    dirs = (rng.standard_normal((P, 3)) * 0.3 + np.array([0, 0, -1.0])).astype(np.float32)
    x18 = np.concatenate([xyz, dirs, np.zeros((P, 12), np.float32)], axis=1)
    fld = dict(x=x18[:, :6].copy(), audio=audio, pose=pose)
    for vname, m in models.items():
        with torch.no_grad():
            xt = torch.from_numpy(x18)
            drv = m.audNet_head(torch.from_numpy(audio).unsqueeze(0)).repeat(P, 1)
            pe36 = m.encode_pose_fn(ref.models.pose_to_euler_trans(torch.from_numpy(pose).unsqueeze(0), "cpu")).repeat(P, 1)
            mapped = m.map_points(xt[:, :3], drv, pe36)
            fld[vname + "_dx"] = (mapped[:, :3] - xt[:, :3]).numpy()
            fld[vname + "_warped"] = mapped[:, :3].numpy()
            fld[vname + "_w"] = mapped[:, 3:].numpy()
            fld[vname + "_grid_coarse"] = m.sample_from_3dgrid("coarse", mapped[..., :3]).numpy()
            for lvl in ("coarse", "fine"):
                fld[vname + "_raw_" + lvl] = m(lvl, xt, torch.from_numpy(audio), torch.from_numpy(pose), None).numpy()
        # the same seams from the reference model in float64 (yardstick: |ref_fp32 - f64| is the reference's own round-off)
        m64 = models64[vname]
        with torch.no_grad(), Float64Yardstick():
            xd, ad, pd = torch.from_numpy(x18).double(), torch.from_numpy(audio).double(), torch.from_numpy(pose).double()
            drv = m64.audNet_head(ad.unsqueeze(0)).repeat(P, 1)
            pe36 = m64.encode_pose_fn(ref.models.pose_to_euler_trans(pd.unsqueeze(0), "cpu")).repeat(P, 1)
            assert drv.dtype == torch.float64 and pe36.dtype == torch.float64
            mapped = m64.map_points(xd[:, :3], drv, pe36)
            fld[vname + "_dx_f64"] = (mapped[:, :3] - xd[:, :3]).numpy()
            fld[vname + "_w_f64"] = mapped[:, 3:].numpy()
            g64 = m64.sample_from_3dgrid("coarse", mapped[..., :3])
            assert g64.dtype == torch.float64
            fld[vname + "_grid_coarse_f64"] = g64.numpy()
            for lvl in ("coarse", "fine"):
                r64 = m64(lvl, xd, ad, pd, None)
                assert r64.dtype == torch.float64
                fld[vname + "_raw_" + lvl + "_f64"] = r64.numpy()
    out["field"] = fld

    # ---- composite: volume_render_radiance_field (volume_rendering_utils.py:7-78) ----
    N, S = 40, 64
    raw = (rng.standard_normal((N, S, 16)) * 1.5).astype(np.float32)
    raw[..., 15] = raw[..., 15] * 8.0 + 2.0
    raw[:4, :, 15] = -5.0  # rays with zero density everywhere: all weight on the background sample
    z = np.sort(rng.uniform(0.48, 1.08, (N, S)).astype(np.float32), axis=1)
    z[5, 10] = z[5, 11]  # a zero-length interval
    rdc = (rng.standard_normal((N, 3)) * 0.2 + np.array([0, 0, -1.0])).astype(np.float32)
    bgc = np.concatenate([rng.uniform(0, 1, (N, 3)), np.ones((N, 1)), np.zeros((N, 11))], axis=1).astype(np.float32)
    noise = (rng.standard_normal((N, S)) * 0.1).astype(np.float32)
    comp = dict(raw=raw, z=z, rd=rdc, bg=bgc, noise=noise)

    def run_comp(tag, use_bg, use_noise, white):
        rf = torch.from_numpy(raw.copy())
        bg_t = torch.from_numpy(bgc) if use_bg else None
        if use_bg:
            rf[:, -1, :-1] = bg_t  # train_utils.py:135-136
        if use_noise:
            orig = torch.randn
            torch.randn = lambda *a, **k: torch.from_numpy(noise / 0.1)
            try:
                r = ref.volume_rendering_utils.volume_render_radiance_field(rf, torch.from_numpy(z), torch.from_numpy(rdc), 0.1, white, bg_t)
            finally:
                torch.randn = orig
        else:
            r = ref.volume_rendering_utils.volume_render_radiance_field(rf, torch.from_numpy(z), torch.from_numpy(rdc), 0.0, white, bg_t)
        for nm, t in zip(("rgb", "disp", "acc", "weights", "depth"), r):
            comp[tag + "_" + nm] = t.numpy()

    run_comp("bg", True, False, False)
    run_comp("bg_noise", True, True, False)
    run_comp("nobg", False, False, False)
    run_comp("nobg_white", False, False, True)
    out["composite"] = comp

    # ---- sample_pdf_2 (nerf_helpers.py:454-497) + cat/sort (train_utils.py:157-166) ----
    Np = 160
    zc = np.sort(rng.uniform(0.48, 1.08, (Np, 64)).astype(np.float32), axis=1)
    wts = rng.uniform(0, 1, (Np, 64)).astype(np.float32) ** 8
    wts[:16] = 0.0              # flat pdf (all weight would be 1e-5)
    wts[16:32, 20] = 1.0        # a spike
    wts[16:32, :20] = 0.0
    wts[16:32, 21:] = 0.0
    u = rng.uniform(0, 1, (Np, 64)).astype(np.float32)
    u[0, 0] = 0.0
    bins = 0.5 * (zc[:, 1:] + zc[:, :-1])
    pdf = dict(z=zc, weights=wts, u=u)

    def ref_pdf(tag, det):
        orig = torch.rand
        torch.rand = lambda *a, **k: torch.from_numpy(u)
        try:
            s = ref.nerf_helpers.sample_pdf_2(torch.from_numpy(bins), torch.from_numpy(wts[:, 1:-1]), 64, det=det)
        finally:
            torch.rand = orig
        # inds as the reference computes them (nerf_helpers.py:459-482), re-derived here for the fixture
        w = torch.from_numpy(wts[:, 1:-1]) + 1e-5
        p = w / torch.sum(w, dim=-1, keepdim=True)
        cdf = torch.cat([torch.zeros_like(p[..., :1]), torch.cumsum(p, dim=-1)], dim=-1)
        uu = torch.linspace(0.0, 1.0, steps=64).expand(Np, 64) if det else torch.from_numpy(u)
        inds = torch.searchsorted(cdf.contiguous(), uu.contiguous(), right=True)
        zs, _ = torch.sort(torch.cat((torch.from_numpy(zc), s), dim=-1), dim=-1)
        pdf[tag + "_samples"] = s.numpy()
        pdf[tag + "_inds"] = inds.numpy()
        pdf[tag + "_cdf"] = cdf.numpy()
        pdf[tag + "_z_sorted"] = zs.numpy()

    ref_pdf("rand", False)
    ref_pdf("det", True)
    out["pdf"] = pdf

    # ---- end to end: run_one_iter_of_nerf (train_utils.py:209-321) ----
    He, We = 12, 12
    intr_e = np.array([1200.0 * 12 / 512, 1200.0 * 12 / 512, 0.5, 0.5], np.float32)
    pose_e = np.concatenate([np.eye(3, dtype=np.float32), np.array([[0.0], [0.0], [0.8]], np.float32)], axis=1)
    ro_e, rd_e = ref.nerf_helpers.get_ray_bundle(He, We, intr_e, torch.from_numpy(pose_e))
    bg_e = np.concatenate([rng.uniform(0, 1, (He * We, 3)), np.ones((He * We, 1)), np.zeros((He * We, 11))], axis=1).astype(np.float32)
    mask_e = np.zeros((He, We, 12), np.float32)
    mask_e[..., 0] = 1.0

    def e2e(tag, vname, mode, chunksize, perturb, noise_std):
        model = models[vname]
        c = cfg.clone() if hasattr(cfg, "clone") else cfg
        node = getattr(c.nerf, mode)
        old = (node.chunksize, node.perturb, node.radiance_field_noise_std)
        node.chunksize, node.perturb, node.radiance_field_noise_std = chunksize, perturb, noise_std
        torch.manual_seed(42)
        try:
            with torch.no_grad(), RandCapture() as cap:
                r = ref.train_utils.run_one_iter_of_nerf(
                    He, We, intr_e, model, ro_e, rd_e, c, mode=mode, driving=torch.from_numpy(audio),
                    pose=torch.from_numpy(pose_e), pose_c=None, background_prior=torch.from_numpy(bg_e),
                    latent_code=None, inHead=torch.from_numpy(mask_e))
            # float64 yardstick: the reference driver over the float64 model, double rays, the SAME draws
            with torch.no_grad(), RandFeed(cap.log), Float64Yardstick():
                r64 = ref.train_utils.run_one_iter_of_nerf(
                    He, We, intr_e, models64[vname], ro_e.double(), rd_e.double(), c, mode=mode, driving=torch.from_numpy(audio).double(),
                    pose=torch.from_numpy(pose_e).double(), pose_c=None, background_prior=torch.from_numpy(bg_e).double(),
                    latent_code=None, inHead=torch.from_numpy(mask_e).double())
            assert all(t.dtype == torch.float64 for t in r64)
        finally:
            node.chunksize, node.perturb, node.radiance_field_noise_std = old
        d = dict(H=He, W=We, intrinsics=intr_e, pose=pose_e, audio=audio, bg=bg_e, mode=mode, chunksize=chunksize,
                 perturb=perturb, noise_std=noise_std, near=cfg.dataset.near, far=cfg.dataset.far,
                 num_coarse=node.num_coarse, num_fine=node.num_fine, ro=ro_e.numpy(), rd=rd_e.numpy())
        for nm, t in zip(("rgb_c", "disp_c", "acc_c", "rgb_f", "disp_f", "acc_f", "w_bg", "depth_f"), r):
            d["out_" + nm] = t.numpy()
        for nm, t in zip(("rgb_c", "disp_c", "acc_c", "rgb_f", "disp_f", "acc_f", "w_bg", "depth_f"), r64):
            d["f64_" + nm] = t.numpy()
        for i, (kind, arr) in enumerate(cap.log):
            d["rand_%02d_%s" % (i, kind)] = arr
        d["n_rand"] = len(cap.log)
        d["variant"] = vname
        out[tag] = d

    e2e("e2e_default_val", "default", "validation", 131072, True, 0.0)
    e2e("e2e_boosted_val", "boosted", "validation", 131072, True, 0.0)
    e2e("e2e_boosted_val_2chunks", "boosted", "validation", 128, True, 0.0)
    e2e("e2e_boosted_det", "boosted", "validation", 131072, False, 0.0)
    e2e("e2e_boosted_train_noise", "boosted", "train", 131072, True, 0.1)
    e2e("e2e_hdr_val", "hdr", "validation", 131072, True, 0.0)
    e2e("e2e_hdr_det", "hdr", "validation", 131072, False, 0.0)
    e2e("e2e_hdr_train_noise", "hdr", "train", 131072, True, 0.1)

    # ---- gradients (config[4] semantics): train mode, 32 rays, relu->relu.clone() harness shim ----
    F = torch.nn.functional
    orig_relu = F.relu
    F.relu = lambda x, inplace=False: orig_relu(x).clone()
    try:
        mg = build_model(**VARIANTS["boosted"]).train()
        # WarpFieldMLP/HyperSheetMLP captured F.relu at construction (modules.py:367,441): rebuilt above under the shim
        nr = 32
        sel = rng.choice(He * We, nr, replace=False)
        ro_g = ro_e.reshape(-1, 3)[sel]
        rd_g = rd_e.reshape(-1, 3)[sel]
        bg_g = bg_e[sel]
        mask_g = mask_e.reshape(-1, 12)[sel]
        A = rng.standard_normal((nr, 15)).astype(np.float32)
        B = rng.standard_normal((nr, 15)).astype(np.float32)
        c = cfg
        torch.manual_seed(7)
        audio_t = torch.from_numpy(audio).requires_grad_(True)
        with RandCapture() as cap:
            r = ref.train_utils.run_one_iter_of_nerf(
                He, We, intr_e, mg, ro_g, rd_g, c, mode="train", driving=audio_t, pose=torch.from_numpy(pose_e), pose_c=None,
                background_prior=torch.from_numpy(bg_g), latent_code=None, inHead=torch.from_numpy(mask_g))
        loss = (r[0] * torch.from_numpy(A)).sum() + (r[3] * torch.from_numpy(B)).sum() + r[7].sum() * 0.1
        loss.backward()
        g = dict(sel=sel, ro=ro_g.numpy(), rd=rd_g.numpy(), bg=bg_g, A=A, B=B, audio=audio, pose=pose_e, loss=loss.item(),
                 near=cfg.dataset.near, far=cfg.dataset.far)
        for nm, t in zip(("rgb_c", "disp_c", "acc_c", "rgb_f", "disp_f", "acc_f", "w_bg", "depth_f"), r):
            g["out_" + nm] = t.detach().numpy()
        for i, (kind, arr) in enumerate(cap.log):
            g["rand_%02d_%s" % (i, kind)] = arr
        g["n_rand"] = len(cap.log)
        norms = []
        for k, p in mg.named_parameters():
            gr = p.grad if p.grad is not None else torch.zeros_like(p)
            norms.append(float(gr.double().norm()))
            if p.numel() <= 4096 or k.endswith("bias"):
                g["grad_" + k] = gr.numpy()
            else:
                g["gradsub_" + k] = gr.reshape(-1)[::max(1, p.numel() // 2048)].numpy().copy()
        g["grad_norms"] = np.array(norms, np.float64)
        g["grad_names"] = np.array([k for k, _ in mg.named_parameters()])
        g["grad_audio"] = audio_t.grad.numpy()
        out["train_grads"] = g

        # ---- the reference's loss classes (nerf_helpers.py:14-62), weights as the training script builds them
        #      (train_stage_rays_auto.py:268-271: ones(12), [7:9] = 2); class 5 is empty, class 11 has one pixel ----
        Nl = 200
        cls = rng.integers(0, 12, Nl)
        cls[cls == 5] = 4
        cls[cls == 11] = 10
        cls[0] = 11
        maskl = np.eye(12, dtype=np.float32)[cls]
        pred_rgb = rng.uniform(0, 1, (Nl, 3)).astype(np.float32)
        tgt_rgb = rng.uniform(0, 1, (Nl, 3)).astype(np.float32)
        pred_seg = rng.dirichlet(np.ones(12) * 0.3, Nl).astype(np.float32)
        pred_seg[3] = 0.0                      # a pixel whose predicted distribution is all zeros: log(0 + 1e-10)
        spw = torch.ones(12)
        spw[7:9] = 2
        mse_l, ce_l = ref.nerf_helpers.MaskMSELoss(spw.clone()), ref.nerf_helpers.MaskCrossEntropyLoss(spw.clone())
        lm = mse_l(torch.from_numpy(maskl), torch.from_numpy(pred_rgb), torch.from_numpy(tgt_rgb))
        lc = ce_l(torch.from_numpy(maskl), torch.from_numpy(pred_seg), torch.from_numpy(maskl))
        lm0 = ref.nerf_helpers.MaskMSELoss()(torch.from_numpy(maskl), torch.from_numpy(pred_rgb), torch.from_numpy(tgt_rgb))
        out["losses"] = dict(mask=maskl, pred_rgb=pred_rgb, target_rgb=tgt_rgb, pred_seg=pred_seg, weights=spw.numpy(),
                             mse=lm[0].numpy(), mse_masked=lm[1].numpy(), mse_weighted=lm[2].numpy(),
                             ce=lc[0].numpy(), ce_masked=lc[1].numpy(), ce_weighted=lc[2].numpy(),
                             mse_weighted_noweights=lm0[2].numpy())

        # ---- one training step of train_stage_rays_auto.py:437-468 (loss recipe + sample_prob feedback) over the reference's
        #      own renders, 32 rays, train mode (noise 0.1), hdr weights; fp32 and the float64 yardstick on the same draws ----
        def train_step_capture(model, dt, log=None):
            cls_t = rng_t.integers(0, 12, nr)
            cls_t[cls_t == 5] = 4
            mask_t = torch.from_numpy(np.eye(12, dtype=np.float32)[cls_t]).to(dt)
            target = torch.from_numpy(tgt_t).to(dt)
            a_t = torch.from_numpy(audio).to(dt).requires_grad_(True)
            w12 = torch.ones(12, dtype=dt)
            w12[7:9] = 2
            mse_loss, cross_entropy_loss = ref.nerf_helpers.MaskMSELoss(w12.clone()), ref.nerf_helpers.MaskCrossEntropyLoss(w12.clone())
            ctx = RandCapture() if log is None else RandFeed(log)
            with ctx as cap:
                rr = ref.train_utils.run_one_iter_of_nerf(
                    He, We, intr_e, model, ro_g.to(dt), rd_g.to(dt), cfg, mode="train", driving=a_t, pose=torch.from_numpy(pose_e).to(dt),
                    pose_c=None, background_prior=torch.from_numpy(bg_g).to(dt), latent_code=None, inHead=mask_t)
            rgb_coarse, rgb_fine = rr[0], rr[3]
            # train_stage_rays_auto.py:455-468, restated over the reference's loss classes
            c_l2, m_c_l2, m_c_l2_w = mse_loss(mask_t, rgb_coarse[..., :3], target[..., :3])
            c_ce, m_c_ce, m_c_ce_w = cross_entropy_loss(mask_t, rgb_coarse[..., 3:], mask_t)
            c_loss = c_l2 + 0.02 * c_ce + 0.005 * torch.sum(m_c_l2[7:9] + m_c_ce[7:9])
            f_l2, m_f_l2, m_f_l2_w = mse_loss(mask_t, rgb_fine[..., :3], target[..., :3])
            f_ce, m_f_ce, m_f_ce_w = cross_entropy_loss(mask_t, rgb_fine[..., 3:], mask_t)
            f_loss = f_l2 + 0.02 * f_ce + 0.005 * torch.sum(m_f_l2[7:9] + m_f_ce[7:9])
            sample_prob = (m_c_l2_w + m_c_ce_w + m_f_l2_w + m_f_ce_w) / (m_c_l2_w.sum() + m_c_ce_w.sum() + m_f_l2_w.sum() + m_f_ce_w.sum())
            loss = c_loss + f_loss
            loss.backward()
            return dict(cap=cap, outs=rr, loss=loss, sample_prob=sample_prob, parts=(c_l2, c_ce, f_l2, f_ce), audio_grad=a_t.grad,
                        mask=mask_t, cls=cls_t)

        rng_t = np.random.default_rng(77)
        tgt_t = np.random.default_rng(78).uniform(0, 1, (nr, 3)).astype(np.float32)
        mt = build_model(**VARIANTS["hdr"]).train()
        torch.manual_seed(11)
        t32 = train_step_capture(mt, torch.float32)
        rng_t = np.random.default_rng(77)
        mt64 = build_model(**VARIANTS["hdr"]).double().train()
        with Float64Yardstick():
            t64 = train_step_capture(mt64, torch.float64, log=t32["cap"].log)
        ts = dict(sel=sel, ro=ro_g.numpy(), rd=rd_g.numpy(), bg=bg_g, audio=audio, pose=pose_e, target=tgt_t, mask=t32["mask"].numpy(),
                  near=cfg.dataset.near, far=cfg.dataset.far, loss=t32["loss"].item(), loss_f64=t64["loss"].item(),
                  sample_prob=t32["sample_prob"].detach().numpy(), sample_prob_f64=t64["sample_prob"].detach().numpy(),
                  parts=np.array([float(x) for x in t32["parts"]]), parts_f64=np.array([float(x) for x in t64["parts"]]),
                  grad_audio=t32["audio_grad"].numpy(), grad_audio_f64=t64["audio_grad"].numpy())
        for nm, t, t6 in zip(("rgb_c", "disp_c", "acc_c", "rgb_f", "disp_f", "acc_f", "w_bg", "depth_f"), t32["outs"], t64["outs"]):
            ts["out_" + nm] = t.detach().numpy()
            ts["f64_" + nm] = t6.detach().numpy()
        for i, (kind, arr) in enumerate(t32["cap"].log):
            ts["rand_%02d_%s" % (i, kind)] = arr
        ts["n_rand"] = len(t32["cap"].log)
        norms, norms64 = [], []
        for (k, p_), (_, p6) in zip(mt.named_parameters(), mt64.named_parameters()):
            gr = p_.grad if p_.grad is not None else torch.zeros_like(p_)
            g6 = p6.grad if p6.grad is not None else torch.zeros_like(p6)
            norms.append(float(gr.double().norm()))
            norms64.append(float(g6.norm()))
            if p_.numel() <= 4096 or k.endswith("bias"):
                ts["grad_" + k], ts["grad64_" + k] = gr.numpy(), g6.numpy()
            else:
                st = max(1, p_.numel() // 2048)
                ts["gradsub_" + k], ts["gradsub64_" + k] = gr.reshape(-1)[::st].numpy().copy(), g6.reshape(-1)[::st].numpy().copy()
        ts["grad_norms"], ts["grad_norms_f64"] = np.array(norms, np.float64), np.array(norms64, np.float64)
        ts["grad_names"] = np.array([k for k, _ in mt.named_parameters()])
        out["train_step_hdr"] = ts
    finally:
        F.relu = orig_relu

    # ---- eval post-processing (eval_stage_rays.py:116-151 torch_normal_map; nerf/utils.py:112-140 label2color, :5-66 shrink) ----
    # The eval script imports imageio/torchvision (absent here) and names from nerf/__init__; neither is touched by
    # torch_normal_map, so empty stand-in modules / None names are enough to import it (same recipe as the pytorch3d stub).
    for n in ("imageio", "torchvision"):
        if n not in sys.modules:
            sys.modules[n] = types.ModuleType(n)
    nerf_pkg = sys.modules["nerf"]
    nerf_pkg.utils = importlib.import_module("nerf.utils")
    for nm, val in (("CfgNode", ref.cfgnode.CfgNode), ("get_ray_bundle", ref.nerf_helpers.get_ray_bundle),
                    ("get_ray_bundle_by_mask", ref.nerf_helpers.get_ray_bundle_by_mask), ("load_flame_data", None), ("load_llff_data", None),
                    ("models", ref.models), ("get_embedding_function", ref.nerf_helpers.get_embedding_function),
                    ("run_one_iter_of_nerf", ref.train_utils.run_one_iter_of_nerf), ("meshgrid_xy", ref.nerf_helpers.meshgrid_xy)):
        setattr(nerf_pkg, nm, val)
    sys.path.insert(0, REF)
    ev = importlib.import_module("eval_stage_rays")
    Hn = 24
    disp = torch.from_numpy((1.0 / rng.uniform(0.5, 1.0, (Hn, Hn))).astype(np.float32))
    wbg = torch.from_numpy(rng.uniform(0, 0.5, (Hn, Hn)).astype(np.float32))
    focal = np.array([1200.0 * Hn / 512, 1150.0 * Hn / 512, 0.48, 0.52], np.float32)
    seg = torch.from_numpy(rng.standard_normal((Hn, Hn, 12)).astype(np.float32))
    onehot = np.eye(12, dtype=np.float32)[rng.integers(0, 12, (Hn, Hn))]
    out["post"] = dict(disp=disp.numpy(), w_bg=wbg.numpy(), focal=focal, seg=seg.numpy(), onehot=onehot,
                       normals_clean=ev.torch_normal_map(disp.clone(), focal, wbg.clone(), clean=True).numpy(),
                       normals_raw=ev.torch_normal_map(disp.clone(), focal, None, clean=False).numpy(),
                       normals_central=ev.torch_normal_map(disp.clone(), focal, wbg.clone(), clean=True, central_difference=True).numpy(),
                       seg_color=nerf_pkg.utils.label2color(seg).numpy(), shrink=nerf_pkg.utils.shrink(onehot),
                       disp_img=ev.cast_to_disparity_image(disp))

    for name, d in out.items():
        d = dict(d)
        if name.startswith("e2e") or name.startswith("train_"):
            v = "default" if "default" in name else ("hdr" if "hdr" in name else "boosted")
            d.update({"weights_" + k: val for k, val in VARIANTS[v].items()})
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
        print("%-28s %8.1f KB" % (name + ".npz", os.path.getsize(os.path.join(HERE, name + ".npz")) / 1024))


if __name__ == "__main__":
    main()
