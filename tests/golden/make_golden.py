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

    VARIANTS = {"default": dict(seed=0, density_bias=0.0, density_gain=1.0),
                "boosted": dict(seed=0, density_bias=8.0, density_gain=30.0)}
    models = {k: build_model(**v) for k, v in VARIANTS.items()}
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

    def e2e(tag, model, mode, chunksize, perturb, noise_std):
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
        finally:
            node.chunksize, node.perturb, node.radiance_field_noise_std = old
        d = dict(H=He, W=We, intrinsics=intr_e, pose=pose_e, audio=audio, bg=bg_e, mode=mode, chunksize=chunksize,
                 perturb=perturb, noise_std=noise_std, near=cfg.dataset.near, far=cfg.dataset.far,
                 num_coarse=node.num_coarse, num_fine=node.num_fine, ro=ro_e.numpy(), rd=rd_e.numpy())
        for nm, t in zip(("rgb_c", "disp_c", "acc_c", "rgb_f", "disp_f", "acc_f", "w_bg", "depth_f"), r):
            d["out_" + nm] = t.numpy()
        for i, (kind, arr) in enumerate(cap.log):
            d["rand_%02d_%s" % (i, kind)] = arr
        d["n_rand"] = len(cap.log)
        out[tag] = d

    e2e("e2e_default_val", models["default"], "validation", 131072, True, 0.0)
    e2e("e2e_boosted_val", models["boosted"], "validation", 131072, True, 0.0)
    e2e("e2e_boosted_val_2chunks", models["boosted"], "validation", 128, True, 0.0)
    e2e("e2e_boosted_det", models["boosted"], "validation", 131072, False, 0.0)
    e2e("e2e_boosted_train_noise", models["boosted"], "train", 131072, True, 0.1)

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
        if name.startswith("e2e") or name == "train_grads":
            v = "default" if "default" in name else "boosted"
            d.update({"weights_" + k: val for k, val in VARIANTS[v].items()})
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
        print("%-28s %8.1f KB" % (name + ".npz", os.path.getsize(os.path.join(HERE, name + ".npz")) / 1024))


if __name__ == "__main__":
    main()
