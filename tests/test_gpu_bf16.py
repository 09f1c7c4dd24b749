"""bf16 MFMA path (BASELINE.json configs[2]) on the MI355X: error against the CPU oracle at the field seam and
the PSNR protocol of SURVEY.md section 8d on the full W512 frame, with the exact fp32 HIP path (itself checked against
the oracle and the reference's golden outputs in test_gpu_parity.py) as the image reference.
bf16 tolerances are statistical: every layer rounds its activations to 8 significant bits."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import REPO, load_golden, pkg

pytestmark = pytest.mark.gpu
from oracle import oracle  # noqa: E402  (checker only)


def dev():
    return torch.device("cuda:0")


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def psnr(a, b):
    mse = float(torch.mean((a.clamp(0, 1) - b.clamp(0, 1)) ** 2))
    return 99.0 if mse == 0 else -10.0 * np.log10(mse)


# per-variant bounds on the bf16 field error against the fp32 oracle: 3x the observed values (printed by the test; round 2, hdr: colour
# logit rms 0.027, density rms 0.19 on a mean magnitude of ~10): dx max, rgb/seg logit rms, density logit rms relative to its mean magnitude
BF16_BOUNDS = {"boosted": dict(dx_max=2e-3, col_rms=3e-4, sig_rel=5e-3), "hdr": dict(dx_max=2e-3, col_rms=8.1e-2, sig_rel=6e-2)}


@pytest.mark.parametrize("variant", ["boosted", "hdr"])
def test_bf16_field_vs_oracle(flat_weights, variant):
    from conftest import VARIANT_KW
    ops, lib = pkg("ops"), pkg("_lib")
    g = load_golden("cond")
    fw = flat_weights(**VARIANT_KW[variant])
    flat = T(fw)
    packed = ops.pack_weights(flat, lib.SAHS_BF16)
    frame = ops.fold_conditioning(flat, T(g["audio"]), T(g["pose"]))
    rng = np.random.default_rng(11)
    N, S = 61, 64     # ragged: 3904 samples, not a multiple of the 256-sample tile
    rays = np.zeros((N, 8), np.float32)
    rays[:, 0:3] = rng.normal(0, 0.05, (N, 3)) + np.array([0, 0, 0.8])
    rays[:, 3:6] = rng.normal(0, 0.15, (N, 3)) + np.array([0, 0, -1.0])
    z = np.sort(rng.uniform(0.48, 1.08, (N, S)).astype(np.float32), axis=1)
    x6 = np.concatenate([rays[:, None, 0:3] + rays[:, None, 3:6] * z[..., None], np.broadcast_to(rays[:, None, 3:6], (N, S, 3))], -1).reshape(-1, 6)
    drv, p36 = oracle.audionet(fw, g["audio"]), oracle.pose_encoding(g["pose"])
    stats = {}
    for level in (0, 1):
        ref, rdx, rw, rgrid = oracle.field_forward(fw, level, x6.astype(np.float32), drv, p36, debug=True)
        raw, dx, w, grid = ops.field_forward(packed, frame, level, T(rays), T(z), precision=lib.SAHS_BF16, debug=True)
        raw = raw.view(-1, 16).cpu().numpy()
        e_dx = np.abs(dx.view(-1, 3).cpu().numpy() - rdx)
        e_w = np.abs(w.view(-1, 2).cpu().numpy() - rw)
        e_col = np.abs(raw[:, :15] - ref[:, :15])
        e_sig = np.abs(raw[:, 15] - ref[:, 15])
        stats[level] = dict(dx_max=float(e_dx.max()), dx_rms=float(np.sqrt((e_dx ** 2).mean())), w_max=float(e_w.max()),
                            col_max=float(e_col.max()), col_rms=float(np.sqrt((e_col ** 2).mean())), sig_max=float(e_sig.max()),
                            sig_rms=float(np.sqrt((e_sig ** 2).mean())), sig_scale=float(np.abs(ref[:, 15]).mean()),
                            grid_max=float(np.abs(grid.view(-1, 32).cpu().numpy() - rgrid).max()))
        stats[level]["col_scale"] = float(np.sqrt((ref[:, :15] ** 2).mean()))
        assert np.isfinite(raw).all()
    print(variant, json.dumps(stats))
    b = BF16_BOUNDS[variant]
    for level in (0, 1):
        st = stats[level]
        assert st["dx_max"] < b["dx_max"] and st["col_rms"] < b["col_rms"], (variant, st)
        assert st["sig_rms"] < b["sig_rel"] * st["sig_scale"], (variant, st)


def test_bf16_full_frame_psnr(weights_mod):
    sahs = pkg()
    cfg = sahs.default_config()
    H = W = 512
    R = H * W
    # the HIGH-DYNAMIC-RANGE network (O(1) activations, logits of sigma ~2.5, semi-transparent volume): on default-scale weights
    # every logit is < 0.1 and a bf16-vs-fp32 PSNR of 100 dB says nothing.  Pseudo-target: the SAME network
    # rendered in fp32 under other random draws.
    fw = weights_mod.flatten_state_dict(weights_mod.hash_state_dict(0, 2.0, 30.0, hdr=True))
    m32 = sahs.AudioFaceModel(cfg, precision="fp32").to(dev()).load_flat(fw)
    m16 = sahs.AudioFaceModel(cfg, precision="bf16").to(dev()).load_flat(fw)
    rng = np.random.default_rng(42)
    audio, pose = T(rng.standard_normal((16, 29)).astype(np.float32)), T(np.concatenate([np.eye(3), [[0.0], [0.0], [0.8]]], 1).astype(np.float32))
    intr = np.array([1200.0, 1200.0, 0.5, 0.5], np.float32)
    bg = torch.cat([torch.rand(R, 3, device=dev()), torch.ones(R, 1, device=dev()), torch.zeros(R, 11, device=dev())], 1)
    ro, rd = sahs.get_ray_bundle(H, W, intr, pose)

    def render(model, seed=7):
        torch.manual_seed(seed)   # same random draws for the two kernels
        with torch.no_grad():
            return sahs.run_one_iter_of_nerf(H, W, intr, model, ro, rd, cfg, mode="validation", driving=audio, pose=pose, background_prior=bg)

    o32, o16, ot = render(m32), render(m16), render(m32, seed=8)
    rgb32, rgb16, rgbt = o32[3][..., :3], o16[3][..., :3], ot[3][..., :3]
    res = dict(psnr_bf16_vs_fp32=psnr(rgb16, rgb32), psnr_fp32_vs_target=psnr(rgb32, rgbt), psnr_bf16_vs_target=psnr(rgb16, rgbt),
               max_abs_rgb=float((rgb16 - rgb32).abs().max()), seg_max_abs=float((o16[3][..., 3:] - o32[3][..., 3:]).abs().max()),
               depth_max_abs=float((o16[7] - o32[7]).abs().max()), acc_max_abs=float((o16[5] - o32[5]).abs().max()))
    res["delta_psnr"] = abs(res["psnr_bf16_vs_target"] - res["psnr_fp32_vs_target"])
    print(json.dumps(res))
    res["w_bg_mean"] = float(o32[6].mean())
    assert 0.02 < res["w_bg_mean"] < 0.9, "the volume must be semi-transparent for the protocol to mean anything"
    assert res["delta_psnr"] <= 0.05, res      # north_star: PSNR within 0.05 dB of the reference
    assert res["psnr_bf16_vs_fp32"] > 35.0, res


# ---- SAHS_BF16X3: near-fp32 on the bf16 matrix pipe (every net with bf16 hi + lo operands, three MFMAs per product; round 2 kept the
# deformation nets on the fp32 kernel, SAHS_X3_DEFORM=f32 still does) ---------------------------------------------------------------
@pytest.mark.parametrize("variant", ["boosted", "hdr"])
def test_bf16x3_field_vs_oracle(flat_weights, variant):
    """Field seam against the CPU oracle: (x', w) from the split-operand deformation launch, the radiance nets' raw output on top of them;
    every operand carries 16-17 significant bits: bounds ~4x the observed error, 300-500x below plain bf16's."""
    from conftest import VARIANT_KW
    ops, lib = pkg("ops"), pkg("_lib")
    g = load_golden("cond")
    fw = flat_weights(**VARIANT_KW[variant])
    flat = T(fw)
    packed = ops.pack_weights(flat, lib.SAHS_BF16X3)
    frame = ops.fold_conditioning(flat, T(g["audio"]), T(g["pose"]))
    rng = np.random.default_rng(11)
    N, S = 61, 64     # ragged: 3904 samples, not a multiple of the 128-sample tile
    rays = np.zeros((N, 8), np.float32)
    rays[:, 0:3] = rng.normal(0, 0.05, (N, 3)) + np.array([0, 0, 0.8])
    rays[:, 3:6] = rng.normal(0, 0.15, (N, 3)) + np.array([0, 0, -1.0])
    z = np.sort(rng.uniform(0.48, 1.08, (N, S)).astype(np.float32), axis=1)
    x6 = np.concatenate([rays[:, None, 0:3] + rays[:, None, 3:6] * z[..., None], np.broadcast_to(rays[:, None, 3:6], (N, S, 3))], -1).reshape(-1, 6)
    drv, p36 = oracle.audionet(fw, g["audio"]), oracle.pose_encoding(g["pose"])
    stats = {}
    for level in (0, 1):
        ref, rdx, rw, _ = oracle.field_forward(fw, level, x6.astype(np.float32), drv, p36, debug=True)
        xw = torch.zeros(N, S, 8, device=dev())
        raw = ops.field_forward_split(packed, frame, level, ops.FIELD_ALL, T(rays), xw, z=T(z), precision=lib.SAHS_BF16X3).view(-1, 16).cpu().numpy()
        xwn = xw.view(-1, 8).cpu().numpy()
        e_x = np.abs(xwn[:, :3] - (x6[:, :3] + rdx)).max()
        e_w = np.abs(xwn[:, 3:5] - rw).max()
        e_col, e_sig = np.abs(raw[:, :15] - ref[:, :15]), np.abs(raw[:, 15] - ref[:, 15])
        stats[level] = dict(xprime_max=float(e_x), w_max=float(e_w), col_max=float(e_col.max()), col_rms=float(np.sqrt((e_col ** 2).mean())),
                            col_scale=float(np.sqrt((ref[:, :15] ** 2).mean())), sig_max=float(e_sig.max()),
                            sig_rms=float(np.sqrt((e_sig ** 2).mean())), sig_scale=float(np.abs(ref[:, 15]).mean()))
        assert np.isfinite(raw).all()
    print(variant, json.dumps(stats))
    for level in (0, 1):
        st = stats[level]
        # deformation nets on the split-operand pipe as well (round 3): x' = x + tanh(.) adds a small correction to an exact x (observed
        # 6.1e-7; the fp32 kernel: 1e-7), w is a raw network output (observed 9.8e-6 on a scale of ~1); bounds = 4x observed
        assert st["xprime_max"] < 2.5e-6 and st["w_max"] < 4e-5, st
        assert st["col_rms"] < 6e-5 * max(1.0, st["col_scale"]) and st["sig_rms"] < 1.2e-4 * max(1.0, st["sig_scale"]), st
        assert st["col_max"] < 6e-4 * max(1.0, st["col_scale"]) and st["sig_max"] < 1.2e-3 * max(1.0, st["sig_scale"]), st


def test_bf16x3_frame_vs_fp32_within_4x_of_its_tolerance(weights_mod):
    """End to end (drop-in driver, validation mode, HDR weights): the bf16x3 frame against the fp32 path's on the same rays and draws at
    FOUR times the fp32 path's tolerance (rtol 4e-4 / atol 4e-5; SURVEY.md section 8d states 1e-4 / 1e-5 for fp32), a few rays excepted
    in the fine pass (a resampled depth that lands on the other side of a cdf knot -- as between any two fp32 implementations)."""
    sahs = pkg()
    d = dev()
    cfg = sahs.default_config()
    fw = weights_mod.flatten_state_dict(weights_mod.hash_state_dict(0, 2.0, 30.0, hdr=True))
    H = W = 48
    rng = np.random.default_rng(3)
    audio = T(rng.standard_normal((16, 29)).astype(np.float32))
    pose = T(np.concatenate([np.eye(3), [[0.0], [0.0], [0.8]]], 1).astype(np.float32))
    intr = np.array([1200.0 * W / 512, 1200.0 * W / 512, 0.5, 0.5], np.float32)
    bg = T(np.concatenate([rng.uniform(0, 1, (H * W, 3)), np.ones((H * W, 1)), np.zeros((H * W, 11))], 1).astype(np.float32))
    outs = {}
    for prec in ("fp32", "bf16x3"):
        model = sahs.AudioFaceModel(cfg, precision=prec).to(d).load_flat(fw).eval()
        ro, rd = sahs.get_ray_bundle(H, W, intr, pose)
        with torch.no_grad(), sahs.train_utils.partition_invariant_rng(7):
            outs[prec] = sahs.run_one_iter_of_nerf(H, W, intr, model, ro, rd, cfg, mode="validation", driving=audio, pose=pose, background_prior=bg)
    names = ["rgb_c", "disp_c", "acc_c", "rgb_f", "disp_f", "acc_f", "w_bg", "depth_f"]
    worst = {}
    for nm, a, b in zip(names, outs["fp32"], outs["bf16x3"]):
        a, b = a.reshape(H * W, -1), b.reshape(H * W, -1)
        assert bool(torch.isfinite(b).all()), nm
        bad = ((a - b).abs() > 4e-5 + 4e-4 * a.abs()).any(dim=1)
        worst[nm] = (float((a - b).abs().max()), float(bad.float().mean()))
        # coarse pass: every ray; chained fine pass: the resampled depths differ where a sample sits on a cdf knot (the coarse weights
        # differ by ~1e-5), which moves a depth by a bin -- those rays are excepted here and pinned below on identical depths
        assert float(bad.float().mean()) <= (0.0 if nm.endswith("_c") else 0.08), (nm, worst[nm])
    print(json.dumps({k: v for k, v in worst.items()}))
    assert float(outs["fp32"][6].mean()) > 0.02          # rays do spread their weight (the frame is not all foreground)
    assert psnr(outs["fp32"][3][..., :3], outs["bf16x3"][3][..., :3]) >= 60.0
    # the fine pass on IDENTICAL depths (the fp32 run's): field + compositing, every ray within 4x of the fp32 tolerance
    ops = pkg("ops")
    flat = T(fw)
    frame = ops.fold_conditioning(flat, audio, pose)
    ro, rd = sahs.get_ray_bundle(H, W, intr, pose)
    N = H * W
    rays = torch.cat([ro.reshape(-1, 3), rd.reshape(-1, 3), torch.full((N, 1), float(cfg.dataset.near), device=d),
                      torch.full((N, 1), float(cfg.dataset.far), device=d)], 1)
    g = torch.Generator(device=d).manual_seed(5)
    z_f = torch.sort(torch.rand(N, 128, device=d, generator=g) * 0.6 + 0.4838, dim=1).values
    res = {}
    for prec in ("fp32", "bf16x3"):
        pk = ops.pack_weights(flat, ops.PRECISIONS[prec])
        xw = torch.zeros(N, 128, 8, device=d)
        raw = ops.field_forward_split(pk, frame, 1, ops.FIELD_ALL, rays, xw, z=z_f, precision=ops.PRECISIONS[prec])
        res[prec] = ops.composite_forward(raw, z_f, rays, bg=bg)
    for nm, a, b in zip(("rgb", "disp", "acc", "weights", "depth"), res["fp32"], res["bf16x3"]):
        a, b = a.reshape(N, -1), b.reshape(N, -1)
        assert bool(((a - b).abs() <= 4e-5 + 4e-4 * a.abs()).all()), (nm, float((a - b).abs().max()))


def test_bf16x3_w512_frame_vs_fp32(weights_mod):
    """BASELINE.json's full size (512 x 512 rays, 64 + 128 evaluations) through the drop-in driver, bf16x3 (every net with split bf16 operands)
    against fp32 on the same keyed draws: every coarse output of every ray within 4x the fp32 tolerance, the fine outputs too except the
    resampling-knot rays, image PSNR.  (Observed: coarse max |d| 4.1e-5, 6.6 % knot rays, 70.0 dB.)"""
    sahs = pkg()
    d = dev()
    cfg = sahs.default_config()
    fw = weights_mod.flatten_state_dict(weights_mod.hash_state_dict(0, 2.0, 30.0, hdr=True))
    H = W = 512
    rng = np.random.default_rng(42)
    audio = T(rng.standard_normal((16, 29)).astype(np.float32))
    pose = T(np.concatenate([np.eye(3), [[0.0], [0.0], [0.8]]], 1).astype(np.float32))
    intr = np.array([1200.0, 1200.0, 0.5, 0.5], np.float32)
    bg = T(np.concatenate([rng.uniform(0, 1, (H * W, 3)), np.ones((H * W, 1)), np.zeros((H * W, 11))], 1).astype(np.float32))
    rows = {}
    for prec in ("fp32", "bf16x3"):
        model = sahs.AudioFaceModel(cfg, precision=prec).to(d).load_flat(fw).eval()
        ro, rd = sahs.get_ray_bundle(H, W, intr, pose)
        with torch.no_grad(), sahs.train_utils.partition_invariant_rng(7):
            out = sahs.run_one_iter_of_nerf(H, W, intr, model, ro, rd, cfg, mode="validation", driving=audio, pose=pose, background_prior=bg)
        rows[prec] = torch.cat([o.reshape(H * W, -1) for o in out], dim=1)
        del model
    a, b = rows["fp32"], rows["bf16x3"]
    assert bool(torch.isfinite(b).all())
    bad = (a - b).abs() > 4e-5 + 4e-4 * a.abs()
    res = dict(coarse_max=float((a[:, :17] - b[:, :17]).abs().max()), coarse_bad=float(bad[:, :17].any(dim=1).float().mean()),
               fine_bad=float(bad[:, 17:].any(dim=1).float().mean()), psnr_rgb_fine=psnr(a[:, 17:20], b[:, 17:20]), w_bg_mean=float(a[:, 34].mean()))
    print(json.dumps(res))
    assert res["coarse_bad"] == 0.0 and res["fine_bad"] <= 0.10 and res["psnr_rgb_fine"] >= 60.0 and 0.02 < res["w_bg_mean"] < 0.9, res


def test_bf16_exact_leaky_is_selectable_at_run_time(flat_weights):
    """The SAHS_BF16 kernels' LeakyReLU: the packed-integer form on the bf16 bit patterns (default; slope 0.0095 .. 0.0106) or, selected at run
    time (include/sahs_nerf.h: sahs_bf16_exact_leaky, ops.bf16_exact_leaky), the reference's max(v, 0.01 v) in fp32 before rounding -- so
    that a user with a real checkpoint can A/B the two without rebuilding.  Both forms must stay inside the bf16 error bounds against the
    fp32 oracle; they are different kernels (outputs differ), and the exact form is the one nearer the oracle."""
    from conftest import VARIANT_KW
    ops, lib = pkg("ops"), pkg("_lib")
    g = load_golden("cond")
    fw = flat_weights(**VARIANT_KW["hdr"])
    flat = T(fw)
    packed = ops.pack_weights(flat, lib.SAHS_BF16)
    frame = ops.fold_conditioning(flat, T(g["audio"]), T(g["pose"]))
    rng = np.random.default_rng(12)
    N, S = 64, 64
    rays = np.zeros((N, 8), np.float32)
    rays[:, 0:3] = rng.normal(0, 0.05, (N, 3)) + np.array([0, 0, 0.8])
    rays[:, 3:6] = rng.normal(0, 0.15, (N, 3)) + np.array([0, 0, -1.0])
    z = np.sort(rng.uniform(0.48, 1.08, (N, S)).astype(np.float32), axis=1)
    x6 = np.concatenate([rays[:, None, 0:3] + rays[:, None, 3:6] * z[..., None], np.broadcast_to(rays[:, None, 3:6], (N, S, 3))], -1).reshape(-1, 6)
    ref = oracle.field_forward(fw, 1, x6.astype(np.float32), oracle.audionet(fw, g["audio"]), oracle.pose_encoding(g["pose"]))
    assert ops.bf16_exact_leaky() is False
    out = {}
    try:
        for exact in (False, True):
            assert ops.bf16_exact_leaky(exact) is exact
            out[exact] = ops.field_forward(packed, frame, 1, T(rays), T(z), precision=lib.SAHS_BF16).view(-1, 16).cpu().numpy()
    finally:
        ops.bf16_exact_leaky(False)
    rms = {k: float(np.sqrt(((v[:, :15] - ref[:, :15]) ** 2).mean())) for k, v in out.items()}
    print("bf16 colour-logit rms error vs the fp32 oracle: packed-integer LeakyReLU %.4f, exact %.4f" % (rms[False], rms[True]))
    assert not np.array_equal(out[False], out[True])
    assert rms[True] < BF16_BOUNDS["hdr"]["col_rms"] and rms[False] < BF16_BOUNDS["hdr"]["col_rms"]
    assert rms[True] <= rms[False] * 1.05
