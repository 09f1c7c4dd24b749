"""Pin the CPU oracle (oracle/sahs_oracle.c) to the real reference.

Every golden .npz was produced by tests/golden/make_golden.py importing the unmodified
reference.  Tolerances: the oracle's arithmetic order is sequential fp32 while ATen's CPU kernels
are vectorised (different summation trees, SLEEF transcendentals), so agreement is to fp32
round-off amplified by the path's conditioning (PE frequencies up to 2^9 act on the warped
point, see DESIGN.md "Numerics"), not bit-for-bit -- except the index work (sample_pdf_2: cdf,
searchsorted indices, samples, merged depths), which IS bit-for-bit against ATen.
"""
import numpy as np
import pytest

from conftest import VARIANT_KW, golden_rand, golden_weights_kw, load_golden, yardstick
from oracle import oracle


def close(a, b, rtol, atol, what=""):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.all(err <= tol), "%s: max abs err %.3e (tol %.3e) at %s" % (
        what, err.max(), tol.reshape(-1)[err.argmax()], np.unravel_index(err.argmax(), err.shape))


def test_param_count(weights_mod):
    assert oracle.param_count() == weights_mod.param_count() == 2_775_633


def test_get_ray_bundle():
    g = load_golden("rays")
    ro, rd = oracle.get_ray_bundle(int(g["H"]), int(g["W"]), g["intrinsics"], g["c2w"])
    close(ro, g["ro"], 0, 0, "ro")
    close(rd, g["rd"], 1e-6, 1e-7, "rd")


def test_conditioning(flat_weights):
    g = load_golden("cond")
    close(oracle.audionet(flat_weights(), g["audio"]), g["driving"], 1e-5, 1e-6, "driving")
    close(oracle.pose_encoding(g["pose"]), g["pose36"], 1e-6, 1e-6, "pose36")


def test_positional_encoding():
    g = load_golden("pe")
    close(oracle.positional_encoding(g["x"], 10), g["pe_xyz"], 1e-6, 1e-6, "pe_xyz")
    close(oracle.positional_encoding(g["x"], 4), g["pe_dir"], 1e-6, 1e-6, "pe_dir")
    close(oracle.positional_encoding(g["w"], 4), g["pe_amb"], 1e-6, 1e-6, "pe_amb")
    close(oracle.positional_encoding(g["x"], 3, include_input=False), g["pe_pose"], 1e-6, 1e-6, "pe_pose")


@pytest.mark.parametrize("variant", ["default", "boosted", "hdr"])
def test_field(flat_weights, variant):
    g = load_golden("field")
    flat = flat_weights(**VARIANT_KW[variant])
    drv = oracle.audionet(flat, g["audio"])
    p36 = oracle.pose_encoding(g["pose"])
    raw_c, dx, w, grid = oracle.field_forward(flat, 0, g["x"], drv, p36, debug=True)
    raw_f = oracle.field_forward(flat, 1, g["x"], drv, p36)
    close(dx, g[variant + "_dx"], 1e-4, 2e-6, "dx")
    close(w, g[variant + "_w"], 1e-4, 2e-6, "ambient w")
    # grid features and raw see PE(512 * warped point): round-off in dx (1e-7) is amplified ~512x
    close(grid, g[variant + "_grid_coarse"], 1e-3, 2e-6, "grid features")
    scale = 1.0 if variant == "default" else 30.0
    for lvl, raw in (("coarse", raw_c), ("fine", raw_f)):
        close(raw[:, :15], g[variant + "_raw_" + lvl][:, :15], 1e-3, 1e-4, "raw rgb/seg " + lvl)
        close(raw[:, 15], g[variant + "_raw_" + lvl][:, 15], 1e-3, 1e-4 * scale, "raw sigma " + lvl)
    # accuracy against the reference model in float64: the oracle is as close to it as the reference's own fp32 run
    tag = "oracle field[%s] " % variant
    yardstick(dx, g[variant + "_dx"], g[variant + "_dx_f64"], tag + "dx")
    yardstick(w, g[variant + "_w"], g[variant + "_w_f64"], tag + "w")
    yardstick(grid, g[variant + "_grid_coarse"], g[variant + "_grid_coarse_f64"], tag + "grid")
    for lvl, raw in (("coarse", raw_c), ("fine", raw_f)):
        r32, r64 = g[variant + "_raw_" + lvl], g[variant + "_raw_" + lvl + "_f64"]
        yardstick(raw[:, :15], r32[:, :15], r64[:, :15], tag + "raw rgb/seg " + lvl)
        yardstick(raw[:, 15], r32[:, 15], r64[:, 15], tag + "raw sigma " + lvl)


@pytest.mark.parametrize("tag,use_bg,use_noise,white", [("bg", True, False, False), ("bg_noise", True, True, False),
                                                       ("nobg", False, False, False), ("nobg_white", False, False, True)])
def test_composite(tag, use_bg, use_noise, white):
    g = load_golden("composite")
    rgb, disp, acc, wts, depth = oracle.composite(g["raw"], g["z"], g["rd"], noise=g["noise"] if use_noise else None,
                                                  bg=g["bg"] if use_bg else None, white_background=white)
    close(wts, g[tag + "_weights"], 2e-5, 1e-7, "weights")
    close(rgb, g[tag + "_rgb"], 2e-5, 2e-6, "rgb")
    close(acc, g[tag + "_acc"], 2e-5, 1e-6, "acc")
    close(depth, g[tag + "_depth"], 2e-5, 1e-6, "depth")
    close(disp, g[tag + "_disp"], 5e-5, 1e-6, "disp")


@pytest.mark.parametrize("tag", ["rand", "det"])
def test_sample_pdf(tag):
    """Index work is BIT-EXACT against ATen (north_star: "bit-exact for sample indices"): the oracle adds torch.sum in ATen's own order,
    accumulates torch.cumsum in double and rounds every prefix (ATen CPU's acc_type<float>), and forms torch.linspace's second half as one
    fused multiply-add -- with those three restated, the cdf, the searchsorted indices, the samples and the merged sorted depths all
    equal the reference's, in both modes (det=True puts u = 1.0 exactly on the last knot of every ray: round 2 differed there on 58 rays)."""
    g = load_golden("pdf")
    z, w = g["z"], g["weights"]
    u = g["u"] if tag == "rand" else None
    zs, zsorted, inds = oracle.resample(z, w, 64, u=u)
    assert np.array_equal(inds, g[tag + "_inds"]), "searchsorted index mismatches: %d" % (inds != g[tag + "_inds"]).sum()
    assert np.array_equal(zs, g[tag + "_samples"]), "z_samples differ from the reference in %d places" % (zs != g[tag + "_samples"]).sum()
    assert np.array_equal(zsorted, g[tag + "_z_sorted"])
    assert np.all(np.diff(zsorted, axis=1) >= 0)


def test_aten_sum_and_linspace_restatements():
    """The two ATen summation/generation orders the index work depends on, against torch itself for every length the path can see
    (S, num_fine <= 256).  torch is the reference's arithmetic library; this runs wherever the CPU suite runs."""
    import torch
    rng = np.random.default_rng(5)
    for n in range(1, 301):
        x = (rng.random(n, dtype=np.float32) * np.float32(10.0) ** rng.integers(-6, 1)).astype(np.float32) + np.float32(1e-5)
        assert oracle.aten_sum(x) == torch.sum(torch.from_numpy(x)).numpy(), n
        # as the reference calls it: a (rows, n) tensor reduced over its last dimension
        X = np.stack([x, x[::-1].copy(), np.roll(x, 3)])
        want = torch.sum(torch.from_numpy(X), dim=-1).numpy()
        assert all(oracle.aten_sum(X[r]) == want[r] for r in range(3)), n
        assert np.array_equal(oracle.linspace01(n), torch.linspace(0.0, 1.0, n).numpy()), n


def test_stratified_depths_bit_exact_vs_aten():
    """train_utils.py:93-113 written out in torch (linspace, lerp, mids, perturbation) against the oracle: bit for bit."""
    import torch
    N, S = 50, 64
    near, far = np.full(N, 0.483771, np.float32), np.full(N, 1.083771, np.float32)
    tr = np.random.default_rng(0).random((N, S), dtype=np.float32)
    t = torch.linspace(0.0, 1.0, S)
    z = torch.from_numpy(near)[:, None] * (1.0 - t) + torch.from_numpy(far)[:, None] * t
    mids = 0.5 * (z[..., 1:] + z[..., :-1])
    upper, lower = torch.cat((mids, z[..., -1:]), -1), torch.cat((z[..., :1], mids), -1)
    zp = lower + (upper - lower) * torch.from_numpy(tr)
    assert np.array_equal(oracle.stratified_depths(near, far, S), z.numpy())
    assert np.array_equal(oracle.stratified_depths(near, far, S, t_rand=tr), zp.numpy())


def _chunk_rand(g, nchunks):
    log = golden_rand(g)
    per = len(log) // nchunks
    keys = {4: ["t_rand", "noise_c", "u", "noise_f"], 2: ["t_rand", "u"], 0: []}[per]
    std = float(g["noise_std"]) if "noise_std" in g else 0.1
    out = []
    for c in range(nchunks):
        d = {}
        for k, (kind, arr) in zip(keys, log[c * per:(c + 1) * per]):
            assert kind == ("randn" if k.startswith("noise") else "rand")
            d[k] = arr * np.float32(std) if k.startswith("noise") else arr
        out.append(d)
    return out


@pytest.mark.parametrize("name,nchunks", [("e2e_default_val", 1), ("e2e_boosted_val", 1), ("e2e_boosted_val_2chunks", 2),
                                          ("e2e_boosted_det", 1), ("e2e_boosted_train_noise", 1), ("e2e_hdr_val", 1),
                                          ("e2e_hdr_det", 1), ("e2e_hdr_train_noise", 1)])
def test_end_to_end(flat_weights, name, nchunks):
    g = load_golden(name)
    flat = flat_weights(**golden_weights_kw(g))
    ro, rd = oracle.get_ray_bundle(int(g["H"]), int(g["W"]), g["intrinsics"], g["pose"])
    close(rd, g["rd"], 1e-6, 1e-7)
    outs = oracle.run_one_iter_of_nerf(flat, ro, rd, float(g["near"]), float(g["far"]), int(g["num_coarse"]), int(g["num_fine"]),
                                       g["audio"], g["pose"], background_prior=g["bg"], chunksize=int(g["chunksize"]),
                                       rand=_chunk_rand(g, nchunks))
    names = ["rgb_c", "disp_c", "acc_c", "rgb_f", "disp_f", "acc_f", "w_bg", "depth_f"]
    for nm, o in zip(names, outs):
        ref = g["out_" + nm].reshape(o.shape)
        # end-to-end: field round-off (1e-4 abs on raw) passes through sigmoid/softmax and the composite
        if "hdr" not in name:
            close(o, ref, 2e-3, 2e-4, name + ":" + nm)
        yardstick(o, ref, g["f64_" + nm], "oracle %s:%s" % (name, nm), outlier_rays=0.0 if nm.endswith("_c") else 0.02, scale_floor=1.0,
                  ray_shape=(int(g["H"]) * int(g["W"]),))
    if "boosted" in name:
        assert float(np.mean(g["out_w_bg"])) < 0.5, "density-boosted fixture should terminate rays before the background"


def test_ray_uniforms_known_answer_and_partition_invariance():
    """Philox4x32-10: Random123's published known-answer vector for the all-zero counter and key
    (6627e8d5 e169c58d bc57ac4c 9b00dbd8); a draw is the top 24 bits of its word."""
    u = oracle.ray_uniforms(0, 0, 0, 1, 4)[0]
    assert [int(round(float(v) * 2 ** 24)) for v in u] == [w >> 8 for w in (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)]
    full = oracle.ray_uniforms(7, 1, 0, 10, 13)
    assert np.array_equal(full[4:], oracle.ray_uniforms(7, 1, 4, 6, 13))          # a shard starting at ray 4 sees the same draws
    assert np.array_equal(full[:, :8], oracle.ray_uniforms(7, 1, 0, 10, 8))       # and so does a shorter sample count
    assert not np.array_equal(full, oracle.ray_uniforms(7, 0, 0, 10, 13)) and not np.array_equal(full, oracle.ray_uniforms(8, 1, 0, 10, 13))
    u = oracle.ray_uniforms(42, 0, 0, 4096, 64)
    assert 0.0 <= u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 2e-3 and abs(u.var() - 1 / 12) < 1e-3
