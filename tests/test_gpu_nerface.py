"""SURVEY.md section 8f-3 on the MI355X: the expression-driven NeRFaceModel (config/expression/person_2|3.yml) through the same
C ABI family (sahs_nerface_*), against the reference's golden vectors (tests/golden/nerface_*.npz) and against the NeRFaceModel
build of the CPU oracle.

Tolerances: this model encodes positions with 15 octaves, so round-off in the warp output (1e-7) is amplified 2^14-fold in
the top PE features (vs 2^9 for the audio model): `raw` agrees to 2e-3, and the fine pass is additionally ill-conditioned in
the resampled depths -- the stages are pinned separately, as in tests/test_oracle_nerface_vs_golden.py."""
import numpy as np
import pytest
import torch

from conftest import golden_rand, load_golden, pkg, yardstick
from test_gpu_parity import FeedRand, T, close, dev

pytestmark = pytest.mark.gpu

from oracle import oracle  # noqa: E402  (checker only)


@pytest.fixture(scope="module")
def ops():
    return pkg("ops")


@pytest.fixture(scope="module")
def nf():
    W = pkg("weights")
    ops = pkg("ops")
    cache = {}

    def get(seed=0, density_bias=0.0, density_gain=1.0, arch="nerface"):
        key = (int(seed), float(density_bias), float(density_gain), arch)
        if key not in cache:
            fw = W.flatten_state_dict(W.hash_state_dict(*key[:3], model=arch), model=arch)
            flat = T(fw)
            cache[key] = (fw, flat, ops.pack_weights(flat, arch=arch))
        return cache[key]

    return get


def test_sizes_and_errors(ops):
    assert ops.param_count("nerface") == 2_311_140 and ops.param_count() == 2_775_633 and ops.param_count("nerface_static") == 2_066_976
    with pytest.raises(Exception):
        ops.pack_weights(torch.zeros(2_775_633, device=dev()), arch="nerface")        # the audio model's buffer
    with pytest.raises(Exception):      # the A/B precision ids (development builds) are not in the shipped library
        ops.pack_weights(torch.zeros(2_066_976, device=dev()), precision=2, arch="nerface_static")
    with pytest.raises(Exception):
        ops.pack_weights(torch.zeros(2_775_633, device=dev()), precision=4)
    with pytest.raises(Exception):      # the mixed-precision model's whole-network entry point needs the split chain's workspace
        ops.field_forward(ops.pack_weights(torch.zeros(2_311_140, device=dev()), precision=ops.SAHS_BF16, arch="nerface"),
                          torch.zeros(16384, device=dev()), 0,
                          torch.zeros(4, 8, device=dev()), torch.zeros(4, 2, device=dev()), precision=ops.SAHS_BF16, arch="nerface")


@pytest.mark.parametrize("variant", ["default", "boosted"])
def test_field_vs_golden(ops, nf, variant):
    g = load_golden("nerface_field")
    kw = dict(default=dict(), boosted=dict(density_bias=8.0, density_gain=30.0))[variant]
    fw, flat, packed = nf(**kw)
    frame = ops.fold_conditioning(flat, T(g["expression"]), T(g["pose"]), arch="nerface")
    assert torch.equal(frame[:76].cpu(), torch.from_numpy(g["expression"]))
    x = g["x"]
    P = x.shape[0]
    rays = np.zeros((P, 8), np.float32)
    rays[:, :6] = x
    z = torch.zeros(P, 1, device=dev())
    raw_c, dx, w, grid = ops.field_forward(packed, frame, 0, T(rays), z, debug=True, arch="nerface")
    raw_f = ops.field_forward(packed, frame, 1, T(rays), z, arch="nerface")
    close(dx.view(P, 3), g[variant + "_dx"], 1e-4, 2e-6, "dx")
    close(w.view(P, 2)[:, :1], g[variant + "_w"], 1e-4, 2e-6, "w")
    close(grid.view(P, 32), g[variant + "_grid_coarse"], 1e-3, 2e-6, "grid")
    scale = 30.0 if variant == "boosted" else 1.0
    for lvl, raw in (("coarse", raw_c), ("fine", raw_f)):
        raw = raw.view(P, 16)
        close(raw[:, :15], g[variant + "_raw_" + lvl][:, :15], 2e-3, 2e-3, "raw rgb/seg " + lvl)
        close(raw[:, 15], g[variant + "_raw_" + lvl][:, 15], 2e-3, 2e-3 * scale, "raw sigma " + lvl)
    # float64 yardstick: the reference's own fp32 run is 3e-4 .. 6e-4 from the exact value on this 15-octave network
    tag = "hip nerface field[%s] " % variant
    yardstick(dx.view(P, 3), g[variant + "_dx"], g[variant + "_dx_f64"], tag + "dx")
    yardstick(w.view(P, 2)[:, :1], g[variant + "_w"], g[variant + "_w_f64"], tag + "w")
    yardstick(grid.view(P, 32), g[variant + "_grid_coarse"], g[variant + "_grid_coarse_f64"], tag + "grid")
    for lvl, raw in (("coarse", raw_c), ("fine", raw_f)):
        yardstick(raw.view(P, 16), g[variant + "_raw_" + lvl], g[variant + "_raw_" + lvl + "_f64"], tag + "raw " + lvl)


@pytest.mark.parametrize("N,S", [(37, 64), (19, 128), (5, 1), (3, 192)])
def test_field_vs_oracle(ops, nf, N, S):
    """Ragged sizes, both levels, rays + depths input (the product entry point), against the NeRFaceModel oracle."""
    rng = np.random.default_rng(N * 1000 + S)
    g = load_golden("nerface_field")
    fw, flat, packed = nf(density_bias=8.0, density_gain=30.0)
    frame = ops.fold_conditioning(flat, T(g["expression"]), T(g["pose"]), arch="nerface")
    rays = np.zeros((N, 20), np.float32)
    rays[:, 0:3] = rng.normal(0, 0.05, (N, 3)) + np.array([0, 0, 0.5])
    rays[:, 3:6] = rng.normal(0, 0.15, (N, 3)) + np.array([0, 0, -1.0])
    rays[:, 6], rays[:, 7] = 0.2, 0.8
    z = np.sort(rng.uniform(0.2, 0.8, (N, S)).astype(np.float32), axis=1)
    x6 = np.concatenate([rays[:, None, 0:3] + rays[:, None, 3:6] * z[..., None], np.broadcast_to(rays[:, None, 3:6], (N, S, 3))], axis=-1)
    x6 = x6.reshape(-1, 6).astype(np.float32)
    with oracle.model("nerface"):
        p36 = oracle.pose_encoding(g["pose"])
        refs = [oracle.field_forward(fw, level, x6, g["expression"], p36, debug=True) for level in (0, 1)]
    for level in (0, 1):
        ref, rdx, rw, rgrid = refs[level]
        raw, dx, w, grid = ops.field_forward(packed, frame, level, T(rays), T(z), debug=True, arch="nerface")
        close(dx.reshape(-1, 3), rdx, 1e-4, 2e-6, "dx")
        close(w.reshape(-1, 2)[:, :1], rw, 1e-4, 2e-6, "w")
        close(grid.reshape(-1, 32), rgrid, 1e-3, 2e-6, "grid")
        raw = raw.reshape(-1, 16)
        close(raw[:, :15], ref[:, :15], 2e-3, 2e-3, "raw rgb/seg level %d" % level)
        close(raw[:, 15], ref[:, 15], 2e-3, 6e-2, "raw sigma level %d" % level)


@pytest.mark.parametrize("name", ["nerface_e2e_val", "nerface_e2e_det"])
def test_end_to_end_vs_golden(ops, nf, name):
    sahs = pkg()
    g = load_golden(name)
    cfg = sahs.default_config("expression")
    node = cfg.nerf.validation
    node.perturb, node.radiance_field_noise_std = bool(g["perturb"]), float(g["noise_std"])
    assert abs(cfg.dataset.near - float(g["near"])) < 1e-7 and abs(cfg.dataset.far - float(g["far"])) < 1e-7
    fw, flat, packed = nf(int(g["weights_seed"]), float(g["weights_density_bias"]), float(g["weights_density_gain"]))
    model = sahs.NeRFaceModel(cfg).to(dev()).load_flat(fw)
    pose = T(g["pose"])
    H, Wd = int(g["H"]), int(g["W"])
    ro, rd = sahs.get_ray_bundle(H, Wd, g["intrinsics"], pose)
    with torch.no_grad(), FeedRand(golden_rand(g)) as feed:
        outs = sahs.run_one_iter_of_nerf(H, Wd, g["intrinsics"], model, ro, rd, cfg, mode="validation", driving=T(g["expression"]),
                                         pose=pose, pose_c=None, background_prior=T(g["bg"]), latent_code=None,
                                         inHead=torch.zeros(H, Wd, 12, device=dev()))
        assert not feed.log, "the driver must consume exactly the reference's random draws"
    names = ["rgb_c", "disp_c", "acc_c", "rgb_f", "disp_f", "acc_f", "w_bg", "depth_f"]
    o = dict(zip(names, outs))
    for nm in names:
        assert tuple(o[nm].shape) == tuple(g["out_" + nm].shape), (nm, o[nm].shape)
    for nm in ("rgb_c", "disp_c", "acc_c"):
        close(o[nm], g["out_" + nm], 2e-4, 1e-4, name + ":" + nm)
    for nm in ("rgb_f", "disp_f", "acc_f", "w_bg", "depth_f"):
        close(o[nm], g["out_" + nm], 1e-2, 3e-3, name + ": chained " + nm)
    for nm in names:       # and against the float64 run of the reference (its own fp32 disparity is 5e-2 off on this network)
        yardstick(o[nm], g["out_" + nm], g["f64_" + nm], "hip %s:%s" % (name, nm), outlier_rays=0.0 if nm.endswith("_c") else 0.02, scale_floor=1.0,
                  ray_shape=(H * Wd,))
    # fine pass on the reference's own depths: field + composite, tight
    N = H * Wd
    rays = torch.cat([ro.reshape(-1, 3), rd.reshape(-1, 3), torch.full((N, 1), float(g["near"]), device=dev()),
                      torch.full((N, 1), float(g["far"]), device=dev())], 1).contiguous()
    frame = model.frame(T(g["expression"]), pose)
    zf = T(g["z_fine"])
    raw = ops.field_forward(packed, frame, 1, rays, zf, arch="nerface")
    rgb, disp, acc, wts, depth = ops.composite_forward(raw, zf, rays, None, T(g["bg"]), False)
    for nm, v in (("rgb_f", rgb), ("disp_f", disp), ("acc_f", acc), ("depth_f", depth), ("w_bg", wts[:, -1])):
        close(v.reshape(g["out_" + nm].shape), g["out_" + nm], 5e-4, 1e-4, name + ": fine pass on reference depths: " + nm)


def test_model_seam_and_config_guard(nf):
    sahs = pkg()
    cfg = sahs.default_config("expression")
    fw, flat, packed = nf(density_bias=8.0, density_gain=30.0)
    model = sahs.NeRFaceModel(cfg).to(dev()).load_flat(fw)
    g = load_golden("nerface_field")
    x = T(np.concatenate([g["x"], np.zeros((256, 12), np.float32)], 1))
    with torch.no_grad():
        raw = model("coarse", x, T(g["expression"]), T(g["pose"]), None)
    close(raw[:, :15], g["boosted_raw_coarse"][:, :15], 2e-3, 2e-3, "model(...) seam")
    with pytest.raises(NotImplementedError):
        sahs.NeRFaceModel(sahs.default_config("audio"))
    with pytest.raises((NotImplementedError, KeyError)):
        sahs.NeRFaceModel(cfg, precision="bf16_2w")
    assert sahs.NeRFaceModel(cfg, precision="bf16").precision == pkg("ops").SAHS_BF16


# ---- config/expression/person_1.yml: NeRFaceModel without deformation nets (use_warp False, use_ambient False) ----
@pytest.mark.parametrize("variant", ["default", "boosted"])
def test_static_field_vs_golden(ops, nf, variant):
    g = load_golden("nerface_static_field")
    kw = dict(default=dict(), boosted=dict(density_bias=8.0, density_gain=30.0))[variant]
    fw, flat, packed = nf(arch="nerface_static", **kw)
    frame = ops.fold_conditioning(flat, T(g["expression"]), T(g["pose"]), arch="nerface_static")
    x = g["x"]
    P = x.shape[0]
    rays = np.zeros((P, 8), np.float32)
    rays[:, :6] = x
    z = torch.zeros(P, 1, device=dev())
    raw_c, dx, w, grid = ops.field_forward(packed, frame, 0, T(rays), z, debug=True, arch="nerface_static")
    raw_f = ops.field_forward(packed, frame, 1, T(rays), z, arch="nerface_static")
    assert not bool(dx.any())
    close(grid.view(P, 32), g[variant + "_grid_coarse"], 1e-5, 1e-7, "grid")
    scale = 30.0 if variant == "boosted" else 1.0
    for lvl, raw in (("coarse", raw_c), ("fine", raw_f)):
        raw = raw.view(P, 16)
        close(raw[:, :15], g[variant + "_raw_" + lvl][:, :15], 1e-3, 1e-4, "raw rgb/seg " + lvl)
        close(raw[:, 15], g[variant + "_raw_" + lvl][:, 15], 1e-3, 1e-4 * scale, "raw sigma " + lvl)


@pytest.mark.parametrize("N,S", [(37, 64), (3, 192), (5, 1)])
def test_static_field_vs_oracle(ops, nf, N, S):
    rng = np.random.default_rng(N * 1000 + S + 7)
    g = load_golden("nerface_static_field")
    fw, flat, packed = nf(density_bias=8.0, density_gain=30.0, arch="nerface_static")
    frame = ops.fold_conditioning(flat, T(g["expression"]), T(g["pose"]), arch="nerface_static")
    rays = np.zeros((N, 8), np.float32)
    rays[:, 0:3] = rng.normal(0, 0.05, (N, 3)) + np.array([0, 0, 0.5])
    rays[:, 3:6] = rng.normal(0, 0.15, (N, 3)) + np.array([0, 0, -1.0])
    z = np.sort(rng.uniform(0.2, 0.8, (N, S)).astype(np.float32), axis=1)
    x6 = np.concatenate([rays[:, None, 0:3] + rays[:, None, 3:6] * z[..., None], np.broadcast_to(rays[:, None, 3:6], (N, S, 3))], axis=-1)
    x6 = x6.reshape(-1, 6).astype(np.float32)
    with oracle.model("nerface_static"):
        refs = [oracle.field_forward(fw, level, x6, g["expression"], oracle.pose_encoding(g["pose"])) for level in (0, 1)]
    for level in (0, 1):
        raw = ops.field_forward(packed, frame, level, T(rays), T(z), arch="nerface_static").reshape(-1, 16)
        close(raw[:, :15], refs[level][:, :15], 1e-3, 1e-4, "raw rgb/seg level %d" % level)
        close(raw[:, 15], refs[level][:, 15], 1e-3, 3e-3, "raw sigma level %d" % level)


def test_static_end_to_end_vs_golden(nf):
    sahs = pkg()
    g = load_golden("nerface_static_e2e_val")
    cfg = sahs.default_config("expression_static")
    fw, flat, packed = nf(int(g["weights_seed"]), float(g["weights_density_bias"]), float(g["weights_density_gain"]), arch="nerface_static")
    model = sahs.NeRFaceModel(cfg).to(dev()).load_flat(fw)
    assert model.arch == "nerface_static"
    pose = T(g["pose"])
    H, Wd = int(g["H"]), int(g["W"])
    ro, rd = sahs.get_ray_bundle(H, Wd, g["intrinsics"], pose)
    with torch.no_grad(), FeedRand(golden_rand(g)) as feed:
        outs = sahs.run_one_iter_of_nerf(H, Wd, g["intrinsics"], model, ro, rd, cfg, mode="validation", driving=T(g["expression"]),
                                         pose=pose, background_prior=T(g["bg"]), inHead=torch.zeros(H, Wd, 12, device=dev()))
        assert not feed.log
    names = ["rgb_c", "disp_c", "acc_c", "rgb_f", "disp_f", "acc_f", "w_bg", "depth_f"]
    for nm, o in zip(names, outs):   # 10 octaves, no warp: the audio model's end-to-end tolerance
        close(o, g["out_" + nm], 2e-3, 2e-4, "static e2e:" + nm)


# ---- training path of the NeRFaceModels: HIP backward vs the reference's own autograd (golden) and vs eager autograd ----
@pytest.mark.parametrize("arch", ["nerface", "nerface_static"])
def test_gradients_vs_golden(nf, arch):
    """Train mode, noise 0.1, loss = <rgb_c,A> + <rgb_f,B> + 0.1*sum(depth_f) through ops.RenderRaysFn, on the reference's rays and
    random draws; gradients against the reference's (tests/golden/make_golden_nerface.py).  The 15-octave model's fine pass (and
    with it the gradients) is ill-conditioned in the resampled depths -- see test_oracle_nerface_vs_golden.py -- hence its looser
    bound; the 10-octave one is held to the audio model's."""
    sahs = pkg()
    g = load_golden(arch + "_train_grads")
    cfg = sahs.default_config("expression" if arch == "nerface" else "expression_static")
    fw, flat, packed = nf(0, 8.0, 30.0, arch=arch)
    model = sahs.NeRFaceModel(cfg).to(dev()).load_flat(fw).train()
    expr = T(g["expression"]).requires_grad_(True)
    with FeedRand(golden_rand(g)) as feed:
        outs = sahs.run_one_iter_of_nerf(10, 10, None, model, T(g["ro"]), T(g["rd"]), cfg, mode="train", driving=expr, pose=T(g["pose"]),
                                         background_prior=T(g["bg"]), inHead=torch.zeros(32, 12, device=dev()))
        assert not feed.log
    # The chained 15-octave gradients (resampled depths + kink flips + the 2^14-amplified, cancelling PE derivative in front of the
    # deformation nets) differ between ANY two fp32 evaluations by tens of percents on single entries -- the reference's CPU run
    # against its own GPU run alike -- and the atomics make this run itself non-deterministic at that level.  What is stable is
    # checked here (loss, gradient norms of the radiance nets and the grid); the exactness of the backward is established where
    # it is well conditioned, by test_field_backward_seam_vs_autograd.  The 10-octave model is held to the audio model's bounds.
    chained15 = arch == "nerface"
    rn, ra = (1.5e-1, None) if chained15 else (5e-3, 1e-2)
    loss = (outs[0] * T(g["A"])).sum() + (outs[3] * T(g["B"])).sum() + outs[7].sum() * 0.1
    assert abs(float(loss) - float(g["loss"])) <= (5e-2 if chained15 else rn) * abs(float(g["loss"])) + 1e-3
    loss.backward()
    params = dict(model.named_parameters())
    worst = 0.0
    for k, ref_norm in zip([str(n) for n in g["grad_names"]], g["grad_norms"]):
        gr = params[k].grad
        assert gr is not None and bool(torch.isfinite(gr).all()), k
        if chained15 and not (k.startswith("nerf_mlps.") or k == "spatial_embeddings"):
            continue
        n = float(gr.double().norm())
        assert abs(n - ref_norm) <= rn * ref_norm + 1e-7, (k, n, ref_norm)
        if ra is not None and "grad_" + k in g:
            ref = g["grad_" + k]
            err = float(np.abs(gr.cpu().numpy() - ref).max()) / (float(np.abs(ref).max()) + 1e-12)
            worst = max(worst, err)
            assert err <= ra, (k, err)
    ge = expr.grad.cpu().numpy()
    assert np.isfinite(ge).all()
    if ra is not None:
        assert float(np.abs(ge - g["grad_expression"]).max()) <= ra * float(np.abs(g["grad_expression"]).max()), "d expression"
    print("worst entry error / tensor scale: %.2e" % worst)


@pytest.mark.parametrize("arch", ["nerface", "nerface_static"])
def test_train_step_vs_eager_autograd(nf, arch):
    """256 rays, identical inputs and draws: the HIP backward against plain torch autograd of the eager restatement (itself pinned
    to the reference's gradients on CPU).  Same depths on both sides, so this comparison is tight for both architectures."""
    from oracle import torch_eager as TE
    sahs = pkg()
    W = pkg("weights")
    cfg = sahs.default_config("expression" if arch == "nerface" else "expression_static")
    sd_np = W.hash_state_dict(0, 8.0, 30.0, model=arch)
    fw, flat, packed = nf(0, 8.0, 30.0, arch=arch)
    model = sahs.NeRFaceModel(cfg).to(dev()).load_flat(fw).train()
    sd_t = {k: torch.from_numpy(v).to(dev()).requires_grad_(True) for k, v in sd_np.items()}
    field = TE.EagerField(sd_t, arch=arch)
    R = 256
    gen = torch.Generator(device=dev()).manual_seed(5)
    expr = (torch.randn(76, device=dev(), generator=gen) * 0.5)
    pose = T(np.concatenate([np.eye(3), [[0.0], [0.0], [0.5]]], 1).astype(np.float32))
    ro = torch.zeros(R, 3, device=dev())
    ro[:, 2] = 0.5
    rd = torch.randn(R, 3, device=dev(), generator=gen) * 0.15 + torch.tensor([0, 0, -1.0], device=dev())
    bg = torch.cat([torch.rand(R, 3, device=dev(), generator=gen), torch.ones(R, 1, device=dev()), torch.zeros(R, 11, device=dev())], 1)
    A, B = torch.randn(R, 15, device=dev(), generator=gen), torch.randn(R, 15, device=dev(), generator=gen)
    rand = dict(t_rand=torch.rand(R, 64, device=dev(), generator=gen), noise_c=torch.randn(R, 64, device=dev(), generator=gen) * 0.1,
                u=torch.rand(R, 64, device=dev(), generator=gen), noise_f=torch.randn(R, 128, device=dev(), generator=gen) * 0.1)
    feed = [("rand", rand["t_rand"].cpu().numpy()), ("randn", (rand["noise_c"] / 0.1).cpu().numpy()), ("rand", rand["u"].cpu().numpy()),
            ("randn", (rand["noise_f"] / 0.1).cpu().numpy())]
    e1 = expr.clone().requires_grad_(True)
    with FeedRand(feed):
        outs = sahs.run_one_iter_of_nerf(0, 0, None, model, ro, rd, cfg, mode="train", driving=e1, pose=pose, background_prior=bg)
    loss = (outs[0] * A).sum() + (outs[3] * B).sum() + outs[7].sum() * 0.1
    loss.backward()
    e2 = expr.clone().requires_grad_(True)
    o2 = TE.run_one_iter(field, ro, rd, cfg.dataset.near, cfg.dataset.far, e2, pose, bg=bg, rand=[rand], perturb=True, noise_std=0.1)
    l2 = (o2[0] * A).sum() + (o2[3] * B).sum() + o2[7].sum() * 0.1
    l2.backward()
    assert abs(float(loss) - float(l2)) <= 2e-2 * abs(float(l2)) + 1e-2
    for k, p in model.named_parameters():
        ref = sd_t[k].grad
        if arch == "nerface":      # see test_gradients_vs_golden: norms of the radiance nets / grid only
            if k.startswith("nerf_mlps.") or k == "spatial_embeddings":
                assert abs(float(p.grad.norm()) - float(ref.norm())) <= 1.5e-1 * float(ref.norm()) + 1e-7, k
            continue
        scale = float(ref.abs().max()) + 1e-12
        err = float((p.grad - ref).abs().max()) / scale
        assert err <= 2e-2, "%s: %.3e of scale" % (k, err)
    if arch != "nerface":
        assert float((e1.grad - e2.grad).abs().max()) <= 2e-2 * float(e2.grad.abs().max())


@pytest.mark.parametrize("arch,level", [("nerface", 0), ("nerface", 1), ("nerface_static", 1), ("audio", 1)])
def test_field_backward_seam_vs_autograd(ops, nf, arch, level):
    """sahs_model_field_forward_save + sahs_model_field_backward on explicit rays/depths against torch autograd of the eager field on
    the same points with the same upstream gradient: no resampling in the loop, so this is well conditioned for every architecture
    (what is left of the 15-octave amplification is the 1e-7 round-off of x' times 2^14).  The yardstick is float64 autograd evaluated on the
    SAME side of every (leaky-)ReLU kink as the HIP forward (its saved activations): two correct fp32 forwards that differ by 1e-5
    put a handful of pre-activations on different sides of zero, and with density-boosted weights one such sample moves a
    bias-gradient entry by percents -- that, not the backward arithmetic, is what separates the HIP gradients from PyTorch's in
    the end-to-end gradient tests (observed there: up to 1.7e-2 of scale).  With the branches pinned the bound is 2e-3 (observed: audio 1.5e-5,
    static 5e-6, 15-octave model 5e-4)."""
    from oracle import torch_eager as TE
    W = pkg("weights")
    S = 64 if level == 0 else 128
    N = 24
    gen = torch.Generator(device=dev()).manual_seed(11 + level)
    if arch == "audio":
        fw = W.flatten_state_dict(W.hash_state_dict(0, 8.0, 30.0))
        flat = T(fw)
        packed = ops.pack_weights(flat)
        driving = torch.randn(16, 29, device=dev(), generator=gen)
        near, far, cam = 0.48, 1.08, 0.8
        sd_np = W.hash_state_dict(0, 8.0, 30.0)
    else:
        fw, flat, packed = nf(0, 8.0, 30.0, arch=arch)
        driving = torch.randn(76, device=dev(), generator=gen) * 0.5
        near, far, cam = 0.2, 0.8, 0.5
        sd_np = W.hash_state_dict(0, 8.0, 30.0, model=arch)
    pose = T(np.concatenate([np.eye(3), [[0.0], [0.0], [cam]]], 1).astype(np.float32))
    rays = torch.zeros(N, 8, device=dev())
    rays[:, 2] = cam
    rays[:, 3:6] = torch.randn(N, 3, device=dev(), generator=gen) * 0.15 + torch.tensor([0, 0, -1.0], device=dev())
    z = torch.sort(torch.rand(N, S, device=dev(), generator=gen) * (far - near) + near, dim=1).values
    d_raw = torch.randn(N * S, 16, device=dev(), generator=gen)
    frame = ops.fold_conditioning(flat, driving, pose, arch=arch)
    raw, act = ops.field_forward_save(packed, frame, level, rays, z, arch)
    grad_flat = torch.zeros_like(flat)
    grad_cond = torch.zeros(128, device=dev())
    ops.field_backward(flat, frame, level, act, d_raw, grad_flat, grad_cond, arch)
    x6 = torch.cat([rays[:, None, 0:3] + rays[:, None, 3:6] * z[..., None], rays[:, None, 3:6].expand(N, S, 3)], -1).reshape(-1, 6)
    lvl = "coarse" if level == 0 else "fine"

    # which side of zero every hidden unit is on, as the HIP forward saw it (saved post-activations; sahs::act layout)
    kbx, kba, trl = {"audio": (4, 2, 8), "nerface": (6, 2, 4), "nerface_static": (4, 0, 4)}[arch]
    WH = 16 * kbx; HH = WH + 6 * 128 + 16; Tt = HH + 6 * 64 + 32 + 16 * kbx + 16 * kba; C = Tt + trl * 256 + 256 + 64; Ss = C + 512
    assert act.shape[1] == Ss + 512
    P = N * S
    flat_act = act.reshape(-1)
    arr = lambda c, w: flat_act[c * P:(c + w) * P].view(P, w)      # one dense [P x width] array per layer, array (c, w) at c * P
    masks = {}
    for i in range(6):
        masks["warp.%d" % i] = arr(WH + 128 * i, 128) > 0
        masks["hyper.%d" % i] = arr(HH + 64 * i, 64) > 0
    for i in range(trl):
        masks["trunk.%d" % i] = arr(Tt + 256 * i, 256) > 0
    for i in range(4):
        masks["dir.%d" % i] = arr(C + 128 * i, 128) > 0
        masks["seg.%d" % i] = arr(Ss + 128 * i, 128) > 0

    def autograd(dtype, use_masks):
        sd_t = {k: torch.from_numpy(v).to(dev()).to(dtype).requires_grad_(True) for k, v in sd_np.items()}
        drv = driving.to(dtype).clone().requires_grad_(True)
        raw_e = TE.EagerField(sd_t, arch=arch, masks=masks if use_masks else None).forward(lvl, x6.to(dtype), drv, pose.to(dtype))
        (raw_e * d_raw.to(dtype)).sum().backward()
        return raw_e.detach(), sd_t, drv.grad

    raw32, g32, dd32 = autograd(torch.float32, False)     # plain fp32 PyTorch: its own branch decisions
    raw64, g64, dd64 = autograd(torch.float64, True)      # the yardstick: exact arithmetic on the HIP forward's side of every kink
    close(raw.reshape(-1, 16)[:, :15], raw64[:, :15].cpu().numpy(), 2e-3, 2e-3, "raw")
    off = W.canonical_offsets(arch)
    worst = ("", 0.0, 0.0)
    for k, (o, shape) in off.items():
        if ("nerf_mlps." in k and lvl not in k) or k.startswith("audNet_head"):
            continue                      # the other level's net gets no gradient; AudioNet is sahs_conditioning_backward's
        ref = g64[k].grad
        got = grad_flat[o:o + ref.numel()].view_as(ref).double()
        scale = float(ref.abs().max()) + 1e-30
        err = float((got - ref).abs().max()) / scale
        err32 = float((g32[k].grad.double() - ref).abs().max()) / scale
        if err > worst[1]:
            worst = (k, err, err32)
        # 2e-3, except for sums that cancel (the single fc_ambient bias under 15 octaves: plain fp32 autograd is off by 31 % there,
        # the HIP value by 1 %): never worse than half of what fp32 PyTorch makes of the same tensor
        assert err <= max(2e-3, 0.5 * err32), "%s: %.3e of scale (plain fp32 autograd, own branches: %.3e)" % (k, err, err32)
    print("worst: %s %.3e of scale (plain fp32 autograd with its own branch decisions: %.3e)" % worst)
    if arch != "audio":
        sc = float(dd64.abs().max())
        e, e32 = float((grad_cond[:76].double() - dd64).abs().max()) / sc, float((dd32.double() - dd64).abs().max()) / sc
        assert e <= max(2e-3, 0.5 * e32), (e, e32)



X3_XPRIME_BOUND, X3_W_BOUND = 4e-6, 6e-5      # ~4x observed (8.9e-7; 1.55e-5 on a scale of 1.2); the fp32 kernel: 1e-7


def test_mixed_precision_bf16_vs_fp32(ops, nf):
    """NeRFaceModel in mixed precision (precision "bf16": deformation nets with split bf16 operands, plain-bf16 radiance nets; DESIGN.md
    section 7b) against its fp32 path on the same rays, weights and draws: the deformed points agree to the split-operand accuracy, the
    radiance nets' raw output differs by bf16 rounding, and the rendered frame stays within the PSNR bound of the audio model's
    bf16 path.  Observed values are printed; the bounds are ~3x of them."""
    import json
    W = pkg("weights")
    d = dev()
    # high-dynamic-range weights with the density logit placed so that a ray spreads its weight (mean background weight ~0.25)
    fw = W.flatten_state_dict(W.hash_state_dict(0, -3.0, 10.0, model="nerface", hdr=True), model="nerface")
    flat = T(fw)
    packs = {p: ops.pack_weights(flat, ops.PRECISIONS[p], arch="nerface") for p in ("fp32", "bf16")}
    g = torch.Generator(device=d).manual_seed(7)
    frame = ops.fold_conditioning(flat, torch.randn(76, device=d, generator=g) * 0.5,
                                  T(np.concatenate([np.eye(3), [[0.0], [0.0], [0.5]]], 1).astype(np.float32)), arch="nerface")
    N, nc, nfine = 1500, 64, 64
    rays = torch.zeros(N, 8, device=d)
    rays[:, 2] = 0.5
    rays[:, 3:6] = torch.randn(N, 3, device=d, generator=g) * 0.15 + torch.tensor([0, 0, -1.0], device=d)
    rays[:, 6], rays[:, 7] = 0.2, 0.8
    bg = torch.cat([torch.rand(N, 3, device=d, generator=g), torch.ones(N, 1, device=d), torch.zeros(N, 11, device=d)], 1)
    t_rand, u = torch.rand(N, nc, device=d, generator=g), torch.rand(N, nfine, device=d, generator=g)
    out = {}
    for p in ("fp32", "bf16"):
        rows = torch.full((N, 36), float("nan"), device=d)
        ws = {}
        ops.render_rays_rows(packs[p], frame, rays, nc, nfine, rows, precision=ops.PRECISIONS[p], bg=bg, t_rand=t_rand, u=u, workspace=ws, arch="nerface")
        out[p] = (rows, ws["xw"].clone(), ws["raw"].clone(), ws["z_f"].clone())
        assert bool(torch.isfinite(rows).all()), p
    # coarse pass: same depths; the mixed path's deformation launch runs on the split-operand pipe (round 3), so its (x', w) are the fp32
    # launch's to ~16 bits per operand: x' = x + tanh(.) adds a small correction to an exact x, w is a raw network output
    dxw = (out["fp32"][1][:, :nc, :5] - out["bf16"][1][:, :nc, :5]).abs()
    seam = dict(xprime_max=float(dxw[..., :3].max()), w_max=float(dxw[..., 3:5].max()), w_scale=float(out["fp32"][1][:, :nc, 3:5].abs().max()))
    print(json.dumps(seam))
    assert seam["xprime_max"] <= X3_XPRIME_BOUND and seam["w_max"] <= X3_W_BOUND * max(1.0, seam["w_scale"]), seam
    a, b = out["fp32"][0], out["bf16"][0]
    mse = lambda x, y: float(((x - y) ** 2).mean())
    psnr = lambda m: -10.0 * np.log10(max(m, 1e-30))
    res = dict(psnr_rgb_coarse=psnr(mse(a[:, 0:3], b[:, 0:3])), psnr_rgb_fine=psnr(mse(a[:, 17:20], b[:, 17:20])),
               max_abs_rgb_fine=float((a[:, 17:20] - b[:, 17:20]).abs().max()), seg_max_abs=float((a[:, 20:32] - b[:, 20:32]).abs().max()),
               acc_max_abs=float((a[:, 33] - b[:, 33]).abs().max()), depth_rms=float(((a[:, 35] - b[:, 35]) ** 2).mean().sqrt()),
               w_bg_mean=float(a[:, 34].mean()))
    print(json.dumps(res))
    assert res["psnr_rgb_coarse"] >= 38.0 and res["psnr_rgb_fine"] >= 33.0, res
    assert res["acc_max_abs"] <= 1e-3 and 0.02 < res["w_bg_mean"] < 0.98, res
    # the B2 seam of a mixed-precision model: model(level, x) = split-operand deformation launch + bf16 radiance launch
    sahs = pkg()
    cfg = sahs.default_config("expression")
    m32 = sahs.NeRFaceModel(cfg).to(d).load_flat(fw).eval()
    m16 = sahs.NeRFaceModel(cfg, precision="bf16").to(d).load_flat(fw).eval()
    x = torch.cat([torch.rand(300, 3, device=d, generator=g) * 0.4 - 0.2, torch.randn(300, 3, device=d, generator=g)], 1)
    drv, pose = torch.randn(76, device=d, generator=g) * 0.5, T(np.concatenate([np.eye(3), [[0.0], [0.0], [0.5]]], 1).astype(np.float32))
    with torch.no_grad():
        r32, r16 = m32("fine", x, drv, pose), m16("fine", x, drv, pose)
    rel = float((r32 - r16).abs().max() / r32.abs().max())
    print(json.dumps(dict(seam_raw_rel_max=rel)))
    assert rel <= 0.1, rel


def test_static_bf16_vs_fp32(ops):
    """NeRFaceModel without deformation nets (config/expression/person_1.yml) on the bf16 kernel (the whole network is the radiance
    net, queried at the raw point) against its fp32 path: raw output within bf16 rounding, rendered frame PSNR."""
    import json
    W = pkg("weights")
    d = dev()
    g = torch.Generator(device=d).manual_seed(9)
    drv, pose = torch.randn(76, device=d, generator=g) * 0.5, T(np.concatenate([np.eye(3), [[0.0], [0.0], [0.5]]], 1).astype(np.float32))
    N, nc, nfine = 700, 64, 64
    rays = torch.zeros(N, 8, device=d)
    rays[:, 2] = 0.5
    rays[:, 3:6] = torch.randn(N, 3, device=d, generator=g) * 0.15 + torch.tensor([0, 0, -1.0], device=d)
    rays[:, 6], rays[:, 7] = 0.2, 0.8
    z = torch.sort(torch.rand(N, nc, device=d, generator=g) * 0.6 + 0.2, dim=1).values
    # high-dynamic-range weights; the density bias is placed from a first evaluation so that the mean density logit is ~2 (its spread is
    # ~3): rays then spread their weight over many samples instead of saturating or vanishing
    bias = 0.0
    for _ in range(2):
        fw = W.flatten_state_dict(W.hash_state_dict(0, bias, 10.0, model="nerface_static", hdr=True), model="nerface_static")
        flat = T(fw)
        frame = ops.fold_conditioning(flat, drv, pose, arch="nerface_static")
        probe = ops.field_forward(ops.pack_weights(flat, arch="nerface_static"), frame, 1, rays, z, arch="nerface_static")
        bias += 2.0 - float(probe[..., 15].mean())
    packs = {p: ops.pack_weights(flat, ops.PRECISIONS[p], arch="nerface_static") for p in ("fp32", "bf16")}
    r32 = ops.field_forward(packs["fp32"], frame, 1, rays, z, arch="nerface_static")
    r16 = ops.field_forward(packs["bf16"], frame, 1, rays, z, precision=ops.SAHS_BF16, arch="nerface_static")
    assert bool(torch.isfinite(r16).all())
    col = float((r32[..., :15] - r16[..., :15]).abs().max() / r32[..., :15].abs().max())
    sig = float((r32[..., 15] - r16[..., 15]).abs().max() / r32[..., 15].abs().max())
    bg = torch.cat([torch.rand(N, 3, device=d, generator=g), torch.ones(N, 1, device=d), torch.zeros(N, 11, device=d)], 1)
    t_rand, u = torch.rand(N, nc, device=d, generator=g), torch.rand(N, nfine, device=d, generator=g)
    o32 = ops.render_rays(packs["fp32"], frame, rays, nc, nfine, bg=bg, t_rand=t_rand, u=u, arch="nerface_static")
    o16 = ops.render_rays(packs["bf16"], frame, rays, nc, nfine, precision=ops.SAHS_BF16, bg=bg, t_rand=t_rand, u=u, arch="nerface_static")
    mse = float(((o32[3][:, :3] - o16[3][:, :3]) ** 2).mean())
    res = dict(raw_colour_rel_max=col, raw_sigma_rel_max=sig, psnr_rgb_fine=-10.0 * np.log10(max(mse, 1e-30)), w_bg_mean=float(o32[6].mean()))
    print(json.dumps(res))
    assert col <= 0.05 and sig <= 0.05 and res["psnr_rgb_fine"] >= 33.0 and 0.02 < res["w_bg_mean"] < 0.9, res
