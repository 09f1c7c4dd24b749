"""configs[4] on the MI355X: a 2048-ray training step through the HIP autograd op, checked against plain PyTorch autograd
of the eager restatement (oracle/torch_eager.py, itself pinned to the reference's golden gradients) and timed; then a few
optimiser steps of the drop-in training loop."""
import json
import os
import time

import numpy as np
import pytest
import torch

from conftest import REPO, golden_rand, golden_weights_kw, load_golden, pkg, yardstick

pytestmark = pytest.mark.gpu


def test_train_step_2048_rays_vs_eager_autograd(weights_mod):
    from oracle import torch_eager as TE
    sahs = pkg()
    dev = torch.device("cuda:0")
    cfg = sahs.default_config()
    sd_np = weights_mod.hash_state_dict(0, 8.0, 30.0)
    model = sahs.AudioFaceModel(cfg).to(dev).load_flat(weights_mod.flatten_state_dict(sd_np)).train()
    sd_t = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in sd_np.items()}
    field = TE.EagerField(sd_t)
    rng = np.random.default_rng(3)
    R = 2048
    audio = torch.from_numpy(rng.standard_normal((16, 29)).astype(np.float32)).to(dev)
    pose = torch.from_numpy(np.concatenate([np.eye(3), [[0.0], [0.0], [0.8]]], 1).astype(np.float32)).to(dev)
    ro = torch.zeros(R, 3, device=dev)
    ro[:, 2] = 0.8
    rd = torch.randn(R, 3, device=dev) * 0.15 + torch.tensor([0, 0, -1.0], device=dev)
    bg = torch.cat([torch.rand(R, 3, device=dev), torch.ones(R, 1, device=dev), torch.zeros(R, 11, device=dev)], 1)
    A, B = torch.randn(R, 15, device=dev), torch.randn(R, 15, device=dev)
    rand = [dict(t_rand=torch.rand(R, 64, device=dev), noise_c=torch.randn(R, 64, device=dev) * 0.1, u=torch.rand(R, 64, device=dev),
                 noise_f=torch.randn(R, 128, device=dev) * 0.1)]
    feed = [("rand", rand[0]["t_rand"]), ("randn", rand[0]["noise_c"] / 0.1), ("rand", rand[0]["u"]), ("randn", rand[0]["noise_f"] / 0.1)]

    def hip_step():
        log = list(feed)
        o_rand, o_randn = torch.rand, torch.randn
        torch.rand = lambda *a, **k: log.pop(0)[1]
        torch.randn = lambda *a, **k: log.pop(0)[1]
        try:
            outs = sahs.run_one_iter_of_nerf(0, 0, None, model, ro, rd, cfg, mode="train", driving=audio, pose=pose, background_prior=bg)
        finally:
            torch.rand, torch.randn = o_rand, o_randn
        loss = (outs[0] * A).sum() + (outs[3] * B).sum() + outs[7].sum() * 0.1
        model.zero_grad(set_to_none=True)
        loss.backward()
        return loss

    def eager_step():
        outs = TE.run_one_iter(field, ro, rd, cfg.dataset.near, cfg.dataset.far, audio, pose, bg=bg, rand=rand, perturb=True, noise_std=0.1)
        loss = (outs[0] * A).sum() + (outs[3] * B).sum() + outs[7].sum() * 0.1
        for v in sd_t.values():
            v.grad = None
        loss.backward()
        return loss

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            loss = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps, loss

    t_hip, l_hip = timed(hip_step, 3)
    t_eager, l_eager = timed(eager_step, 2)
    assert abs(float(l_hip) - float(l_eager)) <= 1e-3 * abs(float(l_eager)) + 1e-2
    worst = 0.0
    for k, p in model.named_parameters():
        ref = sd_t[k].grad
        scale = float(ref.abs().max()) + 1e-12
        err = float((p.grad - ref).abs().max()) / scale
        worst = max(worst, err)
        assert err <= 2e-2, "%s: %.3e of scale" % (k, err)
    res = dict(rays=R, hip_step_s=t_hip, eager_step_s=t_eager, hip_rays_per_s=R / t_hip, eager_rays_per_s=R / t_eager,
               speedup=t_eager / t_hip, worst_grad_err_rel_scale=worst)
    print(json.dumps(res))


@pytest.mark.parametrize("fused,prec", [(False, "bf16x3"), (True, "bf16x3"), (True, "fp32")])
def test_train_step_vs_reference_fixture(flat_weights, fused, prec):
    """fused: the objective evaluated by sahs_stage1_loss_forward and its gradient formed inside composite_backward_kernel
    (run_one_iter_of_nerf(..., _loss=...)); otherwise the torch statement of the loss modules + autograd.  Same bounds.
    prec: the arithmetic of the backward walk -- the default split-bf16 operands, or exact fp32 products (the reference's own; the fused
    walk's f32 kernels: field_bwd_chain_f32.hip, gemm_tn_jobs*_f32_kernel).
    f-2 pinned: one training step (train_stage_rays_auto.py:437-468) on the high-dynamic-range network -- train-mode render
    with the reference's captured draws, the loss recipe, the sample_prob feedback and backward through the HIP kernels --
    against what the REFERENCE computed on the same 32 rays (tests/golden/train_step_hdr.npz: its own MaskMSELoss /
    MaskCrossEntropyLoss with the script's weights, its own autograd), with the float64 run of the reference as yardstick.
    The reference's own fp32 gradients are ~1 % (norms) / 2.7 % (entries, of the tensor's scale) away from its float64 ones on this
    network ((leaky-)ReLU kinks); the HIP gradients must be within twice that of the float64 ones."""
    sahs, Tr = pkg(), pkg("training")
    dev = torch.device("cuda:0")
    g = load_golden("train_step_hdr")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    cfg = sahs.default_config()
    model = sahs.AudioFaceModel(cfg).to(dev).load_flat(flat_weights(**golden_weights_kw(g))).train()
    audio = T(g["audio"]).requires_grad_(True)
    log = golden_rand(g)
    o_rand, o_randn = torch.rand, torch.randn

    def feed(kind):
        def f(*a, **k):
            knd, arr = log.pop(0)
            assert knd == kind
            return T(arr)
        return f

    torch.rand, torch.randn = feed("rand"), feed("randn")
    try:
        extra = dict(_loss=(T(g["target"])[:, :3], T(g["mask"]), Tr.sample_prob_weights(dev))) if fused else {}
        outs = sahs.run_one_iter_of_nerf(12, 12, None, model, T(g["ro"]), T(g["rd"]), cfg, mode="train", driving=audio, pose=T(g["pose"]),
                                         background_prior=T(g["bg"]), inHead=T(g["mask"]), **extra)
    finally:
        torch.rand, torch.randn = o_rand, o_randn
    assert not log, "the driver must consume exactly the reference's random draws"
    for nm, o in zip(["rgb_c", "disp_c", "acc_c", "rgb_f", "disp_f", "acc_f", "w_bg", "depth_f"], outs):
        yardstick(o, g["out_" + nm], g["f64_" + nm], "hip train_step_hdr:" + nm, outlier_rays=0.0 if nm.endswith("_c") else 0.04, scale_floor=1.0)
    loss, prob, fine_mse = Tr.stage1_loss(outs[0], outs[3], T(g["target"]), T(g["mask"]))
    if fused:       # the kernel's value against the torch statement on the same maps, then use the kernel's
        assert len(outs) == 10
        st = outs[9]
        assert abs(float(outs[8]) - float(loss)) <= 2e-6 * abs(float(loss)), (float(outs[8]), float(loss))
        assert float((st[2:14] - prob.detach()).abs().max()) <= 2e-6
        assert abs(float(st[1]) - float(fine_mse)) <= 2e-6 * float(fine_mse)
        loss, prob = outs[8], st[2:14]
    l64, l32 = float(g["loss_f64"]), float(g["loss"])
    assert abs(float(loss) - l64) <= 3.0 * abs(l32 - l64) + 1e-5 * abs(l64), (float(loss), l32, l64)
    p64, p32 = g["sample_prob_f64"], g["sample_prob"].astype(np.float64)
    e_hip, e_ref = np.abs(prob.detach().cpu().numpy() - p64).max(), np.abs(p32 - p64).max()
    assert e_hip <= 3.0 * e_ref + 1e-6, (e_hip, e_ref)
    ops = pkg("ops")
    ops.backward_gemm_precision(prec)
    try:
        loss.backward()
        torch.cuda.synchronize()
    finally:
        ops.backward_gemm_precision("bf16x3")
    params = dict(model.named_parameters())
    names = [str(n) for n in g["grad_names"]]
    n32, n64 = g["grad_norms"], g["grad_norms_f64"]
    ref_norm_err = float(np.max(np.abs(n32 - n64) / n64))
    worst_norm, worst_entry, ref_entry = 0.0, 0.0, 0.0
    for k, r32, r64 in zip(names, n32, n64):
        gr = params[k].grad
        assert gr is not None, k
        worst_norm = max(worst_norm, abs(float(gr.double().norm()) - r64) / r64)
        for pre, sub in (("grad", False), ("gradsub", True)):
            if pre + "_" + k in g:
                ref32, ref64 = g[pre + "_" + k], g[pre + "64_" + k]
                mine = gr.reshape(-1)[::max(1, gr.numel() // 2048)].cpu().numpy() if sub else gr.cpu().numpy()
                scale = float(np.abs(ref64).max()) + 1e-30
                worst_entry = max(worst_entry, float(np.abs(mine.reshape(ref64.shape) - ref64).max()) / scale)
                ref_entry = max(ref_entry, float(np.abs(ref32 - ref64).max()) / scale)
    ga64 = g["grad_audio_f64"]
    e_audio = float(np.abs(audio.grad.cpu().numpy() - ga64).max() / np.abs(ga64).max())
    ref_audio = float(np.abs(g["grad_audio"] - ga64).max() / np.abs(ga64).max())
    res = dict(loss=float(loss), loss_ref=l32, loss_f64=l64, sample_prob_err=float(e_hip), sample_prob_err_ref=float(e_ref),
               grad_norm_err=worst_norm, grad_norm_err_ref=ref_norm_err, grad_entry_err=worst_entry, grad_entry_err_ref=ref_entry,
               grad_audio_err=e_audio, grad_audio_err_ref=ref_audio)
    print(json.dumps(res))
    assert worst_norm <= 2.0 * ref_norm_err + 1e-4, res
    assert worst_entry <= 2.0 * ref_entry + 1e-4, res
    assert e_audio <= 2.0 * ref_audio + 1e-4, res


@pytest.mark.parametrize("arch", ["audio", "nerface"])
def test_shared_deformation_training_matches_plain_chain(arch, weights_mod):
    """Training with the deformation nets evaluated once per depth (forward launches FIELD_ALL / FIELD_DEFORM / FIELD_RADIANCE with saved
    activations; backward cut at the (x', w) seam, the fine pass's seam gradient routed through the merge permutation into the coarse
    pass's backward) against the plain chain (whole network per level): identical outputs bit for bit, gradients equal up to the
    summation order of the float atomics."""
    sahs, ops = pkg(), pkg("ops")
    dev = torch.device("cuda:0")
    model_name = "audio" if arch == "audio" else "nerface"
    fw = weights_mod.flatten_state_dict(weights_mod.hash_state_dict(0, 2.0, 30.0, model=model_name, hdr=(arch == "audio")), model=model_name)
    gen = torch.Generator(device=dev).manual_seed(11)
    N, nc, nf = 193, 64, 128 if arch == "audio" else 64
    drv = torch.randn(16, 29, device=dev, generator=gen) if arch == "audio" else torch.randn(76, device=dev, generator=gen) * 0.5
    cam = 0.8 if arch == "audio" else 0.5
    pose = torch.from_numpy(np.concatenate([np.eye(3), [[0.0], [0.0], [cam]]], 1).astype(np.float32)).to(dev)
    near, far = (0.483771, 1.083771) if arch == "audio" else (0.2, 0.8)
    rays = torch.zeros(N, 8, device=dev)
    rays[:, 2] = cam
    rays[:, 3:6] = torch.randn(N, 3, device=dev, generator=gen) * 0.15 + torch.tensor([0, 0, -1.0], device=dev)
    rays[:, 6], rays[:, 7] = near, far
    bg = torch.cat([torch.rand(N, 3, device=dev, generator=gen), torch.ones(N, 1, device=dev), torch.zeros(N, 11, device=dev)], 1)
    t_rand, u = torch.rand(N, nc, device=dev, generator=gen), torch.rand(N, nf, device=dev, generator=gen)
    noise_c, noise_f = torch.randn(N, nc, device=dev, generator=gen) * 0.1, torch.randn(N, nc + nf, device=dev, generator=gen) * 0.1
    A = [torch.randn(N, 15, device=dev, generator=gen) for _ in range(2)]
    res = {}
    try:
        for share in (False, True):
            ops.RenderRaysFn.SHARE_DEFORMATION = share
            flat = torch.from_numpy(fw).to(dev).requires_grad_(True)
            d = drv.clone().requires_grad_(True)
            packed = ops.pack_weights(flat.detach(), arch=arch)
            outs = ops.RenderRaysFn.apply(flat, d, pose, rays, bg, t_rand, noise_c, u, noise_f, packed, nc, nf, False, False, arch)
            loss = (outs[0] * A[0]).sum() + (outs[3] * A[1]).sum() + 0.3 * outs[7].sum() + 0.2 * outs[6].sum() + 0.1 * outs[1].sum()
            loss.backward()
            res[share] = ([o.detach().clone() for o in outs], flat.grad.clone(), d.grad.clone())
    finally:
        ops.RenderRaysFn.SHARE_DEFORMATION = True
    for a, b in zip(res[False][0], res[True][0]):
        assert torch.equal(a, b)
    for k, nm in ((1, "parameters"), (2, "driving input")):
        a, b = res[False][k], res[True][k]
        scale = float(a.abs().max())
        assert scale > 0 and float((a - b).abs().max()) <= 2e-4 * scale, (nm, float((a - b).abs().max()), scale)
    # per parameter tensor too (a small tensor must not hide behind the global scale): norms within 1e-3
    off = 0
    for name, shape in weights_mod.canonical_spec(model_name):
        n = int(np.prod(shape))
        ga, gb = res[False][1][off:off + n], res[True][1][off:off + n]
        na = float(ga.norm())
        assert abs(na - float(gb.norm())) <= 1e-3 * na + 1e-12, (name, na, float(gb.norm()))
        off += n


def test_stage1_loss_kernel_vs_loss_modules():
    """sahs_stage1_loss_forward against the torch statement of the reference's loss modules (pinned to the reference's own classes by
    tests/golden/losses.npz on CPU): ragged ray count, an empty class, a one-ray class, coarse-only; and its gradient, formed inside
    composite_backward_kernel, against autograd of that statement through ops.CompositeFn."""
    ops, Tr, sahs = pkg("ops"), pkg("training"), pkg()
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(5)
    N, S = 777, 40
    cls = torch.randint(0, 12, (N,), device=dev, generator=gen)
    cls[cls == 5] = 4
    cls[cls == 11] = 10
    cls[3] = 11
    mask = torch.nn.functional.one_hot(cls, 12).float()
    target = torch.rand(N, 3, device=dev, generator=gen)
    w = Tr.sample_prob_weights(dev)
    maps = []
    for _ in range(2):
        m = torch.cat([torch.rand(N, 3, device=dev, generator=gen), torch.softmax(torch.randn(N, 12, device=dev, generator=gen) * 2, -1)], 1)
        maps.append(m)
    for mc, mf in ((maps[0], maps[1]), (maps[0], None)):
        st = ops.stage1_loss_forward(mc, mf, target, mask, w)
        loss, prob, mse = Tr.stage1_loss(mc, mf, target, mask)
        assert abs(float(st[0]) - float(loss)) <= 2e-6 * abs(float(loss))
        assert float((st[2:14] - prob).abs().max()) <= 1e-6 and abs(float(st[1]) - float(mse)) <= 2e-6 * float(mse)
        assert float(st[14 + 5]) == 1.0 and float(st[14 + 11]) == 1.0 and float(st[26]) == N        # empty class counts as 1
    # gradient: d loss / d raw through one compositing level, fused against autograd of the statement
    raw = (torch.randn(N, S, 16, device=dev, generator=gen) * 1.5).requires_grad_(True)
    z = torch.sort(torch.rand(N, S, device=dev, generator=gen) * 0.6 + 0.48, dim=1).values
    rays = torch.zeros(N, 8, device=dev)
    rays[:, 3:6] = torch.randn(N, 3, device=dev, generator=gen) * 0.2 + torch.tensor([0, 0, -1.0], device=dev)
    rgb = sahs.volume_render_radiance_field(raw, z, rays[:, 3:6], radiance_field_noise_std=0.0)[0]      # no prior: 15 sigmoid channels
    loss, _, _ = Tr.stage1_loss(rgb, None, target, mask)
    (3.0 * loss).backward()
    st = ops.stage1_loss_forward(rgb.detach(), None, target, mask, w)
    d_raw = ops.composite_backward(raw.detach(), z, rays, None, None, False, None, None, None, None, None,
                                   loss=(rgb.detach().contiguous(), target, mask, st, torch.tensor([3.0], device=dev)))
    scale = float(raw.grad.abs().max())
    assert float((d_raw - raw.grad).abs().max()) <= 2e-5 * scale, (float((d_raw - raw.grad).abs().max()), scale)


def test_training_loop_reduces_loss(weights_mod):
    sahs = pkg()
    Tr = pkg("training")
    dev = torch.device("cuda:0")
    cfg = sahs.default_config()
    cfg.nerf.train.num_random_rays = 512
    model = sahs.AudioFaceModel(cfg).to(dev).load_flat(weights_mod.flatten_state_dict(weights_mod.hash_state_dict(0, 8.0, 30.0))).train()
    opt = torch.optim.Adam(model.parameters(), lr=cfg.optimizer.lr)
    g = torch.Generator(device=dev).manual_seed(0)
    H = W = 32
    image = torch.rand(H, W, 3, device=dev, generator=g) * 0.2 + 0.4
    mask = torch.zeros(H, W, 12, device=dev)
    mask[..., 0] = 1.0
    mask[8:16, 8:16] = 0.0
    mask[8:16, 8:16, 7] = 1.0
    bgp = torch.cat([torch.rand(H, W, 3, device=dev, generator=g), torch.ones(H, W, 1, device=dev), torch.zeros(H, W, 11, device=dev)], -1)
    audio = torch.randn(16, 29, device=dev, generator=g)
    pose = torch.from_numpy(np.concatenate([np.eye(3), [[0.0], [0.0], [0.8]]], 1).astype(np.float32)).to(dev)
    intr = np.array([1200.0 * H / 512, 1200.0 * H / 512, 0.5, 0.5], np.float32)
    sp = torch.ones(12, device=dev) / 12
    torch.manual_seed(0)
    losses = []
    for step in range(8):
        r = Tr.train_step(model, opt, cfg, step, image, mask, pose, intr, audio, bgp, sp, generator=g)
        sp = r["sample_prob"]
        losses.append(r["loss"])
        assert np.isfinite(r["loss"]) and abs(float(sp.sum()) - 1) < 1e-5
    assert min(losses[4:]) < losses[0], losses


@pytest.mark.parametrize("S,use_bg,white", [(64, True, False), (128, True, False), (100, False, True)])
def test_composite_backward_vs_autograd(S, use_bg, white):
    """Seam-level: sahs_composite_backward against torch autograd of the eager compositing (volume_rendering_utils.py:7-78)."""
    from oracle import torch_eager as TE
    ops = pkg("ops")
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(S)
    N = 257
    raw = (torch.randn(N, S, 16, device=dev, generator=g) * 1.5)
    raw[..., 15] = raw[..., 15] * 6 + 1.0
    z = torch.sort(torch.rand(N, S, device=dev, generator=g) * 0.6 + 0.48, dim=1).values
    rays = torch.zeros(N, 8, device=dev)
    rays[:, 3:6] = torch.randn(N, 3, device=dev, generator=g) * 0.2 + torch.tensor([0, 0, -1.0], device=dev)
    bg = torch.rand(N, 15, device=dev, generator=g) if use_bg else None
    noise = torch.randn(N, S, device=dev, generator=g) * 0.1
    gr = [torch.randn(N, 15, device=dev, generator=g)] + [torch.randn(N, device=dev, generator=g) for _ in range(4)]
    x = raw.clone().requires_grad_(True)
    xin = x
    if use_bg:
        xin = torch.cat((x[:, :-1], torch.cat((bg, x[:, -1, -1:]), dim=-1).unsqueeze(1)), dim=1)
    rgb, disp, acc, w, depth = TE.volume_render(xin, z, rays[:, 3:6], noise, white, use_bg)
    loss = (rgb * gr[0]).sum() + (disp * gr[1]).sum() * 1e-3 + (acc * gr[2]).sum() + (depth * gr[3]).sum() + (w[:, -1] * gr[4]).sum()
    loss.backward()
    d_raw = ops.composite_backward(raw, z, rays, noise, bg, white, gr[0], gr[1] * 1e-3, gr[2], gr[3], gr[4])
    ref = x.grad
    scale = float(ref.abs().max())
    err = float((d_raw - ref).abs().max())
    assert err <= 2e-4 * scale + 1e-6, (err, scale)


@pytest.mark.parametrize("use_bg,white,noise_std", [(True, False, 0.0), (False, True, 0.1), (True, False, 0.1)])
def test_volume_render_seam_is_differentiable(use_bg, white, noise_std):
    """Seam B3: volume_render_radiance_field(...) is differentiable w.r.t. radiance_field like the reference's
    (volume_rendering_utils.py:7-78), including the gradient of the full `weights` output and of the verbatim last sample."""
    from oracle import torch_eager as TE
    sahs = pkg()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(3)
    N, S = 65, 96
    raw = torch.randn(N, S, 16, device=dev, generator=g) * 1.5
    raw[..., 15] = raw[..., 15] * 6 + 1.0
    if use_bg:
        raw[:, -1, :15] = torch.rand(N, 15, device=dev, generator=g)
    z = torch.sort(torch.rand(N, S, device=dev, generator=g) * 0.6 + 0.48, dim=1).values
    rd = torch.randn(N, 3, device=dev, generator=g) * 0.2 + torch.tensor([0, 0, -1.0], device=dev)
    gr = [torch.randn(N, 15, device=dev, generator=g), torch.randn(N, device=dev, generator=g) * 1e-3, torch.randn(N, device=dev, generator=g),
          torch.randn(N, S, device=dev, generator=g), torch.randn(N, device=dev, generator=g)]
    noise = torch.randn(N, S, device=dev, generator=g)
    x = raw.clone().requires_grad_(True)
    o_randn = torch.randn
    torch.randn = lambda *a, **k: noise
    try:
        outs = sahs.volume_render_radiance_field(x, z, rd, noise_std, white, x[:, -1, :15] if use_bg else None)
    finally:
        torch.randn = o_randn
    sum(((o * w).sum() for o, w in zip(outs, gr))).backward()
    y = raw.clone().requires_grad_(True)
    ref = TE.volume_render(y, z, rd, noise * noise_std if noise_std > 0 else None, white, use_bg)
    sum(((o * w).sum() for o, w in zip(ref, gr))).backward()
    for a, b in zip(outs, ref):
        assert torch.allclose(a, b, rtol=2e-4, atol=2e-5)
    scale = float(y.grad.abs().max())
    assert float((x.grad - y.grad).abs().max()) <= 2e-4 * scale, float((x.grad - y.grad).abs().max()) / scale


def test_model_seam_is_differentiable(weights_mod):
    """Seam B2: model(level, x, driving, pose) back-propagates to the parameters and the audio window (ops.FieldFn), so a driver
    written against the reference's seams (run_network -> model -> volume_render_radiance_field) trains through the HIP kernels."""
    from oracle import torch_eager as TE
    sahs = pkg()
    dev = torch.device("cuda:0")
    cfg = sahs.default_config()
    sd_np = weights_mod.hash_state_dict(0, 8.0, 30.0)
    model = sahs.AudioFaceModel(cfg).to(dev).load_flat(weights_mod.flatten_state_dict(sd_np)).train()
    g = torch.Generator(device=dev).manual_seed(9)
    P = 64 * 6
    x = torch.cat([torch.rand(P, 3, device=dev, generator=g) * 0.5 - 0.25, torch.randn(P, 3, device=dev, generator=g) * 0.2], 1)
    audio = torch.randn(16, 29, device=dev, generator=g).requires_grad_(True)
    pose = torch.from_numpy(np.concatenate([np.eye(3), [[0.0], [0.0], [0.8]]], 1).astype(np.float32)).to(dev)
    wgt = torch.randn(P, 16, device=dev, generator=g)
    raw = model("coarse", x, audio, pose, None)
    (raw * wgt).sum().backward()
    sd_t = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in sd_np.items()}
    a2 = audio.detach().clone().requires_grad_(True)
    ref = TE.EagerField(sd_t).forward("coarse", x, a2, pose)
    (ref * wgt).sum().backward()
    assert torch.allclose(raw, ref, rtol=2e-3, atol=2e-3)
    for k, p in model.named_parameters():
        r = sd_t[k].grad
        if r is None:
            assert p.grad is None or not bool(p.grad.any()), k     # the fine net gets no gradient from a coarse query
            continue
        assert float((p.grad - r).abs().max()) <= 3e-2 * float(r.abs().max()) + 1e-9, k
    assert float((audio.grad - a2.grad).abs().max()) <= 3e-2 * float(a2.grad.abs().max())


def test_blockwise_backward_matches_kept_activations(weights_mod, monkeypatch):
    """Ray chunks larger than RenderRaysFn.BLOCK_RAYS keep only the depths and re-run the field block by block in backward; the
    gradients must equal those of the path that keeps the activations from the forward (same kernels, other summation split)."""
    sahs = pkg()
    ops = pkg("ops")
    dev = torch.device("cuda:0")
    cfg = sahs.default_config()
    fw = weights_mod.flatten_state_dict(weights_mod.hash_state_dict(0, 8.0, 30.0))
    g = torch.Generator(device=dev).manual_seed(21)
    R = 300
    audio = torch.randn(16, 29, device=dev, generator=g)
    pose = torch.from_numpy(np.concatenate([np.eye(3), [[0.0], [0.0], [0.8]]], 1).astype(np.float32)).to(dev)
    ro = torch.zeros(R, 3, device=dev)
    ro[:, 2] = 0.8
    rd = torch.randn(R, 3, device=dev, generator=g) * 0.15 + torch.tensor([0, 0, -1.0], device=dev)
    bg = torch.cat([torch.rand(R, 3, device=dev, generator=g), torch.ones(R, 1, device=dev), torch.zeros(R, 11, device=dev)], 1)
    A, B = torch.randn(R, 15, device=dev, generator=g), torch.randn(R, 15, device=dev, generator=g)
    draws = [torch.rand(R, 64, device=dev, generator=g), torch.randn(R, 64, device=dev, generator=g), torch.rand(R, 64, device=dev, generator=g),
             torch.randn(R, 128, device=dev, generator=g)]

    def grads(block):
        monkeypatch.setattr(ops.RenderRaysFn, "BLOCK_RAYS", block)
        model = sahs.AudioFaceModel(cfg).to(dev).load_flat(fw).train()
        a = audio.clone().requires_grad_(True)
        log = list(draws)
        o_rand, o_randn = torch.rand, torch.randn
        torch.rand = lambda *x, **k: log.pop(0)
        torch.randn = lambda *x, **k: log.pop(0)
        try:
            outs = sahs.run_one_iter_of_nerf(0, 0, None, model, ro, rd, cfg, mode="train", driving=a, pose=pose, background_prior=bg)
        finally:
            torch.rand, torch.randn = o_rand, o_randn
        ((outs[0] * A).sum() + (outs[3] * B).sum() + outs[7].sum() * 0.1).backward()
        return [o.detach() for o in outs], {k: p.grad.clone() for k, p in model.named_parameters()}, a.grad.clone()

    o_k, g_k, a_k = grads(4096)      # activations kept from the forward
    o_b, g_b, a_b = grads(128)       # 300 rays in blocks of 128: recomputed in backward
    for x, y in zip(o_k, o_b):
        assert torch.equal(x, y)
    for k in g_k:
        scale = float(g_k[k].abs().max()) + 1e-12
        assert float((g_k[k] - g_b[k]).abs().max()) <= 1e-4 * scale, k
    assert float((a_k - a_b).abs().max()) <= 1e-4 * float(a_k.abs().max())


def test_model_seam_gradients_in_blocks(weights_mod, monkeypatch):
    """ops.FieldFn cuts a large batch into blocks in FORWARD (the saved activations are plane-per-layer over the P of one forward
    call, so they cannot be row-sliced afterwards): gradients of a 3-block evaluation equal those of the single-block one."""
    sahs, ops = pkg(), pkg("ops")
    dev = torch.device("cuda:0")
    cfg = sahs.default_config()
    fw = weights_mod.flatten_state_dict(weights_mod.hash_state_dict(0, 8.0, 30.0))
    g = torch.Generator(device=dev).manual_seed(4)
    P = 700
    x = torch.cat([torch.rand(P, 3, device=dev, generator=g) * 0.5 - 0.25, torch.randn(P, 3, device=dev, generator=g) * 0.2], 1)
    pose = torch.from_numpy(np.concatenate([np.eye(3), [[0.0], [0.0], [0.8]]], 1).astype(np.float32)).to(dev)
    audio0 = torch.randn(16, 29, device=dev, generator=g)
    wgt = torch.randn(P, 16, device=dev, generator=g)

    def run(block):
        monkeypatch.setattr(ops.FieldFn, "BLOCK", block)
        model = sahs.AudioFaceModel(cfg).to(dev).load_flat(fw).train()
        a = audio0.clone().requires_grad_(True)
        raw = model("fine", x, a, pose, None)
        (raw * wgt).sum().backward()
        return raw.detach(), {k: (p.grad.clone() if p.grad is not None else None) for k, p in model.named_parameters()}, a.grad.clone()

    r1, g1, a1 = run(2_000_000)
    r3, g3, a3 = run(256)             # 700 points in blocks of 256
    assert torch.equal(r1, r3)
    for k in g1:
        if g1[k] is None:
            assert g3[k] is None or not bool(g3[k].any()), k
            continue
        assert float((g1[k] - g3[k]).abs().max()) <= 1e-4 * (float(g1[k].abs().max()) + 1e-12), k
    assert float((a1 - a3).abs().max()) <= 1e-4 * float(a1.abs().max())


def test_coarse_only_training_gradients(weights_mod):
    """nerf.train.num_fine = 0 (train_utils.py:148-149: the 8-tuple then carries the COARSE pass's weights[:, -1] and depth):
    backward runs, and gradients of a loss on (rgb_coarse, weights[:, -1], depth) match autograd of the eager restatement."""
    from oracle import torch_eager as TE
    sahs = pkg()
    dev = torch.device("cuda:0")
    cfg = sahs.default_config()
    cfg.nerf.train.num_fine = 0
    sd_np = weights_mod.hash_state_dict(0, 8.0, 30.0)
    model = sahs.AudioFaceModel(cfg).to(dev).load_flat(weights_mod.flatten_state_dict(sd_np)).train()
    g = torch.Generator(device=dev).manual_seed(12)
    R = 96
    audio = torch.randn(16, 29, device=dev, generator=g)
    pose = torch.from_numpy(np.concatenate([np.eye(3), [[0.0], [0.0], [0.8]]], 1).astype(np.float32)).to(dev)
    ro = torch.zeros(R, 3, device=dev)
    ro[:, 2] = 0.8
    rd = torch.randn(R, 3, device=dev, generator=g) * 0.15 + torch.tensor([0, 0, -1.0], device=dev)
    bg = torch.cat([torch.rand(R, 3, device=dev, generator=g), torch.ones(R, 1, device=dev), torch.zeros(R, 11, device=dev)], 1)
    A, wl, wd = torch.randn(R, 15, device=dev, generator=g), torch.randn(R, device=dev, generator=g), torch.randn(R, device=dev, generator=g)
    t_rand, noise = torch.rand(R, 64, device=dev, generator=g), torch.randn(R, 64, device=dev, generator=g)
    log = [t_rand, noise]
    o_rand, o_randn = torch.rand, torch.randn
    torch.rand = lambda *a, **k: log.pop(0)
    torch.randn = lambda *a, **k: log.pop(0)
    try:
        outs = sahs.run_one_iter_of_nerf(0, 0, None, model, ro, rd, cfg, mode="train", driving=audio, pose=pose, background_prior=bg)
    finally:
        torch.rand, torch.randn = o_rand, o_randn
    assert not log and outs[3] is None
    ((outs[0] * A).sum() + (outs[6] * wl).sum() + (outs[7] * wd).sum()).backward()
    # eager: coarse pass only
    sd_t = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in sd_np.items()}
    field = TE.EagerField(sd_t)
    near, far = float(cfg.dataset.near), float(cfg.dataset.far)
    rays = torch.cat([ro, rd, torch.full((R, 1), near, device=dev), torch.full((R, 1), far, device=dev)], 1)
    t = torch.linspace(0.0, 1.0, 64, device=dev)
    z = (near * (1.0 - t) + far * t).expand(R, 64)
    mids = 0.5 * (z[..., 1:] + z[..., :-1])
    upper, lower = torch.cat((mids, z[..., -1:]), -1), torch.cat((z[..., :1], mids), -1)
    z = lower + (upper - lower) * t_rand
    pts = ro[:, None, :] + rd[:, None, :] * z[..., None]
    raw = TE.run_network(field, "coarse", pts, rays, 131072, audio, pose)
    raw = torch.cat((raw[:, :-1], torch.cat((bg, raw[:, -1, -1:]), dim=-1).unsqueeze(1)), dim=1)
    rgb, disp, acc, w, depth = TE.volume_render(raw, z, rd, noise * 0.1, False, True)
    for a, b in ((outs[0], rgb), (outs[6], w[:, -1]), (outs[7], depth)):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5)
    ((rgb * A).sum() + (w[:, -1] * wl).sum() + (depth * wd).sum()).backward()
    for k, p in model.named_parameters():
        r = sd_t[k].grad
        if r is None or not bool(r.any()):
            assert p.grad is None or not bool(p.grad.any()), k     # the fine net is never evaluated
            continue
        assert float((p.grad - r).abs().max()) <= 2e-2 * float(r.abs().max()) + 1e-9, k


@pytest.mark.parametrize("arch", ["audio", "nerface", "nerface_static"])
def test_backward_gemms_bf16x3_vs_f32(arch, weights_mod):
    """The backward's dense-layer GEMMs on the bf16 matrix pipe with split operands (the default, include/sahs_nerf.h:
    sahs_backward_gemm_precision) against the same kernels' f32-MFMA form on the same saved activations and upstream gradient: every
    parameter gradient, the seam gradient and the conditioning gradient within 1e-4 of the tensor's largest entry (observed values are
    printed; products carry ~1e-5 relative error, sums of them less) -- two orders below the ReLU-kink noise between two correct fp32
    forwards (1e-2, test_gradients_vs_golden).  Every level and part of the split walk, ragged sample count."""
    ops = pkg("ops")
    W = weights_mod
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(7)
    model_kw = {} if arch == "audio" else dict(model=arch)
    flat = torch.from_numpy(W.flatten_state_dict(W.hash_state_dict(0, 2.0, 30.0, hdr=True, **model_kw), **model_kw)).to(dev)
    packed = ops.pack_weights(flat, arch=arch)
    driving = torch.randn(16, 29, device=dev, generator=gen) if arch == "audio" else torch.randn(76, device=dev, generator=gen) * 0.5
    near, far, cam = (0.48, 1.08, 0.8) if arch == "audio" else (0.2, 0.8, 0.5)
    pose = torch.from_numpy(np.concatenate([np.eye(3), [[0.0], [0.0], [cam]]], 1).astype(np.float32)).to(dev)
    frame = ops.fold_conditioning(flat, driving, pose, arch=arch)
    N, S = 37, 77                 # 2849 samples: ragged against the 128-row tiles and the 512-sample slabs
    rays = torch.zeros(N, 8, device=dev)
    rays[:, 2] = cam
    rays[:, 3:6] = torch.randn(N, 3, device=dev, generator=gen) * 0.15 + torch.tensor([0, 0, -1.0], device=dev)
    z = torch.sort(torch.rand(N, S, device=dev, generator=gen) * (far - near) + near, dim=1).values
    d_raw = torch.randn(N * S, 16, device=dev, generator=gen)
    assert ops.backward_gemm_precision() == "bf16x3"
    worst = {}
    try:
        for level in (0, 1):
            raw, act = ops.field_forward_save(packed, frame, level, rays, z, arch)
            res = {}
            for prec in ("fp32", "bf16x3"):
                ops.backward_gemm_precision(prec)
                gf, gc = torch.zeros_like(flat), torch.zeros(128, device=dev)
                ops.field_backward(flat, frame, level, act, d_raw, gf, gc, arch)
                res[prec] = (gf, gc)
            off = W.canonical_offsets(arch)
            for k, (o, shape) in off.items():
                n = int(np.prod(shape))
                a, b = res["fp32"][0][o:o + n], res["bf16x3"][0][o:o + n]
                scale = float(a.abs().max())
                if scale == 0.0:
                    assert float(b.abs().max()) == 0.0, k
                    continue
                err = float((a - b).abs().max()) / scale
                worst[k] = max(worst.get(k, 0.0), err)
            cs = float(res["fp32"][1].abs().max()) + 1e-30
            worst["grad_cond"] = max(worst.get("grad_cond", 0.0), float((res["fp32"][1] - res["bf16x3"][1]).abs().max()) / cs)
    finally:
        ops.backward_gemm_precision("bf16x3")
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:4]
    print(arch, "bf16x3 vs f32 backward GEMMs, worst |delta| / scale:", ", ".join("%s %.2e" % kv for kv in top))
    assert top[0][1] <= 1e-4, top


@pytest.mark.parametrize("arch,with_loss", [("audio", False), ("audio", True), ("nerface", False)])
def test_two_stream_backward_matches_one_stream(arch, with_loss, weights_mod, monkeypatch):
    """The backward of a kept, shared training batch issues its independent walks pairwise on two streams (ops.RenderRaysFn.backward);
    SAHS_BWD_ONE_STREAM=1 (read per call) selects the serial order.  Same forward, same upstream gradient (plain-gradient form and the
    fused-loss form): every gradient must agree to the order of the float atomics -- a missing stream edge or a non-atomic add into the
    shared gradient buffers would show as an error of the size of a whole walk's contribution."""
    ops = pkg("ops")
    dev = torch.device("cuda:0")
    model_name = "audio" if arch == "audio" else "nerface"
    fw = weights_mod.flatten_state_dict(weights_mod.hash_state_dict(0, 2.0, 30.0, model=model_name, hdr=(arch == "audio")), model=model_name)
    gen = torch.Generator(device=dev).manual_seed(23)
    N, nc, nf = 517, 64, 64
    drv = torch.randn(16, 29, device=dev, generator=gen) if arch == "audio" else torch.randn(76, device=dev, generator=gen) * 0.5
    cam = 0.8 if arch == "audio" else 0.5
    pose = torch.from_numpy(np.concatenate([np.eye(3), [[0.0], [0.0], [cam]]], 1).astype(np.float32)).to(dev)
    near, far = (0.483771, 1.083771) if arch == "audio" else (0.2, 0.8)
    rays = torch.zeros(N, 8, device=dev)
    rays[:, 2] = cam
    rays[:, 3:6] = torch.randn(N, 3, device=dev, generator=gen) * 0.15 + torch.tensor([0, 0, -1.0], device=dev)
    rays[:, 6], rays[:, 7] = near, far
    bg = torch.cat([torch.rand(N, 3, device=dev, generator=gen), torch.ones(N, 1, device=dev), torch.zeros(N, 11, device=dev)], 1)
    t_rand, u = torch.rand(N, nc, device=dev, generator=gen), torch.rand(N, nf, device=dev, generator=gen)
    noise_c, noise_f = torch.randn(N, nc, device=dev, generator=gen) * 0.1, torch.randn(N, nc + nf, device=dev, generator=gen) * 0.1
    A = [torch.randn(N, 15, device=dev, generator=gen) for _ in range(2)]
    target = torch.rand(N, 3, device=dev, generator=gen)
    mask = torch.zeros(N, 12, device=dev)
    mask.scatter_(1, torch.randint(0, 12, (N, 1), device=dev, generator=gen), 1.0)
    cw = pkg("training").sample_prob_weights(dev)
    res = {}
    ops.fused_backward(False)        # (the fused walk of the audio model is issued on one stream; this pins the per-layer walks' two-stream issue)
    for one in (True, False):
        if one:
            monkeypatch.setenv("SAHS_BWD_ONE_STREAM", "1")
        else:
            monkeypatch.delenv("SAHS_BWD_ONE_STREAM", raising=False)
        flat = torch.from_numpy(fw).to(dev).requires_grad_(True)
        d = drv.clone().requires_grad_(True)
        packed = ops.pack_weights(flat.detach(), arch=arch)
        extra = (target, mask, cw) if with_loss else ()
        outs = ops.RenderRaysFn.apply(flat, d, pose, rays, bg, t_rand, noise_c, u, noise_f, packed, nc, nf, False, False, arch, *extra)
        if with_loss:
            loss = outs[8]
        else:
            loss = (outs[0] * A[0]).sum() + (outs[3] * A[1]).sum() + 0.3 * outs[7].sum() + 0.2 * outs[6].sum() + 0.1 * outs[1].sum()
        loss.backward()
        torch.cuda.synchronize()
        res[one] = (flat.grad.clone(), d.grad.clone())
    ops.fused_backward(True)
    for k, nm in ((0, "parameters"), (1, "driving input")):
        a, b = res[True][k], res[False][k]
        scale = float(a.abs().max())
        assert scale > 0 and float((a - b).abs().max()) <= 2e-6 * scale, (nm, float((a - b).abs().max()), scale)
    off = 0
    for name, shape in weights_mod.canonical_spec(model_name):      # per tensor: a small tensor must not hide behind the global scale
        n = int(np.prod(shape))
        ga, gb = res[True][0][off:off + n], res[False][0][off:off + n]
        sc = float(ga.abs().max())
        assert float((ga - gb).abs().max()) <= 2e-5 * sc + 1e-12, (name, float((ga - gb).abs().max()), sc)
        off += n


@pytest.mark.parametrize("prec,N", [("fp32", 2048), ("fp32", 1237), ("bf16x3", 2048)])
def test_fused_walk_at_step_size_vs_per_layer(prec, N, weights_mod):
    """The fused walk at the size of a training step (2,048 rays x (64 + 128) evaluations: 262,144 fine samples, every weight-gradient unit cut
    into many ranges by the item plans, several items per workgroup; 1,237 rays: ranges that do not divide the samples) through the
    autograd op, against the per-layer walk of the same arithmetic on the same forward: parameter and driving-input gradients within
    2e-5 of scale in fp32 products (another order of sums), 1e-4 with split-bf16 operands."""
    ops = pkg("ops")
    dev = torch.device("cuda:0")
    fw = weights_mod.flatten_state_dict(weights_mod.hash_state_dict(0, 2.0, 30.0, hdr=True))
    gen = torch.Generator(device=dev).manual_seed(29)
    nc, nf = 64, 64
    drv = torch.randn(16, 29, device=dev, generator=gen)
    pose = torch.from_numpy(np.concatenate([np.eye(3), [[0.0], [0.0], [0.8]]], 1).astype(np.float32)).to(dev)
    rays = torch.zeros(N, 8, device=dev)
    rays[:, 2] = 0.8
    rays[:, 3:6] = torch.randn(N, 3, device=dev, generator=gen) * 0.15 + torch.tensor([0, 0, -1.0], device=dev)
    rays[:, 6], rays[:, 7] = 0.483771, 1.083771
    bg = torch.cat([torch.rand(N, 3, device=dev, generator=gen), torch.ones(N, 1, device=dev), torch.zeros(N, 11, device=dev)], 1)
    t_rand, u = torch.rand(N, nc, device=dev, generator=gen), torch.rand(N, nf, device=dev, generator=gen)
    noise_c, noise_f = torch.randn(N, nc, device=dev, generator=gen) * 0.1, torch.randn(N, nc + nf, device=dev, generator=gen) * 0.1
    target = torch.rand(N, 3, device=dev, generator=gen)
    mask = torch.zeros(N, 12, device=dev)
    mask.scatter_(1, torch.randint(0, 12, (N, 1), device=dev, generator=gen), 1.0)
    cw = pkg("training").sample_prob_weights(dev)
    res = {}
    try:
        ops.backward_gemm_precision(prec)
        for fused in (False, True):
            ops.fused_backward(fused)
            flat = torch.from_numpy(fw).to(dev).requires_grad_(True)
            d = drv.clone().requires_grad_(True)
            packed = ops.pack_weights(flat.detach())
            outs = ops.RenderRaysFn.apply(flat, d, pose, rays, bg, t_rand, noise_c, u, noise_f, packed, nc, nf, False, False, "audio", target, mask, cw)
            outs[8].backward()
            torch.cuda.synchronize()
            res[fused] = (flat.grad.clone(), d.grad.clone())
    finally:
        ops.backward_gemm_precision("bf16x3")
        ops.fused_backward(True)
    tol = 2e-5 if prec == "fp32" else 1e-4
    worst = {}
    off = 0
    for name, shape in weights_mod.canonical_spec("audio"):      # per tensor: a small tensor must not hide behind the global scale
        n = int(np.prod(shape))
        a, b = res[True][0][off:off + n], res[False][0][off:off + n]
        sc = float(b.abs().max())
        if sc > 0:
            worst[name] = float((a - b).abs().max()) / sc
        else:
            assert float(a.abs().max()) == 0.0, name
        off += n
    worst["driving input"] = float((res[True][1] - res[False][1]).abs().max()) / float(res[False][1].abs().max())
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:4]
    print("fused vs per-layer walk at step size (%s, %d rays), worst |delta| / scale:" % (prec, N), ", ".join("%s %.2e" % kv for kv in top))
    assert top[0][1] <= tol, top


def test_fused_backward_vs_per_layer_walks(weights_mod):
    """The fused backward walk (one sample-major data-gradient chain + one weight-gradient launch per part, masks from the sign bits of
    the saving forward: include/sahs_nerf.h, sahs_model_field_backward_fused) against the per-layer walk on the SAME saved activations and
    upstream gradients -- its split-operand form (same arithmetic, other summation order) and its f32-MFMA form (the reference's
    precision) -- and the fused walk in exact fp32 products against the per-layer f32 walk (1e-5): every parameter gradient, the conditioning gradient and the seam gradients within 1e-4 of the tensor's largest entry.
    Every part the training step uses: radiance / deformation parts of a radiance-only and a deformation-only save, both parts of a
    whole-network save (full_act), and part 3.  Ragged sample counts (1,480 and 2,849 against 128-sample tiles and 1,024-sample ranges)."""
    ops = pkg("ops")
    W = weights_mod
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(19)
    flat = torch.from_numpy(W.flatten_state_dict(W.hash_state_dict(0, 2.0, 30.0, hdr=True))).to(dev)
    packed = ops.pack_weights(flat)
    driving = torch.randn(16, 29, device=dev, generator=gen)
    near, far, cam = 0.48, 1.08, 0.8
    pose = torch.from_numpy(np.concatenate([np.eye(3), [[0.0], [0.0], [cam]]], 1).astype(np.float32)).to(dev)
    frame = ops.fold_conditioning(flat, driving, pose)
    N, nc, nf = 37, 40, 37
    Sf = nc + nf
    rays = torch.zeros(N, 8, device=dev)
    rays[:, 2] = cam
    rays[:, 3:6] = torch.randn(N, 3, device=dev, generator=gen) * 0.15 + torch.tensor([0, 0, -1.0], device=dev)
    zs = lambda S: torch.sort(torch.rand(N, S, device=dev, generator=gen) * (far - near) + near, dim=1).values
    z_c, z_new = zs(nc), zs(nf)
    xw = torch.empty(N, Sf, 8, device=dev)
    src = torch.stack([torch.randperm(Sf, device=dev, generator=gen) for _ in range(N)]).to(torch.int32)
    sb = lambda samples, mode: ops.alloc_sign_bits(samples, mode, "audio", dev)
    bits_c, bits_d, bits_r = sb(N * nc, ops.FIELD_ALL), sb(N * nf, ops.FIELD_DEFORM), sb(N * Sf, ops.FIELD_RADIANCE)
    assert bits_c is not None and bits_c.shape[1] == bits_d.shape[1] + bits_r.shape[1]
    raw_c, act_c = ops.field_forward_split_save(packed, frame, 0, ops.FIELD_ALL, rays, xw, z=z_c, bits=bits_c)
    _, act_d = ops.field_forward_split_save(packed, frame, 1, ops.FIELD_DEFORM, rays, xw, z=z_new, xw_col0=nc, bits=bits_d)
    raw_f, act_r = ops.field_forward_split_save(packed, frame, 1, ops.FIELD_RADIANCE, rays, xw, src=src, bits=bits_r)
    d_raw_c = torch.randn(N * nc, 16, device=dev, generator=gen)
    d_raw_f = torch.randn(N * Sf, 16, device=dev, generator=gen)
    seam = lambda P_: torch.randn(P_, 8, device=dev, generator=gen) * torch.tensor([1, 1, 1, 0, 1, 1, 0, 0.0], device=dev)
    xwg_new, xwg_c = seam(N * nf), seam(N * nc)

    def run(fused, prec):
        ops.backward_gemm_precision(prec)
        ops.fused_backward(fused)
        out = {}
        for name in ("split", "whole"):
            gf, gc = torch.zeros_like(flat), torch.zeros(128, device=dev)
            g_f = ops.field_backward_split(flat, frame, 1, ops.FIELD_RADIANCE, act_r, gf, gc, d_raw=d_raw_f, bits=bits_r)
            ops.field_backward_split(flat, frame, 1, ops.FIELD_DEFORM, act_d, gf, gc, xw_grad_in=xwg_new, bits=bits_d)
            if name == "split":
                g_c = ops.field_backward_split(flat, frame, 0, ops.FIELD_RADIANCE, act_c, gf, gc, d_raw=d_raw_c, full_act=True, bits=bits_c)
                ops.field_backward_split(flat, frame, 0, ops.FIELD_DEFORM, act_c, gf, gc, xw_grad_in=xwg_c + g_c, full_act=True, bits=bits_c)
            else:
                g_c = torch.zeros(1, device=dev)
                ops.field_backward_split(flat, frame, 0, 3, act_c, gf, gc, d_raw=d_raw_c, xw_grad_in=xwg_c, bits=bits_c)
            out[name] = (gf, gc, g_f, g_c)
        torch.cuda.synchronize()
        return out

    try:
        ref32 = run(False, "fp32")
        refx3 = run(False, "bf16x3")
        fused = run(True, "bf16x3")
        fused32 = run(True, "fp32")      # the fused walk in exact fp32 products (field_bwd_chain_f32.hip, gemm_tn_jobs*_f32_kernel)
    finally:
        ops.backward_gemm_precision("bf16x3")
        ops.fused_backward(True)
    off = W.canonical_offsets("audio")
    worst = {}
    for mine, other, tag in ((fused, refx3, "x3"), (fused, ref32, "f32"), (fused32, ref32, "f32 fused vs f32")):
        for name in ("split", "whole"):
            a, b = mine[name], other[name]
            for k, (o, shape) in off.items():
                n = int(np.prod(shape))
                scale = float(b[0][o:o + n].abs().max())
                if scale == 0.0:
                    assert float(a[0][o:o + n].abs().max()) == 0.0, k
                    continue
                worst[(tag, name, k)] = float((a[0][o:o + n] - b[0][o:o + n]).abs().max()) / scale
            worst[(tag, name, "grad_cond")] = float((a[1] - b[1]).abs().max()) / (float(b[1].abs().max()) + 1e-30)
            worst[(tag, name, "seam_fine")] = float((a[2] - b[2]).abs().max()) / float(b[2].abs().max())
            if name == "split":
                worst[(tag, name, "seam_coarse")] = float((a[3] - b[3]).abs().max()) / float(b[3].abs().max())
    # the whole walk must equal the split walk of the same mode (same kernels, the seam added in another place)
    for f in (fused, fused32):
        a, b = f["split"][0], f["whole"][0]
        assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max())
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:5]
    print("fused backward vs per-layer walks, worst |delta| / scale:", ", ".join("%s %.2e" % (str(k), v) for k, v in top))
    assert top[0][1] <= 1e-4, top
    # fp32 products on both sides: only the order of the sums differs (chain vs per-layer GEMM tiles, job ranges vs slabs, float atomics)
    top32 = sorted(((k, v) for k, v in worst.items() if k[0] == "f32 fused vs f32"), key=lambda kv: -kv[1])[:5]
    print("fused f32 walk vs per-layer f32 walk, worst |delta| / scale:", ", ".join("%s %.2e" % (str(k), v) for k, v in top32))
    assert top32[0][1] <= 1e-5, top32


# act:: table of the AudioFaceModel (csrc/sahs_layout.hpp): (first column, width, columns that are written)
_ACT_AUDIO = {"E": (0, 64, 64), "WH": (64, 768, 768), "DX": (832, 16, 3), "HH": (848, 384, 384), "AW": (1232, 16, 2), "XW": (1248, 16, 3),
              "PEX": (1264, 64, 64), "PEW": (1328, 32, 32), "T": (1360, 2048, 2048), "FEAT": (3408, 256, 256), "DIR": (3664, 32, 32),
              "GRID": (3696, 32, 32), "C": (3728, 512, 512), "S": (4240, 512, 512)}
_ACT_PART = {1: ("E", "WH", "DX", "HH", "AW", "XW"), 2: ("XW", "PEX", "PEW", "T", "FEAT", "DIR", "GRID", "C", "S")}
# sign-bit planes (sbits): (first word, layer width, layers, act array of the layers) per part
_BITS_PART = {1: ((0, 128, 6, "WH"), (24, 64, 6, "HH")), 2: ((0, 256, 8, "T"), (64, 128, 4, "C"), (80, 128, 4, "S"))}


def _act_array(act, name, part, layer=None, width=None):
    """array `name` of a saved-activation buffer of `part` (0: whole network) as (P, width): a saved array of column c starts at c * P"""
    P = act.shape[0]
    c, w, _ = _ACT_AUDIO[name]
    c0 = _ACT_AUDIO["XW"][0] if part == 2 else 0
    if layer is not None:
        c, w = c + layer * width, width
    return act.reshape(-1)[(c - c0) * P:(c - c0 + w) * P].view(P, w)


def _expected_sign_words(values):
    """sbits words [P][4 q][NW] of a (P, width) layer: feature 16 t + 4 q + r -> bit 4 (t % 8) + r of word (q, t // 8)"""
    P, w = values.shape
    nw = max(w // 128, 1)
    v = (values > 0).view(P, w // 16, 4, 4).to(torch.int64)                 # [p][t][q][r]
    words = torch.zeros(P, 4, nw, dtype=torch.int64, device=values.device)
    for t in range(w // 16):
        for r in range(4):
            words[:, :, t // 8] |= v[:, t, :, r] << (4 * (t % 8) + r)
    return words, (1 << (4 * min(w // 16, 8))) - 1 if w < 128 else 0xFFFFFFFF


@pytest.mark.parametrize("N,nc,nf", [(37, 40, 37), (1, 3, 2), (5, 64, 64)])
def test_x3_saving_forward_writes_what_the_f32_saving_forward_writes(N, nc, nf, weights_mod):
    """Training with the forward on the split-operand pipe (sahs_model_field_forward_split_save_bits_x3): the saved activations of every
    array of both parts within 2e-4 of the array's largest entry of what the fp32 saving forward keeps -- the encodings of the deformed
    point within 1e-3: sin(2^9 x') amplifies the ~1e-6 the two kernels' x' differ by (the deformation nets on this pipe) --
    and those of the inputs both kernels hold exactly to 1e-6; the raw outputs alike; the sign planes EXACTLY the signs of the values
    this launch saved; and -- the
    deformation + radiance pair into one whole-network save -- the fused backward over these buffers within 2e-4 of the backward over
    the fp32 forward's.  Ragged sample counts (1,480 / 1,369 / 2,849 against 128-sample tiles), a launch smaller than one wave's 32 samples
    (3 / 2 / 5: every lane past the end redoes the last sample) and whole tiles (320 / 320 / 640)."""
    ops = pkg("ops")
    W = weights_mod
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(29)
    flat = torch.from_numpy(W.flatten_state_dict(W.hash_state_dict(0, 2.0, 30.0, hdr=True))).to(dev)
    packed, packed_x3 = ops.pack_weights(flat), ops.pack_weights(flat, ops.SAHS_BF16X3)
    driving = torch.randn(16, 29, device=dev, generator=gen)
    near, far, cam = 0.48, 1.08, 0.8
    pose = torch.from_numpy(np.concatenate([np.eye(3), [[0.0], [0.0], [cam]]], 1).astype(np.float32)).to(dev)
    frame = ops.fold_conditioning(flat, driving, pose)
    Sf = nc + nf
    rays = torch.zeros(N, 8, device=dev)
    rays[:, 2] = cam
    rays[:, 3:6] = torch.randn(N, 3, device=dev, generator=gen) * 0.15 + torch.tensor([0, 0, -1.0], device=dev)
    zs = lambda S: torch.sort(torch.rand(N, S, device=dev, generator=gen) * (far - near) + near, dim=1).values
    z_c, z_new = zs(nc), zs(nf)
    src = torch.stack([torch.randperm(Sf, device=dev, generator=gen) for _ in range(N)]).to(torch.int32)
    ident = torch.arange(nc, dtype=torch.int32, device=dev).repeat(N, 1).contiguous()
    sb = lambda samples, mode: ops.alloc_sign_bits(samples, mode, "audio", dev)

    def save(x3):
        pk, prec = (packed_x3, ops.SAHS_BF16X3) if x3 else (packed, ops.SAHS_F32)
        xw = torch.zeros(N, Sf, 8, device=dev)
        bits_c, bits_d, bits_r = sb(N * nc, ops.FIELD_ALL), sb(N * nf, ops.FIELD_DEFORM), sb(N * Sf, ops.FIELD_RADIANCE)
        if x3:
            act_c = torch.zeros(N * nc, 4752, device=dev)
            ops.field_forward_split_save(pk, frame, 0, ops.FIELD_DEFORM, rays, xw, z=z_c, precision=prec, whole=(act_c, bits_c))
            raw_c, _ = ops.field_forward_split_save(pk, frame, 0, ops.FIELD_RADIANCE, rays, xw, src=ident, precision=prec, whole=(act_c, bits_c))
        else:
            raw_c, act_c = ops.field_forward_split_save(pk, frame, 0, ops.FIELD_ALL, rays, xw, z=z_c, bits=bits_c)
        _, act_d = ops.field_forward_split_save(pk, frame, 1, ops.FIELD_DEFORM, rays, xw, z=z_new, xw_col0=nc, bits=bits_d, precision=prec)
        raw_f, act_r = ops.field_forward_split_save(pk, frame, 1, ops.FIELD_RADIANCE, rays, xw, src=src, bits=bits_r, precision=prec)
        torch.cuda.synchronize()
        return {"raw_c": raw_c, "raw_f": raw_f, "act_c": act_c, "act_d": act_d, "act_r": act_r, "bits_c": bits_c, "bits_d": bits_d, "bits_r": bits_r, "xw": xw}

    ref, got = save(False), save(True)
    worst = {}
    for k in ("raw_c", "raw_f", "xw"):
        worst[k] = float((ref[k] - got[k]).abs().max()) / float(ref[k].abs().max())
    for buf, part, parts in (("act_c", 0, (1, 2)), ("act_d", 1, (1,)), ("act_r", 2, (2,))):
        for pt in parts:
            for name in _ACT_PART[pt]:
                valid = _ACT_AUDIO[name][2]
                a, b = _act_array(ref[buf], name, part)[:, :valid], _act_array(got[buf], name, part)[:, :valid]
                worst[(buf, name)] = float((a - b).abs().max()) / float(a.abs().max())
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:5]
    print("x3 saving forward vs f32 saving forward, worst |delta| / scale:", ", ".join("%s %.2e" % (str(k), v) for k, v in top))
    for k, v in worst.items():
        name = k[1] if isinstance(k, tuple) else k
        assert v <= {"E": 1e-6, "DIR": 1e-6, "PEX": 1e-3, "PEW": 1e-3}.get(name, 2e-4), (k, v, top)
    # the sign planes are the signs of the values saved beside them (bit for bit)
    for buf, bits, part, parts in (("act_c", "bits_c", 0, (1, 2)), ("act_d", "bits_d", 1, (1,)), ("act_r", "bits_r", 2, (2,))):
        P = got[buf].shape[0]
        for pt in parts:
            w0 = 48 if (part == 0 and pt == 2) else 0      # a whole-network save holds [deformation planes | radiance planes]
            for first, width, layers, name in _BITS_PART[pt]:
                nwords = 4 * max(width // 128, 1)
                for l in range(layers):
                    vals = _act_array(got[buf], name, part, layer=l, width=width)
                    exp, mask = _expected_sign_words(vals)
                    b0 = w0 + first + nwords * l
                    have = got[bits].reshape(-1)[b0 * P:(b0 + nwords) * P].view(P, 4, nwords // 4).to(torch.int64) & 0xFFFFFFFF
                    assert torch.equal(have & mask, exp & mask), (buf, name, l)
    # the fused backward over the x3 forward's buffers against the same walk over the fp32 forward's
    d_raw_c = torch.randn(N * nc, 16, device=dev, generator=gen)
    d_raw_f = torch.randn(N * Sf, 16, device=dev, generator=gen)
    xwg_new = torch.randn(N * nf, 8, device=dev, generator=gen) * torch.tensor([1, 1, 1, 0, 1, 1, 0, 0.0], device=dev)

    def walk(s):
        gf, gc = torch.zeros_like(flat), torch.zeros(128, device=dev)
        g_f = ops.field_backward_split(flat, frame, 1, ops.FIELD_RADIANCE, s["act_r"], gf, gc, d_raw=d_raw_f, bits=s["bits_r"])
        ops.field_backward_split(flat, frame, 1, ops.FIELD_DEFORM, s["act_d"], gf, gc, xw_grad_in=xwg_new, bits=s["bits_d"])
        ops.field_backward_split(flat, frame, 0, 3, s["act_c"], gf, gc, d_raw=d_raw_c, bits=s["bits_c"])
        torch.cuda.synchronize()
        return gf, gc, g_f

    assert ops.fused_backward() and ops.backward_gemm_precision() == "bf16x3"
    fused_ref, fused_got = walk(ref), walk(got)
    ops.fused_backward(False)
    try:
        layer_got = walk(got)      # the per-layer walk reads its masks from the saved VALUES: pins planes and values of this save against each other
    finally:
        ops.fused_backward(True)

    def deltas(a, b, norm):
        out = {}
        for k, (o, shape) in W.canonical_offsets("audio").items():
            n = int(np.prod(shape))
            if float(a[0][o:o + n].abs().max()) > 0.0:
                out[k] = norm(a[0][o:o + n], b[0][o:o + n])
        out["grad_cond"], out["seam_fine"] = norm(a[1], b[1]), norm(a[2], b[2])
        return sorted(out.items(), key=lambda kv: -kv[1])[:4]

    top = deltas(layer_got, fused_got, lambda x, y: float((x - y).abs().max()) / float(x.abs().max()))
    print("x3 forward's save, fused backward vs per-layer backward, worst |delta| / scale:", ", ".join("%s %.2e" % kv for kv in top))
    assert top[0][1] <= 1e-4, top
    # against the same walk over the fp32 forward's save: the two forwards differ by ~1e-5 of scale, which flips the mask of the few units
    # that sit at zero (a unit's whole contribution, not a rounding error) -- so in the 2-norm of each gradient, not entry by entry
    top = deltas(fused_ref, fused_got, lambda x, y: float((x - y).norm()) / float(x.norm()))
    print("fused backward over the x3 forward's save vs over the f32 forward's, worst |delta|_2 / |g|_2:", ", ".join("%s %.2e" % kv for kv in top))
    if N * nc >= 1000:      # (a handful of samples: one flipped unit is a few per cent of a gradient)
        assert top[0][1] <= 2e-2, top


def test_training_step_on_the_x3_forward_matches_the_f32_forward(weights_mod):
    """ops.training_forward_precision("bf16x3") through the autograd op (fused-loss form, 517 rays x (64 + 64) samples) against the same
    batch with the saving forward on fp32 MFMAs, by the criteria of the bf16x3 frame (tests/test_gpu_bf16.py): every ray's coarse map
    within four times the fp32 tolerance, the fine map too except the rays whose resampled depths landed on the other side of a cdf
    knot; the loss to 1e-4; every gradient in the 2-norm (a mask that flips at a unit sitting at zero moves single entries)."""
    ops = pkg("ops")
    dev = torch.device("cuda:0")
    fw = weights_mod.flatten_state_dict(weights_mod.hash_state_dict(0, 2.0, 30.0, hdr=True))
    gen = torch.Generator(device=dev).manual_seed(31)
    N, nc, nf = 517, 64, 64
    drv = torch.randn(16, 29, device=dev, generator=gen)
    cam = 0.8
    pose = torch.from_numpy(np.concatenate([np.eye(3), [[0.0], [0.0], [cam]]], 1).astype(np.float32)).to(dev)
    rays = torch.zeros(N, 8, device=dev)
    rays[:, 2] = cam
    rays[:, 3:6] = torch.randn(N, 3, device=dev, generator=gen) * 0.15 + torch.tensor([0, 0, -1.0], device=dev)
    rays[:, 6], rays[:, 7] = 0.483771, 1.083771
    bg = torch.cat([torch.rand(N, 3, device=dev, generator=gen), torch.ones(N, 1, device=dev), torch.zeros(N, 11, device=dev)], 1)
    t_rand, u = torch.rand(N, nc, device=dev, generator=gen), torch.rand(N, nf, device=dev, generator=gen)
    noise_c, noise_f = torch.randn(N, nc, device=dev, generator=gen) * 0.1, torch.randn(N, nc + nf, device=dev, generator=gen) * 0.1
    target = torch.rand(N, 3, device=dev, generator=gen)
    mask = torch.zeros(N, 12, device=dev)
    mask.scatter_(1, torch.randint(0, 12, (N, 1), device=dev, generator=gen), 1.0)
    cw = pkg("training").sample_prob_weights(dev)
    res = {}
    for x3 in (False, True):
        flat = torch.from_numpy(fw).to(dev).requires_grad_(True)
        d = drv.clone().requires_grad_(True)
        packed = ops.pack_weights(flat.detach())
        px3 = ops.pack_weights(flat.detach(), ops.SAHS_BF16X3) if x3 else None
        outs = ops.RenderRaysFn.apply(flat, d, pose, rays, bg, t_rand, noise_c, u, noise_f, packed, nc, nf, False, False, "audio", target, mask, cw, px3)
        outs[8].backward()
        torch.cuda.synchronize()
        res[x3] = (float(outs[8].detach()), outs[0].detach().clone(), outs[3].detach().clone(), flat.grad.clone(), d.grad.clone())
    a, b = res[False], res[True]
    assert abs(a[0] - b[0]) <= 1e-4 * abs(a[0]), (a[0], b[0])
    for i, nm, allowed in ((1, "coarse map", 0.0), (2, "fine map", 0.08)):
        bad = ((a[i] - b[i]).abs() > 4e-5 + 4e-4 * a[i].abs()).any(dim=1).float().mean()
        assert float(bad) <= allowed, (nm, float(bad), float((a[i] - b[i]).abs().max()))
    worst = {}
    off = 0
    for name, shape in weights_mod.canonical_spec("audio"):
        n = int(np.prod(shape))
        ga, gb = a[3][off:off + n], b[3][off:off + n]
        if float(ga.norm()) > 0.0:
            worst[name] = float((ga - gb).norm()) / float(ga.norm())
        off += n
    worst["driving input"] = float((a[4] - b[4]).norm()) / float(a[4].norm())
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:4]
    print("training step, x3 forward vs f32 forward, loss %.6f vs %.6f, worst gradient |delta|_2 / |g|_2:" % (b[0], a[0]), ", ".join("%s %.2e" % kv for kv in top))
    # (measured 2.8e-2 on the warp field's tensors: mask flips as in the test above -- 1.4e-2 there -- plus the rays whose fine depths moved a bin)
    assert top[0][1] <= 6e-2, top
