"""configs[4] on the MI355X: a 2048-ray training step through the HIP autograd op, checked against plain PyTorch autograd
of the eager restatement (oracle/torch_eager.py, itself pinned to the reference's golden gradients) and timed; then a few
optimiser steps of the drop-in training loop."""
import json
import os
import time

import numpy as np
import pytest
import torch

from conftest import REPO, pkg

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
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(REPO, "gpurun_out", "train_step.json"), "w"), indent=1)
    print(json.dumps(res))


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
