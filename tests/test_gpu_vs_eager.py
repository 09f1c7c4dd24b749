"""On the MI355X: the HIP path against the plain PyTorch-ROCm eager restatement of the reference
(oracle/torch_eager.py, pinned to the reference by test_torch_eager_vs_golden.py) on the full W512
frame on the HIGH-DYNAMIC-RANGE weights (mean background weight ~0.2: the rays spread their weight over
the samples, so agreement is not vacuous): per-ray agreement of all 8 outputs, the PSNR protocol of
SURVEY.md section 8d, and the speed ratio that BASELINE.json's north_star targets.  Numbers are printed."""
import json
import time

import numpy as np
import pytest
import torch

from conftest import VARIANT_KW, pkg

pytestmark = pytest.mark.gpu


def psnr(a, b):
    mse = float(torch.mean((a.clamp(0, 1) - b.clamp(0, 1)) ** 2))
    return 99.0 if mse == 0 else -10.0 * np.log10(mse)


def test_full_frame_against_eager_pytorch(weights_mod):
    from oracle import torch_eager as TE
    sahs = pkg()
    dev = torch.device("cuda:0")
    cfg = sahs.default_config()
    H = W = 512
    R = H * W
    sd_np = weights_mod.hash_state_dict(**VARIANT_KW["hdr"])
    model = sahs.AudioFaceModel(cfg).to(dev).load_flat(weights_mod.flatten_state_dict(sd_np))
    field = TE.EagerField({k: torch.from_numpy(v).to(dev) for k, v in sd_np.items()})
    rng = np.random.default_rng(42)
    audio = torch.from_numpy(rng.standard_normal((16, 29)).astype(np.float32)).to(dev)
    pose = torch.from_numpy(np.concatenate([np.eye(3), [[0.0], [0.0], [0.8]]], axis=1).astype(np.float32)).to(dev)
    intr = np.array([1200.0, 1200.0, 0.5, 0.5], np.float32)
    bg = torch.cat([torch.rand(R, 3, device=dev), torch.ones(R, 1, device=dev), torch.zeros(R, 11, device=dev)], 1)
    ro, rd = sahs.get_ray_bundle(H, W, intr, pose)
    gen = torch.Generator(device=dev).manual_seed(1)
    rand = [dict(t_rand=torch.rand(131072, 64, device=dev, generator=gen), u=torch.rand(131072, 64, device=dev, generator=gen)) for _ in range(2)]
    feed = [("rand", r[k]) for r in rand for k in ("t_rand", "u")]

    def hip():
        log = list(feed)
        orig = torch.rand
        torch.rand = lambda *a, **k: log.pop(0)[1]
        try:
            with torch.no_grad():
                return sahs.run_one_iter_of_nerf(H, W, intr, model, ro, rd, cfg, mode="validation", driving=audio, pose=pose,
                                                 background_prior=bg)
        finally:
            torch.rand = orig

    def eager():
        with torch.no_grad():
            return TE.run_one_iter(field, ro, rd, cfg.dataset.near, cfg.dataset.far, audio, pose, bg=bg, rand=rand, perturb=True)

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps, out

    t_hip, o_hip = timed(hip, 2)
    t_eager, o_eager = timed(eager, 1)
    rgb_h, rgb_e = o_hip[3].reshape(-1, 15)[:, :3], o_eager[3][:, :3]
    # pseudo-target: the eager render under a different random stream
    rand2 = [dict(t_rand=torch.rand(131072, 64, device=dev, generator=gen), u=torch.rand(131072, 64, device=dev, generator=gen)) for _ in range(2)]
    with torch.no_grad():
        tgt = TE.run_one_iter(field, ro, rd, cfg.dataset.near, cfg.dataset.far, audio, pose, bg=bg, rand=rand2, perturb=True)[3][:, :3]
    res = dict(rays=R, hip_s=t_hip, eager_s=t_eager, hip_rays_per_s=R / t_hip, eager_rays_per_s=R / t_eager, speedup=t_eager / t_hip,
               psnr_hip_vs_eager=psnr(rgb_h, rgb_e), psnr_hip_vs_target=psnr(rgb_h, tgt), psnr_eager_vs_target=psnr(rgb_e, tgt),
               max_abs_rgb_diff=float((rgb_h - rgb_e).abs().max()), w_bg_mean=float(o_eager[6].mean()))
    # per-ray agreement on all 8 outputs.  The two fp32 implementations differ by ~1e-5 in the coarse weights; a resampled depth that
    # thereby lands on the other side of a cdf knot moves that ray's fine outputs by up to a few 1e-2 on this network -- for the
    # reference's own fp32 run against its float64 run as well (conftest.yardstick: outlier_rays) -- so the criterion is the one
    # bench.py's cpu_baseline leg asserts: >= 99 % of the rays within 1e-3 on every output, none beyond 0.1, coarse outputs 1e-4.
    worst, frac_ok = {}, 1.0
    for nm, a, b in zip(["rgb_c", "disp_c", "acc_c", "rgb_f", "disp_f", "acc_f", "w_bg", "depth_f"], o_hip, o_eager):
        d = (a.reshape(R, -1) - b.reshape(R, -1)).abs().max(dim=1).values
        worst[nm] = float(d.max())
        frac_ok = min(frac_ok, float((d <= 1e-3).float().mean()))
    res.update(rays_within_1e3=frac_ok, worst_abs_diff=worst)
    print(json.dumps(res))
    assert frac_ok >= 0.99 and max(worst.values()) <= 0.1, res
    assert max(worst[k] for k in ("rgb_c", "acc_c")) <= 1e-4, res
    assert res["psnr_hip_vs_eager"] > 60.0, res
    assert abs(res["psnr_hip_vs_target"] - res["psnr_eager_vs_target"]) <= 0.05, res
    assert res["w_bg_mean"] < 0.5, "workload must terminate rays before the background sample"
    assert res["speedup"] > 2.0, res
