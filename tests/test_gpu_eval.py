"""SURVEY.md section 8f-1 on the MI355X: the evaluation frame loop (checkpoint -> per-frame render -> normal / mask / RGB
PNGs) through the HIP path; the rendered frame is checked against the CPU oracle on the same inputs."""
import os

import numpy as np
import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu


def test_eval_frame_loop_checkpoint_to_png(weights_mod, tmp_path):
    from oracle import oracle as O
    sahs = pkg()
    E = pkg("evaluation")
    dev = torch.device("cuda:0")
    cfg = sahs.default_config()
    cfg.nerf.validation.perturb = False          # deterministic depths so frame 0 can be checked against the oracle
    sd_np = weights_mod.hash_state_dict(0, 8.0, 30.0)
    flat = weights_mod.flatten_state_dict(sd_np)
    H = W = 24
    focal = np.array([1200.0 * W / 512, 1200.0 * W / 512, 0.5, 0.5], np.float32)
    rng = np.random.default_rng(11)
    bgimg = torch.cat([torch.rand(H, W, 3), torch.ones(H, W, 1), torch.zeros(H, W, 11)], -1)
    src = sahs.AudioFaceModel(cfg).load_flat(flat)
    ck = str(tmp_path / "checkpoint.ckpt")
    torch.save({"model_state_dict": src.state_dict(), "height": H, "width": W, "focal_length": torch.from_numpy(focal), "background": bgimg}, ck)
    model, extras = E.load_checkpoint(ck, sahs.AudioFaceModel(cfg).to(dev), dev)
    assert extras["height"] == H and torch.equal(extras["background"].cpu(), bgimg)
    frames = []
    for i in range(2):
        pose = np.concatenate([np.eye(3), [[0.01 * i], [0.0], [0.8]]], 1).astype(np.float32)
        frames.append(dict(pose=pose, audio=rng.standard_normal((16, 29)).astype(np.float32), name="img/f_%04d.jpg" % i))
    logs = []
    out = E.render_frames(model, cfg, frames, (H, W, focal), background=extras["background"], savedir=str(tmp_path / "out"),
                          save_disparity=True, log=logs.append)
    assert len(out) == 2 and len(logs) == 2 and logs[0].startswith("Avg time per image")
    for sub in ("", "masks", "normals", "disparity"):
        for i in range(2):
            assert os.path.getsize(tmp_path / "out" / sub / ("f_%04d.png" % i)) > 0
    from PIL import Image
    png = np.asarray(Image.open(tmp_path / "out" / "f_0000.png"))
    assert png.shape == (H, W, 3) and png.dtype == np.uint8
    # frame 0 against the oracle (validation mode: no perturbation, deterministic u)
    ro, rd = O.get_ray_bundle(H, W, focal, frames[0]["pose"])
    ref = O.run_one_iter_of_nerf(flat, ro.reshape(-1, 3), rd.reshape(-1, 3), cfg.dataset.near, cfg.dataset.far, 64, 64,
                                 frames[0]["audio"], frames[0]["pose"], background_prior=bgimg.reshape(-1, 15).numpy())
    rgb_f = ref[3]
    rgb = out[0]["rgb"].reshape(-1, 15).cpu().numpy()
    assert np.abs(rgb - rgb_f).max() < 2e-3
    assert np.abs(png.astype(np.int32) - (np.clip(rgb_f[:, :3], 0, 1) * 255).astype(np.int32).reshape(H, W, 3)).max() <= 1
    assert out[0]["normals"].shape == (H - 1, W - 1, 3) and torch.isfinite(out[0]["normals"]).all()


def test_eval_frame_loop_expression_model(tmp_path):
    """The same frame loop over an expression-driven NeRFaceModel (frames carry `expression` instead of `audio`)."""
    sahs = pkg()
    E = pkg("evaluation")
    W = pkg("weights")
    dev = torch.device("cuda:0")
    cfg = sahs.default_config("expression")
    cfg.nerf.validation.perturb = False
    fw = W.flatten_state_dict(W.hash_state_dict(0, 8.0, 30.0, model="nerface"), model="nerface")
    model = sahs.NeRFaceModel(cfg).to(dev).load_flat(fw).eval()
    H = Wd = 16
    focal = np.array([1100.0 * Wd / 512, 1100.0 * Wd / 512, 0.5, 0.5], np.float32)
    rng = np.random.default_rng(4)
    frames = [dict(pose=np.concatenate([np.eye(3), [[0.0], [0.0], [0.5]]], 1).astype(np.float32),
                   expression=(rng.standard_normal(76) * 0.5).astype(np.float32), name="f_%04d.png" % i) for i in range(2)]
    out = E.render_frames(model, cfg, frames, (H, Wd, focal), background=torch.rand(H, Wd, 15), savedir=str(tmp_path / "o"), log=lambda s: None)
    assert len(out) == 2 and out[0]["rgb"].shape == (H, Wd, 15) and bool(torch.isfinite(out[0]["rgb"]).all())
    assert not torch.equal(out[0]["rgb"], out[1]["rgb"])          # the expression drives the render
    assert os.path.getsize(tmp_path / "o" / "normals" / "f_0001.png") > 0


@pytest.mark.parametrize("kind,arch", [("audio", "audio"), ("expression", "nerface"), ("expression_static", "nerface_static")])
def test_eval_frame_loop_bf16_models(kind, arch, tmp_path):
    """The evaluation frame loop with precision="bf16" models as drop-ins (AudioFaceModel: bf16 kernel; NeRFaceModel with deformation
    nets: mixed precision through the split chain; without: bf16 kernel): same frames as the fp32 model within the PSNR the
    kernel-level tests establish, PNGs written."""
    sahs = pkg()
    E = pkg("evaluation")
    W = pkg("weights")
    dev = torch.device("cuda:0")
    cfg = sahs.default_config(kind)
    cfg.nerf.validation.perturb = False
    fw = W.flatten_state_dict(W.hash_state_dict(0, 8.0, 30.0, model=arch), model=arch)
    Model = sahs.AudioFaceModel if arch == "audio" else sahs.NeRFaceModel
    H = Wd = 20
    focal = np.array([1100.0 * Wd / 512, 1100.0 * Wd / 512, 0.5, 0.5], np.float32)
    rng = np.random.default_rng(5)
    cam = 0.8 if arch == "audio" else 0.5
    drive = dict(audio=rng.standard_normal((16, 29)).astype(np.float32)) if arch == "audio" else dict(expression=(rng.standard_normal(76) * 0.5).astype(np.float32))
    frames = [dict(pose=np.concatenate([np.eye(3), [[0.0], [0.0], [cam]]], 1).astype(np.float32), name="f_0000.png", **drive)]
    bg = torch.rand(H, Wd, 15)
    outs = {}
    for prec in ("fp32", "bf16"):
        model = Model(cfg, precision=prec).to(dev).load_flat(fw).eval()
        outs[prec] = E.render_frames(model, cfg, frames, (H, Wd, focal), background=bg, savedir=str(tmp_path / prec), log=lambda s: None)[0]["rgb"]
        assert outs[prec].shape == (H, Wd, 15) and bool(torch.isfinite(outs[prec]).all())
        assert os.path.getsize(tmp_path / prec / "f_0000.png") > 0
    mse = float(((outs["fp32"][..., :3] - outs["bf16"][..., :3]) ** 2).mean())
    assert mse > 0.0 and -10.0 * np.log10(mse) >= 35.0, mse      # a different kernel ran, and it agrees with the fp32 frame
