"""Stage-I evaluation harness around the HIP renderer: the counterpart of the frame loop of
``eval_stage_rays.py:390-554`` -- checkpoint loading (:302-327), per-frame ``get_ray_bundle`` + ``run_one_iter_of_nerf(mode=
"validation")`` (:454-475), normal map from the fine disparity (:116-151, :477), RGB / segmentation-colour / disparity PNGs
(:526-545, ``nerf/utils.py:112-140``), running mean seconds per image (:554).  Dataset readers (cv2/imageio, no data offline)
are out of scope: frames are handed in as dicts of tensors.
"""
import contextlib
import os
import time

import numpy as np
import torch

from .nerf_helpers import get_ray_bundle
from .train_utils import partition_invariant_rng, run_one_iter_of_nerf

# class index -> colour as written to disk (nerf/utils.py:122-135 reverses each RGB triple before use)
SEG_COLOURS = [(0, 0, 0), (0, 0, 204), (0, 153, 76), (0, 204, 204), (255, 51, 51), (255, 255, 0), (0, 51, 102), (0, 204, 102),
               (0, 255, 255), (204, 0, 0), (51, 153, 255), (0, 204, 0)]


def load_checkpoint(path, model, device):
    """eval_stage_rays.py:302-327: model_state_dict + optional height / width / focal_length / background / pose_c.
    Loaded with weights_only=True: nothing in a foreign checkpoint is executed."""
    ck = torch.load(path, map_location=device, weights_only=True)
    model.load_state_dict(ck["model_state_dict"])
    extras = {k: ck[k] for k in ("height", "width", "focal_length", "background", "pose_c", "latent_codes") if k in ck}
    return model.eval(), extras


def label2color(seg):
    """nerf/utils.py:112-140: argmax over the 12 classes -> (H, W, 3) colour in [0, 1]."""
    idx = torch.argmax(seg, dim=-1)
    table = torch.tensor(SEG_COLOURS, dtype=torch.float32, device=seg.device) / 255.0
    return table[idx]


def shrink(mask):
    """nerf/utils.py:5-66: (H, W, >=12) scores -> one-hot (H, W, 12) int32 of the arg-max class."""
    idx = np.argmax(np.asarray(mask), axis=-1)
    return np.eye(12, dtype=np.int32)[idx]


def normal_map(depthmap, focal, weights=None, clean=True, central_difference=False):
    """eval_stage_rays.py:116-151 (square maps, as there): back-project with the intrinsics, cross product of the forward (or
    central) differences, normalise, map to [0, 255]; ``clean`` whitens pixels whose background weight exceeds 0.22 and blends
    the rest towards white by that weight."""
    W, H = depthmap.shape
    cx, cy, fx, fy = focal[2] * W, focal[3] * H, focal[0], focal[1]
    ii, jj = torch.meshgrid(torch.arange(W, device=depthmap.device), torch.arange(H, device=depthmap.device), indexing="xy")
    pts = torch.stack([((ii - cx) * depthmap) / fx, -((jj - cy) * depthmap) / fy, depthmap], dim=-1)
    d = 2 if central_difference else 1
    dx = pts[d:, :, :] - pts[:-d, :, :]
    dy = pts[:, d:, :] - pts[:, :-d, :]
    n = torch.cross(dy[:-d, :, :], dx[:, :-d, :], dim=2)
    n = n / torch.sqrt(torch.sum(n * n, 2, keepdim=True))
    n = n * 0.5 + 0.5
    if clean and weights is not None:
        m = weights[..., None].expand(-1, -1, 3)[:-d, :-d]
        n = torch.where(m > 0.22, torch.ones_like(n), n)
        n = (1 - m) * n + m * torch.ones_like(n)
    return n * 255


def cast_to_image(t):
    """eval_stage_rays.py:223-229: (H, W, 3) float -> uint8 (clamp, x255, truncate)."""
    return (t.detach().clamp(0.0, 1.0) * 255).to(torch.uint8).cpu().numpy()


def cast_to_disparity_image(t):
    """eval_stage_rays.py:232-235."""
    img = (t - t.min()) / (t.max() - t.min())
    return (img.clamp(0, 1) * 255).detach().cpu().numpy().astype(np.uint8)


def _save_png(path, arr):
    from PIL import Image
    os.makedirs(os.path.dirname(path), exist_ok=True)
    Image.fromarray(arr).save(path)


def render_frames(model, cfg, frames, hwf, background=None, pose_c=None, savedir=None, save_disparity=False, log=print, shard=None):
    """frames: iterable of dict(pose (3|4,4), audio (16,29) [AudioFaceModel] or expression (76,) [NeRFaceModel], mask (H,W,12)
    optional, name optional).
    hwf = (H, W, intrinsics[fx, fy, cx, cy]).  Returns the list of per-frame outputs (dicts of tensors).
    shard (True | process group): the node renders each frame together -- every rank (one process per GPU, torch.distributed
    initialised, the same frames on every rank) renders its block of the frame's rays and one RCCL all-gather assembles the frame on
    all of them (run_one_iter_of_nerf(_shard=...)); rank 0 writes the images.  The draws of frame i are keyed by (randomseed + i, global
    ray index), so the images do not depend on the number of GPUs; shard=None keeps the reference's torch.rand stream."""
    H, W, focal = hwf
    results, times = [], []
    dev = next(model.parameters()).device
    bg = background.reshape(-1, 15).to(dev) if background is not None else None
    sharded = shard is not None and shard is not False
    writer = not (sharded and torch.distributed.is_initialized() and torch.distributed.get_rank(None if shard is True else shard) != 0)
    for i, fr in enumerate(frames):
        t0 = time.time()
        keyed = partition_invariant_rng(int(cfg.experiment.randomseed) + i) if sharded else contextlib.nullcontext()
        with torch.no_grad(), keyed:
            pose = torch.as_tensor(fr["pose"], dtype=torch.float32, device=dev)[:3, :4].contiguous()
            audio = torch.as_tensor(fr["audio"] if "audio" in fr else fr["expression"], dtype=torch.float32, device=dev)
            mask = fr.get("mask")
            mask = torch.as_tensor(shrink(mask)).to(dev) if mask is not None else None
            ro, rd = get_ray_bundle(H, W, focal, pose)
            rgb_c, disp_c, _, rgb_f, disp_f, _, w_bg, depth_f = run_one_iter_of_nerf(
                H, W, focal, model, ro, rd, cfg, mode="validation", driving=audio, pose=pose, pose_c=pose_c, background_prior=bg,
                latent_code=None, inHead=mask, _shard=shard if sharded else None)
            rgb = rgb_f if rgb_f is not None else rgb_c
            normals = normal_map(disp_f, focal, w_bg, clean=True)
        torch.cuda.synchronize() if dev.type == "cuda" else None
        times.append(time.time() - t0)
        if savedir and writer:
            name = str(fr.get("name", "f_%04d.png" % i)).split("/")[-1].split(".")[0] + ".png"
            _save_png(os.path.join(savedir, name), cast_to_image(rgb[..., :3]))
            _save_png(os.path.join(savedir, "masks", name), cast_to_image(label2color(rgb[..., 3:])))
            _save_png(os.path.join(savedir, "normals", name), normals.clamp(0, 255).to(torch.uint8).cpu().numpy())
            if save_disparity:
                _save_png(os.path.join(savedir, "disparity", name), cast_to_disparity_image(disp_f))
        results.append(dict(rgb=rgb, disp=disp_f, depth=depth_f, w_bg=w_bg, normals=normals))
        log("Avg time per image: %s" % (sum(times) / (i + 1)))
    return results
