"""T2048 (configs[4]): time the 2048-ray forward+backward step through the HIP autograd op (no optimiser, no eager leg);
used under rocprofv3 for the per-kernel breakdown of the training path."""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default="audio", choices=["audio", "nerface", "nerface_static"])
    a = ap.parse_args()
    sahs = importlib.import_module("sahs-deformable-nerf_amd")
    W = importlib.import_module("sahs-deformable-nerf_amd.weights")
    dev = torch.device("cuda:0")
    R = a.rays
    g = torch.Generator(device=dev).manual_seed(3)
    if a.model == "audio":
        cfg = sahs.default_config()
        model = sahs.AudioFaceModel(cfg).to(dev).load_flat(W.flatten_state_dict(W.hash_state_dict(0, 8.0, 30.0))).train()
        audio = torch.randn(16, 29, device=dev, generator=g)
        cam, mac = 0.8, W.MAC_PER_SAMPLE
    else:
        cfg = sahs.default_config("expression" if a.model == "nerface" else "expression_static")
        fw = W.flatten_state_dict(W.hash_state_dict(0, 8.0, 30.0, model=a.model), model=a.model)
        model = sahs.NeRFaceModel(cfg).to(dev).load_flat(fw).train()
        audio = torch.randn(76, device=dev, generator=g) * 0.5
        cam, mac = 0.5, (W.NERFACE_MAC_PER_SAMPLE if a.model == "nerface" else W.NERFACE_STATIC_MAC_PER_SAMPLE)
    pose = torch.from_numpy(np.concatenate([np.eye(3), [[0.0], [0.0], [cam]]], 1).astype(np.float32)).to(dev)
    ro = torch.zeros(R, 3, device=dev)
    ro[:, 2] = cam
    rd = torch.randn(R, 3, device=dev, generator=g) * 0.15 + torch.tensor([0, 0, -1.0], device=dev)
    bg = torch.cat([torch.rand(R, 3, device=dev, generator=g), torch.ones(R, 1, device=dev), torch.zeros(R, 11, device=dev)], 1)
    A, B = torch.randn(R, 15, device=dev, generator=g), torch.randn(R, 15, device=dev, generator=g)

    def step():
        outs = sahs.run_one_iter_of_nerf(0, 0, None, model, ro, rd, cfg, mode="train", driving=audio, pose=pose, background_prior=bg)
        loss = (outs[0] * A).sum() + (outs[3] * B).sum() + outs[7].sum() * 0.1
        model.zero_grad(set_to_none=True)
        loss.backward()
        return loss

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    flop = R * 192 * 2 * mac * 3
    print(json.dumps(dict(workload="T%d fwd+bwd (%s)" % (R, a.model), rays=R, ms_per_step=dt * 1e3, rays_per_s=R / dt, tflops_3x_fwd=flop / dt / 1e12)))


if __name__ == "__main__":
    main()
