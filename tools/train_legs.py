"""bench.py's two training legs alone (configs[4]: fp32 backward products, and the default split-bf16 backward), optionally with the
fused backward walk switched off (--per-layer) for a same-box A/B.  Prints one JSON line."""
import argparse
import importlib
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--per-layer", action="store_true", help="keep the per-layer backward walk (ops.fused_backward(False))")
    ap.add_argument("--only", default=None, choices=["fp32", "bf16x3", "x3fwd"])
    a = ap.parse_args()
    pkg = importlib.import_module("sahs-deformable-nerf_amd")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    pkg.ops.fused_backward(not a.per_layer)
    out = {"fused_backward": pkg.ops.fused_backward()}
    for mode in ("fp32", "bf16x3"):
        if a.only in (None, mode):
            out["train_T2048" + ("" if mode == "fp32" else "_bf16x3")] = bench.train_leg(pkg, dev, steps=a.steps, warmup=a.warmup, backward=mode)
    if a.only in (None, "x3fwd"):      # the saving forward on the split-operand kernels too (ops.training_forward_precision)
        out["train_T2048_x3fwd"] = bench.train_leg(pkg, dev, steps=a.steps, warmup=a.warmup, backward="bf16x3", forward="bf16x3")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
