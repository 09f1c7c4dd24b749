#!/usr/bin/env python3
"""Where a wave of the backward GEMMs spends its cycles (diagnostic build -DSAHS_GEMM_STAMP, s_memtime stamps summed over every wave of
every launch of one training step):   python tools/ablate.py build gstamp   (here)   then on the GPU   python tools/stamp_gemm.py [variant]"""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variant = sys.argv[1] if len(sys.argv) > 1 else "gstamp"
os.environ["SAHS_NERF_LIB"] = os.path.join(REPO, "sahs-deformable-nerf_amd", "build", "variants", "libsahs_%s.so" % variant)
sys.argv = [sys.argv[0], "--steps", "1", "--warmup", "2"]
sys.path.insert(0, os.path.join(REPO, "tools"))
import train_bench   # noqa: E402

L = ctypes.CDLL(os.environ["SAHS_NERF_LIB"])
buf = (ctypes.c_ulonglong * 32)()
orig = train_bench.time.perf_counter
state = {"n": 0}


def hook():      # the first perf_counter() call of main() is right after the warm-up: reset the counters there
    if state["n"] == 0:
        assert L.sahs_dbg_gemm_stamps(None, 1) == 0
    state["n"] += 1
    return orig()


train_bench.time.perf_counter = hook
train_bench.main()
assert L.sahs_dbg_gemm_stamps(buf, 0) == 0
for ta, name in ((1, "weight-gradient GEMM (TA)"), (0, "data-gradient GEMM")):
    w, b, body, pro, epi, tot, waves, steps = [buf[8 * ta + i] for i in range(8)]
    print("%s: %d waves, %.1f K-steps per wave; cycles per wave: total %.0f = prologue %.0f + loop (wait vmcnt %.0f + barrier %.0f + body %.0f) + epilogue %.0f"
          % (name, waves, steps / waves, tot / waves, pro / waves, w / waves, b / waves, body / waves, epi / waves))
    print("    per K-step: wait %.0f, barrier %.0f, body %.0f cycles" % (w / steps, b / steps, body / steps))
steps = buf[8 + 7]
if buf[16 + 5]:
    names = ["DMA issue", "stage reads landed", "split + fragment writes (+ sign bits)", "barrier B", "fragment reads landed", "MFMA issue", "wait vmcnt", "barrier A"]
    print("weight-gradient K-step, cycles: " + ", ".join("%s %.0f" % (n, buf[16 + i] / steps) for i, n in enumerate(names)))
if buf[24 + 6]:
    names = ["first barrier", "stage half 0 + barrier", "stores half 0", "barrier + stage half 1 + barrier", "stores half 1", "last barrier"]
    print("data-gradient epilogue, cycles per wave: " + ", ".join("%s %.0f" % (n, buf[24 + i] / buf[24 + 6]) for i, n in enumerate(names)))
