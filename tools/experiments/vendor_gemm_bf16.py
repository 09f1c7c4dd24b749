"""What the vendor library sustains on this box at the precision of the SAHS_BF16 kernels: torch.matmul in bf16 (hipBLASLt) on a large square
GEMM and on the field's own layer shape ([P x 256] x [256 x 256]), with rocm-smi socket power / sclk sampled beside it -- the yardstick for
"the bf16 forward is power-limited at 0.55 of the nominal 2.5 PFLOP/s" (DESIGN.md section 3.2).  Prints one JSON line.
Run on the GPU box:  python tools/experiments/vendor_gemm_bf16.py"""
import json
import subprocess
import threading
import time

import torch

PEAK = 2500.0      # TFLOP/s, dense bf16 (MI355X_MICROARCH.md)


def sample_smi(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5)
            d = json.loads(r.stdout)
            card = d[sorted(d)[0]]
            p = [float(v) for k, v in card.items() if "ower" in k and "(W)" in k]
            c = [v for k, v in card.items() if k.startswith("sclk")]
            mhz = float(str(c[0]).strip("()").lower().replace("mhz", "")) if c else None
            out.append((max(p) if p else None, mhz))
        except Exception:      # the sampler must never stop the measurement
            pass
        time.sleep(0.1)


def run(name, a, b, seconds=3.0):
    for _ in range(5):
        a @ b
    torch.cuda.synchronize()
    stop, samples = threading.Event(), []
    th = threading.Thread(target=sample_smi, args=(stop, samples))
    th.start()
    n, t0 = 0, time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            a @ b
        n += 20
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop.set()
    th.join()
    ms = e0.elapsed_time(e1) / n
    flop = 2.0 * a.shape[0] * a.shape[1] * b.shape[1]
    tf = flop / ms / 1e9
    loaded = [s for s in samples if s[0] is not None and s[0] > 0.8 * max(x[0] for x in samples if x[0] is not None)]
    return {"gemm": name, "ms": ms, "tflops": tf, "frac_of_2500": tf / PEAK, "smi_samples_under_load": len(loaded),
            "mean_power_w": sum(s[0] for s in loaded) / len(loaded) if loaded else None,
            "mean_sclk_mhz": (sum(s[1] for s in loaded if s[1]) / max(1, sum(1 for s in loaded if s[1]))) if loaded else None}


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    mk = lambda m, n: torch.randn(m, n, device=dev, generator=g).to(torch.bfloat16)
    out = [run("8192 x 8192 x 8192", mk(8192, 8192), mk(8192, 8192)),
           run("16384 x 16384 x 8192", mk(16384, 8192), mk(8192, 16384)),
           run("field layer: [2097152 x 256] x [256 x 256]", mk(2097152, 256), mk(256, 256))]
    print(json.dumps({"what": "torch.matmul bf16 (hipBLASLt) on this box; nominal dense bf16 peak 2500 TFLOP/s", "results": out}))


if __name__ == "__main__":
    main()
