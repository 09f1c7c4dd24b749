#!/usr/bin/env python3
"""Condense gpurun_out/prof_r2 (tools/profile_r2.sh) into the tracked summaries under profiles/: per-kernel statistics of the fp32
headline, the bf16 leg and the training step, and the PMC summary (+ a small JSON that bench.py quotes as roofline.traffic)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = os.path.join(REPO, "gpurun_out", "prof_r2")
out = os.path.join(REPO, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r2"


def short(name):
    return name if len(name) <= 150 else name[:147] + "..."


def kernel_stats(sub, dst, header):
    ks = glob.glob(os.path.join(base, sub, "*", "*_kernel_stats.csv"))[0]
    rows = list(csv.DictReader(open(ks)))
    with open(os.path.join(out, dst), "w") as f:
        f.write("# " + header + "\n")
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    return rows


def kind(kn):
    """Which field launch a kernel name is: fp32 whole network / deformation nets / radiance net, or the bf16 kernel."""
    if "field_forward_bf16w" in kn and "sahs_nf" not in kn:
        for m, lab in (("0", "bf16_all"), ("1", "bf16_deform"), ("2", "bf16_radiance")):
            if "kernel<%s>" % m in kn or "kernelILi%sE" % m in kn:
                return lab
    if "field_forward_f32_kernel" in kn and "sahs_n" not in kn:
        for m, lab in (("0", "f32_all"), ("1", "f32_deform"), ("2", "f32_radiance")):
            if "<false, %s>" % m in kn or "ILb0ELi%sE" % m in kn:
                return lab
    return None


rows32 = kernel_stats("trace_f32", tag + "_fp32_kernel_stats.csv",
                      "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-secondary --no-cpu-baseline --steps 2 --warmup 1   (fp32 W512 headline; per chunk: "
                      "field<false,0> = coarse launch (8.39 M samples, whole network), <false,1> = deformation nets on the 8.39 M new depths, <false,2> = radiance net on the 16.78 M fine samples)")
rows16 = kernel_stats("trace_bf16", tag + "_bf16_kernel_stats.csv",
                      "rocprofv3 --kernel-trace --stats -- python3 bench.py --precision bf16 --no-secondary --no-cpu-baseline --steps 4 --warmup 1   (bf16 W512: one-wave-per-SIMD kernel)")
rowstr = kernel_stats("trace_train", tag + "_train_T2048_kernel_stats.csv",
                      "rocprofv3 --kernel-trace --stats -- python3 tools/train_bench.py --steps 5 --warmup 2   (T2048: 2048-ray forward + backward)")
for a, b in (("bench_trace_f32.json", "_fp32_bench_under_rocprof.json"), ("bench_trace_bf16.json", "_bf16_bench_under_rocprof.json"),
             ("bench_trace_train.json", "_train_T2048_bench_under_rocprof.json")):
    shutil.copy(os.path.join(base, a), os.path.join(out, tag + b))

agg = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(base, "pmc_*", "*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = kind(r["Kernel_Name"])
        if k:
            agg.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
dur = {}
for rows in (rows32, rows16):
    for r in rows:
        k = kind(r["Name"])
        if k:
            dur[k] = (float(r["MinNs"]), float(r["MaxNs"]), float(r["AverageNs"]), int(r["Calls"]))
lines = ["# rocprofv3 PMC summary (separate passes per counter group, tools/profile_r2.sh), bench.py --steps 1 --warmup 0: the first frame's dispatches,",
         "# two 131,072-ray chunks; fp32: per chunk coarse (f32_all, 8.39 M samples), deformation nets (f32_deform, 8.39 M), radiance net (f32_radiance, 16.78 M);",
         "# bf16: the same three launches per chunk by field_forward_bf16w_kernel<0|1|2>",
         "kernel,counter,dispatch_0,dispatch_1,dispatch_2,dispatch_3"]
for (k, c), v in agg.items():
    lines.append("%s,%s,%s" % (k, c, ",".join("%.6g" % x for x in v[:4])))
summary = {}
P_FINE = 16777216
for k, label, pick, samples, alg_bytes in (
        ("f32_radiance", "fp32 radiance-net launch (the dominant dispatch: 16.78 M fine samples)", 0, P_FINE, P_FINE * (64 + 32 + 4)),
        ("f32_all", "fp32 whole-network launch (coarse pass: 8.39 M samples)", 0, P_FINE // 2, (P_FINE // 2) * (64 + 4 + 32)),
        ("f32_deform", "fp32 deformation-net launch (8.39 M new depths)", 0, P_FINE // 2, (P_FINE // 2) * (4 + 32)),
        ("bf16_radiance", "bf16 radiance-net launch (16.78 M fine samples)", 0, P_FINE, P_FINE * (64 + 32 + 4)),
        ("bf16_all", "bf16 whole-network launch (coarse pass: 8.39 M samples)", 0, P_FINE // 2, (P_FINE // 2) * (64 + 4 + 32)),
        ("bf16_deform", "bf16 deformation-net launch (8.39 M new depths)", 0, P_FINE // 2, (P_FINE // 2) * (4 + 32))):
    if k not in dur or (k, "GRBM_GUI_ACTIVE") not in agg:
        continue
    g = lambda c: agg[(k, c)][pick] if (k, c) in agg and len(agg[(k, c)]) > pick else float("nan")
    t = dur[k][2] * 1e-9
    wr, fe = g("WRITE_SIZE") * 1024, g("FETCH_SIZE") * 1024
    clk = g("GRBM_GUI_ACTIVE") / 8 / t
    lines += ["", "# %s: %.2f ms under rocprof (%d dispatches in the traced run)" % (label, t * 1e3, dur[k][3]),
              "#   WRITE_SIZE = %.3f GB, FETCH_SIZE = %.3f GB as reported, x2 (gfx950: 16 B/lane streaming reads count half) = %.3f GB; algorithmic bytes %.3f GB"
              % (wr / 1e9, fe / 1e9, 2 * fe / 1e9, alg_bytes / 1e9),
              "#   HBM-side traffic WRITE + 2 x FETCH = %.3f GB = %.0f GB/s: not a bound (reads beyond the algorithmic ones are L2 misses of the weight stream)"
              % ((wr + 2 * fe) / 1e9, (wr + 2 * fe) / t / 1e9)]
    if (k, "TCC_HIT_sum") in agg:
        lines.append("#   L2 hit rate %.1f %%" % (100 * g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))))
    lines.append("#   clock = GRBM_GUI_ACTIVE / 8 / time = %.2f GHz;  MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles) = %.1f %%"
                 % (clk / 1e9, 100 * g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024 * g("GRBM_GUI_ACTIVE") / 8)))
    mops = g("SQ_INSTS_VALU_MFMA_MOPS_BF16") if k.startswith("bf16") else g("SQ_INSTS_VALU_MFMA_MOPS_F32")
    lines.append("#   executed MFMA FLOPs = MOPS x 512 = %.3e" % (mops * 512))
    lines.append("#   wave time: WAIT_ANY %.1f %%, WAIT_INST_ANY %.1f %%, ACTIVE_INST_ANY %.1f %% of SQ_WAVE_CYCLES; LDS bank conflict cycles / LDS instructions = %.3f"
                 % (100 * g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), 100 * g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), 100 * g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES"),
                    g("SQ_LDS_BANK_CONFLICT") / g("SQ_INSTS_LDS")))
    summary[k] = {"ms": t * 1e3, "traffic_bytes": wr + 2 * fe, "algorithmic_bytes": alg_bytes, "write_bytes": wr, "fetch_bytes_corrected": 2 * fe,
                  "clock_ghz": clk / 1e9, "mfma_busy": g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024 * g("GRBM_GUI_ACTIVE") / 8)}
open(os.path.join(out, tag + "_pmc_summary.csv"), "w").write("\n".join(lines) + "\n")
json.dump(summary, open(os.path.join(out, tag + "_pmc_summary.json"), "w"), indent=1)
print("\n".join(lines[-30:]))
