#!/usr/bin/env python3
"""Condense gpurun_out/prof_r1 (tools/profile_r1.sh) into the tracked summaries under profiles/."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = os.path.join(REPO, "gpurun_out", "prof_r1")
out = os.path.join(REPO, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
ks = glob.glob(os.path.join(base, "trace", "*", "*_kernel_stats.csv"))[0]
rows = list(csv.DictReader(open(ks)))
with open(os.path.join(out, tag + "_kernel_stats.csv"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline (fp32 headline + bf16 leg + NeRFaceModel fp32 leg)\n")
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        name = r["Name"]
        if len(name) > 160:
            name = name[:157] + "..."
        w.writerow([name, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
shutil.copy(os.path.join(base, "bench_trace.json"), os.path.join(out, tag + "_bench_under_rocprof.json"))
agg = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(base, "pmc_*", "*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        k = ("nerface_f32" if "sahs_nf::" in kn else "f32") if "field_forward_f32" in kn else ("bf16" if "field_forward_bf16" in kn else None)
        if k:
            agg.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
stats = {r["Name"]: r for r in rows}
dur = {}
for name, r in stats.items():
    if "field_forward_f32" in name:
        dur["nerface_f32" if "sahs_nf::" in name else "f32"] = (float(r["MinNs"]), float(r["MaxNs"]), float(r["AverageNs"]))
    if "field_forward_bf16" in name:
        dur["bf16"] = (float(r["MinNs"]), float(r["MaxNs"]), float(r["AverageNs"]))
lines = ["# rocprofv3 PMC summary (separate passes per counter group, tools/profile_r1.sh), bench.py --steps 1 --warmup 0",
         "# dispatch order per frame: [chunk0 coarse (8.39M samples), chunk0 fine (16.78M), chunk1 coarse, chunk1 fine]; first frame shown",
         "kernel,counter,coarse_0,fine_0,coarse_1,fine_1"]
for (k, c), v in agg.items():
    lines.append("%s,%s,%s" % (k, c, ",".join("%.6g" % x for x in v[:4])))
P_FINE = 16777216
for k, peak_flop_per_mop in (("f32", 512), ("bf16", 512), ("nerface_f32", 512)):
    if k not in dur:
        continue
    t_fine = dur[k][1] * 1e-9   # the longest dispatch is a fine launch
    g = lambda c: agg[(k, c)][1]
    lines.append("")
    lines.append("# %s field kernel, fine launch (P = %d samples, %.2f ms under rocprof; average over all launches %.2f ms):" % (k, P_FINE, t_fine * 1e3, dur[k][2] * 1e-6))
    wr, fe = g("WRITE_SIZE") * 1024, g("FETCH_SIZE") * 1024
    lines.append("#   WRITE_SIZE = %.3f GB (algorithmic raw output P x 64 B = %.3f GB)" % (wr / 1e9, P_FINE * 64 / 1e9))
    lines.append("#   FETCH_SIZE = %.2f GB as reported, x2 gfx950 correction for 16 B/lane streaming = %.2f GB: L2 misses of the LDS-DMA weight stream" % (fe / 1e9, 2 * fe / 1e9))
    hit = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))
    lines.append("#   L2 hit rate %.1f %%; (2 x FETCH + WRITE) / time = %.0f GB/s of fabric traffic: not a bound" % (100 * hit, (2 * fe + wr) / t_fine / 1e9))
    clk = g("GRBM_GUI_ACTIVE") / 8 / t_fine
    lines.append("#   clock = GRBM_GUI_ACTIVE / 8 / time = %.2f GHz" % (clk / 1e9))
    lines.append("#   MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles) = %.1f %%" % (100 * g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024 * g("GRBM_GUI_ACTIVE") / 8)))
    mops = g("SQ_INSTS_VALU_MFMA_MOPS_BF16") if k == "bf16" else g("SQ_INSTS_VALU_MFMA_MOPS_F32")
    lines.append("#   executed MFMA FLOPs = MOPS x 512 = %.3e (algorithmic %.3e)" % (mops * 512, P_FINE * (1438336.0 if k == "nerface_f32" else 1855744.0)))
    lines.append("#   wave time: WAIT_ANY %.1f %%, WAIT_INST_ANY %.1f %%, ACTIVE_INST_ANY %.1f %% of SQ_WAVE_CYCLES; LDS bank conflict cycles / LDS instructions = %.3f" % (
        100 * g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), 100 * g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), 100 * g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES"),
        g("SQ_LDS_BANK_CONFLICT") / g("SQ_INSTS_LDS")))
open(os.path.join(out, tag + "_pmc_summary.csv"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[-22:]))
