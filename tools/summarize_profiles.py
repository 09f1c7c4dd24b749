#!/usr/bin/env python3
"""Condense gpurun_out/prof_<round> (tools/profile.sh <round>) into the tracked summaries under profiles/ (legs that were not run are skipped): per-kernel statistics of the fp32
headline, the bf16 / bf16x3 / NeRFace mixed-precision legs and the training step, the PMC summary of the field kernels' dominant
dispatches (+ the small JSON that bench.py quotes as roofline.traffic), and the per-kernel HBM traffic of the training step."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r4"
base = os.path.join(REPO, "gpurun_out", "prof_" + tag)
out = os.path.join(REPO, "profiles")
B = "python3 bench.py --no-secondary --no-cpu-baseline"


def short(name):
    return name if len(name) <= 150 else name[:147] + "..."


def kernel_stats(sub, dst, header):
    ks = newest(os.path.join(base, sub, "*", "*_kernel_stats.csv"))
    rows = list(csv.DictReader(open(ks)))
    with open(os.path.join(out, dst), "w") as f:
        f.write("# " + header + "\n")
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    return rows


def newest(pattern):
    """gpurun MERGES a call's files into gpurun_out/: an earlier call's run directory (another pid) may still sit beside the new one."""
    return max(glob.glob(pattern), key=os.path.getmtime)


def mode_of(kn, lead):
    for m in "012":
        if "%s%s>" % (lead, m) in kn or "%s%s," % (lead, m) in kn or "ILi%sE" % m in kn or "ELi%sE" % m in kn:
            return m
    return None


def kind(kn, leg):
    """Which field launch a kernel name is, within profiling leg `leg` (f32 | bf16 | bf16x3 | nfmixed)."""
    nf = "sahs_nf" in kn
    if "field_radiance_bf16x3_kernel" in kn:
        return "bf16x3_radiance"
    if "field_deform_bf16x3_kernel" in kn:
        return "nf_x3_deform" if nf else "bf16x3_deform"
    if "field_forward_bf16w" in kn:
        m = mode_of(kn, "kernel<")
        return None if m is None else ("nf_bf16_" if nf else "bf16_") + {"0": "all", "1": "deform", "2": "radiance"}[m]
    if "field_forward_f32_kernel" in kn and ("<false, " in kn or "ILb0E" in kn):
        m = mode_of(kn, "<false, ")
        if m is None:
            return None
        lab = {"0": "all", "1": "deform", "2": "radiance"}[m]
        return ("nf_f32_" if nf else ("x3leg_f32_" if leg == "bf16x3" else "f32_")) + lab
    return None


stats = {}
for leg, args, steps, what in (("f32", "--precision fp32", 8, "fp32 W512 headline; per 131,072-ray chunk: field<false,0> = coarse launch (8.39 M samples, whole network), "
                                "<false,1> = deformation nets on the 8.39 M new depths, <false,2> = radiance nets on the 16.78 M fine samples"),
                               ("bf16", "--precision bf16", 100, "bf16 W512: field_forward_bf16w_kernel<0|1|2>, same three launches per chunk"),
                               ("bf16x3", "--precision bf16x3", 25, "bf16x3 W512: field_deform_bf16x3_kernel (deformation nets, coarse and new depths: 8.39 M samples each) + "
                                "field_radiance_bf16x3_kernel on the 8.39 M coarse and the 16.78 M fine samples of a chunk"),
                               ("nfmixed", "--arch nerface --precision bf16", 80, "NeRFaceModel (config/expression/person_2.yml) in mixed precision: deformation launches with split bf16 "
                                "operands (sahs_nf::hx3::field_deform_bf16x3_kernel) + sahs_nf::field_forward_bf16w_kernel<2> radiance launches")):
    if not glob.glob(os.path.join(base, "trace_" + leg, "*", "*_kernel_stats.csv")):
        continue
    stats[leg] = kernel_stats("trace_" + leg, "%s_%s_kernel_stats.csv" % (tag, {"f32": "fp32"}.get(leg, leg)),
                              "rocprofv3 --kernel-trace --stats -- %s %s --steps %d --warmup 1   (%s)" % (B, args, steps, what))
    shutil.copy(os.path.join(base, "bench_trace_%s.json" % leg), os.path.join(out, "%s_%s_bench_under_rocprof.json" % (tag, {"f32": "fp32"}.get(leg, leg))))
for sub, name, arg, what in (("trace_train", "train_T2048_bf16x3", "bf16x3", "the default backward: field_backward_chain_{rad,def}_kernel = the sample-major data-gradient chains, "
                              "gemm_tn_jobs_kernel = every weight gradient of a part in one launch, both on the bf16 pipe with split operands"),
                             ("trace_train_x3fwd", "train_T2048_x3fwd", "x3fwd", "opt-in: the saving forward on the split-operand kernels too -- field_{radiance,deform}_bf16x3_kernel<true>"),
                             ("trace_train_f32", "train_T2048", "fp32", "backward products in f32 (the reference's arithmetic): the fused walk in f32 -- field_backward_chain_{rad,def}_f32_kernel, "
                              "gemm_tn_jobs_f32_kernel / gemm_tn_jobs256_f32_kernel")):
    if glob.glob(os.path.join(base, sub, "*", "*_kernel_stats.csv")):
        stats[sub] = kernel_stats(sub, "%s_%s_kernel_stats.csv" % (tag, name),
                                  "rocprofv3 --kernel-trace --stats -- python3 tools/train_legs.py --only %s --steps 5 --warmup 2   (T2048: 2048-ray forward + backward, 7 steps; %s)" % (arg, what))
        shutil.copy(os.path.join(base, "bench_" + sub + ".json"), os.path.join(out, "%s_%s_bench_under_rocprof.json" % (tag, name)))

agg = collections.OrderedDict()
for leg in ("f32", "bf16", "bf16x3", "nfmixed"):
    for f in sorted(newest(os.path.join(d, "*", "*counter_collection.csv")) for d in glob.glob(os.path.join(base, "pmc_%s_*" % leg))):
        for r in csv.DictReader(open(f)):
            k = kind(r["Kernel_Name"], leg)
            if k:
                agg.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
dur = {}
for leg in ("f32", "bf16", "bf16x3", "nfmixed"):
    for r in stats.get(leg, []):
        k = kind(r["Name"], leg)
        if k:
            dur[k] = (float(r["MinNs"]), float(r["MaxNs"]), float(r["AverageNs"]), int(r["Calls"]))
lines = ["# rocprofv3 PMC summary (separate passes per counter group, tools/profile.sh), bench.py --steps 1 --warmup 0: the first frame's dispatches,",
         "# two 131,072-ray chunks.  Per kernel and counter: the values of its first dispatches in launch order.",
         "kernel,counter,dispatch_0,dispatch_1,dispatch_2,dispatch_3"]
for (k, c), v in agg.items():
    lines.append("%s,%s,%s" % (k, c, ",".join("%.6g" % x for x in v[:4])))
summary = {}
P_FINE = 16777216
# (kind, label, which dispatch of the first frame, samples, algorithmic HBM bytes)
for k, label, pick, samples, alg_bytes in (
        ("f32_radiance", "fp32 radiance-net launch (the dominant dispatch: 16.78 M fine samples)", 0, P_FINE, P_FINE * (64 + 32 + 4)),
        ("f32_all", "fp32 whole-network launch (coarse pass: 8.39 M samples)", 0, P_FINE // 2, (P_FINE // 2) * (64 + 4 + 32)),
        ("f32_deform", "fp32 deformation-net launch (8.39 M new depths)", 0, P_FINE // 2, (P_FINE // 2) * (4 + 32)),
        ("bf16_radiance", "bf16 radiance-net launch (16.78 M fine samples)", 0, P_FINE, P_FINE * (64 + 32 + 4)),
        ("bf16_all", "bf16 whole-network launch (coarse pass: 8.39 M samples)", 0, P_FINE // 2, (P_FINE // 2) * (64 + 4 + 32)),
        ("bf16_deform", "bf16 deformation-net launch (8.39 M new depths)", 0, P_FINE // 2, (P_FINE // 2) * (4 + 32)),
        ("bf16x3_radiance", "bf16x3 radiance-net launch over the 16.78 M fine samples (second dispatch of a chunk; the first is the coarse pass's 8.39 M)",
         1, P_FINE, P_FINE * (64 + 32 + 4)),
        ("bf16x3_deform", "bf16x3 deformation-net launch (8.39 M depths)", 0, P_FINE // 2, (P_FINE // 2) * (4 + 32)),
        ("nf_x3_deform", "NeRFaceModel (mixed precision) deformation-net launch with split bf16 operands (8.39 M depths)", 0, P_FINE // 2, (P_FINE // 2) * (4 + 32)),
        ("nf_bf16_radiance", "NeRFaceModel bf16 radiance-net launch over the 16.78 M fine samples (second dispatch of a chunk)", 1, P_FINE, P_FINE * (64 + 32 + 4))):
    if (k, "GRBM_GUI_ACTIVE") not in agg or len(agg[(k, "GRBM_GUI_ACTIVE")]) <= pick:
        continue
    g = lambda c: agg[(k, c)][pick] if (k, c) in agg and len(agg[(k, c)]) > pick else float("nan")
    # this dispatch's own duration under the (untraced) counter pass is not recorded: scale the traced average by the sample count
    wr, fe = g("WRITE_SIZE") * 1024, g("FETCH_SIZE") * 1024
    cyc = g("GRBM_GUI_ACTIVE") / 8
    t = (dur[k][1] if pick == 1 else dur[k][2]) * 1e-9 if k in dur else float("nan")
    if k in ("bf16x3_radiance", "nf_bf16_radiance") and k in dur:
        t = dur[k][1] * 1e-9      # the fine launch is the longer of the kernel's two dispatch sizes
    clk = cyc / t
    lines += ["", "# %s: %.2f ms under rocprof" % (label, t * 1e3),
              "#   WRITE_SIZE = %.3f GB, FETCH_SIZE = %.3f GB as reported, x2 (gfx950: 16 B/lane streaming reads count half) = %.3f GB; algorithmic bytes %.3f GB"
              % (wr / 1e9, fe / 1e9, 2 * fe / 1e9, alg_bytes / 1e9),
              "#   HBM-side traffic WRITE + 2 x FETCH = %.3f GB = %.0f GB/s: not a bound (reads beyond the algorithmic ones are L2 misses of the weight stream)"
              % ((wr + 2 * fe) / 1e9, (wr + 2 * fe) / t / 1e9)]
    if (k, "TCC_HIT_sum") in agg:
        lines.append("#   L2 hit rate %.1f %%" % (100 * g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))))
    lines.append("#   clock = GRBM_GUI_ACTIVE / 8 / time = %.2f GHz;  MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles) = %.1f %%"
                 % (clk / 1e9, 100 * g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024 * cyc)))
    mops = g("SQ_INSTS_VALU_MFMA_MOPS_F32") if k.startswith("f32") else g("SQ_INSTS_VALU_MFMA_MOPS_BF16")
    lines.append("#   executed MFMA FLOPs = MOPS x 512 = %.3e" % (mops * 512))
    lines.append("#   wave time: WAIT_ANY %.1f %%, WAIT_INST_ANY %.1f %%, ACTIVE_INST_ANY %.1f %% of SQ_WAVE_CYCLES; LDS bank conflict cycles / LDS instructions = %.3f"
                 % (100 * g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), 100 * g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), 100 * g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES"),
                    g("SQ_LDS_BANK_CONFLICT") / g("SQ_INSTS_LDS")))
    summary[k] = {"ms": t * 1e3, "traffic_bytes": wr + 2 * fe, "algorithmic_bytes": alg_bytes, "write_bytes": wr, "fetch_bytes_corrected": 2 * fe,
                  "clock_ghz": clk / 1e9, "mfma_busy": g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024 * cyc)}

# ---- training step: HBM traffic per kernel (FETCH x2-corrected + WRITE), last step of the counter runs ----
def last_step(rows):
    idx = [i for i, r in enumerate(rows) if "conditioning_backward" in r["Kernel_Name"]]
    return rows[idx[-2] + 1: idx[-1] + 1]


for pmc_tag, leg_arg, key in (("pmc_train", "bf16x3", "train_T2048"), ("pmc_trainx3", "x3fwd", "train_T2048_x3fwd"), ("pmc_trainf32", "fp32", "train_T2048_f32")):
  try:
    fs = last_step(list(csv.DictReader(open(newest(os.path.join(base, pmc_tag + "_FETCH_SIZE", "*", "*counter_collection.csv"))))))
    ws = last_step(list(csv.DictReader(open(newest(os.path.join(base, pmc_tag + "_WRITE_SIZE", "*", "*counter_collection.csv"))))))
    tr = collections.OrderedDict()
    for a, b in zip(fs, ws):
        n = a["Kernel_Name"]
        nm = ("weight gradients over job tables, f32: gemm_tn_jobs_f32_kernel + gemm_tn_jobs256_f32_kernel" if "gemm_tn_jobs" in n and "_f32_kernel" in n else
              "weight gradients over job tables: gemm_tn_jobs_kernel + gemm_tn_jobs256_kernel" if "gemm_tn_jobs" in n else "data-gradient chain field_backward_chain_*_kernel" if "field_backward_chain" in n else
              "weight-gradient GEMM gemm_tn_split_kernel" if "gemm_tn_split_kernel" in n else "weight-gradient GEMM gemm_dma_kernel<true,*>" if "gemm_dma_kernel<true" in n else "data-gradient GEMM gemm_dma_kernel<false,true>" if "gemm_dma_kernel<false" in n
              else "field_forward_f32_kernel<true,*> (activation-saving forward)" if "field_forward" in n
              else "field_{radiance,deform}_bf16x3_kernel<true> (activation-saving forward, split operands)" if "bf16x3_kernel" in n else "gemm_f32_kernel" if "gemm_f32" in n else "other")
        d = tr.setdefault(nm, [0, 0.0, 0.0, 0.0])
        d[0] += 1
        d[1] += float(a["Counter_Value"]) * 2048
        d[2] += float(b["Counter_Value"]) * 1024
        d[3] += (int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) * 1e-9
    lines += ["", "# training step T2048 (tools/train_legs.py --only %s), one step: HBM-side bytes per kernel group (2 x FETCH_SIZE + WRITE_SIZE) and the rate over the kernels' time" % leg_arg,
              "train_kernel_group,launches,fetch_GB_x2,write_GB,time_ms,TB_per_s"]
    tot = [0.0, 0.0, 0.0]
    for nm, d in tr.items():
        lines.append("%s,%d,%.3f,%.3f,%.3f,%.2f" % (nm, d[0], d[1] / 1e9, d[2] / 1e9, d[3] * 1e3, (d[1] + d[2]) / d[3] / 1e12))
        tot = [tot[0] + d[1], tot[1] + d[2], tot[2] + d[3]]
    lines.append("total,,%.3f,%.3f,%.3f,%.2f" % (tot[0] / 1e9, tot[1] / 1e9, tot[2] * 1e3, (tot[0] + tot[1]) / tot[2] / 1e12))
    summary[key] = {"fetch_bytes_corrected": tot[0], "write_bytes": tot[1], "kernel_time_ms": tot[2] * 1e3}
  except (IndexError, OSError) as e:
    lines.append("# training-step counters missing (%s): %r" % (key, e))
# ---- GPU power / clock sampled by rocm-smi beside each traced leg (tools/profile.sh) ----
def smi_summary(path):
    """-> list of (power W, sclk MHz) samples"""
    out = []
    for ln in open(path):
        try:
            d = json.loads(ln)
        except ValueError:
            continue
        for card in d.values():
            if not isinstance(card, dict):
                continue
            pw = ck = None
            for k, v in card.items():
                kl = k.lower()
                try:
                    if "power" in kl and "(w)" in kl:
                        pw = float(v)
                    elif kl.startswith("sclk clock speed"):
                        ck = float(str(v).strip("()").lower().replace("mhz", ""))
                except ValueError:
                    pass
            if pw is not None and ck is not None:
                out.append((pw, ck))
    return out


for leg in ("f32", "bf16", "bf16x3", "nfmixed"):
    f = os.path.join(base, "smi_%s.jsonl" % leg)
    if os.path.exists(f):
        sm = smi_summary(f)
        if sm:
            top = max(p for p, _ in sm)
            busy = [(p, c) for p, c in sm if p > 0.8 * top]
            mp, mc = sum(p for p, _ in busy) / len(busy), sum(c for _, c in busy) / len(busy)
            lines += ["", "# rocm-smi beside the %s trace leg: %d samples, %d of them under load (socket power > 0.8 x its maximum %.0f W): mean %.0f W at a mean sclk of %.0f MHz "
                      "(min %.0f, max %.0f MHz under load)" % (leg, len(sm), len(busy), top, mp, mc, min(c for _, c in busy), max(c for _, c in busy))]
            summary.setdefault("smi", {})[leg] = {"samples_under_load": len(busy), "power_max_w": top, "power_mean_w": mp, "sclk_mean_mhz": mc}
open(os.path.join(out, tag + "_pmc_summary.csv"), "w").write("\n".join(lines) + "\n")
json.dump(summary, open(os.path.join(out, tag + "_pmc_summary.json"), "w"), indent=1)
print("\n".join(lines[-40:]))
