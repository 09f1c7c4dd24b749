#!/bin/bash
# SQ / L2 counters of the training step's kernels (separate passes; kernel trace only beside them):  tools/pmc_train_sq.sh [x3fwd|bf16x3|fp32]
LEG=${1:-x3fwd}
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_train_sq; rm -rf $OUT; mkdir -p $OUT
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/p$i -o t -- python3 tools/train_legs.py --only $LEG --steps 2 --warmup 1 > $OUT/p$i.json 2> $OUT/p$i.err || echo "pass $i failed"
done
python3 - <<'PY'
import csv,glob,collections
agg=collections.OrderedDict()
for f in sorted(glob.glob("gpurun_out/pmc_train_sq/p*/**/*counter_collection.csv", recursive=True)):
    rows=list(csv.DictReader(open(f)))
    cond=sorted({int(r["Dispatch_Id"]) for r in rows if "conditioning_backward" in r["Kernel_Name"]})      # one per step
    if len(cond) < 2: continue
    for r in [r for r in rows if cond[-2] < int(r["Dispatch_Id"]) <= cond[-1]]:
        n=r["Kernel_Name"]
        k=("tn_jobs256" if "jobs256" in n else "tn_jobs" if "tn_jobs" in n else "chain_rad" if "chain_rad" in n else "chain_def" if "chain_def" in n else
           "grid_bwd" if "grid_backward" in n else "encode_bwd" if "encode_backward" in n else "fwd_radiance" if "radiance_bf16x3" in n else "fwd_deform" if "deform_bf16x3" in n else "fwd_f32" if "field_forward_f32" in n else None)
        if k is None: continue
        d=agg.setdefault(k, collections.OrderedDict())
        d[r["Counter_Name"]]=d.get(r["Counter_Name"],0.0)+float(r["Counter_Value"])
        d["_ns_"+r["Counter_Name"]]=d.get("_ns_"+r["Counter_Name"],0.0)+int(r["End_Timestamp"])-int(r["Start_Timestamp"])
for k,d in agg.items():
    g=lambda c: d.get(c,0.0)
    ns=g("_ns_SQ_BUSY_CYCLES") or 1
    print("%-12s time %.3f ms | clock %.2f GHz | mfma_busy/busy %.3f | wait_any/wave_cycles %.3f | active_any/wave_cycles %.3f | lds active/valu active %.2f | bank conflict cycles/lds active %.3f | L2 hit %.3f | EA rd %.3g wr %.3g"
          % (k, ns/1e6, g("GRBM_GUI_ACTIVE")/8.0/ns, g("SQ_VALU_MFMA_BUSY_CYCLES")/(g("SQ_BUSY_CYCLES") or 1), g("SQ_WAIT_INST_ANY")/(g("SQ_WAVE_CYCLES") or 1), g("SQ_ACTIVE_INST_ANY")/(g("SQ_WAVE_CYCLES") or 1),
             g("SQ_ACTIVE_INST_LDS")/(g("SQ_ACTIVE_INST_VALU") or 1), g("SQ_LDS_BANK_CONFLICT")/(g("SQ_ACTIVE_INST_LDS") or 1), g("TCC_HIT_sum")/((g("TCC_HIT_sum")+g("TCC_MISS_sum")) or 1), g("TCC_EA_RDREQ_sum"), g("TCC_EA_WRREQ_sum")))
    print("             raw:", {c: v for c, v in d.items() if not c.startswith("_ns_")})
PY
