#!/bin/bash
set -e
export TMPDIR=/tmp
OUT=gpurun_out/prof_bf16_${1:-a}
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -- python3 bench.py --precision bf16 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/b1.json 2> $OUT/p1.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/p2 -- python3 bench.py --precision bf16 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/b2.json 2> $OUT/p2.err || echo p2 failed
python3 - <<'PY'
import csv, glob, collections
import sys
out = sys.argv[1] if len(sys.argv) > 1 else None
for d in sorted(glob.glob("gpurun_out/prof_bf16_*/p*/*/*counter_collection.csv")):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(d)):
        if "field_forward" in r["Kernel_Name"]:
            agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    print(d)
    for k, v in agg.items():
        print("   %-28s %s" % (k, " ".join("%.4g" % x for x in v[-4:])))
PY
