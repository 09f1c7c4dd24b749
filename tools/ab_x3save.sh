#!/bin/bash
# Same-box A/B of saving-forward builds (training step with the forward on the split-operand kernels) under rocprofv3:
#   tools/ab_x3save.sh default x3saveplain ...     (variants built beforehand by `python tools/ablate.py build x3saveplain,...`)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ab_x3save
rm -rf $OUT; mkdir -p $OUT
for V in "$@"; do
  if [ "$V" = default ]; then unset SAHS_NERF_LIB; else export SAHS_NERF_LIB=$GRAFT_REPO_ROOT/sahs-deformable-nerf_amd/build/variants/libsahs_$V.so; fi
  rocprofv3 --kernel-trace --stats -d $OUT/$V -o t -- python3 tools/train_legs.py --only x3fwd --steps 10 --warmup 3 > $OUT/$V.json 2> $OUT/$V.err
  echo "== $V: $(python3 -c "import json;d=json.load(open('$OUT/$V.json'));print(round(d['train_T2048_x3fwd']['ms_per_step'],3),'ms/step')")"
  python3 tools/kstats_db.py $OUT/$V/t_results.db 8 13 | grep -i "bf16x3_kernel\|chain\|tn_jobs\|total"
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${V}_w -- python3 tools/train_legs.py --only x3fwd --steps 2 --warmup 1 > /dev/null 2> $OUT/${V}_w.err
  python3 - <<PY
import csv,glob
rows=list(csv.DictReader(open(glob.glob("$OUT/${V}_w/*/*counter_collection.csv")[0])))
idx=[i for i,r in enumerate(rows) if "conditioning_backward" in r["Kernel_Name"]]
print("   forward kernels' WRITE_SIZE (GB):", [round(float(r["Counter_Value"])*1024/1e9,3) for r in rows[idx[-2]+1:idx[-1]+1] if "bf16x3_kernel" in r["Kernel_Name"]])
PY
done
