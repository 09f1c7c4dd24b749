#!/bin/bash
# Same-box A/B of weight-gradient builds (training step, default form) under rocprofv3:  tools/ab_tn.sh default tnnodma ...
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ab_tn
rm -rf $OUT; mkdir -p $OUT
for V in "$@"; do
  if [ "$V" = default ]; then unset SAHS_NERF_LIB; else export SAHS_NERF_LIB=$GRAFT_REPO_ROOT/sahs-deformable-nerf_amd/build/variants/libsahs_$V.so; fi
  rocprofv3 --kernel-trace --stats -d $OUT/$V -o t -- python3 tools/train_legs.py --only x3fwd --steps 10 --warmup 3 > $OUT/$V.json 2> $OUT/$V.err
  echo "== $V"
  python3 tools/kstats_db.py $OUT/$V/t_results.db 8 13 | grep -i "tn_jobs\|total\|chain\|bf16x3_kernel"
done
