#!/bin/bash
# Same-box A/B of backward-chain builds under rocprofv3 (kernel trace only): tools/ab_chain.sh default cnostore cnomask ...
# (variants built beforehand by `python tools/ablate.py build cnostore,...`; ablated builds give WRONG results by construction)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ab_chain
rm -rf $OUT; mkdir -p $OUT
for V in "$@"; do
  if [ "$V" = default ]; then unset SAHS_NERF_LIB; else export SAHS_NERF_LIB=$GRAFT_REPO_ROOT/sahs-deformable-nerf_amd/build/variants/libsahs_$V.so; fi
  rocprofv3 --kernel-trace --stats -d $OUT/$V -o t -- python3 tools/time_bwd_parts.py --reps 5 > $OUT/$V.json 2> $OUT/$V.err
  echo "== $V: $(cut -c1-160 $OUT/$V.json)"
  python3 tools/kstats_db.py $OUT/$V/t_results.db 6 6 | grep -i "chain\|tn_jobs\|total"
done
