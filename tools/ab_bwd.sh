#!/bin/bash
# Same-box A/B of backward-GEMM builds under rocprofv3 (kernel trace only): tools/ab_bwd.sh base gnocompute gnodma ...
# (variants built beforehand by `python tools/ablate.py build base,gnocompute,...`; ablated builds give WRONG results by construction)
set -e
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ab_bwd
rm -rf $OUT; mkdir -p $OUT
for V in "$@"; do
  export SAHS_NERF_LIB=$GRAFT_REPO_ROOT/sahs-deformable-nerf_amd/build/variants/libsahs_$V.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$V -- python3 tools/train_bench.py --steps 4 --warmup 2 > $OUT/$V.json 2> $OUT/$V.err
  echo "== $V: $(cut -c1-120 $OUT/$V.json)"
  python3 tools/kstats.py $OUT/$V 6 4 | grep -i "total\|gemm"
done
