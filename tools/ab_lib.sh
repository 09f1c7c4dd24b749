#!/bin/bash
# Same-box A/B of library builds on the training step (rocprofv3 kernel trace):  tools/ab_lib.sh <leg: x3fwd|bf16x3|fp32> <pattern> default prev ...
# ("default" = the shipped library, anything else = build/variants/libsahs_<name>.so from `python tools/ablate.py build <name>`); prints ms per step
# and the kernels whose names match <pattern>
LEG=$1; PAT=$2; shift 2
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p /tmp/ab
for V in "$@"; do
  if [ "$V" = default ]; then unset SAHS_NERF_LIB; else export SAHS_NERF_LIB=$GRAFT_REPO_ROOT/sahs-deformable-nerf_amd/build/variants/libsahs_$V.so; fi
  rm -rf /tmp/ab/$V
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/ab/$V -o t -- python3 tools/train_legs.py --only $LEG --steps 10 --warmup 3 > /tmp/ab/$V.json 2> /tmp/ab/$V.err
  echo "== $V $(python3 -c "import json;d=json.load(open('/tmp/ab/$V.json'));print([round(v['ms_per_step'],3) for v in d.values() if isinstance(v,dict)])") ms/step"
  python3 tools/kstats_db.py /tmp/ab/$V/t_results.db 10 13 | grep -i "$PAT"
done
