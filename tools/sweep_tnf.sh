#!/bin/bash
# Sweep of the fp32 weight-gradient launches' item cost models on the training step (tuning aid):  tools/sweep_tnf.sh
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" bash tools/ab_lib.sh fp32 "jobs" default 2>&1 | grep -v "^== default" | cut -c1-60,100-170; env "$@" python3 tools/train_legs.py --only fp32 --steps 10 --warmup 3 2>/dev/null | python3 -c "import json,sys;print(round(json.load(sys.stdin)['train_T2048']['ms_per_step'],3),'ms/step')"; }
for c in 0.55 0.9 1.3 2.0 3.0; do run SAHS_TNF_C0N=$c; done
for c in 0.27 0.6 1.0 1.6; do run SAHS_TNF_C0W=$c; done
for r in 1 3 4; do run SAHS_TNF_ROUNDS=$r SAHS_TNF_C0N=0.9; done
