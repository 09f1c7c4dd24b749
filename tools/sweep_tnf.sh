#!/bin/bash
# Sweep of the fp32 weight-gradient launches' item cost models on the training step (tuning aid):  tools/sweep_tnf.sh
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" bash tools/ab_lib.sh fp32 "jobs_f32" default 2>&1 | grep -v "^== default" | cut -c1-60,100-170; }
for c in 0.1 0.2 0.4 0.7 1.0 1.3 1.8; do run SAHS_TNF_C0N=$c; done
for r in 2 4; do run SAHS_TNF_ROUNDS=$r; done
for c in 0.15 0.27 0.45; do run SAHS_TNF_C0W=$c; done
