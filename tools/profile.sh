#!/bin/bash
# rocprofv3 passes of a round (run on the GPU box from the repo root; the program itself follows "--", no wrapper):
#   tools/profile.sh r4 [legs...]      legs: f32 bf16 bf16x3 nfmixed train pmc   (default: all)
# kernel traces of the fp32 headline, the bf16 / bf16x3 / NeRFace mixed-precision legs and the training step, then separate PMC passes
# (counters never together with trace domains other than the kernel trace).  Condensed into profiles/<round>_* by tools/summarize_profiles.py.
TAG=${1:-r4}; shift
LEGS=${@:-f32 bf16 bf16x3 nfmixed train pmc}
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
B="python3 bench.py --no-secondary --no-cpu-baseline"
has() { [[ " $LEGS " == *" $1 "* ]]; }
for L in "f32|--precision fp32|8" "bf16|--precision bf16|100" "bf16x3|--precision bf16x3|25" "nfmixed|--arch nerface --precision bf16|80"; do
  IFS="|" read NAME ARGS STEPS <<< "$L"
  has $NAME || continue
  rm -rf $OUT/trace_$NAME
  # GPU power / clock while the leg runs ("power-limited" as a measurement): rocm-smi sampled beside the profiled program
  ( while true; do rocm-smi --showpower --showclocks --json 2>/dev/null | tr -d '\n'; echo; sleep 0.1; done ) > $OUT/smi_$NAME.jsonl &
  SMI=$!
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$NAME -- $B $ARGS --steps $STEPS --warmup 1 > $OUT/bench_trace_$NAME.json 2> $OUT/trace_$NAME.err
  kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
  echo "[profile] $NAME trace done"
done
if has train; then
  rm -rf $OUT/trace_train
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_train -- python3 tools/train_legs.py --only bf16x3 --steps 5 --warmup 2 > $OUT/bench_trace_train.json 2> $OUT/trace_train.err
  rm -rf $OUT/trace_train_f32
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_train_f32 -- python3 tools/train_legs.py --only fp32 --steps 5 --warmup 2 > $OUT/bench_trace_train_f32.json 2> $OUT/trace_train_f32.err
  rm -rf $OUT/trace_train_x3fwd
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_train_x3fwd -- python3 tools/train_legs.py --only x3fwd --steps 5 --warmup 2 > $OUT/bench_trace_train_x3fwd.json 2> $OUT/trace_train_x3fwd.err
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/pmc_train_$C $OUT/pmc_trainx3_$C $OUT/pmc_trainf32_$C
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_trainf32_$C -- python3 tools/train_legs.py --only fp32 --steps 2 --warmup 1 > $OUT/pmc_trainf32_$C.json 2> $OUT/pmc_trainf32_$C.err
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_train_$C -- python3 tools/train_legs.py --only bf16x3 --steps 2 --warmup 1 > $OUT/pmc_train_$C.json 2> $OUT/pmc_train_$C.err
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_trainx3_$C -- python3 tools/train_legs.py --only x3fwd --steps 2 --warmup 1 > $OUT/pmc_trainx3_$C.json 2> $OUT/pmc_trainx3_$C.err
  done
  echo "[profile] training trace + counters done"
fi
if has pmc; then
  for L in "f32|--precision fp32|SQ_INSTS_VALU_MFMA_MOPS_F32" "bf16|--precision bf16|SQ_INSTS_VALU_MFMA_MOPS_BF16" "bf16x3|--precision bf16x3|SQ_INSTS_VALU_MFMA_MOPS_BF16" "nfmixed|--arch nerface --precision bf16|SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
    IFS="|" read P ARGS MOPS <<< "$L"
    has $P || continue
    CMD="$B $ARGS"
    rm -rf $OUT/pmc_${P}_*
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_${P}_fetch -- $CMD --steps 1 --warmup 0 > $OUT/b_${P}_fetch.json 2> $OUT/${P}_fetch.err
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_${P}_write -- $CMD --steps 1 --warmup 0 > $OUT/b_${P}_write.json 2> $OUT/${P}_write.err
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY $MOPS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_${P}_sq -- $CMD --steps 1 --warmup 0 > $OUT/b_${P}_sq.json 2> $OUT/${P}_sq.err || echo "sq pass failed" >> $OUT/${P}_sq.err
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_${P}_sq2 -- $CMD --steps 1 --warmup 0 > $OUT/b_${P}_sq2.json 2> $OUT/${P}_sq2.err || echo "sq2 pass failed" >> $OUT/${P}_sq2.err
    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_${P}_l2 -- $CMD --steps 1 --warmup 0 > $OUT/b_${P}_l2.json 2> $OUT/${P}_l2.err || echo "l2 pass failed" >> $OUT/${P}_l2.err
    echo "[profile] $P counters done"
  done
fi
find $OUT -name "*.csv" | wc -l
python3 tools/summarize_profiles.py $TAG
