#!/bin/bash
# rocprofv3 passes of round 2 (run on the GPU box from the repo root; the program itself follows "--", no wrapper): kernel traces of
# the fp32 headline, the bf16 leg and the training step, then separate PMC passes (counters never together with trace domains).
set -e
export TMPDIR=/tmp
OUT=gpurun_out/prof_r2
rm -rf $OUT; mkdir -p $OUT
B32="python3 bench.py --no-secondary --no-cpu-baseline"
B16="python3 bench.py --precision bf16 --no-secondary --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_f32 -- $B32 --steps 2 --warmup 1 > $OUT/bench_trace_f32.json 2> $OUT/trace_f32.err
echo "[profile] fp32 trace done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_bf16 -- $B16 --steps 4 --warmup 1 > $OUT/bench_trace_bf16.json 2> $OUT/trace_bf16.err
echo "[profile] bf16 trace done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_train -- python3 tools/train_bench.py --steps 5 --warmup 2 > $OUT/bench_trace_train.json 2> $OUT/trace_train.err
echo "[profile] training trace done"
for P in f32 bf16; do
  if [ $P = f32 ]; then CMD="$B32"; MOPS=SQ_INSTS_VALU_MFMA_MOPS_F32; else CMD="$B16"; MOPS=SQ_INSTS_VALU_MFMA_MOPS_BF16; fi
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_${P}_fetch -- $CMD --steps 1 --warmup 0 > $OUT/b_${P}_fetch.json 2> $OUT/${P}_fetch.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_${P}_write -- $CMD --steps 1 --warmup 0 > $OUT/b_${P}_write.json 2> $OUT/${P}_write.err
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY $MOPS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_${P}_sq -- $CMD --steps 1 --warmup 0 > $OUT/b_${P}_sq.json 2> $OUT/${P}_sq.err || echo "sq pass failed" >> $OUT/${P}_sq.err
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_${P}_sq2 -- $CMD --steps 1 --warmup 0 > $OUT/b_${P}_sq2.json 2> $OUT/${P}_sq2.err || echo "sq2 pass failed" >> $OUT/${P}_sq2.err
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_${P}_l2 -- $CMD --steps 1 --warmup 0 > $OUT/b_${P}_l2.json 2> $OUT/${P}_l2.err || echo "l2 pass failed" >> $OUT/${P}_l2.err
  echo "[profile] $P counters done"
done
find $OUT -name "*.csv" | wc -l
python3 tools/summarize_profiles_r2.py r2
