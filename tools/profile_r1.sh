#!/bin/bash
# rocprofv3 passes for the round-1 bench (run on the GPU box from the repo root): kernel trace + separate PMC passes.
# bench.py's default run measures the fp32 headline and the bf16 leg, so one pass profiles both field kernels.
set -e
export TMPDIR=/tmp
OUT=gpurun_out/prof_r1
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_write.json 2> $OUT/write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_sq.json 2> $OUT/sq.err || echo "sq pass failed" >> $OUT/sq.err
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_sq2.json 2> $OUT/sq2.err || echo "sq2 pass failed" >> $OUT/sq2.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_l2.json 2> $OUT/l2.err || echo "l2 pass failed" >> $OUT/l2.err
find $OUT -name "*.csv" | wc -l
