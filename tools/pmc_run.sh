#!/bin/bash
# PMC collection for the two dominant kernels (run on the GPU box through gpurun).
# Counters are collected in separate passes with --kernel-trace only, as the pool requires.
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_${1:-r01}
mkdir -p $OUT
CMD="python $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 ${2:-}"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT -o sq1 -- $CMD > $OUT/sq1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $OUT -o sq2 -- $CMD > $OUT/sq2.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT -o fetch -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT -o write -- $CMD > $OUT/write.log 2>&1
ls $OUT
