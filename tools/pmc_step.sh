#!/bin/bash
# Memory-path counters of the diffusion kernels: pmc_step.sh <tag> <n_cells> <n_samples>
# (each hardware block has only a few counter slots: small passes, each under its own timeout)
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmcstep_$1
mkdir -p $OUT
CMD="python $R/tools/kstep.py $2 $3"
pass() { n=$1; shift; timeout -k 5 150 rocprofv3 --kernel-trace --output-format csv -d $OUT -o $n --pmc "$@" -- $CMD > $OUT/$n.log 2>&1 || echo "pass $n failed"; }
pass a TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE
pass b TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum
pass c TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass d TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum
pass e TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
pass f TD_TD_BUSY_sum TD_TC_STALL_sum TA_FLAT_READ_WAVEFRONTS_sum
python $R/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
