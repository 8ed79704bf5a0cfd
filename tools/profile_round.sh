#!/bin/bash
# Everything profiles/ needs for one round, on the GPU box:  bash tools/profile_round.sh r01
set -u
R=$GRAFT_REPO_ROOT; TAG=${1:-r01}; OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/bench_C2.json 2> $OUT/bench_C2.err
python $R/bench.py --workload C3 --no-cpu-baseline --steps 20 --warmup 5 > $OUT/bench_C3.json 2> $OUT/bench_C3.err
python $R/bench.py --workload C4 --no-cpu-baseline --steps 3 --warmup 2 > $OUT/bench_C4.json 2> $OUT/bench_C4.err
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- python $R/bench.py --no-cpu-baseline > $OUT/stats.log 2>&1
pass() { n=$1; shift; timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -o $n --pmc "$@" -- python $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $OUT/$n.log 2>&1 || echo "pass $n failed"; }
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE
pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
pass tcp1 TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum
pass tcp2 TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
python $R/tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
ls $OUT | head -50
