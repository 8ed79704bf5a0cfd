#!/bin/bash
# Everything profiles/ needs for round 5, on the GPU box:  bash tools/profile_round5.sh
# (rocprofv3 --kernel-trace --stats of the bench command; PMC passes -- counters only, each in its own
# run -- for HBM-side traffic (FETCH_SIZE / WRITE_SIZE) and the SQ / TCC view of the dominant kernels)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_r05; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SHORT="--no-cpu-baseline --no-extra --steps 6 --warmup 2"
for W in ${WORKLOADS:-C4 C3 C2 C5}; do
  timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats_$W -- python $R/bench.py --workload $W $SHORT > $OUT/stats_$W.log 2>&1
  pass() { n=$1; shift; timeout -k 5 400 rocprofv3 --kernel-trace --output-format csv -d $OUT -o ${n}_$W --pmc "$@" -- python $R/bench.py --workload $W $SHORT > $OUT/${n}_$W.log 2>&1 || echo "pass $n $W failed"; }
  pass fetch FETCH_SIZE
  pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
  if [ $W != C2 ] && [ $W != C5 ]; then
    pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
    pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
    pass tcp1 TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum
  fi
done
python $R/tools/pmc_traffic.py $OUT > $OUT/pmc_traffic.log 2>&1
ls $OUT | head -80
