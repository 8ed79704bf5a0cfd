#!/bin/bash
# Everything profiles/ needs for round 6, on the GPU box:  bash tools/profile_round6.sh
# (rocprofv3 --kernel-trace --stats of the bench command; PMC passes -- counters only, each in its own
# run -- for HBM-side traffic (FETCH_SIZE / WRITE_SIZE) and the SQ / TCC view of the dominant kernels)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_r06; mkdir -p $OUT
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
PMC_TRAFFIC_NAME=r06_pmc_traffic.json python $R/tools/pmc_traffic.py $OUT > $OUT/pmc_traffic.log 2>&1
# per-kernel statistics of the timed region only (the last 6 analyses), and the summaries that go to profiles/
mkdir -p $R/gpurun_out/r06_profiles
for W in ${WORKLOADS:-C4 C3 C2 C5}; do
  t=$(ls $OUT/stats_${W}_kernel_trace.csv $OUT/*/stats_${W}_kernel_trace.csv 2>/dev/null | head -1)
  [ -n "$t" ] && python $R/tools/kernel_stats_timed.py $t $R/gpurun_out/r06_profiles/r06_kernel_stats_$W.csv 6
  [ -f $OUT/pmc_summary_$W.txt ] && cp $OUT/pmc_summary_$W.txt $R/gpurun_out/r06_profiles/r06_pmc_summary_$W.txt
  tail -3 $OUT/stats_$W.log > $R/gpurun_out/r06_profiles/r06_stats_bench_line_$W.txt 2>/dev/null
done
cp $OUT/r06_pmc_traffic.json $OUT/pmc_traffic.log $R/gpurun_out/r06_profiles/ 2>/dev/null
rm -rf $OUT          # (the raw traces: hundreds of MB)
ls -la $R/gpurun_out/r06_profiles
