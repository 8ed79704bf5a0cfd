#!/bin/bash
# What profiles/ holds from the last day of round 6, on the GPU box:  bash tools/evidence_round6.sh
# (the default bench line + details, a 2-rank and a 4-rank real-RCCL plumbing line on ONE GPU, host timelines of the two-call
# path and of the general path, the GPU-side timeline of one 200 000 x 50 step)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_evidence; mkdir -p $O; cd $R
python bench.py > $O/r06_bench_default.json 2> $O/bench_default.err; cp bench_details.json $O/r06_bench_details.json
RANK_TIMEOUT=900 tools/rccl_ranks_one_gpu.sh 2 --steps 3 --warmup 2 --no-cpu-baseline > $O/rccl2.txt 2>&1
tail -1 gpurun_out/rccl2_r0.log > $O/r06_bench_rccl2_one_gpu_plumbing_C4.json; cp bench_details.json $O/r06_bench_rccl2_one_gpu_plumbing_C4_details.json
RANK_TIMEOUT=900 tools/rccl_ranks_one_gpu.sh 4 --steps 2 --warmup 1 --no-cpu-baseline > $O/rccl4.txt 2>&1
tail -1 gpurun_out/rccl4_r0.log > $O/r06_bench_rccl4_one_gpu_plumbing_C4.json
for s in "200000 50 C2" "250000 200 block8" "1000000 100 C3" "2000000 200 C4"; do set -- $s
  { echo "# tools/host_trace.py $1 $2 (graph pinned): the two-call path"; TRACE_PIN=1 python tools/host_trace.py $1 $2;
    echo; echo "# the same, general path (CNA_ONE_CALL=0)"; CNA_ONE_CALL=0 TRACE_PIN=1 python tools/host_trace.py $1 $2 | grep -v "^ *[0-9.]* *+" ;
    echo; echo "# the same, graph NOT pinned (content hashed inside cna_assoc_finish)"; python tools/host_trace.py $1 $2 | grep "^step\|stages"; } > $O/r06_host_trace_$3.txt 2>&1
done
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- env TRACE_PIN=1 python tools/host_trace.py 200000 50 > /dev/null 2>&1
{ echo "# GPU-side timeline of the last 200 000 x 50 step of tools/host_trace.py under rocprofv3 --kernel-trace (tools/kernel_timeline.py)"; python tools/kernel_timeline.py $O/kt; } > $O/r06_timeline_C2.txt 2>&1
rm -rf $O/kt
ls -la $O
