#!/bin/bash
# bench.py's multi-rank modes with real RCCL ranks on one GPU (tools/rccl_ranks_one_gpu.sh): one line per mode with the
# p-value and the communicator's report; the single-GPU p-value of the same workload first.
cd "$(dirname "$0")/.."
out=gpurun_out/rccl_sweep.txt
: > $out
line() { python - "$1" "$2" <<'PY' >> gpurun_out/rccl_sweep.txt
import json, sys
tag, path = sys.argv[1:3]
rows = [l for l in open(path) if l.startswith('{')]
if not rows:
    print('%-44s NO RESULT LINE' % tag); sys.exit()
d = json.loads(rows[-1]); c = d['config']
print('%-44s n=%d ms/step=%8.2f p=%.12g comm=%s parallelism=%s' % (tag, d['n_gpus'], d['ms_per_step'], c['p_value'], json.dumps(c.get('communicator')), c.get('parallelism', '')[:110]))
PY
}
for w in C3 C5; do
  timeout 300 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/sweep_one_$w.log 2>&1
  line "one GPU $w" gpurun_out/sweep_one_$w.log
done
run() { tag=$1; n=$2; shift 2; RANK_TIMEOUT=300 tools/rccl_ranks_one_gpu.sh $n "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2>&1; cp gpurun_out/rccl${n}_r0.log "gpurun_out/sweep_$(echo $tag | tr ' /' '__').log"; line "$tag" gpurun_out/rccl${n}_r0.log; }
run "rccl 4 ranks C3 sharded populations" 4 --workload C3
run "rccl 4 ranks C3 sharded caller order" 4 --workload C3 --partition caller
run "rccl 3 ranks C3 replicated inputs" 3 --workload C3 --inputs replicated
run "rccl 4 ranks C3 weak scaling" 4 --workload C3 --scaling weak
run "rccl 8 ranks C5 sharded populations" 8 --workload C5
CNA_NO_HALO_COMM=1 run "rccl 4 ranks C3 no halo communicator" 4 --workload C3
CNA_HALO=0 run "rccl 4 ranks C3 all-gather of the state" 4 --workload C3
cat $out
