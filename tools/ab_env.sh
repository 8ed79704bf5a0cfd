#!/bin/bash
# A/B of environment switches on a bench line: tools/ab_env.sh <out file> <workload> "VAR=1 VAR2=2" "VAR=3" ...
out=$1; wl=$2; shift 2
mkdir -p "$(dirname "$out")"
for cfg in "$@"; do
  echo "## $cfg" >> "$out"
  env $cfg python bench.py --workload "$wl" --no-extra --no-cpu-baseline --steps 20 --warmup 5 2>>"$out.err" | python -c '
import json,sys
d=json.loads(sys.stdin.readline())
k=d["kernels"]
print("ms/step %.3f  kernels %.3f  host %.3f | " % (d["ms_per_step"], d["gpu_kernel_ms_per_step"], d["host_ms_per_step"]) + "  ".join("%s %.0f x%d" % (n, v["avg_us"], v["launches"]//d["steps"]) for n,v in k.items() if n in ("nam_first","nam_step_sparse","nam_step","gram","gram_reduce","null_local","global_test","select")) + "  p=%r" % d["config"]["p_value"])' >> "$out"
done
cat "$out"
