#!/bin/bash
# A/B of the Gram matrix under the walk's last step (CNA_GRAM_OVERLAP / CNA_GRAM_CUS / CNA_GRAM_PRIO) on the C4 bench line.
# usage: tools/ab_gram_overlap.sh <out file> [workload]
out=${1:-gpurun_out/ab_gram_overlap.txt}
wl=${2:-C4}
mkdir -p "$(dirname "$out")"
run() {
  echo "## $*" >> "$out"
  env "$@" python bench.py --workload "$wl" --no-extra --no-cpu-baseline --steps 20 --warmup 5 2>>"$out.err" | python -c '
import json,sys
d=json.loads(sys.stdin.readline())
k=d["kernels"]
print("ms/step %.3f  kernels %.3f  host %.3f | " % (d["ms_per_step"], d["gpu_kernel_ms_per_step"], d["host_ms_per_step"]) + "  ".join("%s %.0f x%d" % (n, v["avg_us"], v["launches"]//d["steps"]) for n,v in k.items() if n in ("nam_first","nam_step_sparse","nam_step","gram","gram_reduce","null_local","global_test","select")) + "  p=%r" % d["config"]["p_value"])' >> "$out"
}
run CNA_GRAM_OVERLAP=0
run CNA_GRAM_OVERLAP=4
run CNA_GRAM_OVERLAP=8
run CNA_GRAM_OVERLAP=2
run CNA_GRAM_OVERLAP=4 CNA_GRAM_PRIO=1
run CNA_GRAM_OVERLAP=4 CNA_GRAM_PRIO=-1
run CNA_GRAM_OVERLAP=4 CNA_GRAM_CUS=4
run CNA_GRAM_OVERLAP=4 CNA_GRAM_CUS=6
run CNA_GRAM_OVERLAP=4 CNA_GRAM_CUS=8
run CNA_GRAM_OVERLAP=0
cat "$out"
