#!/bin/bash
# counters of the staging-rate probe (tools/micro/stage_rate.hip): bytes from behind the L2 and L2 hit rate of the two
# schedules at 500k x 200; every profiler pass under its own time limit.  Run on the GPU box from the repo root.
out=${1:-gpurun_out/stage_rate_pmc.txt}
mkdir -p "$(dirname "$out")"
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
B=$PWD/tools/micro/stage_rate
d=/tmp/wt_pmc
python tools/micro/walk_tiles.py $d 500000 8 16 192 2>&1 | grep -E "order" >> "$out"
for only in 0 32; do
  for ctr in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum"; do
    rm -rf /tmp/pmc_sr
    timeout -k 5 90 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_sr -o sr -- $B $d 200 8 192 256 $only > /tmp/pmc_sr.log 2>&1
    echo "only=$only $ctr rc=$?" >> "$out"
    python - >> "$out" <<'PY'
import csv, glob
agg = {}
for f in glob.glob('/tmp/pmc_sr/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_stage' in r.get('Kernel_Name', ''):
            agg.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
print('   ', {k: '%.4g per launch (%d launches)' % (sum(v) / len(v), len(v)) for k, v in agg.items()})
PY
  done
done
cat "$out"
