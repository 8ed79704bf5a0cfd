#!/bin/bash
# round 4: staging-rate probe of the sliced LDS walk (tools/micro/stage_rate.hip); run on the GPU box from the repo root
out=${1:-gpurun_out/stage_rate.txt}
n=${2:-500000}
mkdir -p "$(dirname "$out")"
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
B=tools/micro/stage_rate
for cfg in "8 16 192" "8 8 192" "8 32 192"; do
  set -- $cfg
  d=/tmp/wt_${n}_$1_$2_$3
  python tools/micro/walk_tiles.py $d $n $1 $2 $3 2>&1 | grep -E "order|graph" >> "$out"
  $B $d 200 $1 $3 256 >> "$out" 2>&1
done
d=/tmp/wt_${n}_8_16_192
echo "# counters: block walks its chunks (xcd_chunk 4) / groups of 32 blocks, chunk-major" >> "$out"
for only in 0 32; do
  rm -rf /tmp/pmc_sr
  rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --kernel-trace -d /tmp/pmc_sr -o sr -- $B $d 200 8 192 256 $only > /dev/null 2>&1
  python - "$only" >> "$out" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob('/tmp/pmc_sr/**/*counter_collection.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
agg = {}
for r in rows:
    if 'k_stage' not in r.get('Kernel_Name', ''):
        continue
    agg.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
print('only=%s' % sys.argv[1], {k: '%.3g per launch (%d launches)' % (sum(v) / len(v), len(v)) for k, v in agg.items()})
PY
  rm -rf /tmp/pmc_sr
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d /tmp/pmc_sr -o sr -- $B $d 200 8 192 256 $only > /dev/null 2>&1
  python - "$only" >> "$out" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob('/tmp/pmc_sr/**/*counter_collection.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
agg = {}
for r in rows:
    if 'k_stage' not in r.get('Kernel_Name', ''):
        continue
    agg.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
print('only=%s' % sys.argv[1], {k: '%.3g per launch (%d launches)' % (sum(v) / len(v), len(v)) for k, v in agg.items()})
PY
done
cat "$out"
