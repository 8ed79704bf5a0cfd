#!/bin/bash
# Memory-side counters of the walk probes (tools/micro/walk_tiles): pmc_walk_probe.sh <tag> <n_cells> <N> <NW> <R> <S> <halves>
# Separate rocprofv3 passes (few counter slots per block), per-kernel averages printed by the awk at the end.
set -u
R=$GRAFT_REPO_ROOT
cd $R/tools/micro
D=/tmp/wtp_$1
python walk_tiles.py $D $2 $4 $5 $6 512 > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmcwalk_$1
mkdir -p $OUT
CMD="$R/tools/micro/walk_tiles $D $3 $4 $5 $6 $7"
pass() { n=$1; shift; timeout -k 5 120 rocprofv3 --kernel-trace --output-format csv -d $OUT -o $n --pmc "$@" -- $CMD > $OUT/$n.log 2>&1 || echo "pass $n failed"; }
pass a FETCH_SIZE
pass b WRITE_SIZE
pass c TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass d TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE
pass e TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
python - $OUT <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/*counter_collection.csv'):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0][:70]
        agg[k][row['Counter_Name']].append(float(row['Counter_Value']))
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print('    %-28s avg %.4g over %d dispatches' % (c, sum(v) / len(v), len(v)))
PY
