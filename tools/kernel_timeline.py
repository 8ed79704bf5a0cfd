#!/usr/bin/env python3
"""GPU-side timeline of the LAST analysis in a rocprofv3 --kernel-trace CSV: start / end of every kernel relative to the
first kernel of that analysis, the gap to the previous kernel's end, and the queue it ran on.
    rocprofv3 --kernel-trace --output-format csv -d OUT -- python tools/host_trace.py 200000 50
    python tools/kernel_timeline.py OUT [first-kernel-substring]"""
import csv, glob, os, sys
root = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else 'nam_first'
files = glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?'), r.get('Stream_Id', '?')))
rows.sort()
starts = [i for i, r in enumerate(rows) if first in r[2]]
if not starts:
    sys.exit('no kernel matching %r in %d rows' % (first, len(rows)))
i0 = starts[-1]
t0 = rows[i0][0]
prev_end = t0
busy = 0
for s, e, name, q, st in rows[i0:]:
    print('%9.1f us  +%7.1f gap  %8.1f us  q%s s%s  %s' % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, q, st, name[:90]))
    prev_end = max(prev_end, e)
    busy += e - s
print('span %.1f us, kernel time %.1f us' % ((prev_end - t0) / 1e3, busy / 1e3))
