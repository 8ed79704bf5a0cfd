#!/usr/bin/env python3
"""GPU-side span of one analysis from a rocprofv3 --kernel-trace of bench.py: first walk kernel's start to the FDR
table kernel's end, mean over the timed steps, and the idle time inside it.  usage: gpu_span.py <kernel_trace.csv> [steps]"""
import sys
import pandas as pd
df = pd.read_csv(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
firsts = df.loc[df['Kernel_Name'].str.contains('k_nam_first'), 'Start_Timestamp'].sort_values().tolist()
ends = df.loc[df['Kernel_Name'].str.contains('k_fdr_table'), 'End_Timestamp'].sort_values().tolist()
spans, busy = [], []
for t0 in firsts[-steps:]:
    t1 = min(e for e in ends if e > t0)
    d = df[(df['Start_Timestamp'] >= t0) & (df['End_Timestamp'] <= t1) & (df['Stream_Id'] == df.loc[df['Start_Timestamp'] == t0, 'Stream_Id'].iloc[0])]
    spans.append((t1 - t0) / 1e3)
    busy.append(((d['End_Timestamp'] - d['Start_Timestamp']).sum()) / 1e3)
print('GPU span first walk kernel -> FDR table: mean %.1f us (min %.1f, max %.1f); main-stream kernels busy %.1f us' % (
    sum(spans) / len(spans), min(spans), max(spans), sum(busy) / len(busy)))
