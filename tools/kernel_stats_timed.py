#!/usr/bin/env python3
"""Per-kernel statistics of the TIMED region of a rocprofv3 --kernel-trace of bench.py (the last STEPS analyses): what
rocprofv3 --stats reports, without the first call (a large graph is analysed in the caller's cell order first), the
call that adopts the device order and the warm-up.  usage: kernel_stats_timed.py <kernel_trace.csv> <out.csv> [steps]"""
import sys
import pandas as pd
src, dst = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
df = pd.read_csv(src)
firsts = df.loc[df['Kernel_Name'].str.contains('k_nam_first'), 'Start_Timestamp'].sort_values()
if len(firsts) > steps:
    df = df[df['Start_Timestamp'] >= firsts.iloc[-steps]]
df['ns'] = df['End_Timestamp'] - df['Start_Timestamp']
g = df.groupby('Kernel_Name')['ns'].agg(Calls='count', TotalDurationNs='sum', AverageNs='mean', MinNs='min', MaxNs='max', StdDev='std')
g['Percentage'] = 100.0 * g['TotalDurationNs'] / g['TotalDurationNs'].sum()
g = g.sort_values('TotalDurationNs', ascending=False).reset_index().rename(columns={'Kernel_Name': 'Name'})
g[['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs', 'StdDev']].to_csv(dst, index=False)
