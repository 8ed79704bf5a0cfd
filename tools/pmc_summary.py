#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (mean per dispatch)."""
import glob
import sys
import pandas as pd

d = sys.argv[1]
rows = []
for f in sorted(glob.glob(d + '/*counter_collection.csv')):
    df = pd.read_csv(f)
    df['k'] = df['Kernel_Name'].str.replace(r'\(anonymous namespace\)::', '', regex=True).str.replace(r'\(.*', '', regex=True).str.replace('void ', '')
    g = df.groupby(['k', 'Counter_Name'])['Counter_Value'].agg(['mean', 'count']).reset_index()
    rows.append(g)
out = pd.concat(rows).pivot_table(index='k', columns='Counter_Name', values='mean')
pd.set_option('display.width', 250, 'display.max_columns', 50, 'display.float_format', lambda v: '%.4g' % v)
keep = [k for k in out.index if k.startswith('k_')]
print(out.loc[keep].T.to_string())
