#!/usr/bin/env python3
"""From the rocprofv3 --pmc CSVs of tools/profile_round2.sh: per-kernel counter means (pmc_summary_<W>.txt) and
profiles-style HBM-side traffic per bench kernel launch (r03_pmc_traffic.json):
    bytes = 2 * FETCH_SIZE [KB] * 1e3 + WRITE_SIZE [KB] * 1e3   (MI355X_MICROARCH.md: FETCH_SIZE reads half)
summed over the device kernels behind one bench kernel name, per launch of that name."""
import glob
import json
import os
import re
import sys

import pandas as pd

d = sys.argv[1]
STEPS = int(os.environ.get('PMC_STEPS', '6'))      # --steps of the profiled bench command (tools/profile_round4.sh)
# device kernel -> bench kernel name (cna_prof_*), and how many bench launches one step makes
NAMES = [('k_nam_first', 'nam_first'), ('k_nam_step_sparse', 'nam_step_sparse'), ('k_nam_step', 'nam_step'), ('k_null', 'null_local'), ('k_hist_reduce', 'null_local'), ('k_quant_', 'null_local'), ('k_y_', 'null_local'),
         ('k_i8_', 'null_local'),
         ('k_gram_reduce', 'gram_reduce'), ('k_gram', 'gram'), ('k_select_std', 'select'), ('k_select', 'select'),
         ('k_xb', 'resid_xb'), ('k_standardize', 'standardize'), ('k_ncorrs', 'ncorrs'), ('k_cond', 'condition'),
         ('k_gt_', 'global_test'), ('k_obs_counts', 'obs_counts'), ('k_percell', 'percell_fdr')]
PER_STEP = {'percell_fdr': 2}        # (round 3: the dense and the compressed walk step are separate bench kernels)


def bench_name(k):
    for pre, name in NAMES:
        if k.startswith(pre):
            return name
    return None


out = {}
for W in os.environ.get('PMC_WORKLOADS', 'C4,C3,C2,C5').split(','):
    frames = []
    for f in sorted(glob.glob('%s/*_%s_counter_collection.csv' % (d, W)) + glob.glob('%s/*/*_%s_counter_collection.csv' % (d, W))):
        df = pd.read_csv(f)
        # the timed region only: the last STEPS analyses (the first call of a large graph runs in the caller's cell
        # order, the next one re-uploads the graph -- engine.ensure_graph -- and the warm-up follows)
        firsts = df.loc[df['Kernel_Name'].str.contains('k_nam_first'), 'Dispatch_Id'].drop_duplicates().sort_values()
        if len(firsts) > STEPS:
            df = df[df['Dispatch_Id'] >= firsts.iloc[-STEPS]]
        df['k'] = (df['Kernel_Name'].str.replace(r'\(anonymous namespace\)::', '', regex=True)
                   .str.replace('void ', '').str.replace(r'\(.*', '', regex=True))
        frames.append(df)
    if not frames:
        continue
    df = pd.concat(frames)
    tab = df.groupby(['k', 'Counter_Name'])['Counter_Value'].agg(['mean', 'sum', 'count']).reset_index()
    piv = tab.pivot_table(index='k', columns='Counter_Name', values='mean')
    pd.set_option('display.width', 250, 'display.max_columns', 50, 'display.float_format', lambda v: '%.4g' % v)
    keep = [k for k in piv.index if k.startswith('k_')]
    with open(os.path.join(d, 'pmc_summary_%s.txt' % W), 'w') as fh:
        fh.write('# mean per dispatch, workload %s, rocprofv3 --pmc (one counter group per run) of bench.py\n' % W)
        fh.write(piv.loc[keep].T.to_string() + '\n')
    # traffic per bench launch: steps = dispatches of k_nam_first (one per step)
    sums = tab.pivot_table(index='k', columns='Counter_Name', values='sum')
    cnts = tab.pivot_table(index='k', columns='Counter_Name', values='count')
    if 'FETCH_SIZE' not in sums or 'WRITE_SIZE' not in sums:
        continue
    first = [k for k in sums.index if k.startswith('k_nam_first')]
    steps_f = cnts.loc[first, 'FETCH_SIZE'].sum()
    steps_w = cnts.loc[first, 'WRITE_SIZE'].sum()
    traffic = {}
    for k in sums.index:
        name = bench_name(k)
        if name is None:
            continue
        f = sums.loc[k, 'FETCH_SIZE'] / steps_f if steps_f else 0.0
        w = sums.loc[k, 'WRITE_SIZE'] / steps_w if steps_w else 0.0
        traffic[name] = traffic.get(name, 0.0) + (2.0 * f + w) * 1e3
    out[W] = {name: round(v / PER_STEP.get(name, 1), 0) for name, v in traffic.items()}
with open(os.path.join(d, os.environ.get('PMC_TRAFFIC_NAME', 'r05_pmc_traffic.json')), 'w') as fh:
    json.dump(out, fh, indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
