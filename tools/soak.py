#!/usr/bin/env python3
"""Soak: the same analysis over and over on one engine (two phenotypes in turn, NAM cache off), every result compared bit
for bit with the first of its kind -- global p, chosen k, both data.obs columns.  usage: soak.py cells samples calls"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter('ignore')
import numpy as np, pandas as pd
import cna_amd as cna
from cna_amd import synth
from cna_amd.engine import get_engine
n, N, calls = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
data, meta = synth.make_dataset(n, N, k=30, seed=0)
eng = get_engine(); eng.reuse_nam = False
ys = [meta['y'], pd.Series(np.random.RandomState(3).randn(N), index=meta['y'].index)]
kw = dict(nsteps=3, Nnull=1000, seed=0)
first = {}
t0 = time.time()
for it in range(calls):
    j = it & 1
    res = cna.tl.association(data, ys[j], 'id', return_full=(it % 97 == 0), **kw)
    p = res if not hasattr(res, 'p') else res.p
    got = (p, data.obs['coef'].values.copy(), data.obs['coef_fdr'].values.copy())
    if j not in first:
        first[j] = got
        continue
    assert got[0] == first[j][0], (it, got[0], first[j][0])
    assert np.array_equal(got[1], first[j][1]) and np.array_equal(got[2], first[j][2], equal_nan=True), it
print('%d cells x %d samples: %d calls in %.1f s, every result bit-identical to the first of its phenotype (p = %r, %r); device bytes %.2f GB'
      % (n, N, calls, time.time() - t0, first[0][0], first[1][0], eng.device_bytes() / 1e9))
