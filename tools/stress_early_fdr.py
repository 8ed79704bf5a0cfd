#!/usr/bin/env python3
"""Repeat association() on one dataset of the large-input schedule and compare both data.obs columns with the
first call's, bit for bit: the FDR column is copied by the helper thread while the main thread is in the SVD
(cna_percell_fdr_copy_early), so a race would show up here.  stress_early_fdr.py [n_cells] [n_samples] [repeats]"""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter('ignore')
import numpy as np
import cna_amd as cna
from cna_amd import synth
from cna_amd.engine import get_engine
from cna_amd.tools import _association as A
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
data, meta = synth.make_dataset(n, N, k=30, seed=0)
eng = get_engine(); eng.reuse_nam = False
A._EARLY_FDR = False
p0 = cna.tl.association(data, meta['y'], 'id', nsteps=3, Nnull=1000, seed=0)
c0, f0 = data.obs['coef'].values.copy(), data.obs['coef_fdr'].values.copy()
A._EARLY_FDR = True
served = 0
real = eng.percell_fdr_copied_early
def counted():
    global served
    r = real(); served += bool(r); return r
eng.percell_fdr_copied_early = counted
bad = 0
for i in range(reps):
    if i % 3 == 0:
        del data.obs['coef_fdr']                 # fresh storage for the column now and then
    p = cna.tl.association(data, meta['y'], 'id', nsteps=3, Nnull=1000, seed=0)
    ok = p == p0 and np.array_equal(data.obs['coef'].values, c0) and np.array_equal(data.obs['coef_fdr'].values, f0)
    bad += not ok
print('repeats %d  early copies served %d  mismatches %d  (fdr < 1 in %d cells)' % (reps, served, bad, int((f0 < 1).sum())))
sys.exit(1 if bad or served != reps else 0)
