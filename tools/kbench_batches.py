#!/usr/bin/env python3
"""Kernel times of the covariate / batch paths at scale: kbench_batches.py n_cells n_samples n_covs n_batches"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter('ignore')
import numpy as np
import cna_amd as cna
from cna_amd import synth
from cna_amd.engine import get_engine
n, N, nc, nb = (int(v) for v in sys.argv[1:5])
data, meta = synth.make_dataset(n, N, k=30, seed=0, n_covs=nc, n_batches=nb)
eng = get_engine(); eng.reuse_nam = False
kw = dict(covs=meta['covs'], batches=meta['batches'], nsteps=3, Nnull=1000, seed=0)
for _ in range(2): cna.tl.association(data, meta['y'], 'id', **kw)
eng.prof_reset(); eng.prof_enable(True); eng.sync()
t0 = time.perf_counter()
for _ in range(3): cna.tl.association(data, meta['y'], 'id', **kw)
eng.sync(); dt = (time.perf_counter() - t0) / 3
eng.prof_enable(False)
print('%.2f ms/step' % (dt * 1e3), {k: (round(v[0] / v[1] * 1e3), v[1] // 3) for k, v in eng.prof().items() if v[0] / v[1] > 0.05})
