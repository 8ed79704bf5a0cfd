#!/usr/bin/env python3
"""Step time and kernel times of less common call paths at scale: kbench_paths.py n_cells n_samples"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter('ignore')
import numpy as np, pandas as pd
import cna_amd as cna
from cna_amd import synth
from cna_amd.engine import get_engine
n, N = int(sys.argv[1]), int(sys.argv[2])
eng = get_engine(); eng.reuse_nam = False
def run(tag, data, y, **kw):
    for _ in range(2): cna.tl.association(data, y, 'id', **kw)
    eng.prof_reset(); eng.prof_enable(True); eng.sync()
    t0 = time.perf_counter()
    for _ in range(3): r = cna.tl.association(data, y, 'id', **kw)
    eng.sync(); dt = (time.perf_counter() - t0) / 3
    eng.prof_enable(False)
    k = {a: (round(v[0] / v[1] * 1e3), v[1] // 3) for a, v in eng.prof().items() if v[0] / v[1] > 0.2}
    print('%-28s %8.2f ms/step  gpu %.1f ms  %s' % (tag, dt * 1e3, sum(v[0] for v in eng.prof().values()) / 3, k))
data, meta = synth.make_dataset(n, N, k=30, seed=0)
y = meta['y']
run('plain nsteps=3', data, y, nsteps=3, Nnull=1000, seed=0)
run('auto-stop (nsteps=None)', data, y, Nnull=1000, seed=0)
run('return_full', data, y, nsteps=3, Nnull=1000, seed=0, return_full=True)
run('Nnull=200', data, y, nsteps=3, Nnull=200, seed=0)
run('ks=[1,3,5,9]', data, y, nsteps=3, Nnull=1000, seed=0, ks=[1, 3, 5, 9])
donor = pd.Series(np.arange(N) // 2, index=y.index)
yd = pd.Series(y.groupby(donor).transform('first').values, index=y.index)
run('donorids', data, yd, nsteps=3, Nnull=1000, seed=0, donorids=donor)
data64, meta64 = synth.make_dataset(n, N, k=30, seed=0, graph_dtype=np.float64)
run('float64 graph', data64, meta64['y'], nsteps=3, Nnull=1000, seed=0)
