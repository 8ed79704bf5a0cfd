#!/usr/bin/env python3
import os, sys, time, warnings, cProfile, pstats, io
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
pr = cProfile.Profile(); pr.enable()
for _ in range(3): cna.tl.association(data, meta['y'], 'id', **kw)
pr.disable()
buf = io.StringIO(); pstats.Stats(pr, stream=buf).sort_stats('cumtime').print_stats(28); print(buf.getvalue()[:4500])
