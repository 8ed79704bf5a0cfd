#!/usr/bin/env python3
"""Where the first (cold) association() call spends its time: cold_profile.py n_cells n_samples"""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import warnings; warnings.simplefilter('ignore')
import cna_amd as cna
from cna_amd import synth
from cna_amd.engine import get_engine
n, N = int(sys.argv[1]), int(sys.argv[2])
data, meta = synth.make_dataset(n, N, k=30, seed=0)
eng = get_engine(); eng.sync()
pr = cProfile.Profile(); t = time.perf_counter(); pr.enable()
p = cna.tl.association(data, meta['y'], 'id', nsteps=3, Nnull=1000, seed=0)
pr.disable(); print('cold call %.3f s' % (time.perf_counter() - t))
buf = io.StringIO(); pstats.Stats(pr, stream=buf).sort_stats('cumulative').print_stats(28); print(buf.getvalue()[:6000])
