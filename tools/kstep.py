#!/usr/bin/env python3
"""Diffusion kernels only (for rocprofv3 --pmc passes): kstep.py n_cells n_samples [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cna_amd import synth
from cna_amd.engine import get_engine
from cna_amd.tools._nam import sample_codes
n, N = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
data, meta = synth.make_dataset(n, N, k=30, seed=0)
codes, labels = sample_codes(data.obs['id'])
C = np.bincount(codes, minlength=N).astype(float)
eng = get_engine()
eng.ensure_graph(data.obsp['connectivities'].tocsr()); eng.colsums(1)
for rep in range(reps):
    eng.set_samples(codes, N, C)
    if rep == 1: eng.prof_reset(); eng.prof_enable(True)
    eng.nam_step(False, True, False); eng.nam_step(False, True, False); eng.nam_step(False, False, True)
eng.sync(); eng.prof_enable(False)
print({k: round(v[0] / v[1] * 1e3, 1) for k, v in eng.prof().items()})
