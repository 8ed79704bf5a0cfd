#!/usr/bin/env python3
"""Kernel micro-benchmark on the GPU box: time the diffusion kernels alone (HIP events)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cna_amd import synth
from cna_amd.engine import get_engine
from cna_amd.tools._nam import sample_codes
n, N = int(sys.argv[1]), int(sys.argv[2])
data, meta = synth.make_dataset(n, N, k=30, seed=0)
eng = get_engine()
A = data.obsp['connectivities']
eng.ensure_graph(A); eng.colsums(1)
codes, labels = sample_codes(data.obs['id'])
C = np.bincount(codes, minlength=N).astype(float)
for rep in range(3):
    eng.set_samples(codes, N, C)
    if rep == 1:
        eng.prof_reset(); eng.prof_enable(True)
    eng.nam_step(False, True, False); eng.nam_step(False, True, False); eng.nam_step(False, False, True)
eng.sync(); eng.prof_enable(False)
print(os.environ.get('CNA_STEP_VARIANT', '0'), {k: round(v[0] / v[1] * 1e3, 1) for k, v in eng.prof().items()})
