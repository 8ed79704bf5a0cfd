#!/usr/bin/env python3
"""How close is the device walk to scipy's csr_matvecs, bit for bit?  (oracle mode 'f64')"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from cna_amd import synth
from cna_amd.engine import get_engine
from cna_amd.tools._nam import sample_codes
from oracle import cna_oracle as orc
for n, N, dt in ((20000, 50, np.float32), (20000, 130, np.float32), (8000, 40, np.float64)):
    data, meta = synth.make_dataset(n, N, k=15, seed=1, graph_dtype=dt)
    A = sp.csr_matrix(data.obsp['connectivities'])
    codes, labels = sample_codes(data.obs['id'])
    C = np.bincount(codes, minlength=N).astype(float)
    eng = get_engine()
    eng.ensure_graph(data.obsp['connectivities']); eng.colsums(1); eng.set_samples(codes, N, C)
    S = np.zeros((n, N), dtype=bool); S[np.arange(n), codes] = True
    s = S; cs = orc.column_sums(A, 1, 'f64')
    for i in range(3):
        s = orc.diffusion_step(A, s, cs, 1, first_onehot=(i == 0), mode='f64')
        eng.nam_step(False, i < 2, True)
        got = eng.nam_full(); want = s / C
        d = got != want
        print('n=%d N=%d %s step %d: %d of %d entries differ, max rel %.2e' % (n, N, np.dtype(dt).name, i + 1, d.sum(), d.size, np.abs(got - want).max() / np.abs(want).max()))
