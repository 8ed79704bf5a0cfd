#!/usr/bin/env python3
"""Time the Gram kernels alone: kbench_gram.py n_cells n_samples"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cna_amd.engine import get_engine
n, N = int(sys.argv[1]), int(sys.argv[2])
rs = np.random.RandomState(0)
X = rs.randn(n, N)
eng = get_engine(); eng.upload_x(X); eng.standardize(center=True)
G = eng.gram()
eng.prof_reset(); eng.prof_enable(True)
for _ in range(5): G = eng.gram()
eng.prof_enable(False)
p = eng.prof()
ms = p['gram'][0] / p['gram'][1]
print(os.environ.get('CNA_GRAM_NW16', '0'), 'gram %.1f us (%.1f TFLOP/s)  reduce %.1f us  checksum %.10e' % (ms * 1e3, 2.0 * n * N * N / (ms * 1e-3) / 1e12, p['gram_reduce'][0] / p['gram_reduce'][1] * 1e3, np.abs(G).sum()))
