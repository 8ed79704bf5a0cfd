#!/usr/bin/env python3
"""Time the residualisation kernel ((X - mean).M^T on f64 MFMA) alone: kbench_xb.py n_cells n_samples"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cna_amd.engine import get_engine
n, N = int(sys.argv[1]), int(sys.argv[2])
rs = np.random.RandomState(0)
X = rs.randn(n, N)
C = rs.randn(N, 5)
M = np.eye(N) - C.dot(np.linalg.solve(C.T.dot(C), C.T))
eng = get_engine(); eng.upload_x(X)
eng.resid_apply(M, center=True)
eng.prof_reset(); eng.prof_enable(True)
for _ in range(5): eng.resid_apply(M, center=True)
eng.sync(); eng.prof_enable(False)
ms, cnt = eng.prof()['resid_xb']
out = eng.fetch_matrix(1)
print('resid_xb %.1f us (%.1f TFLOP/s)  checksum %.12e' % (ms / cnt * 1e3, 2.0 * n * N * N / (ms / cnt * 1e-3) / 1e12, np.abs(out).sum()))
