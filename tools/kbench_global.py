#!/usr/bin/env python3
"""Sample-space kernels alone (nothing else on the GPU): condition + global F-tests.
    kbench_global.py N P [r]   -> us per launch"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cna_amd.engine import get_engine
N, P = int(sys.argv[1]), int(sys.argv[2])
r = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rs = np.random.RandomState(0)
eng = get_engine()
eng.upload_x(rs.randn(4 * N, N))
Q, _ = np.linalg.qr(rs.randn(N, N))
C = rs.randn(N, max(r, 1))
M = np.eye(N) - C.dot(np.linalg.solve(C.T.dot(C), C.T)) if r else np.eye(N)
Y = rs.randn(N, P)
incr = max(int(0.02 * N), 1)
ks = np.arange(incr, max(min(4 * incr, int(N / 5)), 1) + 1, incr)
for rep in range(4):
    if rep == 1:
        eng.prof_reset(); eng.prof_enable(True)
    eng.condition(M, Y)
    eng.global_test(Q, ks, r)
eng.sync(); eng.prof_enable(False)
print('N=%d P=%d r=%d ks=%s' % (N, P, r, list(ks)), {k: round(v[0] / v[1] * 1e3, 1) for k, v in eng.prof().items()})
