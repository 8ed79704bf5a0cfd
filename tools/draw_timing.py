#!/usr/bin/env python3
"""Host cost of the permutation draw, pieces timed on this machine: numpy's legacy randn, the
library's restatement of the same stream, argsort, and the whole _draw_null."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cna_amd.tools import _stats, _association
def t(f, n=300):
    f(); a = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - a) / n * 1e3
np.random.seed(0)
for m in (50, 100, 200):
    r = np.random.randn(m, 1000)
    idx = np.argsort(r, axis=0)
    Y = np.random.randn(m)
    B = np.ones(m)
    def fresh():
        np.random.seed(0); return _stats.legacy_randn(m, 1000, clean=True)
    print('m=%d  numpy randn %.3f  C stream (get/set state) %.3f  C stream in place %.3f  argsort(axis0) %.3f  gather %.3f  '
          '_draw_null %.3f ms' % (m, t(lambda: np.random.randn(m, 1000)), t(lambda: _stats.legacy_randn(m, 1000)), t(fresh),
                                 t(lambda: np.argsort(r, axis=0)), t(lambda: Y[idx]),
                                 t(lambda: _association._draw_null(Y, B, None, Nnull=1000, seed=0))))
