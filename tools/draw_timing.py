#!/usr/bin/env python3
"""Host cost of the permutation draw (numpy legacy RNG + argsort), pieces timed on this machine."""
import numpy as np, time
def t(f, n=300):
    f(); a = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - a) / n * 1e3
np.random.seed(0)
for m in (50, 100, 200):
    r = np.random.randn(m, 1000)
    idx = np.argsort(r, axis=0)
    Y = np.random.randn(m)
    print('m=%d  randn %.3f ms  argsort(axis0) %.3f ms  gather %.3f ms' % (
        m, t(lambda: np.random.randn(m, 1000)), t(lambda: np.argsort(r, axis=0)), t(lambda: Y[idx])))
