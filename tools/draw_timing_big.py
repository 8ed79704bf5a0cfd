#!/usr/bin/env python3
"""Large permutation draws (200 samples x 10 000 permutations): numpy's pieces against the threaded host helpers."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cna_amd import _ffi
from cna_amd.tools import _stats
lib = _ffi.load()
m, num = 200, 10000
Y = np.random.RandomState(0).randn(m)
def best(f, n=5):
    f(); ts = []
    for _ in range(n):
        a = time.perf_counter(); f(); ts.append(time.perf_counter() - a)
    return min(ts) * 1e3
R = np.random.randn(m, num)
print('numpy: randn %.1f ms  argsort(axis=0) %.1f ms  gather %.1f ms' % (best(lambda: np.random.randn(m, num)), best(lambda: np.argsort(R, axis=0)), best(lambda: Y[np.argsort(R, axis=0)]) - best(lambda: np.argsort(R, axis=0))))
out = np.zeros((m, num))
for nt in (1, 2, 4, 8, 16):
    lib.cna_host_set_threads(nt); _stats._threads_set = True
    def rn():
        np.random.seed(0); return _stats.legacy_randn(m, num, True)
    t1 = best(rn)
    t2 = best(lambda: lib.cna_host_argsort_gather(_ffi.ptr(R), m, num, _ffi.ptr(Y), _ffi.ptr(out), num, None))
    def whole():
        np.random.seed(0); return _stats.conditional_permutation(np.ones(m), Y, num, clean=True)
    print('threads %2d: C randn %.1f ms  C argsort+gather %.1f ms  conditional_permutation %.1f ms' % (nt, t1, t2, best(whole)))
assert np.array_equal(out, Y[np.argsort(R, axis=0)])
