#!/usr/bin/env python3
"""Phases of the permutation draw on this machine: the legacy normal stream and the argsort + gather, by thread count."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cna_amd import _ffi
lib = _ffi.load()
def t(f, n=40):
    f(); ts = []
    for _ in range(n):
        a = time.perf_counter(); f(); ts.append(time.perf_counter() - a)
    return np.median(ts) * 1e3
for m in (50, 200):
    num = 1000
    y = np.random.RandomState(1).randn(m)
    out = np.empty((m, num)); R = np.empty((m, num))
    for th in (1, 2, 4):
        lib.cna_host_set_threads(th)
        def randn():
            np.random.seed(0)
            st = np.random.get_state()
            key = st[1].copy(); pos = C.c_int(st[2]); hg = C.c_int(0); g = C.c_double(0)
            lib.cna_host_legacy_randn(key.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(pos), C.byref(hg), C.byref(g), m * num, R.ctypes.data_as(C.POINTER(C.c_double)))
        def seedonly():
            np.random.seed(0); st = np.random.get_state(); st[1].copy()
        def sort():
            lib.cna_host_argsort_gather(R.ctypes.data_as(C.POINTER(C.c_double)), m, num, y.ctypes.data_as(C.POINTER(C.c_double)), out.ctypes.data_as(C.POINTER(C.c_double)), num, None)
        randn()
        print('m=%d threads %d: seed+get_state %.3f  + stream %.3f   argsort+gather %.3f ms' % (m, th, t(seedonly), t(randn), t(sort)))
lib.cna_host_set_threads(1)
