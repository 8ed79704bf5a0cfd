"""Host-side timing of the permutation draw's parts on this box (no GPU): numpy's argsort + gather against
csrc/host_rng.c:cna_host_argsort_gather (sorting network for <= 128 rows), and the legacy normal stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cna_amd import _ffi
from cna_amd.tools import _stats
lib = _ffi.load()
for m, num in ((50, 1000), (100, 1000), (128, 1000), (200, 1000)):
    R = np.random.randn(m, num); y = np.random.randn(m); out = np.empty((m, num))
    lib.cna_host_set_threads(1)
    t = time.perf_counter()
    for _ in range(50): lib.cna_host_argsort_gather(_ffi.ptr(R), m, num, _ffi.ptr(y), _ffi.ptr(out), num, None)
    tc = (time.perf_counter() - t) / 50
    t = time.perf_counter()
    for _ in range(50): o2 = y[np.argsort(R, axis=0)]
    tn = (time.perf_counter() - t) / 50
    t = time.perf_counter()
    for _ in range(50):
        np.random.seed(0); _stats.legacy_randn(m, num, clean=True)
    tr = (time.perf_counter() - t) / 50
    t = time.perf_counter()
    for _ in range(50):
        h = _stats.native_draw_start(np.ones(m), y, num, 0); h.wait()
    td = (time.perf_counter() - t) / 50
    print('%d x %d: argsort+gather C %.3f ms, numpy %.3f ms (%s); legacy randn %.3f ms; whole native draw %.3f ms' % (
        m, num, tc * 1e3, tn * 1e3, np.array_equal(out, o2), tr * 1e3, td * 1e3))
