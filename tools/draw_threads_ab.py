import sys, time
sys.path.insert(0, '.')
import numpy as np
from cna_amd.tools import _stats
for N in (50, 100, 200):
    Y = np.random.RandomState(0).randn(N); B = np.ones(N)
    ref = None
    for th in (1, 2, 4, 8):
        ts = []
        for rep in range(30):
            t0 = time.perf_counter()
            d = _stats.native_draw_start(B, Y, 1000, 0, threads=th)
            tab = d.wait()
            ts.append(time.perf_counter() - t0)
        if ref is None: ref = tab.copy()
        assert np.array_equal(ref, tab)
        print(N, 'threads', th, 'min %.3f ms  median %.3f ms' % (min(ts) * 1e3, sorted(ts)[15] * 1e3))
