#!/usr/bin/env python3
"""np.linalg.svd of an N x N Gram matrix on the box's host cores under 1 / 2 / 4 / 8 BLAS threads
(threadpoolctl), ms per call (min and median of 30): svd_threads.py [N ...]"""
import sys, time, numpy as np
from threadpoolctl import threadpool_limits
for N in [int(a) for a in sys.argv[1:]] or [50, 100, 200]:
    rng = np.random.RandomState(0)
    X = rng.randn(20000, N); X -= X.mean(0); X /= X.std(0)
    G = X.T @ X
    ref = None
    for th in (1, 2, 4, 8):
        with threadpool_limits(limits=th, user_api='blas'):
            ts = []
            for _ in range(30):
                t = time.perf_counter(); U, s, _ = np.linalg.svd(G); ts.append(time.perf_counter() - t)
        if ref is None: ref = (U, s)
        print('N %4d  threads %d  svd min %.2f ms  median %.2f ms   max|dU| %.1e  signs equal %s' % (
            N, th, min(ts) * 1e3, np.median(ts) * 1e3, np.abs(np.abs(U) - np.abs(ref[0])).max(),
            bool(np.all(np.sign(U[0]) == np.sign(ref[0][0])))), flush=True)
