#!/usr/bin/env python3
"""Time the local-null kernel alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cna_amd.engine import get_engine
n, N, P = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rs = np.random.RandomState(0)
X = rs.randn(n, N); X -= X.mean(1, keepdims=True); X /= X.std(1, ddof=1)[:, None]
eng = get_engine(); eng.upload_x(X)
y = rs.randn(N); nc, m = eng.ncorrs(y, fetch=True)
Yc = rs.randn(N, P); Yc /= Yc.std(0, ddof=1)
thr = np.arange(m / 4, m, m / 400); edges = thr**2 - 1e-8 - 1e-5 * thr**2
eng.null_local(Yc, edges)
eng.prof_reset(); eng.prof_enable(True)
for _ in range(5): t = eng.null_local(Yc, edges)
eng.prof_enable(False)
ms, cnt = eng.prof()['null_local']
print(os.environ.get('CNA_NULL_DEBUG', '0'), 'null_local %.1f us  (%.1f TFLOP/s) frac counted %.3f' % (ms / cnt * 1e3, 2.0 * n * N * P / (ms / cnt * 1e-3) / 1e12, t[:, 0].mean() / n))
