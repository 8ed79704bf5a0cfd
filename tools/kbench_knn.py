#!/usr/bin/env python3
"""Input generator timing: device kNN-graph builder vs the host builder.  kbench_knn.py n [d] [k]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cna_amd import synth
from cna_amd.engine import get_engine
n = int(sys.argv[1]); d = int(sys.argv[2]) if len(sys.argv) > 2 else 8; k = int(sys.argv[3]) if len(sys.argv) > 3 else 30
X, _ = synth.mixture_points(n, dim=d)
eng = get_engine()
eng.knn_graph(X[:5000], k)
t = time.time(); A = eng.knn_graph(X, k); tg = time.time() - t
print('n=%d d=%d k=%d: device builder %.2f s, nnz/row %.2f' % (n, d, k, tg, A.nnz / n), flush=True)
if n <= 1_000_000:
    t = time.time(); B = synth.fuzzy_knn_graph(X, k=k, builder='cpu'); tc = time.time() - t
    print('   host builder %.2f s, nnz/row %.2f, edges differing %d' % (tc, B.nnz / n, abs((A != 0).astype(np.int8) - (B != 0).astype(np.int8)).sum()))
