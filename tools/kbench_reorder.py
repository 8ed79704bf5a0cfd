#!/usr/bin/env python3
"""Experiment: does a locality-preserving cell order (reverse Cuthill-McKee of the kNN graph)
speed up the diffusion kernels?  The library is oblivious: we just feed it the permuted graph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from scipy.sparse.csgraph import reverse_cuthill_mckee
from cna_amd import synth
from cna_amd.engine import get_engine
from cna_amd.tools._nam import sample_codes
n, N = int(sys.argv[1]), int(sys.argv[2])
data, meta = synth.make_dataset(n, N, k=30, seed=0)
A = data.obsp['connectivities'].tocsr()
codes, labels = sample_codes(data.obs['id'])
C = np.bincount(codes, minlength=N).astype(float)
eng = get_engine()
def run(A_, codes_, tag):
    eng.ensure_graph(A_); eng.colsums(1)
    for rep in range(3):
        eng.set_samples(codes_, N, C)
        if rep == 1: eng.prof_reset(); eng.prof_enable(True)
        eng.nam_step(False, True, False); eng.nam_step(False, True, False); eng.nam_step(False, False, True)
    eng.sync(); eng.prof_enable(False)
    print(tag, {k: round(v[0] / v[1] * 1e3, 1) for k, v in eng.prof().items()})
run(A, codes, 'generator order (cluster sorted)')
t = time.time(); perm = reverse_cuthill_mckee(A, symmetric_mode=True); t_rcm = time.time() - t
t = time.time()
inv = np.empty(n, dtype=np.int64); inv[perm] = np.arange(n)
deg = np.diff(A.indptr)[perm]
indptr = np.zeros(n + 1, dtype=np.int64); np.cumsum(deg, out=indptr[1:])
src = np.repeat(A.indptr[:-1][perm].astype(np.int64) - indptr[:-1], deg) + np.arange(indptr[-1])
Ap = sp.csr_matrix((A.data[src], inv[A.indices[src]].astype(np.int32), indptr), shape=A.shape)
t_perm = time.time() - t
print('rcm %.2fs  permute %.2fs' % (t_rcm, t_perm))
run(Ap, codes[perm], 'RCM order')
rs = np.random.RandomState(0); p2 = rs.permutation(n)
inv2 = np.empty(n, dtype=np.int64); inv2[p2] = np.arange(n)
deg = np.diff(A.indptr)[p2]; indptr = np.zeros(n + 1, dtype=np.int64); np.cumsum(deg, out=indptr[1:])
src = np.repeat(A.indptr[:-1][p2].astype(np.int64) - indptr[:-1], deg) + np.arange(indptr[-1])
Ar = sp.csr_matrix((A.data[src], inv2[A.indices[src]].astype(np.int32), indptr), shape=A.shape)
run(Ar, codes[p2], 'random order')
