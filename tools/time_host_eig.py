"""Host-side timing of the leading eigenpairs of a NAM-like Gram matrix (N x N, the 16 leading of 200): the routines
that could stand on the critical path of a small analysis."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.linalg import lapack, eigh
from scipy.sparse.linalg import eigsh
from cna_amd.tools import _nam
rs = np.random.RandomState(0)
ctx = _nam.host_blas_threads(int(os.environ.get('BLAS_THREADS', '1')))
ctx.__enter__()
for N, k in ((50, 4), (100, 8), (200, 16)):
    # spectrum like a NAM's: slow decay (16th / 1st ~ 0.3)
    Q, _ = np.linalg.qr(rs.randn(N, N)); lam = 1.0 / (1.0 + 0.15 * np.arange(N)) ** 1.0
    G = (Q * lam) @ Q.T * 1e6; G = (G + G.T) / 2
    def t(f, reps=30):
        f(); t0 = time.perf_counter()
        for _ in range(reps): f()
        return (time.perf_counter() - t0) / reps * 1e3
    ref = _nam._top_pcs_lapack(G, k)
    assert _nam._top_pcs_native(G, k) is not None
    v0 = np.ones(N) / np.sqrt(N)
    res = {
        'native (csrc/host_eig.c, checked)': t(lambda: _nam._top_pcs_native(G, k)),
        'LAPACKE dsyevr I (ctypes)': t(lambda: _nam._top_pcs_lapack(G, k)),
        'f2py dsyevr I': t(lambda: lapack.dsyevr(G, compute_v=1, range='I', il=N - k + 1, iu=N, lower=1)),
        'f2py dsyevd': t(lambda: lapack.dsyevd(G, compute_v=1, lower=1)),
        'f2py dsyevx I': t(lambda: lapack.dsyevx(G, compute_v=1, range='I', il=N - k + 1, iu=N, lower=1)),
        'np.linalg.svd': t(lambda: np.linalg.svd(G)),
        'eigsh k (ARPACK, tol 0)': t(lambda: eigsh(G, k=k, which='LA', v0=v0, tol=0)),
    }
    w, v = eigsh(G, k=k, which='LA', v0=v0, tol=0)
    Un = _nam._top_pcs_native(G, k)
    print('   native vs dsyevr projector difference %.1e' % np.abs(ref @ ref.T - Un @ Un.T).max())
    P1 = ref @ ref.T; P2 = v @ v.T
    print(N, k, '  '.join('%s %.3f ms' % kv for kv in res.items()), ' | projector difference ARPACK vs dsyevr %.1e' % np.abs(P1 - P2).max())
