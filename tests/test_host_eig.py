"""csrc/host_eig.c: the leading eigenpairs of the samples x samples Gram matrix without LAPACK (tridiagonalisation,
bisection, inverse iteration), CHECKED, with LAPACK's dsyevr as the fallback (tools/_nam.py:_top_pcs).

What the global test consumes are the projectors onto the first k vectors for k in ks (_association.py:35-48): those must
equal dsyevr's to rounding on every Gram matrix the goldens produce and on spectra chosen to break the method; where the
leading spectrum is (nearly) degenerate the native solver must step aside.  CPU only: the routine needs no GPU."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from cna_amd import _ffi
from cna_amd.tools import _nam
from helpers import GOLDEN_DIR


def native(G, k):
    lib = _ffi.load()
    G = np.ascontiguousarray(G, dtype=np.float64)
    n = len(G)
    U, lam = np.empty((n, k)), np.empty(k + 1)
    r, o = C.c_double(), C.c_double()
    rc = lib.cna_host_top_eig(G.ctypes.data, n, k, U.ctypes.data, lam.ctypes.data, C.byref(r), C.byref(o))
    if rc != 0:
        return rc, U, lam, np.inf, np.inf
    # the routine reports residual / orthogonality of its tridiagonal stage; the same two figures on G itself, by the
    # library's own check and by numpy, must be at the same level: the larger of all is what the tests see
    r2, o2 = C.c_double(), C.c_double()
    assert lib.cna_host_eig_check(G.ctypes.data, n, k, U.ctypes.data, lam.ctypes.data, C.byref(r2), C.byref(o2)) == 0
    r3 = np.abs(G @ U - U * lam[:k]).max()
    o3 = np.abs(U.T @ U - np.eye(k)).max()
    assert abs(r2.value - r3) <= 1e-12 * max(1.0, np.abs(lam).max()) and abs(o2.value - o3) <= 1e-14
    return rc, U, lam, max(r.value, r2.value, r3), max(o.value, o2.value, o3)


def projector_gap(U, V, ks):
    return max(np.abs(U[:, :k] @ U[:, :k].T - V[:, :k] @ V[:, :k].T).max() for k in ks)


def with_spectrum(lam, seed=0):
    rs = np.random.RandomState(seed)
    Q, _ = np.linalg.qr(rs.randn(len(lam), len(lam)))
    G = (Q * np.asarray(lam, dtype=float)) @ Q.T
    return (G + G.T) / 2, Q


@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(GOLDEN_DIR, '*.npz'))), ids=lambda p: os.path.basename(p)[:-4])
def test_gram_matrices_of_the_goldens(path):
    """Every fixture's residualised NAM (samples x cells, from the reference): G = X X^T, ks as the reference chose them."""
    z = np.load(path)
    key = 'namresid' if 'namresid' in z.files else ('namresid_sub' if 'namresid_sub' in z.files else None)
    if key is None or 'ks' not in z.files:
        pytest.skip('no residualised NAM in this fixture')
    X = np.asarray(z[key], dtype=np.float64)
    G = X @ X.T
    G = (G + G.T) / 2
    ks = [int(k) for k in z['ks']]
    kmax = max(ks)
    if kmax >= len(G):
        pytest.skip('ks reaches the sample count')
    ref = _nam._top_pcs_lapack(G, kmax)
    w = np.linalg.eigvalsh(G)[::-1]
    if 4 * kmax > len(G) or len(G) < 8:
        assert _nam._top_pcs_native(G, kmax) is None           # outside the native solver's range: LAPACK's
        return
    rc, U, lam, resid, ortho = native(G, kmax)
    assert rc == 0
    assert np.abs(lam - w[:kmax + 1]).max() <= 1e-13 * w[0]
    assert resid <= 1e-13 * w[0] and ortho <= 1e-13
    if ((w[:kmax] - w[1:kmax + 1]) > 1e-6 * w[0]).all():
        assert projector_gap(U, ref, ks) < 1e-13
        got = _nam._top_pcs_native(G, kmax)
        assert got is not None and np.array_equal(got, U)
        # individual vectors, up to sign
        sgn = np.sign((U * ref).sum(axis=0))
        assert np.abs(U * sgn - ref).max() < 1e-10
    else:
        assert _nam._top_pcs_native(G, kmax) is None


@pytest.mark.parametrize('n,k', [(8, 1), (8, 2), (9, 2), (24, 4), (50, 4), (50, 10), (100, 8), (137, 34), (200, 16), (200, 50), (333, 20),
                                 (600, 30), (1024, 40)])
def test_nam_like_spectra(n, k):
    """Slowly decaying spectrum (16th / 1st ~ 0.3, like a NAM's), any size up to the limit of the library."""
    G, _ = with_spectrum(1e6 / (1.0 + 0.15 * np.arange(n)), seed=n)
    rc, U, lam, resid, ortho = native(G, k)
    assert rc == 0
    ref = _nam._top_pcs_lapack(G, k)
    w = np.linalg.eigvalsh(G)[::-1]
    assert np.abs(lam - w[:k + 1]).max() <= 1e-13 * w[0]
    assert resid <= 1e-13 * w[0] and ortho <= 1e-13
    assert projector_gap(U, ref, range(1, k + 1)) < 1e-13
    assert _nam._top_pcs_native(G, k) is not None


def test_adversarial_spectra():
    n, k = 120, 12
    cases = {
        # a tight cluster INSIDE the leading group, well separated from the rest: projectors at the cluster's edges agree,
        # the individual vectors inside it are the solver's choice -> the wrapper hands the matrix to LAPACK (gap rule)
        'cluster_inside': np.r_[[10, 9, 8, 7.0000001, 7.0, 6.9999999, 5, 4, 3, 2.5, 2.2, 2.0], np.linspace(1, 0.01, n - 12)],
        # graded over 16 decades, rank deficient at the bottom
        'graded': np.r_[10.0 ** -np.arange(0, 16, 16 / 40.0), np.zeros(n - 40)],
        # the rest of the spectrum one flat plateau just below the leading group
        'plateau_below': np.r_[np.linspace(3, 2, 12), np.full(n - 12, 1.9)],
        # pairs of close eigenvalues (relative gap 1e-5: still distinct for the gap rule)
        'pairs': np.r_[np.repeat(np.linspace(5, 1.5, 6), 2) * np.tile([1, 1 - 1e-5], 6), np.linspace(1, 0.1, n - 12)],
        # negative eigenvalues below (not a Gram matrix, but the solver takes the largest algebraically)
        'indefinite': np.r_[np.linspace(5, 2, 12), np.linspace(1, -6, n - 12)],
    }
    for name, lam_true in cases.items():
        G, Q = with_spectrum(lam_true, seed=3)
        rc, U, lam, resid, ortho = native(G, k)
        assert rc == 0, name
        w = np.sort(lam_true)[::-1]
        assert np.abs(lam - w[:k + 1]).max() <= 2e-13 * abs(w[0]), name
        assert resid <= 2e-13 * abs(w[0]) and ortho <= 1e-12, (name, resid, ortho)
        # projector onto the whole leading group (its lower edge is a real gap in every case)
        P_true = Q[:, np.argsort(-lam_true)[:k]]
        assert np.abs(U @ U.T - P_true @ P_true.T).max() < 1e-9, name
        ref = _nam._top_pcs_lapack(G, k)
        assert np.abs(U @ U.T - ref @ ref.T).max() < 1e-9, name
        got = _nam._top_pcs_native(G, k)
        gaps_ok = ((w[:k] - w[1:k + 1]) > 1e-6 * w[0]).all()
        assert (got is not None) == bool(gaps_ok), name


def test_degenerate_inputs_go_to_lapack():
    n, k = 40, 4
    eye = np.eye(n) * 3.0                                           # every gap zero
    assert _nam._top_pcs_native(eye, k) is None
    rc, U, lam, resid, ortho = native(eye, k)                       # ... while the routine itself still returns valid pairs
    assert rc == 0 and np.allclose(lam, 3.0) and resid < 1e-12 and ortho < 1e-12
    G, _ = with_spectrum(np.r_[[5.0, 5.0, 4.0, 3.0], np.linspace(1, 0.1, n - 4)])   # exactly repeated leading value
    assert _nam._top_pcs_native(G, k) is None
    assert _nam._top_pcs(G, k).shape == (n, k)
    assert _nam._top_pcs_native(np.zeros((n, n)), k) is None        # nothing positive
    bad = G.copy()
    bad[3, 5] = bad[5, 3] = np.nan
    assert _nam._top_pcs(bad, k) is None                            # (the caller takes the SVD and its errors, as before)
    assert _nam._top_pcs_native(G[:6, :6], 1) is None               # below the size where it pays
    assert _nam._top_pcs_native(G, 11) is None                      # k > n / 4
    lib = _ffi.load()
    assert lib.cna_host_top_eig(None, 10, 2, None, None, None, None) == 2


def test_structured_matrices():
    """Diagonal, block-diagonal and tridiagonal inputs: columns with nothing to annihilate (tau = 0), exact splits."""
    n, k = 64, 6
    d = np.linspace(10, 1, n)
    rs = np.random.RandomState(5)
    for name, G in (('diagonal', np.diag(d)),
                    ('blocks', np.kron(np.eye(4), with_spectrum(np.linspace(9, 1, 16), 1)[0]) + np.diag(np.repeat([0.3, 0.2, 0.1, 0.0], 16))),
                    ('tridiagonal', np.diag(d) + np.diag(np.full(n - 1, 0.3), 1) + np.diag(np.full(n - 1, 0.3), -1)),
                    ('arrow', np.diag(d) + np.outer(np.eye(n)[0], rs.rand(n)) + np.outer(rs.rand(n), np.eye(n)[0]) * 0)):
        G = (G + G.T) / 2
        rc, U, lam, resid, ortho = native(G, k)
        w, v = np.linalg.eigh(G)
        w, v = w[::-1], v[:, ::-1]
        assert rc == 0 and np.abs(lam - w[:k + 1]).max() <= 1e-13 * w[0], name
        assert resid <= 1e-13 * w[0] and ortho <= 1e-13, (name, resid, ortho)
        if ((w[:k] - w[1:k + 1]) > 1e-6 * w[0]).all():
            assert projector_gap(U, v, range(1, k + 1)) < 1e-12, name


def test_association_results_do_not_depend_on_the_eigen_solver():
    """The F-tests of the analysis with the native pairs and with LAPACK's: same p-values to rounding (oracle-side
    restatement of _association.py:35-61 on a golden's residualised NAM)."""
    from cna_amd.tools import _stats
    z = np.load(os.path.join(GOLDEN_DIR, 'c01_plain_f32.npz'))
    X = np.asarray(z['namresid'], dtype=np.float64)
    G = X @ X.T
    N = len(G)
    ks = np.asarray(z['ks'])
    rs = np.random.RandomState(0)
    Z = rs.randn(N, 200)
    a = _stats.minp_stats(Z, np.eye(N), _nam._top_pcs_lapack(G, int(ks.max())), ks, 0)
    U = _nam._top_pcs_native(G, int(ks.max()))
    assert U is not None
    b = _stats.minp_stats(Z, np.eye(N), U, ks, 0)
    assert np.array_equal(a[0], b[0])
    np.testing.assert_allclose(a[1], b[1], rtol=1e-10)
    np.testing.assert_allclose(a[2], b[2], rtol=1e-10)
