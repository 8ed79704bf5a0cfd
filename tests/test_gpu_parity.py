"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI,
against (a) the CPU oracle on the same seeded inputs, (b) the golden vectors captured from the
reference, (c) size-independent properties at larger sizes.

Tolerances: integers / masks / step counts exact; floats 1e-5 relative versus the reference
(BASELINE.json north_star), and much tighter (1e-10) versus the float64 oracle since both
evaluate the same float64 formulas.
"""
import numpy as np
import pandas as pd
import pytest
import scipy.sparse as sp

from helpers import messy_names, golden_names, load_case, run_product, assert_matches_golden, relerr, fdr_rows

pytestmark = pytest.mark.gpu

NAMES = golden_names()


@pytest.fixture(scope='module')
def eng():
    from cna_amd.engine import get_engine
    return get_engine()


@pytest.fixture()
def general_path():
    """Tests that reach into the general path of tools/_association.py (failures injected into the engine calls it makes,
    its helper-thread schedules) on REPEATED calls of a shaped analysis: those calls would otherwise go through
    cna_assoc_begin / cna_assoc_finish (tools/_fast.py; its own error paths: tests/test_gpu_fast.py)."""
    from cna_amd.tools import _fast
    keep = _fast.ENABLED
    _fast.ENABLED = False
    yield
    _fast.ENABLED = keep


@pytest.fixture(scope='module')
def orc():
    from oracle import cna_oracle
    return cna_oracle


def test_native_library_is_what_runs(eng):
    import ctypes as C
    from cna_amd import _ffi
    assert _ffi._lib is not None and _ffi.LIB_PATH.endswith('cna_amd/libcna_hip.so')
    with open('/proc/self/maps') as f:
        assert 'libcna_hip.so' in f.read()
    n = C.c_int(0)
    assert _ffi.load().cna_device_count(C.byref(n)) == 0 and n.value >= 1


# ------------------------------------------------------------------ kernel by kernel vs oracle
def _setup_nam(eng, orc, case, nsteps=3, want_kurt=True):
    from cna_amd.tools._nam import sample_codes
    A = sp.csr_matrix(case['data'].obsp['connectivities'])
    eng.ensure_graph(case['data'].obsp['connectivities'])
    eng.colsums(1)
    codes, labels = sample_codes(case['data'].obs[case['sid_name']])
    N = len(labels)
    C = np.bincount(codes, minlength=N).astype(float)
    eng.set_samples(codes, N, C)
    return A, codes, N


@pytest.mark.parametrize('name', ['c01_plain_f32', 'c05_ks_f64', 'c14_selfweight_autostop_unsorted'])
def test_diffusion_steps_vs_oracle_and_golden(eng, orc, name):
    from cna_amd import _ffi
    case = load_case(name)
    z = case['z']
    A, codes, N = _setup_nam(eng, orc, case)
    cs = eng.fetch_colsums()
    # column sums are formed in scipy's order (ascending row, position in row) without float atomics:
    # bit-identical to the float64 oracle for float32 AND float64 graphs, in any device cell order
    np.testing.assert_array_equal(cs, orc.column_sums(A, 1, 'f64'))
    S = np.zeros((A.shape[0], N), dtype=bool)
    S[np.arange(A.shape[0]), codes] = True
    s = S
    colsums = orc.column_sums(A, 1, 'f64')
    nst = len(z['steps'])
    for i in range(nst):
        s = orc.diffusion_step(A, s, colsums, 1, first_onehot=(i == 0), mode='f64')
        eng.nam_step(True, i + 1 < nst, True)
        got = eng.nam_full()
        want = s / S.sum(axis=0)
        assert relerr(got, want) < 1e-13, (i, relerr(got, want))
        # deterministic column sums + kernels that round like scipy's csr_matvecs (unfused multiply and add,
        # CSR order): bit-identical, float64 graphs included
        np.testing.assert_array_equal(got, want)
        assert relerr(got, z['steps'][i] / S.sum(axis=0)) < 1e-5          # the reference itself
        kurt = eng.cell_stat(A.shape[0])
        np.testing.assert_allclose(kurt, orc.row_kurtosis(want), rtol=1e-9, atol=1e-12)
        assert np.median(kurt) == pytest.approx(z['steps_medkurt'][i], rel=1e-5)
    # the walk is column-stochastic: every sample's row of the (samples x cells) NAM sums to 1
    got = eng.nam_full()
    np.testing.assert_allclose(got.sum(axis=0), 1.0, rtol=1e-10)


@pytest.mark.parametrize('dtype', ['float32', 'float64'])
def test_column_sums_are_scipys_sequential_sums(eng, orc, dtype):
    """colsums = A.sum(axis=0) + w (_nam.py:28) bit for bit in float64, for weights spanning 60 binades
    (where the order of addition matters), hub columns with more than 64 and more than 4096 entries,
    empty columns, an asymmetric pattern, and whatever order the device keeps the cells in."""
    rs = np.random.RandomState(7)
    n = 9000
    rows = rs.randint(0, n, size=40 * n)
    cols = rs.randint(0, n, size=40 * n)
    cols[:5000] = 17                       # a hub column (~5000 entries)
    cols[5000:5200] = 23                   # > 64 entries
    cols[cols == 99] = 98                  # an empty column
    vals = np.exp(rs.uniform(-40, 0, size=len(rows)))
    A = sp.csr_matrix((vals, (rows, cols)), shape=(n, n))
    A.sum_duplicates()
    A = A.astype(dtype)
    A.indices = A.indices.astype(np.int32)
    eng.ensure_graph(A)
    for w in (1, 0.25):
        eng.colsums(w)
        got = eng.fetch_colsums()
        want = np.asarray(A.astype(np.float64).sum(axis=0)).ravel() + w
        np.testing.assert_array_equal(got, want)
    assert eng.fetch_colsums()[99] == 0.25


def test_dense_diffusion_and_self_weight(eng, orc):
    import cna_amd as cna
    case = load_case('c01_plain_f32')
    z = case['z']
    A = sp.csr_matrix(case['data'].obsp['connectivities'])
    for w, key in ((1, 'diffuse_out_2'), (0.5, 'diffuse_out_2_sw05')):
        out = cna.tl.diffuse(case['data'], z['diffuse_in'], 2, self_weight=w)
        assert relerr(out, orc.diffuse(A, z['diffuse_in'], 2, self_weight=w, mode='f64')) < 1e-14
        assert relerr(out, z[key]) < 1e-5
    # a single column, and a DataFrame in -> DataFrame out, generator semantics
    one = z['diffuse_in'][:, :1]
    assert relerr(cna.tl.diffuse(case['data'], one, 3), orc.diffuse(A, one, 3, mode='f64')) < 1e-14
    steps = list(cna.tl.diffuse_stepwise(case['data'], pd.DataFrame(z['diffuse_in']), maxnsteps=2))
    assert len(steps) == 2 and isinstance(steps[1], pd.DataFrame)
    assert relerr(steps[1].values, z['diffuse_out_2']) < 1e-5
    case = load_case('c14_selfweight_autostop_unsorted')
    NAM, keep = cna.tl.nam(case['data'], case['sid_name'], nsteps=2, self_weight=2)
    assert relerr(NAM.values, case['z']['tlnam_sw2']) < 1e-5 and keep.all()


@pytest.mark.parametrize('name', ['c03_covs_batches', 'c12_batchy_qc'])
def test_tl_nam_qc(eng, name):
    import cna_amd as cna
    case = load_case(name)
    z = case['z']
    NAM, keep = cna.tl.nam(case['data'], case['sid_name'], batches=case['batches'], nsteps=3)
    assert np.array_equal(keep, z['tlnam_keep'])
    assert NAM.shape == z['tlnam'].shape and relerr(NAM.values, z['tlnam']) < 1e-5
    assert list(NAM.index) == z['tlnam_index'].tolist()


def _random_x(rs, n, N):
    return rs.randn(n, N) * rs.rand(1, N) + rs.randn(n, 1)


@pytest.mark.parametrize('n,N', [(1000, 20), (777, 50), (513, 3), (300, 130), (257, 200), (64, 256), (5, 17),
                                 (301, 160), (130, 192),       # 160 / 192: row stride padded off a 256-byte multiple
                                 (333, 260), (200, 400), (180, 513), (100, 1024)])   # k-split GEMM, multi-pass Gram
def test_dense_row_kernels_vs_numpy(eng, orc, n, N):
    from cna_amd import _ffi
    rs = np.random.RandomState(n + N)
    X = _random_x(rs, n, N)
    # standardize(center) == svd_nam's re-standardisation
    eng.upload_x(X)
    eng.standardize(center=True)
    Xc = X - X.mean(axis=1, keepdims=True)
    Xs = Xc / Xc.std(axis=1, ddof=1)[:, None]
    got = eng.fetch_matrix(_ffi.MAT_X)
    assert relerr(got, Xs) < 1e-13
    assert relerr(eng.fetch_matrix(_ffi.MAT_X, transposed=True), Xs.T) < 1e-13
    # Gram on the matrix cores
    G = eng.gram()
    assert relerr(G, Xs.T.dot(Xs)) < 1e-12
    assert np.array_equal(G, G.T)
    # projection X.W (V = NAM^T U / sqrt(svs)); asymmetric W catches a transposed operand
    W = rs.randn(N, N)
    assert relerr(eng.project(W), Xs.dot(W)) < 1e-12
    W2 = rs.randn(N, 5)
    assert relerr(eng.project(W2), Xs.dot(W2)) < 1e-12
    # residualisation (X - mean).M^T with an asymmetric M, then in place a second time (ridge loop)
    eng.upload_x(X)
    M = rs.randn(N, N)
    eng.resid_apply(M, center=True)
    want = Xc.dot(M.T)
    assert relerr(eng.fetch_matrix(_ffi.MAT_X), want) < 1e-12
    M2 = rs.randn(N, N)
    eng.resid_apply(M2, center=False)
    want = want.dot(M2.T)
    assert relerr(eng.fetch_matrix(_ffi.MAT_X), want) < 1e-12
    eng.standardize(center=False)
    want = want / want.std(axis=1, ddof=1)[:, None]
    assert relerr(eng.fetch_matrix(_ffi.MAT_X), want) < 1e-12
    # batch kurtosis of the working matrix
    nb = min(7, N)
    bc = np.arange(N) % nb
    eng.batch_kurtosis(_ffi.MAT_X, bc, nb)
    np.testing.assert_allclose(eng.x_stat(), orc.batch_kurtosis(want, bc, nb), rtol=1e-9)
    # neighbourhood coefficients
    y = rs.randn(N)
    nc, m = eng.ncorrs(y, fetch=True)
    ref = (want * y[None, :]).sum(axis=1) / N
    assert relerr(nc, ref) < 1e-12 and m == pytest.approx(np.abs(ref).max(), rel=1e-12)


@pytest.mark.parametrize('n,N,r,nb', [(1000, 5, 1, 2), (777, 17, 3, 3), (4097, 64, 16, 16), (3000, 100, 7, 5), (2000, 128, 8, 9),
                                      (1500, 130, 0, 4), (999, 255, 0, 16), (15, 33, 2, 1), (16 * 257 + 3, 48, 5, 7),
                                      (640, 100, 20, 6), (500, 160, 4, 3)])
def test_sixteen_rows_per_wave_passes_vs_numpy(eng, orc, monkeypatch, n, N, r, nb):
    """rows16.hip: batch kurtosis (_nam.py:78-82) and the in-place ridge pass -- centre, M = I - C.W in factored form,
    batch kurtosis of the result, / std, coefficients (_nam.py:122,143-150,159, _association.py:77) -- with sixteen
    rows per wave and the projector on the matrix cores, against numpy and against the wave-per-row kernels
    (CNA_ROWPASS16=0).  Shapes: sample counts off the multiples of 4 and 16, row counts off the multiples of 16, ranks 1
    ... 16, one batch, sixteen batches; 130 / 255 / 160 samples and rank 20 are beyond the pass with a projector and
    must fall through to the wave-per-row kernels with the same results."""
    from cna_amd import _ffi
    import scipy.sparse as sp
    eng.ensure_graph(sp.identity(n, format='csr', dtype=np.float32))      # (the per-cell buffers are sized by the graph)
    rs = np.random.RandomState(n + 3 * N + r)
    X = _random_x(rs, n, N)
    bc = rs.randint(0, nb, size=N)
    bc[:nb] = np.arange(nb)                                   # every batch has a sample
    y = rs.randn(N)
    Xc = X - X.mean(axis=1, keepdims=True)
    out = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('CNA_ROWPASS16', mode)
        eng.upload_x(X)
        eng.batch_kurtosis(_ffi.MAT_X, bc, nb)
        bk0 = eng.x_stat().copy()
        if r > 0:
            Cm = rs.randn(N, r) if mode == '1' else out['C']
            Cm = Cm - Cm.mean(axis=0)
            W = np.linalg.solve(Cm.T.dot(Cm) + 0.3 * N * np.eye(r), Cm.T)
            out['C'] = Cm
            maxabs, med = eng.resid_lowrank_bk(Cm, W, y, bc, nb)
            want = Xc - Xc.dot(W.T).dot(Cm.T)
            bk1 = eng.x_stat().copy()
            ref_bk1 = orc.batch_kurtosis(want, bc, nb)
            want = want / want.std(axis=1, ddof=1)[:, None]
            got = eng.fetch_matrix(_ffi.MAT_X)
            assert relerr(got, want) < 1e-11
            np.testing.assert_allclose(bk1, ref_bk1, rtol=1e-8)
            np.testing.assert_allclose(med, np.median(ref_bk1), rtol=1e-8)         # (one batch: NaN on both sides)
            ref_nc = (want * y[None, :]).sum(axis=1) / N
            assert maxabs == pytest.approx(np.abs(ref_nc).max(), rel=1e-10)
            out[mode] = (bk0, got, bk1, maxabs)
        else:
            out[mode] = (bk0,)
        np.testing.assert_allclose(bk0, orc.batch_kurtosis(X, bc, nb), rtol=1e-9)
    for a, b in zip(out['1'], out['0']):
        np.testing.assert_allclose(a, b, rtol=1e-8, atol=1e-13)


@pytest.mark.parametrize('n,N,P', [(3000, 20, 100), (2049, 50, 200), (1000, 100, 70), (600, 200, 130),
                                   (100, 256, 64), (16, 12, 5), (700, 160, 90), (333, 224, 33),
                                   (500, 192, 40), (250, 157, 70), (260, 221, 65),
                                   (400, 300, 40), (200, 700, 33), (150, 1024, 17)])
def test_local_null_counts_are_exact(eng, n, N, P):
    """tails / ranks / num_detected are integers: they must equal a brute-force count."""
    from cna_amd import _ffi
    rs = np.random.RandomState(n + N + P)
    X = rs.randn(n, N)
    X = (X - X.mean(axis=1, keepdims=True))
    X /= X.std(axis=1, ddof=1)[:, None]
    eng.upload_x(X)
    y = rs.randn(N)
    nc, maxabs = eng.ncorrs(y, fetch=True)
    Yc = rs.randn(N, P)
    Yc /= Yc.std(axis=0, ddof=1)
    maxcorr = max(maxabs, 0.001)
    thr = np.arange(maxcorr / 4, maxcorr, maxcorr / 400)
    edges = thr ** 2 - 1e-8 - 1e-5 * thr ** 2
    tails = eng.null_local(Yc, edges)
    z2 = (np.abs(X.dot(Yc)) / N) ** 2
    want = np.array([len(v) - np.searchsorted(v, edges, side='left') for v in np.sort(z2, axis=0).T])
    # a value within 1 ulp of an edge may land on either side: allow a handful of off-by-ones
    diff = np.abs(tails - want)
    assert diff.max() <= 1 and diff.sum() <= 3, (diff.max(), diff.sum())
    assert (np.diff(tails, axis=1) <= 0).all() and tails.max() <= n
    ranks, numdet = eng.obs_counts(edges, thr)
    assert np.array_equal(ranks, [(nc ** 2 >= e).sum() for e in edges])
    assert np.array_equal(numdet, [(np.abs(nc) > t).sum() for t in thr])


@pytest.mark.parametrize('n,N,P,heavy', [(3000, 20, 100, 0), (2049, 50, 200, 0), (5000, 100, 1000, 0), (4100, 200, 333, 0),
                                         (100, 256, 64, 0), (16, 12, 5, 0), (700, 160, 90, 1), (333, 224, 33, 0),
                                         (2500, 192, 1000, 1), (250, 157, 70, 0), (260, 33, 65, 1), (31, 64, 64, 0)])
def test_local_null_i8_sums_equal_f64_kernel(eng, n, N, P, heavy):
    """The sums-only pass (integer matrix cores, csrc/null_i8.hip) against the f64 kernel's per-permutation
    tails on the same resident phenotypes: the same integers, every threshold.  `heavy`: rows with outliers
    and zero rows (per-row scales), phenotypes with ties."""
    rs = np.random.RandomState(7 * n + N + P)
    X = rs.randn(n, N)
    if heavy:
        X[::7] = rs.standard_t(2, size=X[::7].shape)
        X[3::50] *= 1e-3
    X = (X - X.mean(axis=1, keepdims=True))
    X /= X.std(axis=1, ddof=1)[:, None]
    if heavy:
        X[5::40] = 0.0
    eng.upload_x(X)
    y = rs.randn(N)
    nc, maxabs = eng.ncorrs(y, fetch=True)
    Yc = rs.randn(N, P) if not heavy else rs.randint(0, 3, size=(N, P)).astype(float) + 1e-3 * rs.randn(N, P)
    Yc -= Yc.mean(axis=0)
    Yc /= Yc.std(axis=0, ddof=1)
    maxcorr = max(maxabs, 0.001)
    thr = np.arange(maxcorr / 4, maxcorr, maxcorr / 400)
    edges = thr ** 2 - 1e-8 - 1e-5 * thr ** 2
    tails = eng.null_local(Yc, edges)                        # f64 kernel, leaves Yc resident
    sums = eng.null_local_resident(0, P, edges, sums_only=True)
    used, rechecked, fallback = eng.null_local_i8_stats()
    assert used and not fallback
    assert np.array_equal(sums, tails.sum(axis=0))
    assert rechecked < 0.02 * n * P + 64, rechecked
    # a sub-range of the resident columns
    if P > 70:
        sums = eng.null_local_resident(3, 67, edges, sums_only=True)
        assert np.array_equal(sums, tails[3:70].sum(axis=0))


@pytest.mark.parametrize('name', ['c01_plain_f32', 'c03_covs_batches', 'c05_ks_f64', 'c12_batchy_qc'])
def test_local_null_i8_on_the_planes_written_by_the_selection_pass(eng, name):
    """After an analysis the working matrix, its fixed-point digit planes (written by the selection pass where
    that pass produced X, by k_quant_x otherwise) and the conditioned phenotypes are resident: the integer
    pass on them must return the f64 kernel's integers, and the FDR table of the analysis must be the one those
    integers give."""
    case = load_case(name)
    res, err, _ = run_product(case, eng)
    assert err is None
    thr = res.fdrs.threshold.values
    edges = thr ** 2 - 1e-8 - 1e-5 * thr ** 2
    P = min(1000, int(case['call'].get('Nnull', 1000)))
    tails = eng.null_local_resident(1, P, edges)
    sums = eng.null_local_resident(1, P, edges, sums_only=True)
    used, rechecked, fallback = eng.null_local_i8_stats()
    assert used and not fallback
    assert np.array_equal(sums, tails.sum(axis=0))
    # the FDR table of the analysis is mean_p(tails / ranks): rebuild it from the f64 tails
    ranks = np.array([(res.ncorrs.values ** 2 >= e).sum() for e in edges])
    with np.errstate(all='ignore'):
        want = (tails / ranks).mean(axis=0)
    np.testing.assert_allclose(res.fdrs.fdr.values, want, rtol=1e-12, equal_nan=True)


@pytest.mark.parametrize('kind', ['irregular', 'many', 'near_zero', 'wide'])
def test_local_null_sums_outside_the_integer_path(eng, kind):
    """Passes the integer kernel does not take -- thresholds that are no arithmetic progression, more
    thresholds than its LDS counters hold, a first cut within a step of zero, more than 256 samples -- go to
    the f64 kernel and give the same sums as the per-permutation tails."""
    rs = np.random.RandomState(11)
    n, N, P = (1500, 40, 130) if kind != 'wide' else (300, 300, 70)
    X = rs.randn(n, N)
    X = (X - X.mean(axis=1, keepdims=True))
    X /= X.std(axis=1, ddof=1)[:, None]
    eng.upload_x(X)
    nc, maxabs = eng.ncorrs(rs.randn(N), fetch=True)
    Yc = rs.randn(N, P)
    Yc /= Yc.std(axis=0, ddof=1)
    if kind == 'irregular':
        thr = np.sort(maxabs * (0.2 + 0.8 * rs.rand(120)))
    elif kind == 'many':
        thr = np.arange(maxabs / 4, maxabs, maxabs / 670)
    elif kind == 'near_zero':
        thr = np.arange(maxabs / 1000, maxabs, maxabs / 300)
    else:
        thr = np.arange(maxabs / 4, maxabs, maxabs / 400)
    edges = thr ** 2 - 1e-8 - 1e-5 * thr ** 2
    edges = edges[edges > 0]
    tails = eng.null_local(Yc, edges)
    sums = eng.null_local_resident(0, P, edges, sums_only=True)
    used, rechecked, fallback = eng.null_local_i8_stats()
    assert not used and rechecked == 0
    assert np.array_equal(sums, tails.sum(axis=0))


def test_local_null_i8_falls_back_when_the_queue_overflows(eng, monkeypatch):
    """A recheck queue too small for the outputs near a cut (forced here through CNA_I8_QCAP): the integer
    pass gives up on the device (status word) and the stand-by f64 kernel behind it produces the sums."""
    rs = np.random.RandomState(5)
    n, N, P = 20000, 50, 640
    X = rs.randn(n, N)
    X = (X - X.mean(axis=1, keepdims=True))
    X /= X.std(axis=1, ddof=1)[:, None]
    eng.upload_x(X)
    nc, maxabs = eng.ncorrs(rs.randn(N), fetch=True)
    Yc = rs.randn(N, P)
    Yc /= Yc.std(axis=0, ddof=1)
    thr = np.arange(maxabs / 4, maxabs, maxabs / 400)
    edges = thr ** 2 - 1e-8 - 1e-5 * thr ** 2
    tails = eng.null_local(Yc, edges)
    monkeypatch.setenv('CNA_I8_QCAP', '8')
    sums = eng.null_local_resident(0, P, edges, sums_only=True)
    used, rechecked, fallback = eng.null_local_i8_stats()
    assert used and fallback and rechecked > 8
    assert np.array_equal(sums, tails.sum(axis=0))
    monkeypatch.delenv('CNA_I8_QCAP')
    sums = eng.null_local_resident(0, P, edges, sums_only=True)
    used, rechecked, fallback = eng.null_local_i8_stats()
    assert used and not fallback and np.array_equal(sums, tails.sum(axis=0))


def test_percell_lookup(eng, orc):
    case = load_case('c12_batchy_qc')
    res, err, _ = run_product(case, eng)
    assert err is None
    coef = case['data'].obs['coef'].values
    fdr = case['data'].obs['coef_fdr'].values
    assert np.isnan(coef[~res.kept]).all() and (fdr[~res.kept] == 1).all()
    want = orc.percell_fdr(coef, res.fdrs.threshold.values, res.fdrs.fdr.values)
    np.testing.assert_allclose(fdr, want, rtol=1e-14)


# ----------------------------------------------------------------------- end to end vs reference
@pytest.mark.parametrize('name', NAMES + messy_names())
def test_association_matches_reference(eng, name):
    """Every fixture captured from the reference: the branch-covering cases (c??) and the randomly drawn messy inputs
    (f??_messy: every sample-level input in an order of its own, NaNs, samples the data does not have, unused categories,
    donor groups, ks / ridges / max_frac_pcs) -- the same numbers, or the same exception with the same message."""
    case = load_case(name)
    z = case['z']
    res, err, msgs = run_product(case, eng)
    if z['raised'].item():
        assert err is not None and type(err).__name__ + ': ' + str(err) == z['raised'].item()
        if 'obs_coef' in z:
            np.testing.assert_allclose(case['data'].obs['coef'].values, z['obs_coef'], rtol=0,
                                       atol=1e-5 * np.nanmax(np.abs(z['obs_coef'])), equal_nan=True)
        else:
            assert 'coef' not in case['data'].obs
        return
    assert err is None, repr(err)
    assert_matches_golden(res, case['data'], z, tol=1e-5, name=name)
    # the warnings of the analysis (not pandas' own remark about the reference's misaligned boolean indexer)
    import json
    skip = ('already exists', 'Boolean Series key')
    ref_msgs = [m for m in json.loads(z['warnings'].item()) if not any(t in m for t in skip)]
    assert [m for m in msgs if not any(t in m for t in skip)] == ref_msgs


@pytest.mark.parametrize('name', ['c01_plain_f32', 'c03_covs_batches', 'c09_y_nan_extra_reordered', 'c15_ridges_custom'])
def test_association_matches_f64_oracle_tightly(eng, orc, name):
    case = load_case(name)
    res, err, _ = run_product(case, eng)
    assert err is None
    ref = orc.association(case['data'], case['y'], case['sid_name'], batches=case['batches'], covs=case['covs'],
                          donorids=case['donorids'], mode='f64', **case['call'])
    assert int(res.k) == ref['k'] and res.p == ref['p'] and np.array_equal(res.kept, ref['kept'])
    assert relerr(res.nam.values.T, ref['nam']) < 1e-13
    assert relerr(res.namresid.values.T, ref['namresid']) < 1e-10
    assert relerr(res.ncorrs.values, ref['ncorrs']) < 1e-10
    assert relerr(res.namresid_svs.values, ref['svs']) < 1e-10
    assert relerr(res.nullminps, ref['nullminps']) < 1e-8
    T = fdr_rows(res.fdrs, ref['fdrs'], ref['ncorrs'])
    assert np.array_equal(res.fdrs.num_detected.values[:T], ref['fdrs']['num_detected'][:T])
    np.testing.assert_allclose(res.fdrs.fdr.values[:T], ref['fdrs']['fdr'][:T], rtol=1e-9, atol=1e-13)


def test_demo_like_config1(eng):
    """BASELINE.json configs[0]: the demo recipe (10 000 cells x 50 samples; covs = male, batches = batch,
    y = case; nsteps=3, Nnull=100) against what the reference returned (tests/golden/d01_demo_like.npz)."""
    import cna_amd as cna
    import warnings
    from helpers import load_demo_case, assert_matches_demo
    case = load_demo_case()
    data = case['data']
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        res = cna.tl.association(data, case['y'], 'id', batches=case['batches'], covs=case['covs'], return_full=True,
                                 engine=eng, **case['call'])
    out = dict(k=res.k, ks=res.ks, r=res.r, kept=res.kept, p=res.p, ncorrs=res.ncorrs.values, nullminps=res.nullminps,
               svs=res.namresid_svs.values, nam=res.nam.values.T, namresid=res.namresid.values.T,
               fdrs=dict(threshold=res.fdrs.threshold.values, fdr=res.fdrs.fdr.values,
                         num_detected=res.fdrs.num_detected.values))
    assert_matches_demo(out, case['z'], 1e-5, obs=dict(coef=data.obs['coef'].values, coef_fdr=data.obs['coef_fdr'].values))


def test_svd_nam_public(eng):
    import cna_amd as cna
    z = load_case('c01_plain_f32')['z']
    nam_df = pd.DataFrame(z['nam'], index=z['nam_index'].tolist())
    U, svs, V = cna.tl.svd_nam(nam_df)
    assert relerr(svs.values, z['svd_svs']) < 1e-6
    from helpers import sign_align
    a, b = sign_align(U.values, z['svd_U'], 5)
    assert relerr(a, b) < 1e-5
    a, b = sign_align(V.values, z['svd_V'], 5)
    assert relerr(a, b) < 1e-5


# -------------------------------------------------------------------------- edge cases / properties
def test_isolated_and_heavy_rows(eng, orc):
    """cells without neighbours, a row with > 64 neighbours (multi-chunk), N not a multiple of 4."""
    from cna_amd import _ffi
    rs = np.random.RandomState(3)
    n, N = 400, 7
    A = sp.random(n, n, density=0.02, random_state=rs, format='lil', dtype=np.float64)
    A[5, :] = 0
    A[:, 5] = 0                       # isolated cell
    A[9, :] = rs.rand(n)              # 399 neighbours
    A[9, 5] = 0
    A = sp.csr_matrix(A)
    A.setdiag(0)
    A.eliminate_zeros()
    A = A.astype(np.float32)
    codes = rs.randint(0, N, n).astype(np.int32)
    obs = pd.DataFrame({'id': codes})
    data = type('D', (), {'obs': obs, 'obsp': {'connectivities': A}, 'uns': {}})()
    import cna_amd as cna
    NAM, keep = cna.tl.nam(data, 'id', nsteps=3)
    ref = orc.build_nam(A, codes, N, nsteps=3, mode='f64')
    assert relerr(NAM.values.T, ref['nam']) < 1e-13 and keep.all()
    # the isolated cell keeps all its mass in its own sample
    row = NAM.values[:, 5] * np.bincount(codes, minlength=N)
    assert row[codes[5]] == pytest.approx(1.0) and np.count_nonzero(row) == 1


def test_properties_at_scale(eng):
    """size-independent properties on a larger synthetic problem (oracle-free)."""
    import cna_amd as cna
    from cna_amd import synth
    data, meta = synth.make_dataset(60000, 50, k=30, seed=5, n_covs=2)
    res = cna.tl.association(data, meta['y'], 'id', covs=meta['covs'], nsteps=3, Nnull=500, seed=1, return_full=True)
    nam = res.nam.values                                   # samples x cells
    np.testing.assert_allclose(nam.sum(axis=1), 1.0, rtol=1e-9)      # column-stochastic walk
    X = res.namresid.values
    assert np.abs(X.mean(axis=0)).max() < 1e-10
    np.testing.assert_allclose(X.std(axis=0, ddof=1), 1.0, rtol=1e-10)
    # residualised NAM is orthogonal to the covariates
    cz = (meta['covs'] - meta['covs'].mean()) / meta['covs'].std()
    assert np.abs(cz.values.T.dot(X)).max() < 1e-8
    U = res.namresid_sampleXpc.values
    np.testing.assert_allclose(U.T.dot(U), np.eye(U.shape[1]), atol=1e-10)
    np.testing.assert_allclose(U.dot(np.diag(res.namresid_varexp.values * 50 * X.shape[1])).dot(U.T), X.dot(X.T),
                               rtol=1e-7, atol=1e-6 * X.shape[1])
    f = res.fdrs
    assert (np.diff(f.num_detected.values) <= 0).all()
    assert f.num_detected.values[0] == (np.abs(res.ncorrs.values) > f.threshold.values[0]).sum()
    yz = (meta['y'].values - meta['y'].values.mean()) / meta['y'].values.std()
    np.testing.assert_allclose(res.ncorrs.values, yz.dot(X) / 50, rtol=1e-9, atol=1e-12)
    assert 1 / 501 <= res.p <= 1 and len(res.nullminps) == 500


@pytest.mark.parametrize('selftest', [0, 40])
def test_rccl_path_with_one_rank(orc, monkeypatch, selftest):
    """The collectives of the sharded path (all-reduce / all-gather through librccl.so) on a
    one-rank communicator: same answers as without a communicator.  selftest > 0 additionally
    routes the state exchange between diffusion steps through the halo path (pack kernel, grouped
    ncclSend/ncclRecv -- here to the rank itself -- and unpack kernel) for that many rows."""
    from cna_amd.engine import Engine
    case = load_case('c12_batchy_qc')
    if selftest:
        monkeypatch.setenv('CNA_HALO_SELFTEST', str(selftest))
    e = Engine(device=0, rank=0, nranks=1, unique_id=Engine.new_unique_id())
    try:
        res, err, _ = run_product(case, e)
        assert err is None, repr(err)
        assert (e.halo is not None and e.halo[0] == e.halo[1] >= selftest) if selftest else e.halo is None
        assert_matches_golden(res, case['data'], case['z'], tol=1e-5, name='c12_batchy_qc')
        import cna_amd as cna
        A = sp.csr_matrix(case['data'].obsp['connectivities'])
        s0 = np.random.RandomState(1).rand(A.shape[0], 5)
        assert relerr(cna.tl.diffuse(case['data'], s0, 3, engine=e), orc.diffuse(A, s0, 3, mode='f64')) < 1e-14
        prof_names = e.prof()
    finally:
        e.close()


@pytest.mark.parametrize('N,P,r,ks', [(50, 1001, 0, [1, 2, 3, 4]), (24, 37, 2, [2, 5]), (200, 300, 5, [4, 8, 12, 16]),
                                       (12, 5, 0, [1]), (130, 64, 1, [3, 26]), (200, 10001, 5, [4, 8, 12, 16]),
                                       (600, 333, 2, [12, 24, 36, 48, 100]), (1024, 50, 0, [20, 40])])
def test_global_test_matches_scipy(eng, N, P, r, ks):
    """Device F-tests (incomplete-beta continued fraction) against scipy's fdtrc through the
    host restatement of _minp_stats; includes strong signals (p down to ~1e-60)."""
    from cna_amd.tools._stats import minp_stats
    rs = np.random.RandomState(N + P)
    X = rs.randn(4 * N, N)
    eng.upload_x(X)                                   # only to define the sample axis
    Q, _ = np.linalg.qr(rs.randn(N, N))
    U = Q                                             # orthonormal "PCs"
    C = rs.randn(N, max(r, 1))
    M = np.eye(N) - C.dot(np.linalg.solve(C.T.dot(C), C.T)) if r else np.eye(N)
    Y = rs.randn(N, P)
    Y[:, 0] = 8 * U[:, 0] + 0.05 * rs.randn(N)        # a column almost inside the span of PC1
    Y[:, 1] = U[:, :max(ks)].dot(rs.randn(max(ks))) + 1e-3 * rs.randn(N)
    eng.condition(M, Y)
    kidx, p, r2 = eng.global_test(U, ks, r)
    kidx_ref, p_ref, r2_ref = minp_stats(Y, M, U, np.asarray(ks), r)
    assert np.array_equal(kidx, kidx_ref)
    np.testing.assert_allclose(p, p_ref, rtol=1e-9, atol=0)
    np.testing.assert_allclose(r2, r2_ref, rtol=1e-9, atol=1e-13)
    assert p.min() < 1e-8
    # the resident matrix also feeds the local null: same counts as uploading the columns
    Zc = M.dot(Y)
    Zc /= Zc.std(axis=0, ddof=1)
    y = rs.randn(N)
    _, m = eng.ncorrs(y)
    thr = np.arange(m / 4, m, m / 400)
    edges = thr ** 2 - 1e-8 - 1e-5 * thr ** 2
    Pl = min(P - 1, 70)
    a = eng.null_local_resident(1, Pl, edges)
    b = eng.null_local(np.ascontiguousarray(Zc[:, 1:1 + Pl]), edges)
    assert np.abs(a - b).max() <= 1


@pytest.mark.parametrize('n,N,extra', [(3000, 70, {}), (2500, 130, dict(n_covs=3)), (2000, 200, dict(n_covs=2, n_batches=4)),
                                       (2400, 160, dict(n_covs=1)),
                                       (1500, 256, {}), (4000, 33, dict(n_batches=9, n_covs=1)),
                                       (2000, 300, dict(n_covs=2)), (2500, 520, {}), (3000, 600, dict(n_covs=1, n_batches=3)),
                                       (3000, 1024, {})])
def test_association_wide_sample_axis_vs_oracle(eng, orc, n, N, extra):
    """end to end at sample counts the golden fixtures do not reach (several 64-lane chunks per row,
    deep k-loops in the MFMA kernels, every local-null instantiation family), against the f64 oracle."""
    import cna_amd as cna
    from cna_amd import synth
    import warnings
    data, meta = synth.make_dataset(n, N, k=15, seed=n + N, **extra)
    kw = dict(nsteps=3, Nnull=130, seed=5)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        res = cna.tl.association(data, meta['y'], 'id', covs=meta['covs'], batches=meta['batches'], return_full=True, **kw)
        ref = orc.association(data, meta['y'], 'id', covs=meta['covs'], batches=meta['batches'], mode='f64', **kw)
    assert int(res.k) == ref['k'] and res.p == ref['p'] and np.array_equal(res.kept, ref['kept'])
    assert relerr(res.nam.values.T, ref['nam']) < 1e-13
    assert relerr(res.namresid.values.T, ref['namresid']) < 1e-9
    assert relerr(res.ncorrs.values, ref['ncorrs']) < 1e-9
    assert relerr(res.namresid_svs.values, ref['svs']) < 1e-9
    np.testing.assert_allclose(res.nullminps, ref['nullminps'], rtol=1e-7)
    T = fdr_rows(res.fdrs, ref['fdrs'], ref['ncorrs'])
    assert np.array_equal(res.fdrs.num_detected.values[:T], ref['fdrs']['num_detected'][:T])
    np.testing.assert_allclose(res.fdrs.fdr.values[:T], ref['fdrs']['fdr'][:T], rtol=1e-8, atol=1e-13)
    np.testing.assert_allclose(data.obs['coef_fdr'].values, ref['obs_coef_fdr'], rtol=1e-8, atol=1e-13)
    from helpers import sign_align
    V, Vref = sign_align(res.namresid_nbhdXpc.values, ref['V'], int(res.k))
    assert relerr(V, Vref) < 1e-6


def test_results_do_not_depend_on_device_cell_order(monkeypatch):
    """The engine renumbers cells (reverse Cuthill-McKee) for gather locality; per-row neighbour
    order is preserved, so every field must be bit-identical to a run in the caller's order --
    here a deliberately scrambled one, with QC dropping cells and a non-trivial M."""
    import cna_amd as cna
    from cna_amd import synth
    from cna_amd.engine import Engine
    data, meta = synth.make_dataset(20000, 30, k=15, seed=11, n_covs=1, n_batches=3, cluster_sorted=False)
    out = {}
    for flag in ('0', '1'):
        monkeypatch.setenv('CNA_REORDER', flag)
        e = Engine(device=0)
        try:
            res = cna.tl.association(data, meta['y'], 'id', covs=meta['covs'], batches=meta['batches'], Nnull=200,
                                     seed=3, return_full=True, engine=e)
            assert (e.perm is None) == (flag == '0')
            if flag == '1':
                assert not np.array_equal(e.perm, np.arange(len(e.perm)))
            out[flag] = dict(p=res.p, k=res.k, kept=res.kept.copy(), ncorrs=res.ncorrs.values.copy(),
                             nam=res.nam.values.copy(), namresid=res.namresid.values.copy(),
                             V=res.namresid_nbhdXpc.values.copy(), U=res.namresid_sampleXpc.values.copy(),
                             fdr=res.fdrs.values.copy(), coef=data.obs['coef'].values.copy(),
                             coef_fdr=data.obs['coef_fdr'].values.copy())
            sw = cna.tl.diffuse(data, np.random.RandomState(0).rand(20000, 3), 2, engine=e)
            out[flag]['diffuse'] = sw
        finally:
            e.close()
    a, b = out['0'], out['1']
    assert a['p'] == b['p'] and a['k'] == b['k']
    for key in ('kept', 'nam', 'diffuse'):
        np.testing.assert_array_equal(a[key], b[key])
    # past the Gram matrix the sum over cells runs in a different order: equal to rounding
    for key in ('ncorrs', 'namresid', 'coef', 'coef_fdr', 'fdr'):
        np.testing.assert_allclose(a[key], b[key], rtol=1e-9, atol=1e-12, equal_nan=True)
    # leading PCs only: the trailing ones span the (numerically) null space left by residualisation
    for key in ('U', 'V'):
        np.testing.assert_allclose(a[key][:, :15], b[key][:, :15], rtol=1e-6, atol=1e-9)


def test_large_graph_is_analysed_first_and_reordered_beside_it(monkeypatch):
    """A graph of 100 000 cells or more goes to the device in the caller's order; the cluster order is computed on a
    host thread meanwhile and adopted -- the resident copy renumbered on the device, cna_graph_reorder -- by the first
    later call that finds it done (engine.ensure_graph).  The first result, the adopting call's and a later one's are the same analysis: NAM bit for
    bit, the rest to rounding (the Gram sum runs over the cells in another order); an in-place edit of the matrix
    between the calls is still seen (the order computed from the old content is dropped)."""
    import cna_amd as cna
    from cna_amd import synth, engine as eng_mod
    from cna_amd.engine import Engine
    monkeypatch.setattr(eng_mod, '_REORDER_ASYNC_CELLS', 20000)
    data, meta = synth.make_dataset(30000, 30, k=15, seed=5, cluster_sorted=False)
    e = Engine(device=0)
    e.reuse_nam = False
    try:
        outs = []
        for call in range(3):
            res = cna.tl.association(data, meta['y'], 'id', Nnull=200, seed=3, return_full=True, engine=e)
            outs.append(dict(p=res.p, k=res.k, nam=res.nam.values.copy(), ncorrs=res.ncorrs.values.copy(),
                             fdr=res.fdrs.values.copy(), coef_fdr=data.obs['coef_fdr'].values.copy(),
                             perm=None if e.perm is None else e.perm.copy()))
            if call == 0:
                assert e.perm is None and e.reorder_pending()
                e.wait_reorder()
            else:
                assert e.perm is not None and not e.reorder_pending()
                assert not np.array_equal(e.perm, np.arange(len(e.perm)))
        for o in outs[1:]:
            assert o['p'] == outs[0]['p'] and o['k'] == outs[0]['k']
            np.testing.assert_array_equal(o['nam'], outs[0]['nam'])
            for key in ('ncorrs', 'fdr', 'coef_fdr'):
                np.testing.assert_allclose(o[key], outs[0][key], rtol=1e-9, atol=1e-12, equal_nan=True)
        for key in ('ncorrs', 'fdr', 'coef_fdr', 'nam'):
            np.testing.assert_array_equal(outs[2][key], outs[1][key])          # steady state: the same call twice
        # a new graph, edited in place while its order is being computed: the stale order must not be adopted
        data2, meta2 = synth.make_dataset(30000, 30, k=15, seed=6, cluster_sorted=False)
        A = data2.obsp['connectivities']
        r1 = cna.tl.association(data2, meta2['y'], 'id', Nnull=100, seed=3, return_full=True, engine=e)
        nam1 = r1.nam.values.copy()
        assert e.reorder_pending()
        e.wait_reorder()
        A.data[A.nnz // 3:A.nnz // 3 + 1000] *= 0.5             # (outside the windows of the cheap identity probe)
        r2 = cna.tl.association(data2, meta2['y'], 'id', Nnull=100, seed=3, return_full=True, engine=e)
        fresh = Engine(device=0)
        try:
            monkeypatch.setattr(eng_mod, '_REORDER_ASYNC', False)
            nam2 = r2.nam.values.copy()
            r3 = cna.tl.association(data2, meta2['y'], 'id', Nnull=100, seed=3, return_full=True, engine=fresh)
            np.testing.assert_array_equal(nam2, r3.nam.values)
            assert not np.array_equal(nam1, r3.nam.values)
        finally:
            fresh.close()
    finally:
        e.close()


@pytest.mark.parametrize('graph_dtype', [np.float32, np.float64])
def test_graph_reordered_on_the_device_equals_an_upload_in_that_order(monkeypatch, graph_dtype):
    """cna_graph_reorder (the adoption of the device order by a graph that is already resident) against the other way to
    the same state: an upload of the rows renumbered on the host (CNA_REORDER_ASYNC off).  Same permutation, same column
    sums and the same analysis, every field bit for bit; what is not a permutation is refused and leaves the graph alone."""
    import ctypes as C
    import cna_amd as cna
    from cna_amd import synth, engine as eng_mod
    from cna_amd.engine import Engine
    monkeypatch.setattr(eng_mod, '_REORDER_ASYNC_CELLS', 20000)
    data, meta = synth.make_dataset(40000, 110, k=15, seed=8, cluster_sorted=False, n_covs=1)
    data.obsp['connectivities'] = data.obsp['connectivities'].astype(graph_dtype)
    kw = dict(Nnull=150, seed=2, nsteps=3, covs=meta['covs'], return_full=True)

    def fields(res, d):
        return dict(p=res.p, k=int(res.k), nam=res.nam.values.copy(), ncorrs=res.ncorrs.values.copy(), fdr=res.fdrs.values.copy(),
                    namresid=res.namresid.values.copy(), coef=d.obs['coef'].values.copy(), coef_fdr=d.obs['coef_fdr'].values.copy(),
                    nullminps=np.asarray(res.nullminps).copy())
    a, b = Engine(device=0), Engine(device=0)
    a.reuse_nam = b.reuse_nam = False
    try:
        first = fields(cna.tl.association(data, meta['y'], 'id', engine=a, **kw), data)
        assert a.perm is None and a.reorder_pending()
        cs_before = a.fetch_colsums()
        a.wait_reorder()
        class Spy:                                       # counts the uploads the adopting call makes
            def __init__(self, lib):
                self._lib, self.uploads = lib, 0

            def __getattr__(self, k):
                f = getattr(self._lib, k)
                if k != 'cna_graph_upload':
                    return f

                def counted(*x):
                    self.uploads += 1
                    return f(*x)
                return counted
        spy = a.lib = Spy(a.lib)
        second = fields(cna.tl.association(data, meta['y'], 'id', engine=a, **kw), data)
        assert a.perm is not None and spy.uploads == 0                 # adopted without a second upload
        a.lib = spy._lib
        monkeypatch.setattr(eng_mod, '_REORDER_ASYNC', False)
        ref = fields(cna.tl.association(data, meta['y'], 'id', engine=b, **kw), data)
        assert np.array_equal(a.perm, b.perm)
        np.testing.assert_array_equal(a.fetch_colsums(), b.fetch_colsums())
        np.testing.assert_array_equal(a.fetch_colsums(), cs_before)    # (caller's order both times)
        for key in ref:
            np.testing.assert_array_equal(second[key], ref[key], err_msg=key)
        np.testing.assert_array_equal(first['nam'], ref['nam'])        # (before the adoption: the same walk, other Gram order)
        # the entry point itself: refusals
        lib = eng_mod._ffi.load()
        n = len(a.perm)
        assert lib.cna_graph_reorder(a.h, np.arange(n, dtype=np.int64).ctypes.data) != 0       # already has a device order
        c = Engine(device=0)
        try:
            monkeypatch.setattr(eng_mod, '_REORDER_ASYNC', True)
            monkeypatch.setattr(eng_mod, '_REORDER_ASYNC_CELLS', 10 ** 9)
            monkeypatch.setenv('CNA_REORDER', '0')
            c.ensure_graph(data.obsp['connectivities'])
            bad = np.arange(n, dtype=np.int64)
            bad[5] = 7                                                 # two positions name cell 7
            assert lib.cna_graph_reorder(c.h, bad.ctypes.data) != 0 and b'permutation' in lib.cna_last_error()
            bad[5] = n
            assert lib.cna_graph_reorder(c.h, bad.ctypes.data) != 0
            c.colsums(1)
            np.testing.assert_array_equal(c.fetch_colsums(), cs_before)      # the graph is as it was
            rs = np.random.RandomState(0)
            perm = rs.permutation(n).astype(np.int64)
            assert lib.cna_graph_reorder(c.h, perm.ctypes.data) == 0
            c.perm = perm
            np.testing.assert_array_equal(c.fetch_colsums(), cs_before)      # permuted with the graph, back in the caller's order
        finally:
            c.close()
    finally:
        a.close()
        b.close()


def test_ordered_fetch_on_device_equals_host_reorder(monkeypatch):
    """nam / namresid / V leave the device already restricted to the kept cells and samples, in the
    caller's order and layout (cna_fetch_rows); the sharded path reorders on the host instead.  Both
    must hand back the very same arrays, and bad row / column indices are refused."""
    import cna_amd as cna
    from cna_amd import synth, _order, _ffi
    from cna_amd.engine import Engine
    data, meta = synth.make_dataset(20000, 30, k=15, seed=11, n_covs=1, n_batches=10, cluster_sorted=False)
    # one cluster is populated by batch-0 samples only: its neighbourhoods fail the batch-kurtosis QC
    sid = np.asarray(data.obs['id']).copy()
    target = np.flatnonzero(meta['cluster'] == np.bincount(meta['cluster']).argmax())
    sid[target] = np.random.RandomState(3).choice(np.flatnonzero(meta['batches'].values == 0), size=len(target))
    data.obs['id'] = sid
    y = meta['y'].copy()
    y.iloc[[3, 17]] = np.nan                                # drops two samples: a real column map
    kw = dict(covs=meta['covs'], batches=meta['batches'], Nnull=100, seed=2, return_full=True)
    out = []
    for device_side in (True, False):
        if not device_side:
            monkeypatch.setattr(_order.CellOrder, '_on_device', lambda self: False)
        e = Engine(device=0)
        try:
            res = cna.tl.association(data, y, 'id', engine=e, **kw)
            assert e.perm is not None and not res.kept.all()
            out.append((res.nam.values.copy(), res.namresid.values.copy(), res.namresid_nbhdXpc.values.copy()))
            frame, keep = cna.tl.nam(data, 'id', batches=meta['batches'], engine=e)
            assert frame.shape == (30, keep.sum()) and (frame.columns == data.obs.index[keep]).all()
            out[-1] += (frame.values.copy(),)
            if device_side:
                n = e.matrix_shape(_ffi.MAT_NAM)[0]
                for rows, cols in ((np.array([0, n]), None), (np.array([-1]), None), (None, np.array([0, 30]))):
                    with pytest.raises(RuntimeError):
                        e.fetch_rows(_ffi.MAT_NAM, rows, cols)
                full = e.fetch_matrix(_ffi.MAT_NAM)
                pick = np.array([5, 0, n - 1, 5])
                np.testing.assert_array_equal(e.fetch_rows(_ffi.MAT_NAM, pick, np.array([2, 1])), full[pick][:, [2, 1]])
                np.testing.assert_array_equal(e.fetch_rows(_ffi.MAT_NAM, None, None, True), full.T)
        finally:
            e.close()
    assert out[0][0].shape[0] == 28
    for a, b in zip(*out):
        assert a.shape == b.shape
        np.testing.assert_array_equal(a, b)


def test_fdr_column_copied_by_the_helper_thread_equals_the_late_copy(eng, monkeypatch, general_path):
    """Large-input schedule: the FDR column follows the local null on the device and the helper thread copies it
    into data.obs[key + '_fdr']'s storage while the main thread is in the SVD (cna_percell_fdr_copy_early).  Same
    column, bit for bit, as with the copy at the end of the call; and the early path is really the one taken."""
    import cna_amd as cna
    from cna_amd import synth
    from cna_amd.tools import _association as A
    monkeypatch.setattr(A, '_COEF_FIRST_CELLS', 0)
    data, meta = synth.make_dataset(6000, 24, k=15, seed=5)
    kw = dict(Nnull=200, seed=1, nsteps=3)
    taken = []
    real = eng.percell_fdr_copied_early
    monkeypatch.setattr(eng, 'percell_fdr_copied_early', lambda: taken.append(real()) or taken[-1])
    out = {}
    for early in (True, False, True):
        monkeypatch.setattr(A, '_EARLY_FDR', early)
        for key in ('coef', 'coef_fdr'):
            if key in data.obs:
                del data.obs[key]
        res = cna.tl.association(data, meta['y'], 'id', engine=eng, return_full=True, **kw)
        out.setdefault(early, []).append((res.p, data.obs['coef'].values.copy(), data.obs['coef_fdr'].values.copy(),
                                          res.fdrs.fdr.values.copy()))
    assert taken == [True, True], taken                    # asked twice (the early runs), served from the helper's copy
    p0, c0, f0, t0 = out[False][0]
    assert (f0 < 1).any() and np.isfinite(f0).all()
    for p1, c1, f1, t1 in out[True]:
        assert p1 == p0
        np.testing.assert_array_equal(c1, c0)
        np.testing.assert_array_equal(f1, f0)
        np.testing.assert_array_equal(t1, t0)


@pytest.mark.parametrize('big_path', [False, True])
def test_late_exception_leaves_obs_untouched(eng, monkeypatch, big_path):
    """An exception AFTER the association test has returned -- here the 'already exists' warning of
    _association.py:229 turned into an error (python -W error) -- finds data.obs as the caller left it: the
    columns written early (coefficients under the null kernel; with big_path also the FDR column's storage, which
    the helper thread may still be filling) are put back, and the helper is done with the storage."""
    import warnings
    import cna_amd as cna
    from cna_amd import synth
    from cna_amd.tools import _association as A
    if big_path:
        monkeypatch.setattr(A, '_COEF_FIRST_CELLS', 0)
    data, meta = synth.make_dataset(5000, 24, k=15, seed=3)
    kw = dict(Nnull=100, seed=1, nsteps=3)
    p1 = cna.tl.association(data, meta['y'], 'id', engine=eng, **kw)
    before = data.obs['coef'].values.copy()
    before_fdr = data.obs['coef_fdr'].values.copy()
    y2 = pd.Series(np.random.RandomState(2).randn(24), index=meta['y'].index)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        warnings.filterwarnings('error', message="Key '.*' already exists")
        with pytest.raises(UserWarning, match='already exists'):
            cna.tl.association(data, y2, 'id', engine=eng, **kw)
    np.testing.assert_array_equal(data.obs['coef'].values, before)
    np.testing.assert_array_equal(data.obs['coef_fdr'].values, before_fdr)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        assert cna.tl.association(data, meta['y'], 'id', engine=eng, **kw) == p1
    np.testing.assert_array_equal(data.obs['coef'].values, before)
    np.testing.assert_array_equal(data.obs['coef_fdr'].values, before_fdr)


@pytest.mark.parametrize('big_path', [False, True])
def test_failed_test_leaves_obs_untouched(eng, monkeypatch, big_path, general_path):
    """(big_path: the schedule of large inputs -- coefficient column under the Gram kernels, the FDR column's
    storage made early and filled in place at the end -- forced on this small dataset.)
    data.obs[key_added] is written early (between the two halves of the F-test call, under the
    local-null kernel).  When the
    association test then fails, the column is put back -- absent if it was absent, the old values if
    it existed -- and the next call works: like upstream, an exception leaves data.obs alone."""
    import cna_amd as cna
    from cna_amd import synth
    from cna_amd.tools import _association as A
    assert A._EARLY_COEF
    if big_path:
        monkeypatch.setattr(A, '_COEF_FIRST_CELLS', 0)
    data, meta = synth.make_dataset(5000, 24, k=15, seed=3)
    kw = dict(Nnull=100, seed=1, nsteps=3)

    real = eng.global_test_fetch

    def boom(*a, **k):                       # fails after the coefficient column has been written
        real()
        assert 'coef' in data.obs
        raise FloatingPointError('injected')
    monkeypatch.setattr(eng, 'global_test_fetch', boom)
    with pytest.raises(FloatingPointError):
        cna.tl.association(data, meta['y'], 'id', engine=eng, **kw)
    assert 'coef' not in data.obs and 'coef_fdr' not in data.obs
    monkeypatch.setattr(eng, 'global_test_fetch', real)
    p1 = cna.tl.association(data, meta['y'], 'id', engine=eng, **kw)
    before = data.obs['coef'].values.copy()
    before_fdr = data.obs['coef_fdr'].values.copy()
    assert (before_fdr < 1).any() and (before_fdr == 1).any()
    y2 = pd.Series(np.random.RandomState(2).randn(24), index=meta['y'].index)
    monkeypatch.setattr(eng, 'global_test_fetch', boom)
    with pytest.raises(FloatingPointError):
        cna.tl.association(data, y2, 'id', engine=eng, **kw)
    np.testing.assert_array_equal(data.obs['coef'].values, before)
    np.testing.assert_array_equal(data.obs['coef_fdr'].values, before_fdr)
    monkeypatch.setattr(eng, 'global_test_fetch', real)
    assert cna.tl.association(data, meta['y'], 'id', engine=eng, **kw) == p1
    np.testing.assert_array_equal(data.obs['coef'].values, before)
    np.testing.assert_array_equal(data.obs['coef_fdr'].values, before_fdr)
    # the early / inline path (coefficients ahead of the null, FDR column queued behind it with the
    # FDR table formed on the device) and the one-shot path write the same two columns, bit for bit
    monkeypatch.setattr(A, '_EARLY_COEF', False)
    assert cna.tl.association(data, meta['y'], 'id', engine=eng, **kw) == p1
    np.testing.assert_array_equal(data.obs['coef'].values, before)
    np.testing.assert_array_equal(data.obs['coef_fdr'].values, before_fdr)


def test_fused_selection_call_equals_separate_calls(eng, monkeypatch):
    """cna_select_standardized_fused queues the Gram kernels, the local null's thresholds (numpy's
    arange / edge arithmetic restated in C), the threshold-only half of the null pass and the early
    coefficient column inside the selection call.  Results must be those of the separate calls, bit
    for bit, and the C thresholds must be numpy's for any max|ncorrs|."""
    import ctypes as C
    import cna_amd as cna
    from cna_amd import synth
    from cna_amd.tools import _association as A
    rs = np.random.RandomState(0)
    for trial in range(3000):
        m = float(np.exp(rs.uniform(np.log(1e-4), np.log(5))))
        maxcorr = max(m, 0.001)
        thr = np.arange(maxcorr / 4, maxcorr, maxcorr / 400)
        edges = thr ** 2 - 1e-8 - 1e-5 * thr ** 2
        t, e = np.empty(512), np.empty(512)
        T = eng.lib.cna_reference_thresholds(m, 512, t.ctypes.data, e.ctypes.data)
        assert T == len(thr) and np.array_equal(t[:T], thr) and np.array_equal(e[:T], edges), m
    data, meta = synth.make_dataset(20000, 30, k=15, seed=8)
    out = {}
    for fuse in (True, False):
        monkeypatch.setattr(A, '_FUSE', fuse)
        eng.prof_reset(); eng.prof_enable(True)
        res = cna.tl.association(data, meta['y'], 'id', Nnull=300, seed=5, nsteps=3, return_full=True, engine=eng)
        eng.sync(); eng.prof_enable(False)
        out[fuse] = (res.p, int(res.k), res.ncorrs.values.copy(), res.fdrs.values.copy(), data.obs['coef'].values.copy(),
                     data.obs['coef_fdr'].values.copy(), res.namresid_sampleXpc.values.copy(), eng.prof()['gram'][1])
    assert out[True][0] == out[False][0] and out[True][1] == out[False][1]
    assert out[True][7] == out[False][7] == 1                 # the Gram kernels ran once either way
    for a, b in zip(out[True][2:7], out[False][2:7]):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize('n,N,nsteps,ncov', [(20011, 200, 3, 0), (5000, 100, 3, 0), (9999, 233, 2, 0), (3001, 66, 4, 0),
                                             (2500, 300, 3, 0), (40, 180, 3, 0), (4000, 64, 3, 0), (7001, 200, 3, 5),
                                             (3000, 90, 5, 2)])
def test_selection_as_a_by_product_of_the_last_walk_step(eng, monkeypatch, n, N, nsteps, ncov):
    """More than 64 samples, every cell and sample kept, nothing regressed out: the last step of the walk also does the
    selection pass (_association.py:182 zero-variance count, _nam.py:122,159 centre and / std, _association.py:77
    coefficients, the digit planes of the integer local null) on the row it has just formed -- diffuse.hip:select_tail,
    armed by cna_nam_select_hint -- and cna_select_standardized finds its work done.  The NAM is the same bit for bit;
    everything downstream agrees with the separate pass to rounding (the row sums run over the lanes in another order)
    and the integer results are identical.  With covariates the selection keeps its own pass (it applies the projector).
    40 cells (fewer than 65 samples have any), 64 samples (two rows per wave:
    no by-product) and a two-step walk (the last step is not held back) exercise the separate pass."""
    import cna_amd as cna
    from cna_amd import synth
    from cna_amd.tools import _association as A
    monkeypatch.setattr(A, '_DEFER_LAST_CELLS', 0)         # (the schedule of large inputs)
    data, meta = synth.make_dataset(n, N, k=15, seed=21, n_covs=ncov)
    out = {}
    for byp in (True, False):
        monkeypatch.setenv('CNA_WALK_SELECT', '1' if byp else '0')
        eng.prof_reset(); eng.prof_enable(True)
        res = cna.tl.association(data, meta['y'], 'id', covs=meta['covs'] if ncov else None, Nnull=200, seed=5,
                                 nsteps=nsteps, return_full=True, engine=eng)
        eng.sync(); eng.prof_enable(False)
        out[byp] = (res.p, int(res.k), res.nam.values.copy(), res.kept.copy(), res.fdrs.num_detected.values.copy(),
                    res.ncorrs.values.copy(), res.namresid.values.copy(), res.fdrs.fdr.values.copy(),
                    data.obs['coef'].values.copy(), data.obs['coef_fdr'].values.copy(), res.nullminps.copy(),
                    eng.gram_fetch().copy())
        out[byp, 'select launches'] = eng.prof().get('select', (0, 0))[1]
    expect = data.obs['id'].nunique() > 64 and nsteps >= 3 and not ncov      # (samples without cells are not part of the NAM)
    assert out[False, 'select launches'] == 1 and out[True, 'select launches'] == (0 if expect else 1)
    assert out[True][1] == out[False][1]
    # (np.arange(m/4, m, m/400) has 300 or 301 entries depending on the last bits of m: compare the common prefix)
    T = min(len(out[True][4]), len(out[False][4]))
    np.testing.assert_array_equal(out[True][2], out[False][2])                 # NAM
    np.testing.assert_array_equal(out[True][3], out[False][3])                 # kept
    np.testing.assert_array_equal(out[True][4][:T], out[False][4][:T])         # num_detected
    assert out[True][0] == pytest.approx(out[False][0], rel=1e-9)
    for i, (a, b) in enumerate(zip(out[True][5:], out[False][5:])):
        if i == 2:
            a, b = a[:T], b[:T]
        np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize('n,N,K,nsteps', [(70001, 200, 4, 3), (66001, 100, 2, 3), (41000, 180, 2, 3), (30000, 200, 4, 3)])
def test_gram_under_the_walks_last_step(eng, monkeypatch, n, N, K, nsteps):
    """NAM.dot(NAM.T) (_nam.py:105) is a sum over cells, and the walk's last step writes the standardised rows one by one
    (select_tail): that step runs in K row ranges and the Gram kernel of each range follows it on a second stream, carrying
    its partial tiles from range to range (c_api.hip:ranged_last_step, mfma.hip:launch_gram_range).  Every workgroup adds
    the same slabs in the same order as the one-launch kernel, so the matrix -- and with it every result -- is the same
    bit for bit.  The 3 x 3-block kernel (200 / 180 samples) and the strided one (100), a dense and a compressed last
    step, a ragged last range; 30 000 cells are too few for four ranges (one launch, Gram afterwards)."""
    import cna_amd as cna
    from cna_amd import synth
    from cna_amd.tools import _association as A
    monkeypatch.setattr(A, '_DEFER_LAST_CELLS', 0)
    data, meta = synth.make_dataset(n, N, k=15, seed=33)
    out = {}
    monkeypatch.setattr(eng, 'reuse_nam', False)          # both runs walk
    for ranged in (True, False):
        monkeypatch.setenv('CNA_GRAM_OVERLAP', str(K) if ranged else '0')
        eng.prof_reset(); eng.prof_enable(True)
        res = cna.tl.association(data, meta['y'], 'id', Nnull=200, seed=5, nsteps=nsteps, return_full=True, engine=eng)
        eng.sync(); eng.prof_enable(False)
        prof = eng.prof()
        out[ranged] = (res.p, int(res.k), eng.gram_fetch().copy(), res.ncorrs.values.copy(), res.namresid.values.copy(),
                       res.fdrs.values.copy(), res.nullminps.copy(), res.namresid_sampleXpc.values.copy(),
                       data.obs['coef_fdr'].values.copy(), res.nam.values.copy())
        out[ranged, 'gram launches'] = prof.get('gram', (0, 0))[1]
        assert prof.get('select', (0, 0))[1] == 0           # the by-product in both runs
    assert out[False, 'gram launches'] == 1
    assert out[True, 'gram launches'] == (K if n > 40000 else 1)
    assert out[True][0] == out[False][0] and out[True][1] == out[False][1]
    for a, b in zip(out[True][2:], out[False][2:]):
        np.testing.assert_array_equal(a, b)
    G = out[True][2]
    X = out[True][4]                                      # samples x cells
    np.testing.assert_allclose(G, X @ X.T, rtol=1e-10, atol=1e-7)


@pytest.mark.parametrize('n,N', [(20011, 200), (5000, 170), (9999, 233), (40, 180)])
def test_selection_and_gram_in_one_kernel(eng, monkeypatch, n, N):
    """161 ... 240 samples, all cells kept, samples in place, nothing regressed out: the selection pass
    (_nam.py:122,159: centre, / std; the coefficients of _association.py:77; the digit planes of the integer local
    null) and the Gram matrix (_nam.py:105) can be ONE kernel (mfma.hip:k_selgram_blk, CNA_SELGRAM=1; off by default:
    slower than the two kernels, see the note at gram_fused_ok).  Everything it leaves behind -- X, the coefficients, the
    Gram matrix, and through them every result field -- equals the separate kernels bit for bit; odd cell counts
    exercise the partial last slab, 40 cells (most samples empty, so the selection is not "in place") the fall-back."""
    import cna_amd as cna
    from cna_amd import synth
    from cna_amd._ffi import MAT_X
    data, meta = synth.make_dataset(n, N, k=15, seed=11)
    out = {}
    monkeypatch.setenv('CNA_WALK_SELECT', '0')        # (the selection pass as its own launch: what this test compares)
    for fused in (True, False):
        if fused:
            monkeypatch.setenv('CNA_SELGRAM', '1')
        else:
            monkeypatch.delenv('CNA_SELGRAM', raising=False)
        eng.prof_reset(); eng.prof_enable(True)
        res = cna.tl.association(data, meta['y'], 'id', Nnull=200, seed=5, nsteps=3, return_full=True, engine=eng)
        eng.sync(); eng.prof_enable(False)
        prof = eng.prof()
        out[fused] = (res.p, int(res.k), res.ncorrs.values.copy(), res.fdrs.values.copy(), data.obs['coef'].values.copy(),
                      data.obs['coef_fdr'].values.copy(), res.namresid_sampleXpc.values.copy(), res.namresid.values.copy(),
                      eng.gram_fetch().copy(), res.nullminps.copy())
        out[fused, 'select launches'] = prof.get('select', (0, 0))[1]
    assert out[False, 'select launches'] == 1
    in_place = data.obs['id'].nunique() == N          # (with 40 cells most samples are empty: the selection drops them)
    assert out[True, 'select launches'] == (0 if n >= 32 and in_place else 1)      # the fused kernel is booked under 'gram'
    assert out[True][0] == out[False][0] and out[True][1] == out[False][1]
    for a, b in zip(out[True][2:], out[False][2:]):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize('n,N,seed', [(30000, 40, 3), (8000, 120, 4), (50001, 24, 5)])
def test_walk_with_the_stop_rule_on_the_device(eng, monkeypatch, n, N, seed):
    """nsteps=None (the reference's default, _nam.py:64-68): cna_nam_auto queues steps ahead of the verdict and
    evaluates np.median(kurtosis) and the rule on the device.  Same number of steps, the same medians and the same
    NAM, bit for bit, as the host loop over cna_nam_step + cna_stat_median (CNA_AUTO_HOST=1) -- also when the rule
    is met before the steps already queued (four are queued at once) and when maxnsteps ends the walk."""
    import cna_amd as cna
    from cna_amd import synth
    from cna_amd.tools._nam import _nam_device, sample_codes
    data, meta = synth.make_dataset(n, N, k=15, seed=seed)
    codes, labels = sample_codes(data.obs['id'])
    counts = np.bincount(codes, minlength=len(labels))
    out = {}
    for host in (False, True):
        monkeypatch.setenv('CNA_AUTO_HOST', '1' if host else '0')
        for maxn in (15, 3, 2):
            _, taken = _nam_device(eng, data, 'id', nsteps=None, maxnsteps=maxn, codes_labels=(codes, labels, counts))
            if taken is None:                     # queued, verdict pending: read the step count (nam_full would collect it too)
                taken, _ = eng.nam_auto_finish()
            out[host, maxn] = (taken, eng.nam_full().copy())
    for maxn in (15, 3, 2):
        assert out[False, maxn][0] == out[True, maxn][0], maxn
        np.testing.assert_array_equal(out[False, maxn][1], out[True, maxn][1])
    assert 3 <= out[True, 15][0] <= 15 and out[True, 2][0] == 2
    # the medians themselves
    monkeypatch.setenv('CNA_AUTO_HOST', '0')
    eng.set_samples(codes, len(labels), counts.astype(float))
    taken, med = eng.nam_auto(15)
    eng.set_samples(codes, len(labels), counts.astype(float))
    want = []
    for i in range(taken):
        eng.nam_step(True, True, True)
        want.append(eng.stat_median())
    assert np.isnan(med[0])                        # the rule never reads the first step's median: not computed
    np.testing.assert_array_equal(med[1:], want[1:])
    # a pending walk is finished by whatever reads the NAM next
    eng.set_samples(codes, len(labels), counts.astype(float))
    eng.nam_auto_launch(15)
    np.testing.assert_array_equal(eng.nam_full(), out[True, 15][1])
    # and through the public call
    res = {}
    for host in ('0', '1'):
        monkeypatch.setenv('CNA_AUTO_HOST', host)
        r = cna.tl.association(data, meta['y'], 'id', Nnull=100, seed=1, return_full=True, engine=eng)
        res[host] = (r.p, int(r.k), r.ncorrs.values.copy(), r.nam.values.copy())
    assert res['0'][:2] == res['1'][:2]
    np.testing.assert_array_equal(res['0'][2], res['1'][2])
    np.testing.assert_array_equal(res['0'][3], res['1'][3])


@pytest.mark.parametrize('n,N,nb,nc', [(20000, 40, 5, 2), (9000, 100, 4, 0), (15000, 64, 12, 1)])
def test_covariates_and_batches_fast_path(eng, monkeypatch, n, N, nb, nc):
    """The demo's call shape (covariates AND batches, demo/demo.ipynb:149; _nam.py:85-99,136-156): the QC decision
    (median, threshold, count of failing cells) is taken on the device, and the first ridge of the schedule is one
    pass (residualise + batch kurtosis + device median + / std + coefficients).  Same results as the step-by-step
    path -- QC vector fetched and compared on the host, ridge by ridge with separate kurtosis / median / std /
    coefficient passes."""
    import cna_amd as cna
    from cna_amd import synth
    data, meta = synth.make_dataset(n, N, k=15, seed=21, n_covs=nc, n_batches=nb)
    kw = dict(covs=meta['covs'], batches=meta['batches'], Nnull=200, seed=3, nsteps=3, return_full=True, engine=eng)
    fast = cna.tl.association(data, meta['y'], 'id', **kw)
    f = (fast.p, int(fast.k), fast.ncorrs.values.copy(), fast.kept.copy(), fast.namresid.values.copy(),
         fast.fdrs.values.copy(), data.obs['coef_fdr'].values.copy(), fast.M.values.copy())
    monkeypatch.setenv('CNA_RIDGE_ONEPASS', '0')
    monkeypatch.setenv('CNA_MEDIAN_HOST', '1')
    monkeypatch.delattr(type(eng), 'stat_qc')
    slow = cna.tl.association(data, meta['y'], 'id', **kw)
    assert f[0] == slow.p and f[1] == int(slow.k)
    assert np.array_equal(f[3], slow.kept)
    np.testing.assert_allclose(f[2], slow.ncorrs.values, rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(f[4], slow.namresid.values, rtol=1e-12, atol=1e-14)
    np.testing.assert_array_equal(f[5][:, 2], slow.fdrs.values[:, 2])             # num_detected
    np.testing.assert_allclose(f[5], slow.fdrs.values, rtol=1e-9)
    np.testing.assert_allclose(f[6], data.obs['coef_fdr'].values, rtol=1e-9)
    np.testing.assert_array_equal(f[7], slow.M.values)


@pytest.mark.parametrize('n,N,nb,nc,isolated', [(20000, 40, 5, 2, False), (9000, 100, 4, 0, False), (30000, 128, 7, 3, False),
                                                (12000, 30, 3, 1, True), (15000, 64, 12, 1, False)])
def test_selection_qc_and_first_ridge_in_one_pass(eng, monkeypatch, n, N, nb, nc, isolated):
    """Round 6: covariates AND batches with every sample analysed in place and at most seven batches -- QC
    (cna_batch_kurtosis + cna_stat_qc), selection (cna_select_checked) and the first ridge (cna_resid_lowrank_bk) are ONE
    pass over the NAM (cna_select_resid_bk; same kernel, same arithmetic: bit-identical results).  `isolated`: a graph with
    a ring of cells that only one sample reaches (rows of the NAM with a single non-zero entry); twelve batches, or more
    projector + batch columns than the matrix instruction's sixteen: not covered, nothing is queued."""
    import cna_amd as cna
    from cna_amd import synth
    data, meta = synth.make_dataset(n, N, k=15, seed=23, n_covs=nc, n_batches=nb)
    y, covs, batches = meta['y'].copy(), meta['covs'], meta['batches']
    if isolated:
        sid = data.obs['id'].values.copy()
        sid[:40] = sid[0]                                     # the first cells and their neighbours: one sample only ...
        A = data.obsp['connectivities'].tolil()
        A[:40, :] = 0
        A[:, :40] = 0
        for i in range(40):
            A[i, (i + 1) % 40] = 1.0
            A[(i + 1) % 40, i] = 1.0
        data.obsp['connectivities'] = A.tocsr().astype(np.float32)
        data.obs['id'] = sid
    kw = dict(covs=covs, batches=batches, Nnull=100, seed=5, nsteps=3, return_full=True, engine=eng)
    calls = []
    real = eng.select_resid_bk

    def spy(*a, **k):
        calls.append(real(*a, **k))
        return calls[-1]
    monkeypatch.setattr(eng, 'select_resid_bk', spy)
    out = {}
    for one in ('1', '0'):
        monkeypatch.setenv('CNA_RIDGE_ONEPASS', one)
        res = cna.tl.association(data, y, 'id', **kw)
        out[one] = (res.p, int(res.k), res.kept.copy(), res.ncorrs.values.copy(), res.namresid.values.copy(), res.fdrs.values.copy(),
                    data.obs['coef'].values.copy(), data.obs['coef_fdr'].values.copy(), res.M.values.copy(), res.nullminps.copy())
    if nb > 7:
        assert calls == []                                    # (more than seven batches: never asked)
    elif 2 * nb + nc > 16:
        assert calls == [None]                                # (asked once, with the switch on: not covered)
    else:
        assert len(calls) == 1 and calls[0] is not None and calls[0][0] == 0 and calls[0][1] == 0 and calls[0][3] <= 6, calls
    assert out['1'][:2] == out['0'][:2]
    for a, b in zip(out['1'][2:], out['0'][2:]):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize('n,N', [(5000, 96), (7001, 100), (3000, 112), (4100, 128), (2500, 144), (2049, 160), (3000, 80)])
def test_gram_in_three_by_three_tile_blocks_at_81_to_160_samples(eng, monkeypatch, n, N):
    """Round 6: X^T X with a 3 x 3 block of 16 x 16 tiles per wave also for 6 ... 10 tiles per side (k_gram_blk on 4 / 8 / 12
    waves; it served 161 ... 240 samples before) against the tile-per-wave kernel (CNA_GRAM_BLK_SMALL=0) and numpy."""
    X = np.random.RandomState(N).randn(n, N)
    X -= X.mean(axis=1, keepdims=True)
    out = {}
    for sw in ('1', '0'):
        monkeypatch.setenv('CNA_GRAM_BLK_SMALL', sw)
        eng.upload_x(X)
        out[sw] = eng.gram()
    want = X.T @ X
    for G in out.values():
        np.testing.assert_array_equal(G, G.T)
        np.testing.assert_allclose(G, want, rtol=0, atol=1e-12 * np.abs(want).max())
    np.testing.assert_allclose(out['1'], out['0'], rtol=0, atol=1e-13 * np.abs(want).max())


def test_analysis_with_the_f64_local_null_kernel(eng, monkeypatch):
    """CNA_NULL_F64=1 (README: the documented way to take the integer matrix cores out of the local null): the f64 kernel
    counts the same integers, so every result field is the same."""
    import cna_amd as cna
    from cna_amd import synth
    data, meta = synth.make_dataset(9000, 48, k=15, seed=31)
    kw = dict(nsteps=3, Nnull=200, seed=2, return_full=True, engine=eng)
    out = {}
    for sw in ('0', '1', '0'):
        monkeypatch.setenv('CNA_NULL_F64', sw)
        res = cna.tl.association(data, meta['y'], 'id', **kw)
        used = eng.null_local_i8_stats()[0]
        assert used == (sw == '0')
        out.setdefault(sw, []).append((res.p, res.fdrs.values.copy(), data.obs['coef_fdr'].values.copy(), res.ncorrs.values.copy()))
    a, b = out['0'][0], out['1'][0]
    assert a[0] == b[0]
    for x, y_ in zip(a[1:], b[1:]):
        np.testing.assert_array_equal(x, y_)


def test_nam_cache_on_device(eng):
    """A second phenotype on the same dataset reuses the resident NAM (no diffusion kernels) and gives
    the results of a from-scratch run."""
    import cna_amd as cna
    from cna_amd import synth
    from cna_amd.engine import Engine
    data, meta = synth.make_dataset(30000, 40, k=15, seed=2, n_covs=1)
    y2 = pd.Series(np.random.RandomState(9).randn(40), index=meta['y'].index)
    kw = dict(covs=meta['covs'], nsteps=3, Nnull=200, seed=4, return_full=True)
    e = Engine(device=0)
    try:
        assert e.reuse_nam
        cna.tl.association(data, meta['y'], 'id', engine=e, **kw)
        e.prof_reset(); e.prof_enable(True)
        r2 = cna.tl.association(data, y2, 'id', engine=e, **kw)
        e.sync(); e.prof_enable(False)
        assert not any(k.startswith('nam_') for k in e.prof()), e.prof().keys()
        nam2 = r2.nam.values.copy()
        e.reuse_nam = False
        e.prof_reset(); e.prof_enable(True)
        r3 = cna.tl.association(data, y2, 'id', engine=e, **kw)
        e.sync(); e.prof_enable(False)
        assert e.prof()['nam_step'][1] == 2 and e.prof()['nam_first'][1] == 1
        assert r2.p == r3.p and r2.k == r3.k
        np.testing.assert_array_equal(r2.ncorrs.values, r3.ncorrs.values)
        np.testing.assert_array_equal(r2.fdrs.values, r3.fdrs.values)
        np.testing.assert_array_equal(nam2, r3.nam.values)
    finally:
        e.close()


def test_nam_cache_after_a_walk_that_left_the_selection_instead_of_the_nam(monkeypatch):
    """The schedule of large inputs: the walk's last step does the selection pass and does not write the raw NAM
    (cna_nam_select_hint).  res.nam of that call is one more run of the last step (c_api.hip:need_nam).  A second
    phenotype on the same dataset skips the walk (NAM cache) AND the selection pass -- the standardised NAM does not depend
    on the phenotype and is still on the device (cna_x_identity): only its coefficients are taken -- and gives what a
    from-scratch run gives (the NAM bit for bit)."""
    import cna_amd as cna
    from cna_amd import synth
    from cna_amd.engine import Engine
    from cna_amd.tools import _association as A
    monkeypatch.setattr(A, '_DEFER_LAST_CELLS', 0)
    data, meta = synth.make_dataset(20000, 100, k=15, seed=3)          # (96 samples or more: the second step is the compressed one)
    y2 = pd.Series(np.random.RandomState(9).randn(100), index=meta['y'].index)
    kw = dict(nsteps=3, Nnull=200, seed=4, return_full=True)
    e = Engine(device=0)
    try:
        assert e.reuse_nam and not e.reuse_x
        e.reuse_x = True
        e.prof_reset(); e.prof_enable(True)
        r1 = cna.tl.association(data, meta['y'], 'id', engine=e, **kw)
        e.sync()
        assert 'select' not in e.prof() and e.prof()['nam_step'][1] == 1          # by-product; NAM not materialised yet
        nam1 = r1.nam.values.copy()                                                # ... now it is (one more dense launch)
        e.sync()
        assert e.prof()['nam_step'][1] == 2
        e.prof_reset()
        r2 = cna.tl.association(data, y2, 'id', engine=e, **kw)
        e.sync(); e.prof_enable(False)
        # resident NAM, and the standardised NAM of the same selection is still there too: only new coefficients
        assert not any(k.startswith('nam_') for k in e.prof()), e.prof().keys()
        assert 'select' not in e.prof() and e.prof()['ncorrs'][1] == 1
        nam2 = r2.nam.values.copy()
        e.reuse_nam = False
        r3 = cna.tl.association(data, y2, 'id', engine=e, **kw)
        # (the coefficients come from another kernel than in a from-scratch run: same X, another order of the row sum)
        assert r2.k == r3.k and r2.p == pytest.approx(r3.p, rel=1e-9)
        np.testing.assert_allclose(r2.ncorrs.values, r3.ncorrs.values, rtol=1e-11, atol=1e-15)
        T = min(len(r2.fdrs), len(r3.fdrs))
        np.testing.assert_array_equal(r2.fdrs.num_detected.values[:T], r3.fdrs.num_detected.values[:T])
        np.testing.assert_allclose(r2.fdrs.fdr.values[:T], r3.fdrs.fdr.values[:T], rtol=1e-9, atol=1e-13)
        np.testing.assert_array_equal(nam2, r3.nam.values)
        np.testing.assert_array_equal(nam1, nam2)
    finally:
        e.close()


def test_properties_at_baseline_config2_size(eng):
    """BASELINE.json configs[1] (200k cells x 50 samples, k=30, nsteps=3, Nnull=1000) through
    size-independent properties: oracle-free identities on the fetched matrices, the per-cell columns
    recomputed on the host from the returned tables, and invariance under a random renumbering of the
    cells (same graph, same samples, cells shuffled) -- the global p-value, the chosen k, the FDR
    table and every cell's coefficient must not move."""
    import cna_amd as cna
    from cna_amd import synth
    data, meta = synth.make_dataset(200000, 50, k=30, seed=0)
    y = meta['y']
    kw = dict(nsteps=3, Nnull=1000, seed=0, return_full=True)
    res = cna.tl.association(data, y, 'id', **kw)
    nam = res.nam.values                                               # samples x cells
    np.testing.assert_allclose(nam.sum(axis=1), 1.0, rtol=1e-9)        # column-stochastic walk
    X = res.namresid.values
    assert np.abs(X.mean(axis=0)).max() < 1e-10
    np.testing.assert_allclose(X.std(axis=0, ddof=1), 1.0, rtol=1e-10)
    yz = (y.values - y.values.mean()) / y.values.std()
    coef = data.obs['coef'].values.copy()
    fdrcol = data.obs['coef_fdr'].values.copy()
    np.testing.assert_allclose(coef, yz.dot(X) / 50, rtol=1e-9, atol=1e-12)
    f = res.fdrs
    thr, fdr, num = f.threshold.values, f.fdr.values, f.num_detected.values
    assert np.array_equal(num, (np.abs(coef)[None, :] > thr[:, None]).sum(axis=1))
    assert (np.diff(num) <= 0).all() and ((fdr >= 0) | np.isnan(fdr)).all()
    runmin = np.fmin.accumulate(fdr)
    idx = np.searchsorted(thr, np.abs(coef), side='right') - 1
    want = np.where(idx >= 0, runmin[np.maximum(idx, 0)], 1.0)
    np.testing.assert_allclose(fdrcol, want, rtol=1e-14)
    assert 1 / 1001 <= res.p <= 1 and len(res.nullminps) == 1000 and res.kept.all()

    # the same analysis with the cells renumbered at random
    rs = np.random.RandomState(123)
    perm = rs.permutation(200000)
    A = sp.csr_matrix(data.obsp['connectivities'])
    Ap = A[perm][:, perm].tocsr()
    Ap.sort_indices()
    obs2 = pd.DataFrame({'id': data.obs['id'].values[perm]}, index=data.obs.index[perm])
    data2 = type('D', (), {'obs': obs2, 'obsp': {'connectivities': Ap}, 'uns': {}})()
    res2 = cna.tl.association(data2, y, 'id', **kw)
    assert res2.p == res.p and res2.k == res.k
    # sums over neighbours run in a different order: equal to rounding, counts identical
    np.testing.assert_allclose(data2.obs['coef'].values, coef[perm], rtol=1e-9, atol=1e-13)
    assert np.array_equal(res2.fdrs.num_detected.values, num)
    np.testing.assert_allclose(res2.fdrs.fdr.values, fdr, rtol=1e-9, equal_nan=True)


@pytest.mark.parametrize('N', [7, 50, 100, 130, 200])
def test_compressed_second_step_matches_dense(monkeypatch, N):
    """The second walk step may gather a compressed copy of the state (k_nam_step_sparse): same
    products and sums in the same order as the dense kernel (unfused multiply and add, LDS f64 adds
    round like VALU adds), so the NAM must be bit-identical, and bit-reproducible from run to run.
    Covers rows with more distinct samples than the compressed form holds (a hub cell with 300
    neighbours), isolated cells and every later step."""
    import cna_amd as cna
    from cna_amd.engine import Engine
    rs = np.random.RandomState(N)
    n = 3000
    A = sp.random(n, n, density=12.0 / n, random_state=rs, format='lil', dtype=np.float64)
    A[17, :] = 0
    A[:, 17] = 0                                   # isolated cell
    hub = rs.choice(n, 300, replace=False)
    A[5, hub] = rs.rand(300)                       # far more than 64 distinct samples when N is large
    A[hub[:40], 5] = rs.rand(40)                   # ... and cells that gather the hub's (dense) row
    A = sp.csr_matrix(A)
    A.setdiag(0)
    A.eliminate_zeros()
    A = A.astype(np.float32)
    obs = pd.DataFrame({'id': rs.randint(0, N, n)})
    obs.loc[:N - 1, 'id'] = np.arange(N)           # every sample present
    data = type('D', (), {'obs': obs, 'obsp': {'connectivities': A}, 'uns': {}})()
    out = {}
    for flag in ('0', '1'):
        monkeypatch.setenv('CNA_SPARSE_MIN_N', flag)
        e = Engine(device=0)
        try:
            for nsteps in (2, 3):
                NAM, keep = cna.tl.nam(data, 'id', nsteps=nsteps, engine=e)
                out[flag, nsteps] = NAM.values.copy()
            e.prof_reset(); e.prof_enable(True)
            NAM, _ = cna.tl.nam(data, 'id', nsteps=None, engine=e)          # auto-stop path, with kurtosis
            e.sync(); e.prof_enable(False)
            out[flag, 'auto'] = NAM.values.copy()
            again, _ = cna.tl.nam(data, 'id', nsteps=3, engine=e)
            np.testing.assert_array_equal(again.values, out[flag, 3])       # deterministic
        finally:
            e.close()
    for key in (2, 3, 'auto'):
        np.testing.assert_array_equal(out['1', key], out['0', key])


@pytest.mark.parametrize('N,n', [(50, 3001), (64, 3000), (33, 2999), (7, 500)])
def test_two_rows_per_wave_step_matches_wave_per_row(monkeypatch, N, n):
    """States of at most 64 columns are walked two destination rows per wave (k_nam_step_pair, one
    load instruction gathers a neighbour row for each).  Each row still adds its products in CSR
    order, so NAM, kurtosis-driven auto-stop and the dense walk are bit-identical to the
    wave-per-row kernel (CNA_STEP_WIDE=1).  Odd row counts, an isolated cell, a 300-neighbour hub
    next to a 1-neighbour row, and a run of empty rows exercise the ragged pairing."""
    import cna_amd as cna
    from cna_amd.engine import Engine
    rs = np.random.RandomState(N)
    A = sp.random(n, n, density=12.0 / n, random_state=rs, format='lil', dtype=np.float64)
    A[17, :] = 0
    A[:, 17] = 0
    for r in range(40, 47):
        A[r, :] = 0                                # empty rows (their columns stay referenced)
    hub = rs.choice(n, min(300, n // 2), replace=False)
    A[4, hub] = rs.rand(len(hub))                  # rows 4 and 5 share a wave when no reordering happens
    A[5, :] = 0
    A[5, 9] = 0.5
    A = sp.csr_matrix(A)
    A.setdiag(0)
    A.eliminate_zeros()
    A = A.astype(np.float32)
    obs = pd.DataFrame({'id': rs.randint(0, N, n)})
    obs.loc[:N - 1, 'id'] = np.arange(N)
    data = type('D', (), {'obs': obs, 'obsp': {'connectivities': A}, 'uns': {}})()
    dense0 = rs.randn(n, 3)
    out = {}
    for wide in ('', '1'):
        if wide:
            monkeypatch.setenv('CNA_STEP_WIDE', '1')
        for reorder in ('0', '1'):
            monkeypatch.setenv('CNA_REORDER', reorder)
            e = Engine(device=0)
            try:
                NAM, _ = cna.tl.nam(data, 'id', nsteps=3, engine=e)
                e.prof_reset(); e.prof_enable(True)
                auto, _ = cna.tl.nam(data, 'id', nsteps=None, engine=e)
                e.sync(); e.prof_enable(False)
                out[wide, reorder] = (NAM.values.copy(), auto.values.copy(), e.prof()['nam_step'][1],
                                      cna.tl.diffuse(data, dense0, 3, engine=e))
            finally:
                e.close()
    for reorder in ('0', '1'):
        a, b = out['', reorder], out['1', reorder]
        assert a[2] == b[2]                        # auto-stop took the same number of steps
        for u, v in zip((a[0], a[1], a[3]), (b[0], b[1], b[3])):
            np.testing.assert_array_equal(u, v)
    np.testing.assert_array_equal(out['', '0'][0], out['', '1'][0])


def test_prepared_null_launch_and_misuse(eng):
    """cna_null_local_prepare + launch(edges=None) equals the one-call launch; launching without a
    prepared pass, or with a shape that does not match it, is an error and leaves nothing pending."""
    from cna_amd._ffi import CnaHipError
    rs = np.random.RandomState(4)
    n, N, P = 4000, 30, 90
    X = _random_x(rs, n, N)
    eng.upload_x(X)
    eng.standardize(center=True)
    y = rs.randn(N)
    _, m = eng.ncorrs(y)
    Y = rs.randn(N, P + 1)
    eng.condition(np.eye(N), Y)
    thr = np.arange(m / 4, m, m / 400)
    edges = thr ** 2 - 1e-8 - 1e-5 * thr ** 2
    eng.null_local_launch(1, P, edges, thr)
    a = eng.null_local_fetch()
    eng.null_local_prepare(P, edges, thr)
    eng.null_local_launch(1, P, None)
    b = eng.null_local_fetch()
    for u, v in zip(a, b):
        np.testing.assert_array_equal(u, v)
    ranks, numdet = eng.obs_counts(edges, thr)
    np.testing.assert_array_equal(a[1], ranks)
    np.testing.assert_array_equal(a[2], numdet)
    with pytest.raises(CnaHipError, match='prepare'):
        eng.null_local_launch(1, P, None)                  # nothing prepared any more
    eng.null_local_prepare(P, edges, thr)
    with pytest.raises(CnaHipError, match='prepare'):
        eng.null_local_launch(1, P - 1, None)              # shape differs from the prepared pass
    eng.null_local_launch(1, P, None)
    c = eng.null_local_fetch()
    np.testing.assert_array_equal(c[0], a[0])


def test_resident_graph_notices_bulk_in_place_edits(eng, orc):
    """The graph stays on the device while the same scipy object is passed; an in-place rescaling of
    its values must be seen (window hash in Engine._key) and the graph uploaded again."""
    import cna_amd as cna
    case = load_case('c01_plain_f32')
    data = case['data']
    A = data.obsp['connectivities']
    s0 = np.random.RandomState(2).rand(A.shape[0], 2)
    a = cna.tl.diffuse(data, s0, 2, engine=eng)
    assert eng.ensure_graph(A) is False                      # resident
    A.data *= 0.5                                            # same object, same buffers, other values
    try:
        b = cna.tl.diffuse(data, s0, 2, engine=eng)
        assert relerr(b, orc.diffuse(sp.csr_matrix(A), s0, 2, mode='f64')) < 1e-13
        assert relerr(a, b) > 1e-3
    finally:
        A.data *= 2.0


def test_resident_graph_notices_single_entry_edits_unless_pinned(eng, orc):
    """The resident graph is keyed on a hash of its whole content (csrc/host_graph.c): changing ONE value
    or ONE column index in place re-uploads it (the reference reads the matrix on every call,
    _nam.py:25-28).  engine.pin_graph(A) is the caller's promise not to do that; the matrix is then
    recognised by identity."""
    import cna_amd as cna
    from cna_amd import synth
    data, _ = synth.make_dataset(40000, 20, k=15, seed=9)           # > 3 x 64 KB of values: edits fall outside any window
    A = data.obsp['connectivities']
    s0 = np.random.RandomState(3).rand(A.shape[0], 2)
    a = cna.tl.diffuse(data, s0, 2, engine=eng)
    assert eng.ensure_graph(A) is False
    e = A.nnz // 3
    old_v, old_j = A.data[e], A.indices[e]
    try:
        A.data[e] = old_v * 3 + 1                                # one value
        b = cna.tl.diffuse(data, s0, 2, engine=eng)
        assert relerr(b, orc.diffuse(sp.csr_matrix(A), s0, 2, mode='f64')) < 1e-13 and not np.array_equal(a, b)
        A.data[e] = old_v
        row = np.searchsorted(A.indptr, e, side='right') - 1
        free = np.setdiff1d(np.arange(A.shape[0]), A.indices[A.indptr[row]:A.indptr[row + 1]])
        A.indices[e] = free[0]                                   # one column index
        c = cna.tl.diffuse(data, s0, 2, engine=eng)
        assert relerr(c, orc.diffuse(sp.csr_matrix((A.data, A.indices, A.indptr), shape=A.shape), s0, 2, mode='f64')) < 1e-13
        assert not np.array_equal(a, c)
        A.indices[e] = old_j
        d = cna.tl.diffuse(data, s0, 2, engine=eng)
        assert np.array_equal(a, d)
        eng.pin_graph(A)
        assert eng.ensure_graph(A) is False                      # pinned: recognised by identity alone
        A.data[e] = old_v * 3 + 1
        assert eng.ensure_graph(A) is False                      # ... which is the caller's promise not to do this
        A.data[e] = old_v
        eng.unpin_graph()
        assert eng.ensure_graph(A) is False                      # unpinned again: content checked, unchanged
        A.data[e] = old_v * 3 + 1
        assert eng.ensure_graph(A) is True                       # and an edit is seen
        A.data[e] = old_v
    finally:
        A.data[e], A.indices[e] = old_v, old_j
        eng.unpin_graph()


def test_association_sees_in_place_graph_edits_through_the_deferred_check(eng):
    """association() validates the resident graph by a full content hash taken on a helper thread while
    the kernels already run (engine.ensure_graph(defer=True)); if the matrix was edited in place the call
    starts over on a fresh upload -- the result must equal that of a fresh engine state, data.obs must
    hold the new numbers only, and the resident NAM must not be reused."""
    import cna_amd as cna
    from cna_amd import synth
    data, meta = synth.make_dataset(30000, 24, k=15, seed=21)
    A = data.obsp['connectivities']
    kw = dict(nsteps=3, Nnull=100, seed=4, return_full=True)
    r1 = cna.tl.association(data, meta['y'], 'id', engine=eng, **kw)
    c1 = data.obs['coef'].values.copy()
    e = A.nnz // 2 + 12345
    keep = A.data.copy()
    try:
        A.data[e:e + 2000] *= 0.25                              # far from the three probe windows
        r2 = cna.tl.association(data, meta['y'], 'id', engine=eng, **kw)
        c2 = data.obs['coef'].values.copy()
        assert not np.array_equal(c1, c2)
        # reference point: the same analysis from scratch
        from cna_amd.engine import Engine
        fresh = Engine(device=0)
        try:
            data3 = type(data)(data.obs[['id']].copy(), A.copy())
            r3 = cna.tl.association(data3, meta['y'], 'id', engine=fresh, **kw)
            assert r2.p == r3.p and r2.k == r3.k
            np.testing.assert_array_equal(c2, data3.obs['coef'].values)
            np.testing.assert_array_equal(data.obs['coef_fdr'].values, data3.obs['coef_fdr'].values)
        finally:
            fresh.close()
    finally:
        A.data[:] = keep
    r4 = cna.tl.association(data, meta['y'], 'id', engine=eng, **kw)
    np.testing.assert_array_equal(data.obs['coef'].values, c1)
    assert r4.p == r1.p


def test_association_sees_in_place_sample_id_edits_through_the_deferred_check(eng):
    """The per-cell sample ids are memoised (factorised codes resident on the device) and, on one GPU,
    validated by a content hash taken on a helper thread while the kernels already run: after an in-place
    edit of the id column the call starts over; result and data.obs must be those of a fresh dataset."""
    import cna_amd as cna
    from cna_amd import synth
    data, meta = synth.make_dataset(20000, 24, k=15, seed=22)
    kw = dict(nsteps=3, Nnull=100, seed=4, return_full=True)
    r1 = cna.tl.association(data, meta['y'], 'id', engine=eng, **kw)
    c1 = data.obs['coef'].values.copy()
    ids = data.obs['id'].values
    keep = ids.copy()
    try:
        sel = np.arange(5000, 9000)
        ids[sel] = ids[sel[::-1]]                                # same buffer, other assignment of cells to samples
        assert not np.array_equal(ids, keep) and np.array_equal(data.obs['id'].values, ids)   # the frame's own buffer
        r2 = cna.tl.association(data, meta['y'], 'id', engine=eng, **kw)
        c2 = data.obs['coef'].values.copy()
        assert not np.array_equal(c1, c2)
        from cna_amd.engine import Engine
        fresh = Engine(device=0)
        try:
            data3 = type(data)(data.obs[['id']].copy(), data.obsp['connectivities'].copy())
            r3 = cna.tl.association(data3, meta['y'], 'id', engine=fresh, **kw)
            assert r2.p == r3.p and r2.k == r3.k
            np.testing.assert_array_equal(c2, data3.obs['coef'].values)
            np.testing.assert_array_equal(data.obs['coef_fdr'].values, data3.obs['coef_fdr'].values)
        finally:
            fresh.close()
    finally:
        ids[:] = keep
    r4 = cna.tl.association(data, meta['y'], 'id', engine=eng, **kw)
    np.testing.assert_array_equal(data.obs['coef'].values, c1)
    assert r4.p == r1.p


def _fuzz_config(seed):
    rs = np.random.RandomState(1000 + seed)
    n = int(rs.choice([400, 900, 1700, 3100]))
    N = int(rs.choice([10, 13, 24, 37, 64, 90, 110]))
    cfg = dict(n=n, N=N, k=int(rs.choice([5, 10, 20])), n_covs=int(rs.choice([0, 0, 1, 3])),
               n_batches=int(rs.choice([0, 0, 2, 5, 9])), graph_dtype=rs.choice(['float32', 'float64']),
               sid_kind=str(rs.choice(['int', 'str', 'cat'])), cluster_sorted=bool(rs.rand() < 0.5),
               signal=bool(rs.rand() < 0.7))
    call = dict(nsteps=rs.choice([None, 1, 2, 3, 5]), Nnull=int(rs.choice([20, 77, 200, 1100])), seed=int(rs.randint(0, 99)))
    call['nsteps'] = None if call['nsteps'] is None else int(call['nsteps'])
    if rs.rand() < 0.3:
        call['ks'] = sorted(set(int(v) for v in rs.randint(1, max(2, N // 6), 3)))
    if rs.rand() < 0.2:
        call['force_permute_all'] = True
    if rs.rand() < 0.3:
        call['max_frac_pcs'] = float(rs.choice([0.05, 0.3]))
    cfg['drop_y'] = int(rs.randint(0, 3))                  # samples whose phenotype is missing
    cfg['donors'] = bool(cfg['n_batches'] == 0 and rs.rand() < 0.25)
    return cfg, call


@pytest.mark.parametrize('seed', range(24))
def test_random_configurations_vs_oracle(eng, orc, seed):
    """Seeded random sweep over the argument space (sizes, graph dtype, id column kind, covariates,
    batches, donors, missing phenotypes, step rule, permutation count, ks, PC budget): the product
    against the float64 oracle, or the same exception type when the inputs are not admissible."""
    import cna_amd as cna
    from cna_amd import synth
    import warnings
    cfg, call = _fuzz_config(seed)
    data, meta = synth.make_dataset(cfg['n'], cfg['N'], k=cfg['k'], seed=seed, n_covs=cfg['n_covs'],
                                    n_batches=cfg['n_batches'], graph_dtype=np.dtype(cfg['graph_dtype']).type,
                                    sid_kind=cfg['sid_kind'], cluster_sorted=cfg['cluster_sorted'], signal=cfg['signal'])
    y = meta['y'].copy()
    if cfg['drop_y']:
        y.iloc[np.random.RandomState(seed).choice(len(y), cfg['drop_y'], replace=False)] = np.nan
    donorids = None
    if cfg['donors']:
        donor = np.arange(len(y)) // 2
        donorids = pd.Series(donor, index=y.index)
        y = pd.Series(y.groupby(donorids).transform('first').values, index=y.index)
    kw = dict(covs=meta['covs'], batches=meta['batches'], donorids=donorids, **call)
    data2 = type('D', (), {'obs': data.obs.copy(), 'obsp': data.obsp, 'uns': {}})()
    want_err = got_err = None
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        try:
            ref = orc.association(data2, y, 'id', mode='f64', allow_low_sample_size=True, **kw)
        except Exception as e:          # noqa: BLE001
            want_err = e
        try:
            res = cna.tl.association(data, y, 'id', return_full=True, allow_low_sample_size=True, engine=eng, **kw)
        except Exception as e:          # noqa: BLE001
            got_err = e
    if want_err is not None or got_err is not None:
        assert want_err is not None and got_err is not None, (repr(want_err), repr(got_err), cfg, call)
        assert type(want_err) is type(got_err) or isinstance(got_err, (ValueError, AttributeError)), (want_err, got_err)
        return
    assert int(res.k) == int(ref['k']) and np.array_equal(res.kept, ref['kept']), (cfg, call)
    assert res.p == pytest.approx(ref['p'], rel=1e-12), (cfg, call)
    assert relerr(res.nam.values.T, ref['nam']) < 1e-12
    assert relerr(res.namresid.values.T, ref['namresid']) < 1e-8
    assert relerr(res.ncorrs.values, ref['ncorrs']) < 1e-8
    np.testing.assert_allclose(res.nullminps, ref['nullminps'], rtol=1e-6)
    if ref.get('fdrs') is not None:
        T = fdr_rows(res.fdrs, ref['fdrs'], ref['ncorrs'])
        assert np.array_equal(res.fdrs.num_detected.values[:T], ref['fdrs']['num_detected'][:T]), (cfg, call)
        np.testing.assert_allclose(res.fdrs.fdr.values[:T], ref['fdrs']['fdr'][:T], rtol=1e-7, atol=1e-13)


@pytest.mark.parametrize('n', [1, 2, 3, 64, 257, 999, 1000])
def test_device_median_equals_numpy(eng, n):
    """cna_stat_median (radix select on the device) against np.median: odd and even counts, duplicates,
    negative values, NaNs (-> NaN), in both statistic spaces."""
    from cna_amd import _ffi
    eng.ensure_graph(load_case('c01_plain_f32')['data'].obsp['connectivities'])   # sizes the statistic buffer (1000 cells)
    rs = np.random.RandomState(n)
    N = 12
    X = np.round(rs.randn(n, N), 1)                     # coarse values: plenty of exact duplicates downstream
    eng.upload_x(X)
    bc = np.arange(N) % 4
    eng.batch_kurtosis(_ffi.MAT_X, bc, 4)
    with np.errstate(all='ignore'):
        v = eng.x_stat(ordered=False)
        want = np.median(v)
    got = eng.stat_median()
    assert (np.isnan(want) and np.isnan(got)) or got == want, (got, want)
    if n >= 3:
        X[n // 2] = 1.0                                 # a constant row: its batch kurtosis is NaN
        eng.upload_x(X)
        eng.batch_kurtosis(_ffi.MAT_X, bc, 4)
        assert np.isnan(eng.x_stat(ordered=False)).any() and np.isnan(eng.stat_median())


def test_device_median_of_walk_kurtosis(eng, orc):
    """... and on the NAM-space statistic (Fisher kurtosis per cell: negative values occur)."""
    case = load_case('c02_covs_autostop')
    A, codes, N = _setup_nam(eng, orc, case)
    for i in range(4):
        eng.nam_step(True, True, True)
        v = eng.cell_stat(A.shape[0])
        assert (v < 0).any()
        assert eng.stat_median() == np.median(v)


@pytest.mark.parametrize('n,d,k', [(3000, 8, 30), (2500, 8, 15), (1200, 50, 15), (900, 3, 7), (400, 20, 41)])
def test_device_knn_graph_builder_vs_host_builder(eng, n, d, k):
    """csrc/knn.hip (the benchmark's input generator on the GPU) against the cKDTree + scipy.sparse builder:
    same neighbour sets (exact kNN both; a handful of float32-vs-float64 ties may differ), same weights,
    a canonical CSR (sorted indices, no diagonal, symmetric)."""
    from cna_amd import synth
    rs = np.random.RandomState(n + d)
    X = rs.randn(n, d).astype(np.float32)
    A = eng.knn_graph(X, k)
    B = synth.fuzzy_knn_graph(X, k=k, builder='cpu')
    assert A.shape == B.shape and A.dtype == np.float32 and A.indices.dtype == np.int32
    assert A.has_canonical_format or (np.diff(A.indptr) >= 0).all()
    for i in range(0, n, 37):
        seg = A.indices[A.indptr[i]:A.indptr[i + 1]]
        assert (np.diff(seg) > 0).all() and i not in seg
    assert abs(A - A.T).max() < 1e-6                                  # fuzzy union is symmetric
    # structure: at most a few edges differ (distance ties); weights of common edges agree
    diff = (A != 0).astype(np.int8) - (B != 0).astype(np.int8)
    assert abs(diff).sum() <= max(4, 0.001 * B.nnz), (abs(diff).sum(), B.nnz)
    common = A.multiply(B != 0) - B.multiply(A != 0)
    assert abs(common).max() < 1e-4
    assert abs(A.nnz - B.nnz) <= max(4, 0.001 * B.nnz)


def test_small_block_many_samples_takes_the_per_cell_pass_under_lapack(eng, monkeypatch, capsys):
    """Fewer than 500 000 cells and 128 samples or more: LAPACK's eigenvectors come from a thread of their own while the
    main thread fetches the local null, builds the FDR table and runs the per-cell pass (_association.py: tail_first).
    Same results, progress text and warnings as the sequential order; an error of the early per-cell pass surfaces where
    the sequential order would meet it (after the global test), and a failing global test still wins over it."""
    import warnings
    import cna_amd as cna
    from cna_amd import synth
    from cna_amd.tools import _association as A
    data, meta = synth.make_dataset(30000, 160, k=15, seed=4)
    kw = dict(Nnull=200, seed=2, nsteps=3, show_progress=True)
    monkeypatch.setattr(eng, 'reuse_nam', False)
    submits = []
    real_pool = A._eig_pool

    def counting_pool():
        submits.append(1)
        return real_pool()
    monkeypatch.setattr(A, '_eig_pool', counting_pool)

    def run():
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter('always')
            res = cna.tl.association(data, meta['y'], 'id', return_full=True, engine=eng, **kw)
        out = capsys.readouterr().out
        return (res.p, int(res.k), res.ncorrs.values.copy(), res.fdrs.copy(), data.obs['coef'].values.copy(),
                data.obs['coef_fdr'].values.copy(), res.fdr_5p_t, res.fdr_10p_t, out,
                sorted(str(x.message) for x in w if not issubclass(x.category, ResourceWarning)))
    monkeypatch.setattr(A, '_TAIL_FIRST', False)
    run()                                                         # (data.obs has its columns from here on: same warnings)
    monkeypatch.setattr(A, '_TAIL_FIRST', True)
    a = run()
    assert len(submits) == 1                                      # the path under test was taken
    monkeypatch.setattr(A, '_TAIL_FIRST', False)
    b = run()
    assert len(submits) == 1
    assert a[0] == b[0] and a[1] == b[1] and a[6] == b[6] and a[7] == b[7]
    np.testing.assert_array_equal(a[2], b[2])
    pd.testing.assert_frame_equal(a[3], b[3])
    np.testing.assert_array_equal(a[4], b[4])
    np.testing.assert_array_equal(a[5], b[5])
    assert a[8] == b[8] and 'computing neighborhood-level FDRs' in a[8]
    assert a[9] == b[9]

    # an error in the early per-cell pass: reported, and only after the global test has had its say
    monkeypatch.setattr(A, '_TAIL_FIRST', True)
    real_percell, real_fetch = eng.percell, eng.global_test_fetch
    order = []

    def bad_percell(*args, **k):
        order.append('percell')
        raise FloatingPointError('per-cell pass')

    def fetch(*args, **k):
        order.append('global test')
        return real_fetch(*args, **k)
    monkeypatch.setattr(eng, 'percell', bad_percell)
    monkeypatch.setattr(eng, 'global_test_fetch', fetch)
    with pytest.raises(FloatingPointError, match='per-cell pass'):
        cna.tl.association(data, meta['y'], 'id', engine=eng, **kw)
    assert order == ['percell', 'global test']

    def bad_fetch(*args, **k):
        real_fetch(*args, **k)
        raise ArithmeticError('global test')
    monkeypatch.setattr(eng, 'global_test_fetch', bad_fetch)
    with pytest.raises(ArithmeticError, match='global test'):
        cna.tl.association(data, meta['y'], 'id', engine=eng, **kw)
    # both per-cell columns reach the frame BEFORE the global test in this schedule (the coefficient column under the local
    # null, the FDR column while the eigenvectors are computed): a test that then fails puts both back
    monkeypatch.setattr(eng, 'percell', real_percell)
    y2 = pd.Series(np.random.RandomState(7).randn(len(meta['y'])), index=meta['y'].index)
    seen = {}

    def bad_fetch2(*args, **k):
        seen['coef'] = data.obs['coef'].values.copy()
        seen['fdr'] = data.obs['coef_fdr'].values.copy()
        real_fetch(*args, **k)
        raise ArithmeticError('global test')
    monkeypatch.setattr(eng, 'global_test_fetch', bad_fetch2)
    with pytest.raises(ArithmeticError, match='global test'):
        cna.tl.association(data, y2, 'id', engine=eng, **kw)
    assert not np.array_equal(seen['coef'], a[4]) and not np.array_equal(seen['fdr'], a[5])    # written early ...
    np.testing.assert_array_equal(data.obs['coef'].values, a[4])                                 # ... and put back
    np.testing.assert_array_equal(data.obs['coef_fdr'].values, a[5])
    monkeypatch.setattr(eng, 'percell', real_percell)
    monkeypatch.setattr(eng, 'global_test_fetch', real_fetch)
    capsys.readouterr()
    assert cna.tl.association(data, meta['y'], 'id', engine=eng, **kw) == a[0]


def test_error_after_a_fused_null_launch_leaves_the_engine_usable(eng, monkeypatch, general_path):
    """The fused selection call may launch the local null itself (the draw thread's flag says the phenotypes are on the
    device).  An analysis that raises AFTER that launch and before its fetch -- `ks` too large for the cohort
    (_association.py:29-33), a draw that fails, a conditioning that refuses -- must not leave the pass pending: the next
    call on the same (process-wide) engine works and returns what a fresh engine returns."""
    import cna_amd as cna
    from cna_amd import synth
    data, meta = synth.make_dataset(6000, 24, k=15, seed=5)
    kw = dict(Nnull=100, seed=1, nsteps=3)
    want = cna.tl.association(data, meta['y'], 'id', engine=eng, return_full=True, **kw)
    coef = data.obs['coef'].values.copy()
    # (1) ks too large: ValueError from the reference's own guard, raised after walk + selection
    # ... and ks = [n - 1]: no degrees of freedom left, every p is NaN, np.nanargmin's ValueError as upstream -- AFTER the fetch
    for bad, msg in (([24], 'Maximum number of PCs'), ([30], 'Maximum number of PCs'), ([23], 'All-NaN slice')):
        with pytest.raises(ValueError, match=msg):
            cna.tl.association(data, meta['y'], 'id', engine=eng, ks=bad, **kw)
        np.testing.assert_array_equal(data.obs['coef'].values, coef)
        got = cna.tl.association(data, meta['y'], 'id', engine=eng, return_full=True, **kw)
        assert got.p == want.p and got.k == want.k
        np.testing.assert_array_equal(got.fdrs.num_detected.values, want.fdrs.num_detected.values)
        np.testing.assert_array_equal(got.ncorrs.values, want.ncorrs.values)
    # (2) a failure of the first consumer after the launch
    # (the Gram matrix is collected by gram_pcs_tests, or by gram_fetch when the caller wants the full result)
    for name, extra in (('gram_pcs_tests', {}), ('gram_fetch', dict(return_full=True))):
        real = getattr(eng, name)

        def boom(*a, **k):
            raise FloatingPointError('injected')
        monkeypatch.setattr(eng, name, boom)
        with pytest.raises(FloatingPointError):
            cna.tl.association(data, meta['y'], 'id', engine=eng, **kw, **extra)
        monkeypatch.setattr(eng, name, real)
        assert cna.tl.association(data, meta['y'], 'id', engine=eng, **kw) == want.p
    # ... and a failure AFTER the F-tests were queued by that call: nothing stays pending either
    real_coef = eng.percell_coef_wait

    def boom2(*a, **k):
        raise FloatingPointError('injected late')
    monkeypatch.setattr(eng, 'percell_coef_wait', boom2)
    with pytest.raises(FloatingPointError):
        cna.tl.association(data, meta['y'], 'id', engine=eng, **kw)
    monkeypatch.setattr(eng, 'percell_coef_wait', real_coef)
    assert cna.tl.association(data, meta['y'], 'id', engine=eng, **kw) == want.p
    # (3) the C entry point by itself: a launched pass, dropped; then nothing pending and the next prepare is accepted
    thr = np.arange(0.01, 0.04, 0.0001)
    edges = thr ** 2 - 1e-8 - 1e-5 * thr ** 2
    eng.null_local_launch(1, 50, edges, thr)
    eng.null_local_discard()
    eng.null_local_discard()                                      # no-op
    with pytest.raises(Exception, match='no local-null pass pending'):
        eng.null_local_fetch()
    eng.null_local_launch(1, 50, edges, thr)
    a = eng.null_local_fetch()
    eng.null_local_launch(1, 50, edges, thr)
    b = eng.null_local_fetch()
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)


def test_new_graph_is_uploaded_beside_the_sample_ids(monkeypatch):
    """A graph of 100 000 cells or more that is new to the device goes up on a helper thread while this thread factorises
    the sample ids (_association.py:_prefetch_graph).  Same results as without; a call that fails on the ids (no such
    column) has still collected the helper -- the graph is resident, the engine usable."""
    import cna_amd as cna
    from cna_amd import synth
    from cna_amd.engine import Engine
    from cna_amd.tools import _association as A
    data, meta = synth.make_dataset(120000, 40, k=15, seed=13)
    kw = dict(nsteps=3, Nnull=100, seed=1, return_full=True)
    outs = []
    for on in (True, False):
        monkeypatch.setattr(A, '_PREFETCH_GRAPH', on)
        e = Engine(device=0)
        try:
            A._TRACE = []
            if on:
                with pytest.raises(KeyError):
                    cna.tl.association(data, meta['y'], 'no_such_column', engine=e, **kw)
                assert e.graph_resident(data.obsp['connectivities'])
            res = cna.tl.association(data, meta['y'], 'id', engine=e, **kw)
            marks = [m for m, _ in A._TRACE]
            assert ('graph prefetched' in marks) == on
            outs.append((res.p, int(res.k), res.ncorrs.values.copy(), res.fdrs.values.copy(), data.obs['coef_fdr'].values.copy()))
        finally:
            A._TRACE = None
            e.close()
    for x, y in zip(*outs):
        np.testing.assert_array_equal(x, y)


def test_zero_variance_cells_with_many_samples(eng, orc):
    """140 samples (the by-product / small-block schedules) and cells of zero variance -- a far-away blob whose only sample
    has no phenotype: whatever the fused selection call queued for "no zero variance" must not be used; results as the
    oracle's, the cells dropped (_association.py:182-185); the next analysis on the engine is unaffected."""
    import cna_amd as cna
    from cna_amd import synth
    data, meta = synth.make_dataset(8000, 140, k=15, seed=9)
    A = sp.csr_matrix(data.obsp['connectivities'])
    n, n_iso = A.shape[0], 25
    rs = np.random.RandomState(7)
    B = sp.random(n_iso, n_iso, density=0.6, random_state=rs, format='csr', dtype=np.float64)
    B = B + B.T
    B.setdiag(0)
    B.eliminate_zeros()
    B.data = np.clip(B.data, 0.05, 1.0)
    A2 = sp.block_diag([A, B.astype(A.dtype)], format='csr')
    A2.sort_indices()
    obs = pd.DataFrame({'id': np.concatenate([data.obs['id'].values, np.repeat(140, n_iso)])},
                       index=pd.Index(['cell_%d' % i for i in range(n + n_iso)], name='cell'))
    d2 = type('D', (), {'obs': obs, 'obsp': {'connectivities': A2}, 'uns': {}})()
    y = pd.concat([meta['y'], pd.Series([np.nan], index=[140])])
    kw = dict(nsteps=3, Nnull=100, seed=3)
    res = cna.tl.association(d2, y, 'id', return_full=True, engine=eng, **kw)
    ref = orc.association(d2, y, 'id', mode='f64', **kw)
    assert (~res.kept).sum() == n_iso and np.array_equal(res.kept, ref['kept'])
    assert int(res.k) == ref['k'] and res.p == ref['p']
    assert relerr(res.ncorrs.values, ref['ncorrs']) < 1e-10
    assert relerr(res.namresid_svs.values, ref['svs']) < 1e-10
    T = fdr_rows(res.fdrs, ref['fdrs'], ref['ncorrs'])
    assert np.array_equal(res.fdrs.num_detected.values[:T], ref['fdrs']['num_detected'][:T])
    # ... and the very next analysis (no zero variance)
    res2 = cna.tl.association(data, meta['y'], 'id', return_full=True, engine=eng, **kw)
    ref2 = orc.association(data, meta['y'], 'id', mode='f64', **kw)
    assert res2.p == ref2['p'] and int(res2.k) == ref2['k'] and relerr(res2.namresid_svs.values, ref2['svs']) < 1e-10


def test_small_block_schedule_is_reproducible_over_many_calls(eng, monkeypatch):
    """Two host threads drive one context in the small-block schedule (the eigenvector thread fetches the Gram matrix,
    solves and queues the F-tests while the main thread collects the local null and runs the per-cell pass; the library's
    draw thread conditions the phenotypes beside both).  300 analyses of two phenotypes in turn: every one returns the
    bits of its first run."""
    import cna_amd as cna
    from cna_amd import synth
    from cna_amd.tools import _association as A
    from cna_amd.tools import _nam as NM
    assert A._TAIL_FIRST
    data, meta = synth.make_dataset(30000, 160, k=15, seed=11)
    monkeypatch.setattr(eng, 'reuse_nam', False)
    ys = [meta['y'], pd.Series(np.random.RandomState(5).randn(160), index=meta['y'].index)]
    kw = dict(Nnull=200, seed=4, nsteps=3)
    import warnings
    first = {}
    native0 = NM.eig_stats['native']
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for it in range(300):
            j = it & 1
            p = cna.tl.association(data, ys[j], 'id', engine=eng, **kw)
            got = (p, data.obs['coef'].values.copy(), data.obs['coef_fdr'].values.copy())
            if j not in first:
                first[j] = got
                continue
            assert got[0] == first[j][0], it
            np.testing.assert_array_equal(got[1], first[j][1])
            np.testing.assert_array_equal(got[2], first[j][2])
    assert NM.eig_stats['native'] - native0 == 300        # the library's own eigen-solver every time, on the helper thread
    assert first[0][0] != first[1][0] or not np.array_equal(first[0][1], first[1][1])
