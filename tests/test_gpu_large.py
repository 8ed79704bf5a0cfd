"""GPU parity at BASELINE.json's large configurations (run with -m gpu on an MI355X).

  C3: 1M cells x 100 samples          (configs[2], "HBM roofline run")
  C4: 2M cells x 200 samples          (configs[3], here on ONE GPU)
  C5: C4 + 5 covariates, Nnull=10000  (configs[4])

The CPU oracle cannot run whole problems of this size in test time, so the checks are the
size-independent ones (reference lines: _nam.py:21-76,118-177, _association.py:64-120,223-237):
oracle-free identities on the fetched matrices, every per-cell column recomputed on the host from
the returned tables, invariance under a random renumbering of the cells, and -- the piece that IS
an oracle comparison at full size -- the NAM rows of a slice of cells against the oracle's walk
restricted to the slice's 3-hop neighbourhood (three walk steps cannot see further).  One
reduced-size case (3k cells, 200 samples, 5 covariates, Nnull=10000) is compared field by field with
oracle.association: it covers the 10001-column global test, the P'=1000 cap of the local null and
nullminps at the permutation count of configs[4].

CNA_SKIP_LARGE=1 skips the 1M / 2M cases (development loops)."""
import os
import time
import warnings

import numpy as np
import pandas as pd
import pytest
import scipy.sparse as sp

from helpers import relerr, fdr_rows

pytestmark = pytest.mark.gpu

SIZES = {'C3': (1_000_000, 100), 'C4': (2_000_000, 200)}
SLICE = 20000


@pytest.fixture(scope='module')
def eng():
    from cna_amd.engine import get_engine
    return get_engine()


@pytest.fixture(scope='module')
def orc():
    from oracle import cna_oracle
    return cna_oracle


@pytest.fixture(scope='module', params=['C3', 'C4'])
def big(request):
    if os.environ.get('CNA_SKIP_LARGE'):
        pytest.skip('CNA_SKIP_LARGE set')
    from cna_amd import synth
    n, N = SIZES[request.param]
    t = time.time()
    data, meta = synth.make_dataset(n, N, k=30, seed=0, n_covs=5)       # the generator and seed of bench.py
    print('\n[%s] dataset %d x %d in %.1f s' % (request.param, n, N, time.time() - t))
    return request.param, data, meta


def _recount(coef, thr):
    """num_detected[t] = #{|coef| > thr[t]} (strict, _association.py:107), NaN never counts"""
    a = np.sort(np.abs(coef[~np.isnan(coef)]))
    return len(a) - np.searchsorted(a, thr, side='right')


def _percell_from_table(coef, thr, fdr):
    """_association.py:234-237 vectorised: min fdr over thresholds <= |coef|, else 1"""
    runmin = np.fmin.accumulate(fdr)
    idx = np.searchsorted(thr, np.abs(coef), side='right') - 1
    out = np.where(idx >= 0, runmin[np.maximum(idx, 0)], 1.0)
    out[np.isnan(coef)] = 1.0
    return out


def _check_full_result(res, data, y, N, Nnull, covs=None):
    n = len(data.obs)
    nam = res.nam.values                                               # samples x cells
    assert nam.shape == (N, n)
    np.testing.assert_allclose(nam.sum(axis=1), 1.0, rtol=1e-9)        # column-stochastic walk, _nam.py:28,33,73
    assert nam.min() >= 0.0
    X = res.namresid.values
    if covs is None:
        # Under the default schedule X is the by-product of the walk's last step (diffuse.hip:select_tail) and the raw
        # NAM comes from a SECOND run of that step (c_api.hip:need_nam): the two must be the same matrix -- X is the
        # NAM centred and divided by its std per cell (_nam.py:122,159), entry by entry, at full size
        for c0 in range(0, n, 250_000):
            blk = nam[:, c0:c0 + 250_000]
            want = blk - blk.mean(axis=0)
            want /= want.std(axis=0, ddof=1)
            np.testing.assert_allclose(X[:, c0:c0 + 250_000], want, rtol=1e-9, atol=1e-11)
        del blk, want
    del nam
    assert np.abs(X.mean(axis=0)).max() < 1e-10                        # _nam.py:122
    np.testing.assert_allclose(X.std(axis=0, ddof=1), 1.0, rtol=1e-10)  # _nam.py:159
    if covs is not None:
        cz = (covs - covs.mean()) / covs.std()
        assert np.abs(cz.values.T.dot(X)).max() < 1e-7                 # M projects the covariates out, _nam.py:128-135
    yz = (y.values - y.values.mean()) / y.values.std()
    coef = data.obs['coef'].values.copy()
    fdrcol = data.obs['coef_fdr'].values.copy()
    np.testing.assert_allclose(coef, yz.dot(X) / N, rtol=1e-9, atol=1e-12)       # _association.py:77
    np.testing.assert_allclose(res.ncorrs.values, coef, rtol=0, atol=0)
    U = res.namresid_sampleXpc.values
    np.testing.assert_allclose(U.T.dot(U), np.eye(U.shape[1]), atol=1e-9)
    G = X.dot(X.T)
    # U diag(svs) U^T is the Gram matrix it was computed from (_nam.py:105,168-175: varexp = svs / N / n_cells)
    np.testing.assert_allclose((U * (res.namresid_varexp.values * N * X.shape[1])).dot(U.T), G, rtol=1e-6,
                               atol=1e-7 * X.shape[1])
    del X, G
    f = res.fdrs
    thr, fdr, num = f.threshold.values, f.fdr.values, f.num_detected.values
    assert np.array_equal(num, _recount(coef, thr))
    assert (np.diff(num) <= 0).all() and ((fdr >= 0) | np.isnan(fdr)).all()
    np.testing.assert_allclose(fdrcol, _percell_from_table(coef, thr, fdr), rtol=1e-14)
    assert 1 / (Nnull + 1) <= res.p <= 1 and len(res.nullminps) == Nnull and res.kept.all()
    assert ((res.nullminps > 0) & (res.nullminps <= 1)).all()
    return coef, thr, fdr, num


def test_large_properties_and_renumbering(big, eng):
    name, data, meta = big
    import cna_amd as cna
    n, N = SIZES[name]
    y = meta['y']
    kw = dict(nsteps=3, Nnull=1000, seed=0, return_full=True)
    t = time.time()
    res = cna.tl.association(data, y, 'id', **kw)
    print('[%s] first association() incl. graph preparation and upload: %.2f s' % (name, time.time() - t))
    coef, thr, fdr, num = _check_full_result(res, data, y, N, 1000)
    p, k = res.p, res.k
    ks = list(res.ks)
    assert k in ks
    # the local null of that analysis ran on the integer matrix cores (csrc/null_i8.hip): on the state it left on
    # the device, the f64 kernel must count the very same integers, threshold by threshold, at full size
    edges = thr ** 2 - 1e-8 - 1e-5 * thr ** 2
    tails = eng.null_local_resident(1, 1000, edges)
    sums = eng.null_local_resident(1, 1000, edges, sums_only=True)
    used, rechecked, fallback = eng.null_local_i8_stats()
    assert used and not fallback and rechecked < 1e-3 * n * 1000
    assert np.array_equal(sums, tails.sum(axis=0))
    ranks = np.array([(np.nan_to_num(coef) ** 2 >= e).sum() for e in edges])
    with np.errstate(all='ignore'):
        np.testing.assert_allclose(fdr, (tails / ranks).mean(axis=0), rtol=1e-12, equal_nan=True)
    # ... and an INDEPENDENT count at full size (no HIP kernel on this side): the oracle's permutation draw
    # (_stats.py:4-18, numpy's global RNG), its tail_counts (_stats.py:34-62) and numpy's GEMV on the fetched
    # residualised NAM, for the first eight permutations -- against the f64 kernel's rows for those columns and
    # against the integer pass restricted to them.  A value within an ulp of an edge may fall on either side
    # of it (numpy adds the 200 products in another order): a handful of off-by-ones, as in
    # test_local_null_counts_are_exact.
    from oracle import cna_oracle as orc_
    X = res.namresid.values                                            # samples x cells
    yz = (y.values - y.values.mean()) / y.values.std()
    np.random.seed(kw['seed'])
    ynull = orc_.conditional_permutation(np.ones(N), yz, 1000)[:, :8]  # M = I: the conditioned phenotype is y_ / std
    zc = ynull / ynull.std(axis=0, ddof=1)
    znull = np.abs(zc.T.dot(X) / N).T                                  # cells x 8, _association.py:96-99
    want = orc_.tail_counts(thr, znull)
    diff = np.abs(tails[:8] - want)
    assert diff.max() <= 1 and diff.sum() <= 8, (diff.max(), diff.sum())
    sums8 = eng.null_local_resident(1, 8, edges, sums_only=True)
    assert np.array_equal(sums8, tails[:8].sum(axis=0))
    print('[%s] independent recount of 8 null columns at full size: %d of %d counts off by one' % (
        name, int(diff.sum()), diff.size))
    del res, tails, X, znull

    # the same analysis with the cells renumbered at random: the global p-value, the chosen k, the FDR
    # table and every cell's coefficient must not move
    rs = np.random.RandomState(123)
    perm = rs.permutation(n)
    A = sp.csr_matrix(data.obsp['connectivities'])
    Ap = A[perm][:, perm].tocsr()
    Ap.sort_indices()
    obs2 = pd.DataFrame({'id': data.obs['id'].values[perm]}, index=data.obs.index[perm])
    data2 = type('D', (), {'obs': obs2, 'obsp': {'connectivities': Ap}, 'uns': {}})()
    res2 = cna.tl.association(data2, y, 'id', **kw)
    assert res2.p == p and res2.k == k
    # sums over neighbours run in a different order: equal to rounding, counts identical
    np.testing.assert_allclose(data2.obs['coef'].values, coef[perm], rtol=1e-9, atol=1e-13)
    # thresholds = np.arange(m/4, m, m/400) has 300 or 301 entries depending on the last bit of m = max |coef|
    # (numpy's ceil((stop - start) / step) at 300.0000...), which the order of the sums may move: compare the
    # common ones
    T2 = len(res2.fdrs)
    assert abs(T2 - len(num)) <= 1
    T = min(T2, len(num))
    np.testing.assert_allclose(res2.fdrs.threshold.values[:T], thr[:T], rtol=1e-12)
    assert np.array_equal(res2.fdrs.num_detected.values[:T], num[:T])
    np.testing.assert_allclose(res2.fdrs.fdr.values[:T], fdr[:T], rtol=1e-9, equal_nan=True)


def test_large_nam_slice_vs_oracle(big, eng, orc):
    """NAM rows of SLICE consecutive cells against the oracle's three walk steps on the sub-graph of the
    slice's 3-hop out-neighbourhood; the column sums (which need every in-edge) are checked over the
    whole graph first and then handed to the oracle."""
    name, data, meta = big
    import cna_amd as cna
    from cna_amd.tools._nam import sample_codes
    n, N = SIZES[name]
    A = sp.csr_matrix(data.obsp['connectivities'])
    NAM, keep = cna.tl.nam(data, 'id', nsteps=3)                  # lazy frame: nothing cells-sized is fetched yet
    assert keep.all()
    cs = eng.fetch_colsums()
    want = np.asarray(A.astype(np.float64).sum(axis=0)).ravel() + 1.0
    assert np.array_equal(cs, want)                               # float32 weights: the f64 sum is exact in any order
    codes, labels = sample_codes(data.obs['id'])
    C = np.bincount(codes, minlength=N).astype(np.float64)
    a0 = (n // 2 // SLICE) * SLICE
    mask = np.zeros(n, dtype=bool)
    mask[a0:a0 + SLICE] = True
    reach = mask.copy()
    for _ in range(3):
        rows = np.flatnonzero(reach)
        lo, hi = A.indptr[rows], A.indptr[rows + 1]
        # columns referenced by the rows reached so far
        take = np.repeat(lo - np.concatenate([[0], np.cumsum(hi - lo)[:-1]]), hi - lo) + np.arange(int((hi - lo).sum()))
        reach[A.indices[take]] = True
    sub = np.flatnonzero(reach)
    print('[%s] 3-hop closure of %d cells: %d cells' % (name, SLICE, len(sub)))
    Asub = A[sub][:, sub].tocsr()
    # rows three hops out have lost neighbours outside the closure: their values are wrong and never
    # reach the slice within three steps
    S = np.zeros((len(sub), N), dtype=bool)
    S[np.arange(len(sub)), codes[sub]] = True
    s = S
    for i in range(3):
        s = orc.diffusion_step(Asub, s, cs[sub], 1, first_onehot=(i == 0), mode='f64')
    pos = np.searchsorted(sub, np.arange(a0, a0 + SLICE))
    ref = s[pos] / C
    got = eng.nam_full(keep=mask)                                  # cells x samples, caller's order (cna_fetch_rows)
    assert got.shape == ref.shape
    err = relerr(got, ref)
    print('[%s] NAM slice vs oracle: relerr %.3g, bit-identical entries %.6f' % (name, err, np.mean(got == ref)))
    assert err < 1e-13


def test_large_config5(big, eng):
    """configs[4]: 5 covariates, Nnull = 10000, local FDR pass on (the graph of the fixture is reused)."""
    name, data, meta = big
    import cna_amd as cna
    n, N = SIZES[name]
    y, covs = meta['y'], meta['covs']
    t = time.time()
    res = cna.tl.association(data, y, 'id', covs=covs, nsteps=3, Nnull=10000, seed=0, return_full=True)
    print('[%s] config-5 style association(): %.2f s' % (name, time.time() - t))
    assert res.r == 5
    coef, thr, fdr, num = _check_full_result(res, data, y, N, 10000, covs=covs)
    # M is the projector off the covariates (_nam.py:128-134)
    M = res.M.values
    cz = ((covs - covs.mean()) / covs.std()).values
    np.testing.assert_allclose(M.dot(cz), 0, atol=1e-10)
    np.testing.assert_allclose(M.dot(M), M, atol=1e-10)
    # the permutation p-value follows from nullminps and the observed min-p (_association.py:85): recompute the
    # observed statistic on the host from the returned PCs
    from cna_amd.tools._stats import minp_stats
    yz = (y.values - y.values.mean()) / y.values.std()
    kidx, pobs, _ = minp_stats(yz[:, None], M, res.namresid_sampleXpc.values, np.asarray(res.ks), res.r)
    assert res.ks[kidx[0]] == res.k
    assert res.p == (np.sum(res.nullminps <= pobs[0] + 1e-8) + 1) / 10001


def test_reduced_config5_vs_oracle(eng, orc):
    """3000 cells x 200 samples, 5 covariates, Nnull = 10000 against oracle.association: the
    10001-column global test, nullminps, the P' = 1000 cap of the local null (_association.py:94)."""
    import cna_amd as cna
    from cna_amd import synth
    data, meta = synth.make_dataset(3000, 200, k=15, seed=77, n_covs=5)
    kw = dict(nsteps=3, Nnull=10000, seed=3)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        res = cna.tl.association(data, meta['y'], 'id', covs=meta['covs'], return_full=True, **kw)
        ref = orc.association(data, meta['y'], 'id', covs=meta['covs'], mode='f64', **kw)
    assert int(res.k) == ref['k'] and res.p == ref['p'] and np.array_equal(res.kept, ref['kept'])
    assert len(res.nullminps) == 10000
    np.testing.assert_allclose(res.nullminps, ref['nullminps'], rtol=1e-7)
    assert relerr(res.nam.values.T, ref['nam']) < 1e-13
    assert relerr(res.namresid.values.T, ref['namresid']) < 1e-9
    assert relerr(res.ncorrs.values, ref['ncorrs']) < 1e-9
    T = fdr_rows(res.fdrs, ref['fdrs'], ref['ncorrs'])
    assert np.array_equal(res.fdrs.num_detected.values[:T], ref['fdrs']['num_detected'][:T])
    np.testing.assert_allclose(res.fdrs.fdr.values[:T], ref['fdrs']['fdr'][:T], rtol=1e-8, atol=1e-13)
    np.testing.assert_allclose(data.obs['coef_fdr'].values, ref['obs_coef_fdr'], rtol=1e-8, atol=1e-13)
    assert res.nullr2_mean == pytest.approx(ref['nullr2_mean'], rel=1e-9)
    assert res.nullr2_std == pytest.approx(ref['nullr2_std'], rel=1e-9)
