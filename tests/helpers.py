"""Shared test helpers: rebuild the inputs stored in a golden fixture."""
import glob
import json
import os

import numpy as np
import pandas as pd
import scipy.sparse as sp

from cna_amd.synth import CellData

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def golden_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, 'c*.npz')))


def messy_names():
    """Fixtures of randomly drawn messy sample-level inputs (tests/golden/make_golden.py: f??_messy)."""
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, 'f*.npz')))


def _index(arr):
    return pd.Index(arr.tolist())


def load_case(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    n = len(z['in_indptr']) - 1
    A = sp.csr_matrix((z['in_data'], z['in_indices'], z['in_indptr']), shape=(n, n))
    A.indptr = A.indptr.astype(np.int32)
    sid_name = z['sid_name'].item()
    if 'in_sid_codes' in z:
        col = pd.Categorical.from_codes(z['in_sid_codes'], categories=z['in_sid_categories'].tolist())
    else:
        col = z['in_sid'].tolist()
    obs = pd.DataFrame({sid_name: col}, index=pd.Index(['cell_%d' % i for i in range(n)], name='cell'))
    data = CellData(obs, A)
    y = pd.Series(z['in_y'], index=_index(z['in_y_index']))
    covs = batches = donorids = None
    if 'in_covs' in z:
        covs = pd.DataFrame(z['in_covs'], index=_index(z['in_covs_index']), columns=z['in_covs_columns'].tolist())
    if 'in_batches' in z:
        batches = pd.Series(z['in_batches'], index=_index(z['in_batches_index']))
    if 'in_donorids' in z:
        donorids = pd.Series(z['in_donorids'], index=_index(z['in_donorids_index']))
    call = json.loads(z['call'].item())
    return dict(name=name, data=data, y=y, covs=covs, batches=batches, donorids=donorids,
                sid_name=sid_name, call=call, z=z)


def sign_align(U, Uref, ncols):
    """Flip columns of U so each has non-negative inner product with Uref's column."""
    U = np.array(U[:, :ncols], dtype=np.float64)
    R = np.asarray(Uref[:, :ncols])
    sgn = np.sign((U * R).sum(axis=0))
    sgn[sgn == 0] = 1
    return U * sgn, R


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = max(np.nanmax(np.abs(b)), 1e-300) if b.size else 1.0
    return float(np.nanmax(np.abs(a - b)) / scale) if a.size else 0.0


def assert_elementwise(a, b, rtol=1e-5, floor=1e-12, what=''):
    """|a - b| <= rtol |b| + floor max|b| for EVERY entry (NaNs must coincide): BASELINE.json's "float within 1e-5
    rel" taken entry by entry, with an absolute floor for entries that are differences of much larger numbers."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if not a.size:
        return
    assert np.array_equal(np.isnan(a), np.isnan(b)), what
    scale = np.nanmax(np.abs(b)) if np.isfinite(b).any() else 0.0
    bad = np.abs(a - b) > rtol * np.abs(b) + floor * scale
    assert not bad.any(), '%s: %d entries off, worst |a-b|/|b| = %.3g, worst |a-b|/max|b| = %.3g' % (
        what, int(bad.sum()), float(np.nanmax(np.abs(a - b)[bad] / np.maximum(np.abs(b)[bad], 1e-300))),
        float(np.nanmax(np.abs(a - b)[bad]) / max(scale, 1e-300)))


def golden_floors(z, name=''):
    """Absolute floors (fractions of max |reference|) of the entry-wise comparison with a golden fixture.  The NAM is a
    sum of non-negative products: purely relative.  Coefficients and the residualised NAM are differences of such sums;
    on a float32 graph the reference takes column sums and the first walk step in float32 (_nam.py:28,33 on scipy's
    float32 CSR) and this implementation in float64 -- a 3e-7 relative difference of the NAM that cancellation turns
    into up to 6e-8 / 8e-8 of the largest entry (measured over the fixtures with the float64 oracle, which the GPU
    matches to 1e-10; 1.5e-6 where the ridge loop of c16 amplifies it).  On a float64 graph: 1e-12."""
    f64_graph = z['in_data'].dtype == np.float64
    if f64_graph:
        return dict(nam=1e-12, ncorrs=1e-12, namresid=1e-12)
    return dict(nam=1e-12, ncorrs=2e-7, namresid=5e-6 if 'ridge_loop' in str(name) else 2e-7)


def assert_odd_row_harmless(thr_a, nd_a, thr_b, nd_b, maxabs, what=''):
    """The one result field whose SHAPE may differ from the reference's: `np.arange(maxcorr/4, maxcorr, maxcorr/400)`
    (_association.py:101-102) has ceil(300 +- rounding) = 300 or 301 entries depending on the last bits of maxcorr =
    max(max|ncorrs|, 1e-3), and any path that is not bit-identical in max|ncorrs| (the reference's float32 first walk
    step against float64 here: ~1e-7) can land on the other side.  Returns the length of the common prefix after
    asserting that the odd row is the degenerate one: its threshold IS maxcorr (to 1e-6 relative; observed 4e-15), and
    nothing but possibly the one cell that attains max|ncorrs| lies above it (num_detected 0, or 1 when the threshold
    rounds to just below maxcorr -- fixture f12).  INTEGRATION.md "Result fields whose shape may differ by one row"."""
    la, lb = len(thr_a), len(thr_b)
    T = min(la, lb)
    assert abs(la - lb) <= 1 and T >= 300, (what, la, lb)
    if la != lb:
        thr, nd = (thr_a, nd_a) if la > lb else (thr_b, nd_b)
        maxcorr = max(float(maxabs), 1e-3)
        assert abs(float(np.asarray(thr)[-1]) - maxcorr) <= 1e-6 * maxcorr, (what, float(np.asarray(thr)[-1]), maxcorr)
        assert int(np.asarray(nd)[-1]) in (0, 1), (what, int(np.asarray(nd)[-1]))
    return T


def fdr_rows(frame, ref_fdrs, ref_ncorrs, what=''):
    """Common rows of a result's FDR table (DataFrame) and the oracle's (dict), the odd row checked (see above)."""
    return assert_odd_row_harmless(frame.threshold.values, frame.num_detected.values, ref_fdrs['threshold'],
                                   ref_fdrs['num_detected'], np.nanmax(np.abs(np.asarray(ref_ncorrs))), what)


# --------------------------------------------------------------------------------------
# running the product API on a fixture and comparing with the reference's outputs
def run_product(case, engine, **overrides):
    """cna_amd.tl.association on a golden case; returns (result or None, exception or None, warnings)."""
    import warnings
    import cna_amd as cna
    call = dict(case['call'])
    call.update(overrides)
    res, err = None, None
    with warnings.catch_warnings(record=True) as wl:
        warnings.simplefilter('always')
        try:
            res = cna.tl.association(case['data'], case['y'], case['sid_name'], batches=case['batches'],
                                     covs=case['covs'], donorids=case['donorids'], return_full=True,
                                     engine=engine, **call)
        except Exception as e:   # noqa: BLE001 - mirrored reference behaviour is checked by the caller
            err = e
    msgs = [str(w.message) for w in wl if issubclass(w.category, UserWarning)]
    return res, err, msgs


def assert_matches_golden(res, data, z, tol=1e-5, check_lazy=True, name=''):
    """Every result field of SURVEY.md §8a a20 against the reference's values.
    ints / masks exact; floats within `tol` relative (BASELINE.json: 1e-5) -- observed maxima over the 19
    fixtures are 2e-8 ... 7e-7 (the reference's float32 first walk step on float32 graphs).  One exception, the
    empirical FDRs: `fdr[t] = sum_p tails[p, t] / ranks[t] / P` is a ratio of COUNTS, and a null coefficient that sits
    within that 1e-7 of a threshold moves a count by one (c14: 3.8e-5 relative on an entry with ~26 000 counted
    outputs): 1e-4 there and on the per-cell FDR column that is looked up in the table."""
    import pytest
    assert int(res.k) == int(z['k'])
    assert np.array_equal(np.asarray(res.ks), z['ks'])
    assert int(res.r) == int(z['r'])
    assert np.array_equal(res.kept, z['kept'])
    assert float(res.p) == pytest.approx(float(z['p']), rel=1e-12)
    fl = golden_floors(z, name)
    assert_elementwise(res.ncorrs.values, z['ncorrs'], tol, fl['ncorrs'], 'ncorrs')
    assert relerr(res.M.values, z['M']) < tol
    assert relerr(res.nullminps, z['nullminps']) < tol
    assert relerr(res.namresid_svs.values, z['svs']) < tol
    assert relerr(res.namresid_varexp.values, z['varexp']) < tol
    assert relerr(np.asarray(res.yresid), z['yresid']) < tol
    assert float(res.r2) == pytest.approx(float(z['r2']), rel=tol)
    assert float(res.nullr2_mean) == pytest.approx(float(z['nullr2_mean']), rel=tol)
    assert float(res.nullr2_std) == pytest.approx(float(z['nullr2_std']), rel=tol)
    kk = int(z['k'])
    U, Uref = sign_align(res.namresid_sampleXpc.values, z['U'], kk)
    assert relerr(U, Uref) < tol
    assert relerr(np.abs(res.beta), np.abs(z['beta'])) < tol
    assert relerr(res.r2_perpc, z['r2_perpc']) < tol
    assert relerr(np.abs(res.yresid_hat), np.abs(z['yresid_hat'])) < tol
    np.testing.assert_allclose(data.obs['coef'].values, z['obs_coef'], rtol=0,
                               atol=tol * np.nanmax(np.abs(z['obs_coef'])), equal_nan=True)
    if 'fdr_fdr' in z:
        f = res.fdrs
        # (300 / 301 rows: the odd one must be the degenerate `threshold = maxcorr` row; the 5 % / 10 % thresholds and
        # the per-cell column -- EVERY cell -- are compared in full below, whichever side has it)
        T = assert_odd_row_harmless(f.threshold.values, f.num_detected.values, z['fdr_threshold'], z['fdr_num_detected'],
                                    np.nanmax(np.abs(z['ncorrs'])), name)
        assert relerr(f.threshold.values[:T], z['fdr_threshold'][:T]) < tol
        assert np.array_equal(f.num_detected.values[:T], z['fdr_num_detected'][:T])
        assert relerr(f.fdr.values[:T], z['fdr_fdr'][:T]) < tol * 10                 # ratio of counts, see above
        for key in ('fdr_5p_t', 'fdr_10p_t'):
            ref = float(z[key])
            got = getattr(res, key)
            if np.isnan(ref):
                assert got is None
            else:
                assert got == pytest.approx(ref, rel=tol)
        np.testing.assert_allclose(data.obs['coef_fdr'].values, z['obs_coef_fdr'], rtol=tol * 10, atol=1e-12)
    if check_lazy:
        assert list(res.nam.index) == z['nam_index'].tolist()
        assert_elementwise(res.nam.values, z['nam'], tol, fl['nam'], 'nam')
        assert_elementwise(res.namresid.values, z['namresid'], tol, fl['namresid'], 'namresid')
        V, Vref = sign_align(res.namresid_nbhdXpc.values, z['V'], kk)
        assert relerr(V, Vref) < tol
        assert res.nam.shape == z['nam'].shape and list(res.nam.columns) == list(data.obs.index[z['kept']])


def load_demo_case():
    """tests/golden/d01_demo_like.npz: BASELINE.json configs[0] -- the demo recipe (10 000 cells x 50 samples,
    makedata.ipynb) analysed as in demo.ipynb (y = case, covs = male, batches = batch), nsteps=3, Nnull=100."""
    z = np.load(os.path.join(GOLDEN_DIR, 'd01_demo_like.npz'))
    n = len(z['in_indptr']) - 1
    A = sp.csr_matrix((z['in_data'], z['in_indices'], z['in_indptr']), shape=(n, n))
    A.indptr = A.indptr.astype(np.int32)
    obs = pd.DataFrame({'id': z['in_sid']}, index=pd.Index(['cell_%d' % i for i in range(n)], name='cell'))
    N = 50
    samplem = pd.DataFrame(index=pd.Index(np.arange(N), name='id'))
    samplem['case'] = [0] * (N // 2) + [1] * (N - N // 2)
    q = int(2 * N / 8)
    samplem['male'] = [0] * q + [1] * q + [0] * q + [1] * (N - 3 * q)
    samplem['batch'] = np.tile(np.arange(5), N // 5)
    return dict(data=CellData(obs, A), y=samplem['case'].astype(float), covs=samplem[['male']].astype(float),
                batches=samplem['batch'], call=json.loads(z['call'].item()), z=z)


def assert_matches_demo(out, z, tol, obs=None):
    """out: dict-like with the oracle's keys (p, k, ks, r, kept, ncorrs, nullminps, svs, nam, namresid,
    fdrs{...}); cells-sized matrices are compared on the fixture's subsample of cells."""
    sub = z['sub']
    assert int(out['k']) == int(z['k']) and np.array_equal(out['ks'], z['ks']) and int(out['r']) == int(z['r'])
    assert np.array_equal(out['kept'], z['kept'])
    assert abs(float(out['p']) - float(z['p'])) < 1e-12
    assert relerr(out['ncorrs'], z['ncorrs']) < tol
    assert relerr(out['nullminps'], z['nullminps']) < tol * 10
    assert relerr(out['svs'], z['svs']) < tol
    assert relerr(np.asarray(out['nam'])[sub].T, z['nam_sub']) < tol
    assert relerr(np.asarray(out['namresid'])[sub].T, z['namresid_sub']) < tol
    f = out['fdrs']
    T = assert_odd_row_harmless(f['threshold'], f['num_detected'], z['fdr_threshold'], z['fdr_num_detected'],
                                np.nanmax(np.abs(z['ncorrs'])), 'demo')
    assert np.array_equal(np.asarray(f['num_detected'])[:T], z['fdr_num_detected'][:T])
    assert relerr(np.asarray(f['fdr'])[:T], z['fdr_fdr'][:T]) < tol * 10
    if obs is not None:
        np.testing.assert_allclose(obs['coef'], z['obs_coef'], rtol=0, atol=tol * np.nanmax(np.abs(z['obs_coef'])))
        np.testing.assert_allclose(obs['coef_fdr'], z['obs_coef_fdr'], rtol=tol * 10, atol=1e-12)


# --------------------------------------------------------------------------------------
# d02_config2: BASELINE.json configs[1] at FULL size through the reference (tests/golden/make_golden.py:run_config2)
def load_config2_case(name='d02_config2'):
    """(also d03_config3 = configs[2], same layout.)  The fixture holds results only; the inputs are regenerated here -- `synth.make_dataset(**dataset, builder='cpu')`,
    the host builder (cKDTree + scipy.sparse: the same graph on every machine of this image) -- and recognised by the
    digest of their CSR arrays and sample ids.  Returns dict(data, y, call, z, same_inputs)."""
    import hashlib
    from cna_amd import synth
    z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    ds = json.loads(z['dataset'].item())
    data, meta = synth.make_dataset(builder='cpu', **ds)
    A = data.obsp['connectivities']
    same = (synth.graph_digest(A) == z['graph_digest'].item() and
            hashlib.sha256(np.asarray(data.obs['id'].values, dtype=np.int64).tobytes()).hexdigest() == z['sid_digest'].item() and
            np.array_equal(meta['y'].values, z['in_y']))
    return dict(data=data, y=meta['y'], call=json.loads(z['call'].item()), z=z, same_inputs=bool(same))


def assert_matches_config2(out, z, tol, obs=None, floors=None, exact_counts=True):
    """out: dict with p, k, ks, r, n_kept, nullminps, svs, U, M, yresid, yresid_hat, r2, r2_perpc, nullr2_mean, nullr2_std,
    ncorrs (all cells), nam / namresid (cells x samples or None), fdrs{threshold, fdr, num_detected}, fdr_5p_t, fdr_10p_t.
    Integers exact, floats within `tol` relative (entry by entry with the golden floors for the cells-sized fields),
    empirical FDRs within 10 tol (ratios of counts), PCs up to sign for the first k."""
    sub = z['sub']
    floors = floors or dict(nam=1e-12, ncorrs=2e-7, namresid=2e-7)
    assert int(out['k']) == int(z['k']) and np.array_equal(out['ks'], z['ks']) and int(out['r']) == int(z['r'])
    assert int(out['n_kept']) == int(z['n_kept'])
    assert abs(float(out['p']) - float(z['p'])) < 1e-12
    assert relerr(out['nullminps'], z['nullminps']) < tol * 10
    assert relerr(out['svs'], z['svs']) < tol
    assert relerr(out['M'], z['M']) < tol
    assert relerr(out['yresid'], z['yresid']) < tol and relerr(out['yresid_hat'], z['yresid_hat']) < tol
    for key in ('r2', 'nullr2_mean', 'nullr2_std'):
        assert abs(float(out[key]) - float(z[key])) <= tol * abs(float(z[key])), key
    assert relerr(out['r2_perpc'], z['r2_perpc']) < tol
    kk = int(z['k'])
    a, b = sign_align(out['U'], z['U'], kk)
    assert relerr(a, b) < tol
    nc = np.asarray(out['ncorrs'])
    assert_elementwise(nc[sub], z['ncorrs_sub'], tol, floors['ncorrs'], 'ncorrs (every 100th cell)')
    assert abs(np.abs(nc).max() - float(z['ncorrs_absmax'])) <= tol * float(z['ncorrs_absmax'])
    if out.get('nam') is not None:
        assert_elementwise(np.asarray(out['nam'])[sub].T, z['nam_sub'], tol, floors['nam'], 'nam (every 100th cell)')
    if out.get('namresid') is not None:
        assert_elementwise(np.asarray(out['namresid'])[sub].T, z['namresid_sub'], tol, floors['namresid'], 'namresid (every 100th cell)')
    f = out['fdrs']
    T = assert_odd_row_harmless(f['threshold'], f['num_detected'], z['fdr_threshold'], z['fdr_num_detected'],
                                float(z['ncorrs_absmax']), 'config')
    assert relerr(np.asarray(f['threshold'])[:T], z['fdr_threshold'][:T]) < tol
    if exact_counts:
        assert np.array_equal(np.asarray(f['num_detected'])[:T], z['fdr_num_detected'][:T])
    else:
        assert np.abs(np.asarray(f['num_detected'])[:T] - z['fdr_num_detected'][:T]).max() <= 3
    np.testing.assert_allclose(np.asarray(f['fdr'])[:T], z['fdr_fdr'][:T], rtol=tol * 10, atol=1e-12, equal_nan=True)
    for key in ('fdr_5p_t', 'fdr_10p_t'):
        ref = float(z[key])
        if np.isnan(ref):
            assert out[key] is None
        else:
            assert out[key] is not None and abs(out[key] - ref) <= tol * abs(ref)
    if obs is not None:
        np.testing.assert_allclose(np.asarray(obs['coef'])[sub], z['obs_coef_sub'], rtol=0, atol=tol * float(z['ncorrs_absmax']))
        np.testing.assert_allclose(np.asarray(obs['coef_fdr'])[sub], z['obs_coef_fdr_sub'], rtol=tol * 10, atol=1e-12)
        if exact_counts:
            assert int((np.asarray(obs['coef_fdr']) < 1).sum()) == int(z['obs_coef_fdr_below_1'])
