"""The fixed-shape analysis in two library calls (cna_amd/tools/_fast.py, csrc/assoc.hip: cna_assoc_begin / cna_assoc_finish)
against the general path of tools/_association.py: the same entry points in the same order, so every result field, both
data.obs columns, the warnings and numpy's generator state must come out bit for bit the same -- and, through
helpers.assert_matches_golden, equal to the reference's own outputs.  Run with -m gpu on an MI355X."""
import warnings

import numpy as np
import pandas as pd
import pytest

from helpers import load_case, run_product, assert_matches_golden, load_config2_case, assert_matches_config2

pytestmark = pytest.mark.gpu

# fixtures whose call has the shape (nsteps given, no batches / donor groups, a seed, local test on); which of them really
# take the two-call path depends on the sample-level inputs (y indexed by exactly the data's samples)
SHAPED = ['c01_plain_f32', 'c05_ks_f64', 'c06_nnull_cap', 'c10_categorical_ids', 'c11_string_ids_null_y', 'c13_zero_variance',
          'c17_low_sample_size', 'c09_y_nan_extra_reordered', 'f08_messy', 'f16_messy', 'f19_messy', 'f23_messy', 'f25_messy',
          'f26_messy', 'f33_constant_phenotype',
          # nsteps=None: the reference's stop rule (evaluated on the device; the selection pass collects the verdict)
          'c02_covs_autostop', 'c14_selfweight_autostop_unsorted', 'c20_unused_category_autostop', 'f09_messy', 'f13_messy']
MUST_TAKE = {'c01_plain_f32', 'c05_ks_f64', 'c06_nnull_cap', 'c10_categorical_ids', 'c11_string_ids_null_y', 'c17_low_sample_size',
             'c02_covs_autostop', 'c14_selfweight_autostop_unsorted'}


@pytest.fixture(scope='module')
def eng():
    from cna_amd.engine import get_engine
    e = get_engine()
    keep = e.reuse_nam
    e.reuse_nam = False
    yield e
    e.reuse_nam = keep


@pytest.fixture()
def fast():
    from cna_amd.tools import _fast
    keep = _fast.ENABLED
    yield _fast
    _fast.ENABLED = keep


FIELDS = ('p', 'k', 'r', 'r2', 'nullr2_mean', 'nullr2_std', 'fdr_5p_t', 'fdr_10p_t')
ARRAYS = ('ks', 'kept', 'nullminps', 'yresid_hat', 'r2_perpc', 'beta')
FRAMES = ('M', 'ncorrs', 'fdrs', 'namresid_sampleXpc', 'namresid_svs', 'namresid_varexp', 'yresid', 'nam', 'namresid',
          'namresid_nbhdXpc')


def same_results(a, b):
    for f in FIELDS:
        x, y = getattr(a, f), getattr(b, f)
        assert (x is None and y is None) or x == y or (x != x and y != y), (f, x, y)
    for f in ARRAYS:
        np.testing.assert_array_equal(np.asarray(getattr(a, f)), np.asarray(getattr(b, f)), err_msg=f)
    for f in FRAMES:
        x, y = getattr(a, f), getattr(b, f)
        assert type(x) is type(y), f
        np.testing.assert_array_equal(np.asarray(x), np.asarray(y), err_msg=f)
        assert list(x.index) == list(y.index), f
        if isinstance(x, pd.DataFrame):
            assert list(x.columns) == list(y.columns), f
    assert sorted(k for k in vars(a) if not k.startswith('_')) == sorted(k for k in vars(b) if not k.startswith('_'))


def both_paths(fast, case, eng, **over):
    """-> ((res, err, warnings, obs columns, RNG state) of the general path, the same of the two-call path, taken?)"""
    out = []
    for on in (False, True):
        fast.ENABLED = on
        for key in ('coef', 'coef_fdr'):
            if key in case['data'].obs:
                del case['data'].obs[key]
        before = dict(fast.stats)
        np.random.seed(12345)
        res, err, msgs = run_product(case, eng, **over)
        if res is not None:
            res.materialize()
        obs = {k: case['data'].obs[k].values.copy() for k in ('coef', 'coef_fdr') if k in case['data'].obs}
        out.append((res, err, msgs, obs, np.random.get_state()))
        taken = fast.stats['taken'] - before['taken']
    return out[0], out[1], bool(taken)


@pytest.mark.parametrize('name', SHAPED)
def test_two_call_path_equals_general_path_and_reference(eng, fast, name):
    case = load_case(name)
    g, f, taken = both_paths(fast, case, eng)
    assert taken or name not in MUST_TAKE
    assert (g[1] is None) == (f[1] is None)
    if g[1] is not None:
        assert type(g[1]) is type(f[1]) and str(g[1]) == str(f[1])
    else:
        same_results(g[0], f[0])
        if not case['z']['raised'].item():
            assert_matches_golden(f[0], case['data'], case['z'], name=name)
    assert g[2] == f[2]                                         # same warnings, same order
    assert g[3].keys() == f[3].keys()
    for k in g[3]:
        np.testing.assert_array_equal(g[3][k], f[3][k])
    for x, y in zip(g[4], f[4]):                                # numpy's generator where the reference leaves it
        np.testing.assert_array_equal(np.asarray(x), np.asarray(y))


def _synthetic(n, N, covs=0, seed=3, k=15):
    from cna_amd import synth
    return synth.make_dataset(n, N, k=k, seed=seed, n_covs=covs)


@pytest.mark.parametrize('n,N,covs,nan_at', [(20000, 80, 0, None), (20000, 80, 2, None), (20000, 80, 2, 7), (20000, 130, 0, 3),
                                             (160000, 80, 0, None), (160000, 96, 0, None), (160000, 70, 3, None)])
def test_covariates_subsets_and_the_walk_by_product(eng, fast, n, N, covs, nan_at):
    """Covariates (M = I - C.W inside the selection pass), a sample dropped by a NaN phenotype (a column map), and blocks
    of 150 000 cells or more with a wide sample axis (the last walk step leaves the selection pass's results)."""
    import cna_amd as cna
    data, meta = _synthetic(n, N, covs)
    y = meta['y'].copy()
    if nan_at is not None:
        y.iloc[nan_at] = np.nan
    kw = dict(nsteps=3, Nnull=200, seed=5, covs=meta.get('covs') if covs else None)
    res = {}
    # (a graph of 100 000 cells or more is analysed in the caller's cell order first and adopts the device order in a later
    # call: the Gram matrix sums the cells in device order, so the comparison starts once that order is in place)
    fast.ENABLED = False
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        cna.tl.association(data, y, 'id', engine=eng, **kw)
        eng.wait_reorder()
        cna.tl.association(data, y, 'id', engine=eng, **kw)
    assert not eng.reorder_pending()
    for on in (False, True, True):
        fast.ENABLED = on
        before = fast.stats['taken']
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            r = cna.tl.association(data, y, 'id', return_full=True, engine=eng, **kw)
        r.materialize()
        res[on] = (r, data.obs['coef'].values.copy(), data.obs['coef_fdr'].values.copy(), fast.stats['taken'] - before)
    assert res[False][3] == 0 and res[True][3] == 1
    same_results(res[False][0], res[True][0])
    np.testing.assert_array_equal(res[False][1], res[True][1])
    np.testing.assert_array_equal(res[False][2], res[True][2])
    assert res[True][0].nam.shape == (N - (nan_at is not None), n)


def test_return_value_and_existing_key_warning(eng, fast):
    import cna_amd as cna
    data, meta = _synthetic(15000, 40)
    kw = dict(nsteps=3, Nnull=100, seed=1, engine=eng)
    fast.ENABLED = False
    p0 = cna.tl.association(data, meta['y'], 'id', **kw)
    fast.ENABLED = True
    before = fast.stats['taken']
    held = data.obs['coef']                                     # the previous column object must stay as it was
    held_values = held.values.copy()
    with warnings.catch_warnings(record=True) as wl:
        warnings.simplefilter('always')
        p1 = cna.tl.association(data, meta['y'], 'id', **kw)
    assert fast.stats['taken'] == before + 1
    assert isinstance(p1, float) and p1 == p0
    assert any("Key 'coef' already exists in data.obs. Overwriting." in str(w.message) for w in wl)
    np.testing.assert_array_equal(held.values, held_values)
    np.testing.assert_array_equal(data.obs['coef'].values, held_values)


def test_eigen_solver_steps_aside(eng, fast):
    """ks beyond a quarter of the samples: cna_gram_pcs_tests does not accept its own pairs (status NEED_PCS), LAPACK's
    are supplied from Python; same numbers as the general path, which takes the same detour."""
    import cna_amd as cna
    data, meta = _synthetic(12000, 30)
    kw = dict(nsteps=2, Nnull=100, seed=2, ks=[3, 9], engine=eng, return_full=True)
    out = {}
    for on in (False, True):
        fast.ENABLED = on
        before = dict(fast.stats)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            out[on] = cna.tl.association(data, meta['y'], 'id', **kw).materialize()
        if on:
            assert fast.stats['taken'] == before['taken'] + 1 and fast.stats['need_pcs'] == before['need_pcs'] + 1
    same_results(out[False], out[True])


def test_zero_variance_cells_go_to_the_general_path(eng, fast):
    """A cell whose NAM entries are constant over the samples: the selection pass reports it (status GENERAL), nothing of
    the attempt is left in data.obs or in numpy's generator, and the general path drops the cell as the reference does."""
    case = load_case('c13_zero_variance')
    g, f, taken = both_paths(fast, case, eng)
    assert not taken and fast.stats['general'] >= 1
    same_results(g[0], f[0])
    assert (~f[0].kept).sum() > 0


def test_stale_graph_is_noticed(fast):
    """Unpinned graph edited in place between two calls: the optimistic attempt on the resident copy is dropped before
    anything gets out, and the call returns what a fresh engine returns for the edited graph."""
    import cna_amd as cna
    from cna_amd.engine import Engine
    data, meta = _synthetic(15000, 40, seed=11)
    kw = dict(nsteps=3, Nnull=100, seed=3, return_full=True)
    e1 = Engine()
    e1.reuse_nam = False
    fast.ENABLED = True
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        cna.tl.association(data, meta['y'], 'id', engine=e1, **kw)
        before = dict(fast.stats)
        cna.tl.association(data, meta['y'], 'id', engine=e1, **kw)
        assert fast.stats['taken'] == before['taken'] + 1
        A = data.obsp['connectivities']
        A.data[: len(A.data) // 2] *= 0.5                       # in place: same object, same buffers
        before = dict(fast.stats)
        r1 = cna.tl.association(data, meta['y'], 'id', engine=e1, **kw).materialize()
        assert fast.stats['stale'] == before['stale'] + 1
        e2 = Engine()
        e2.reuse_nam = False
        fast.ENABLED = False
        r2 = cna.tl.association(data, meta['y'], 'id', engine=e2, **kw).materialize()
    same_results(r2, r1)
    e1.close()
    e2.close()


def test_an_exception_leaves_no_trace(eng, fast, monkeypatch):
    import cna_amd as cna
    from cna_amd.tools import _association as A_
    data, meta = _synthetic(15000, 40, seed=12)
    kw = dict(nsteps=3, Nnull=100, seed=4, engine=eng)
    fast.ENABLED = True
    data.obs['coef'] = np.arange(len(data.obs), dtype=float)
    cna.tl.association(data, meta['y'], 'id', key_added='other', **kw)          # graph resident from here on
    calls = []

    def boom(*a, **k):
        calls.append(1)
        raise IndexError('index 0 is out of bounds for axis 0 with size 0')
    monkeypatch.setattr(A_, '_fdr_tables', boom)
    with pytest.raises(IndexError):
        cna.tl.association(data, meta['y'], 'id', return_full=True, **kw)
    assert calls
    # (the reference seeds and draws before it fails there: the generator is where its draw leaves it)
    got = np.random.get_state()
    np.random.seed(4)
    np.random.randn(40, 100)
    for a_, b_ in zip(got, np.random.get_state()):
        np.testing.assert_array_equal(np.asarray(a_), np.asarray(b_))
    np.testing.assert_array_equal(data.obs['coef'].values, np.arange(len(data.obs), dtype=float))
    assert 'coef_fdr' not in data.obs
    monkeypatch.undo()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        p = cna.tl.association(data, meta['y'], 'id', **kw)
    fast.ENABLED = False
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        assert cna.tl.association(data, meta['y'], 'id', **kw) == p


def test_errors_inside_and_after_the_library_call(eng, fast, monkeypatch):
    """The reference's own late errors through the two-call path -- every p-value NaN (no degrees of freedom left:
    np.nanargmin's ValueError) -- and a failure injected right after cna_assoc_finish: data.obs as it was, nothing pending
    on the engine, the next call returns what it returned before."""
    import cna_amd as cna
    data, meta = _synthetic(12000, 24, seed=15)
    kw = dict(nsteps=3, Nnull=100, seed=8, engine=eng)
    fast.ENABLED = True
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        p0 = cna.tl.association(data, meta['y'], 'id', **kw)
        coef, fdr = data.obs['coef'].values.copy(), data.obs['coef_fdr'].values.copy()
        before = dict(fast.stats)
        with pytest.raises(ValueError, match='All-NaN slice'):
            cna.tl.association(data, meta['y'], 'id', ks=[23], **kw)
        np.testing.assert_array_equal(data.obs['coef'].values, coef)
        np.testing.assert_array_equal(data.obs['coef_fdr'].values, fdr)
        real = eng.assoc_finish

        def boom(*a, **k):
            real(*a, **k)
            raise FloatingPointError('injected')
        monkeypatch.setattr(eng, 'assoc_finish', boom)
        with pytest.raises(FloatingPointError):
            cna.tl.association(data, meta['y'], 'id', **kw)
        monkeypatch.setattr(eng, 'assoc_finish', real)
        np.testing.assert_array_equal(data.obs['coef'].values, coef)
        np.testing.assert_array_equal(data.obs['coef_fdr'].values, fdr)
        assert cna.tl.association(data, meta['y'], 'id', **kw) == p0
        assert fast.stats['taken'] == before['taken'] + 1
        fast.ENABLED = False
        assert cna.tl.association(data, meta['y'], 'id', **kw) == p0


def test_integer_null_that_gives_up_is_redone_in_f64(eng, fast, monkeypatch):
    """CNA_I8_QCAP=8 makes the integer local null overflow its recheck queue: its status word comes back with the sums, and
    whoever collects the pass (cna_null_local_fetch; inside cna_assoc_finish on the two-call path) reruns it on the f64
    kernel and looks the per-cell FDR column up again -- the table that followed the abandoned pass on the device is void.
    Same results as the undisturbed call, through both paths."""
    import cna_amd as cna
    data, meta = _synthetic(20000, 50, seed=17)
    kw = dict(nsteps=3, Nnull=640, seed=9, engine=eng, return_full=True)
    out = {}
    for on in (False, True):
        fast.ENABLED = on
        for cap in (None, '8'):
            if cap:
                monkeypatch.setenv('CNA_I8_QCAP', cap)
            else:
                monkeypatch.delenv('CNA_I8_QCAP', raising=False)
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                res = cna.tl.association(data, meta['y'], 'id', **kw)
            used, rechecked, fallback = eng.null_local_i8_stats()
            assert used and fallback == bool(cap)
            out[(on, cap)] = (res.p, res.fdrs.values.copy(), data.obs['coef'].values.copy(), data.obs['coef_fdr'].values.copy())
    monkeypatch.delenv('CNA_I8_QCAP', raising=False)
    base = out[(False, None)]
    for key, got in out.items():
        assert got[0] == base[0], key
        for a, b in zip(got[1:], base[1:]):
            np.testing.assert_array_equal(a, b, err_msg=str(key))


def test_nam_cache_skips_the_walk(fast):
    """engine.reuse_nam (the default for users): a second phenotype on the resident dataset queues no walk step."""
    import cna_amd as cna
    from cna_amd.engine import Engine
    data, meta = _synthetic(30000, 60, seed=13)
    y2 = pd.Series(np.random.RandomState(5).randn(60), index=meta['y'].index)
    e = Engine()
    e.reuse_nam = True
    fast.ENABLED = True
    kw = dict(nsteps=3, Nnull=100, seed=6, engine=e, return_full=True)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        cna.tl.association(data, meta['y'], 'id', **kw)
        a1 = cna.tl.association(data, meta['y'], 'id', **kw).materialize()
        e.prof_reset()
        e.prof_enable(True)
        before = fast.stats['taken']
        a2 = cna.tl.association(data, y2, 'id', **kw).materialize()
        e.sync()
        e.prof_enable(False)
        assert fast.stats['taken'] == before + 1
        assert not any(k.startswith('nam_') for k in e.prof()), e.prof()
        e.reuse_nam = False
        fast.ENABLED = False
        b2 = cna.tl.association(data, y2, 'id', **kw).materialize()
    same_results(b2, a2)
    np.testing.assert_array_equal(a1.nam.values, a2.nam.values)
    e.close()


def test_one_library_call(eng):
    """cna_assoc_run = begin + finish: the whole fixed-shape analysis behind ONE entry point, checked against the two-call
    sequence (same outputs)."""
    import cna_amd as cna
    from cna_amd.tools._stats import native_draw_start
    data, meta = _synthetic(15000, 40, seed=14)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        cna.tl.association(data, meta['y'], 'id', nsteps=3, Nnull=100, seed=7, engine=eng)      # graph + codes resident
    yv = meta['y'].values
    y_std = (yv - yv.mean()) / yv.std()
    outs = []
    for one_call in (False, True):
        state = np.random.get_state()
        draw = native_draw_start(None, y_std, 100, 7, single_level=True)
        eng.lib.cna_restart_nam(eng.h)
        eng.nam_epoch += 1
        if not one_call:
            eng.assoc_begin(3, None)
        o = eng.assoc_finish(y_std, np.eye(40), np.array([1, 2]), 100, draw.table, draw_pending=True,
                             run_steps=3 if one_call else None)
        draw.abandon()
        np.random.set_state(state)
        assert o['status'] == 0
        outs.append({k_: (np.array(v) if isinstance(v, np.ndarray) else v) for k_, v in o.items()})   # (views of the call's block)
    for key in ('G', 'U', 'minp', 'r2', 'kidx', 'thr', 'fdr', 'tail_sums', 'ranks', 'num_detected'):
        np.testing.assert_array_equal(outs[0][key], outs[1][key], err_msg=key)
    np.testing.assert_array_equal(np.array(outs[1]['coef']), data.obs['coef'].values)


def test_config2_through_the_two_call_path(eng, fast):
    """BASELINE.json configs[1] at full size (200k x 50, Nnull 1000) against the reference's own run -- the fixture
    tests/golden/d02_config2.npz -- through the two-call path."""
    import cna_amd as cna
    case = load_config2_case()
    if not case['same_inputs']:
        pytest.skip('synthetic inputs differ from the fixture (another numpy / scipy build)')
    data, y, z = case['data'], case['y'], case['z']
    fast.ENABLED = True
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        cna.tl.association(data, y, 'id', engine=eng, **case['call'])
        before = fast.stats['taken']
        res = cna.tl.association(data, y, 'id', engine=eng, return_full=True, **case['call'])
    assert fast.stats['taken'] == before + 1
    out = dict(p=res.p, k=res.k, ks=res.ks, r=res.r, n_kept=int(res.kept.sum()), nullminps=res.nullminps,
               svs=res.namresid_svs.values, U=res.namresid_sampleXpc.values, M=res.M.values, yresid=np.asarray(res.yresid),
               yresid_hat=res.yresid_hat, r2=res.r2, r2_perpc=res.r2_perpc, nullr2_mean=res.nullr2_mean,
               nullr2_std=res.nullr2_std, ncorrs=data.obs['coef'].values, nam=None, namresid=None,
               fdrs=dict(threshold=res.fdrs.threshold.values, fdr=res.fdrs.fdr.values, num_detected=res.fdrs.num_detected.values),
               fdr_5p_t=res.fdr_5p_t, fdr_10p_t=res.fdr_10p_t)
    assert_matches_config2(out, z, 1e-5, obs=dict(coef=data.obs['coef'].values, coef_fdr=data.obs['coef_fdr'].values))
