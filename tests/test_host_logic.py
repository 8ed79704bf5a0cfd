"""Host-side logic of cna_amd.tools on CPU: the real orchestration code driven through a
test double of the device engine (tests/fake_engine.py, oracle arithmetic).  Checks that
input validation, sample reindexing/filtering, QC, the ridge schedule, the permutation
null, result fields, warnings and progress text reproduce the reference's golden outputs.
The HIP kernels themselves are checked on the GPU (tests/test_gpu_parity.py)."""
import contextlib
import io
import re

import numpy as np
import pandas as pd
import pytest

import cna_amd as cna
from fake_engine import FakeEngine
from helpers import messy_names, golden_names, load_case, run_product, assert_matches_golden, relerr

NAMES = golden_names()


ORDERS = [(n, None) for n in NAMES] + [(n, 'rcm') for n in NAMES] + [(n, None) for n in messy_names()] + \
         [(n, 'random') for n in ('c01_plain_f32', 'c03_covs_batches', 'c12_batchy_qc', 'c13_zero_variance')]


@pytest.mark.parametrize('name,order', ORDERS)
def test_association_host_logic(name, order):
    """order: how the engine numbers the cells internally (tests/fake_engine.py) -- results must not
    depend on it."""
    case = load_case(name)
    z = case['z']
    res, err, msgs = run_product(case, FakeEngine(order=order))
    if z['raised'].item():
        assert err is not None and type(err).__name__ == z['raised'].item().split(':')[0]
        assert str(err) == z['raised'].item().split(': ', 1)[1]
        if 'obs_coef' in z:
            np.testing.assert_allclose(case['data'].obs['coef'].values, z['obs_coef'], rtol=0,
                                       atol=1e-5 * np.nanmax(np.abs(z['obs_coef'])), equal_nan=True)
        else:
            assert 'coef' not in case['data'].obs
        return
    assert err is None, repr(err)
    assert_matches_golden(res, case['data'], z, name=name)
    import json
    # (pandas' own remark about the reference's misaligned boolean indexer -- "Boolean Series key will be reindexed" -- is
    # an artefact of its implementation, not a message of the analysis)
    skip = ('already exists', 'Boolean Series key')
    ref_msgs = [m for m in json.loads(z['warnings'].item()) if not any(t in m for t in skip)]
    assert [m for m in msgs if not any(t in m for t in skip)] == ref_msgs


def _numbers(text):
    return [float(x) for x in re.findall(r'[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?|nan', text)]


@pytest.mark.parametrize('name', ['c01_plain_f32', 'c02_covs_autostop', 'c03_covs_batches', 'c12_batchy_qc',
                                  'c15_ridges_custom'])
def test_progress_text(name):
    case = load_case(name)
    ref = case['z']['stdout'].item()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        cna.tl.association(case['data'], case['y'], case['sid_name'], batches=case['batches'], covs=case['covs'],
                           donorids=case['donorids'], show_progress=True, engine=FakeEngine(), **case['call'])
    got = buf.getvalue()
    strip = lambda s: re.sub(r'[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?|nan', '#', s)
    assert strip(got) == strip(ref)
    a, b = np.array(_numbers(got)), np.array(_numbers(ref))
    np.testing.assert_allclose(a, b, rtol=1e-5, equal_nan=True)


@pytest.mark.parametrize('ncov', [0, 2])
def test_schedule_of_large_inputs_holds_the_last_walk_step_back(monkeypatch, ncov):
    """Large inputs, three or more steps, more than 64 samples: the walk's last step is queued after validation and
    planning, with the call's own standardised phenotype as a hint for the selection pass when nothing is filtered or
    regressed out (tools/_nam.py:_nam_device, _association.py:compute_nam_and_reindex).  Through the test double: same
    calls in the same order with and without the schedule except for the hint, same results, and the oracle's."""
    from cna_amd import synth
    from cna_amd.tools import _association as A
    from oracle import cna_oracle as orc
    data, meta = synth.make_dataset(1500, 70, k=10, seed=5, n_covs=ncov)
    kw = dict(covs=meta['covs'] if ncov else None, nsteps=3, Nnull=100, seed=3)
    out = {}
    for defer in (True, False):
        monkeypatch.setattr(A, '_DEFER_LAST_CELLS', 0 if defer else 10 ** 9)
        eng = FakeEngine()
        res = cna.tl.association(data, meta['y'], 'id', return_full=True, engine=eng, **kw)
        out[defer] = (res, [c[0] for c in eng.calls], [c for c in eng.calls if c[0] == 'nam_select_hint'])
    hints = out[True][2]
    assert not out[False][2] and len(hints) == (0 if ncov else 1)
    if not ncov:
        yv = meta['y'].values
        np.testing.assert_array_equal(hints[0][1], (yv - yv.mean()) / yv.std())
    steps = [c for c in out[True][1] if c in ('nam_step', 'nam_select_hint')]
    assert steps == (['nam_step'] * 3 if ncov else ['nam_step', 'nam_step', 'nam_select_hint', 'nam_step'])
    a, b = out[True][0], out[False][0]
    assert a.p == b.p and a.k == b.k
    np.testing.assert_array_equal(a.ncorrs.values, b.ncorrs.values)
    np.testing.assert_array_equal(a.fdrs.values, b.fdrs.values)
    ref = orc.association(data, meta['y'], 'id', mode='f64', **kw)
    assert int(a.k) == int(ref['k']) and a.p == pytest.approx(ref['p'], rel=1e-9)
    np.testing.assert_allclose(a.ncorrs.values, ref['ncorrs'], rtol=1e-9, atol=1e-13)


def test_ridge_loop_runs_the_whole_schedule():
    case = load_case('c16_ridge_loop')
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        cna.tl.association(case['data'], case['y'], case['sid_name'], batches=case['batches'], covs=case['covs'],
                           show_progress=True, engine=FakeEngine(), **case['call'])
    ref_lines = [l for l in case['z']['stdout'].item().splitlines() if 'with ridge' in l]
    got_lines = [l for l in buf.getvalue().splitlines() if 'with ridge' in l]
    assert len(got_lines) == len(ref_lines) == 11
    # the last ridges regress the batch effect down to rounding noise: compare the stable ones
    for g, r in list(zip(got_lines, ref_lines))[:8]:
        assert _numbers(g)[0] == _numbers(r)[0]
        assert _numbers(g)[1] == pytest.approx(_numbers(r)[1], rel=1e-6)


def test_tl_nam_and_svd_and_diffuse():
    case = load_case('c12_batchy_qc')
    z = case['z']
    NAM, keep = cna.tl.nam(case['data'], case['sid_name'], batches=case['batches'], nsteps=3, engine=FakeEngine())
    assert isinstance(NAM, pd.DataFrame) and NAM.index.name == case['sid_name']
    assert np.array_equal(keep, z['tlnam_keep'])
    assert relerr(NAM.values, z['tlnam']) < 1e-5
    assert list(NAM.columns) == list(case['data'].obs.index[keep])

    case = load_case('c01_plain_f32')
    z = case['z']
    eng = FakeEngine()
    out = cna.tl.diffuse(case['data'], z['diffuse_in'], 2, engine=eng)
    assert isinstance(out, np.ndarray) and relerr(out, z['diffuse_out_2']) < 1e-6
    out = cna.tl.diffuse(case['data'], z['diffuse_in'], 2, self_weight=0.5, engine=eng)
    assert relerr(out, z['diffuse_out_2_sw05']) < 1e-6
    steps = list(cna.tl.diffuse_stepwise(case['data'], pd.DataFrame(z['diffuse_in']), maxnsteps=2, engine=eng))
    assert len(steps) == 2 and isinstance(steps[0], pd.DataFrame)
    nam_df = pd.DataFrame(z['nam'], index=z['nam_index'].tolist())
    U, svs, V = cna.tl.svd_nam(nam_df, engine=eng)
    assert relerr(svs.values, z['svd_svs']) < 1e-6
    assert list(U.columns[:2]) == ['PC1', 'PC2'] and list(U.index) == list(nam_df.index)
    assert V.shape == z['svd_V'].shape

    case = load_case('c14_selfweight_autostop_unsorted')
    NAM, _ = cna.tl.nam(case['data'], case['sid_name'], nsteps=2, self_weight=2, engine=FakeEngine())
    assert relerr(NAM.values, case['z']['tlnam_sw2']) < 1e-5


def test_autostop_step_flags():
    """nsteps=None: every step from the 3rd on must be allowed to be the last one."""
    case = load_case('c02_covs_autostop')
    eng = FakeEngine()
    cna.tl.association(case['data'], case['y'], case['sid_name'], covs=case['covs'], engine=eng, **case['call'])
    steps = [c for c in eng.calls if c[0] == 'nam_step']
    assert len(steps) == case['z']['stdout'].item().count('median kurtosis')
    assert all(c[1] for c in steps)                    # kurtosis needed for the stop rule
    assert [c[3] for c in steps] == [False, False] + [True] * (len(steps) - 2)


def test_input_validation_matches_reference_messages():
    case = load_case('c01_plain_f32')
    d, y = case['data'], case['y']
    eng = FakeEngine()
    with pytest.raises(TypeError, match="'y' must be a pandas Series"):
        cna.tl.association(d, y.values, 'id', engine=eng)
    with pytest.raises(TypeError, match="'covs' must be a pandas DataFrame"):
        cna.tl.association(d, y, 'id', covs=y, engine=eng)
    with pytest.raises(TypeError, match="'batches' must be a pandas Series"):
        cna.tl.association(d, y, 'id', batches=y.values, engine=eng)
    with pytest.raises(ValueError, match="contains values not present in the index of 'y'"):
        cna.tl.association(d, y.iloc[:-1], 'id', engine=eng)
    with pytest.raises(ValueError, match='do not currently support conditioning on batch'):
        cna.tl.association(d, y, 'id', batches=y, donorids=y, engine=eng)
    few = y.copy()
    few.iloc[5:] = np.nan
    with pytest.raises(ValueError, match='fewer than 10 samples'):
        cna.tl.association(d, few, 'id', engine=eng)
    with pytest.raises(ValueError, match='Maximum number of PCs plus number of covariates'):
        cna.tl.association(d, y, 'id', ks=[20], nsteps=1, engine=eng)
    with pytest.raises(TypeError, match="unexpected keyword argument 'self_weight'"):
        cna.tl.association(d, y, 'id', self_weight=2, engine=eng)
    with pytest.warns(UserWarning, match="Key 'coef' already exists"):
        cna.tl.association(d, y, 'id', nsteps=1, Nnull=10, seed=0, engine=eng)
        cna.tl.association(d, y, 'id', nsteps=1, Nnull=10, seed=0, engine=eng)


def test_lazy_fields_guard_against_stale_device_state():
    case = load_case('c01_plain_f32')
    eng = FakeEngine()
    res = cna.tl.association(case['data'], case['y'], 'id', return_full=True, engine=eng, **case['call'])
    _ = res.namresid                    # read while resident: fine
    cna.tl.association(case['data'], case['y'], 'id', engine=eng, **case['call'])
    with pytest.raises(RuntimeError, match='later cna_amd call'):
        _ = res.namresid_nbhdXpc
    with pytest.raises(RuntimeError, match='later cna_amd call'):
        _ = res.nam


def test_obs_to_sample():
    obs = pd.DataFrame({'id': [2, 2, 1, 1, 1], 'age': [10., 10., 30., 30., 30.], 'x': [1., 3., 0., 3., 6.]})
    d = type('D', (), {'obs': obs})()
    out = cna.ut.obs_to_sample(d, ['age', 'x'], 'id')
    assert list(out.index) == [2, 1]
    assert out.loc[1, 'x'] == 3.0 and out.loc[2, 'age'] == 10.0
    assert cna.ut.obs_to_sample(d, 'x', 'id', aggregate='max').loc[1, 'x'] == 6.0


def test_sample_code_memo_sees_in_place_edits():
    from cna_amd.tools._nam import sample_codes_cached
    ids = pd.Series(np.array([3, 1, 2, 1, 3, 3], dtype=np.int64))
    c1, l1, n1, t1 = sample_codes_cached(ids)
    c2, l2, n2, t2 = sample_codes_cached(ids)
    assert t1 == t2 and c1 is c2 and list(l1) == [1, 2, 3] and list(n1) == [2, 1, 3]
    ids.values[0] = 2                                   # same buffer, different content
    c3, l3, n3, t3 = sample_codes_cached(ids)
    assert t3 != t1 and list(c3) == [1, 0, 1, 0, 2, 2] and list(n3) == [2, 2, 2]
    ids.values[[0, 1]] = ids.values[[1, 0]]             # same buffer, same multiset of values, other order
    c4, l4, n4, t4 = sample_codes_cached(ids)
    assert t4 != t3 and list(c4) == [0, 1, 1, 0, 2, 2] and list(n4) == [2, 2, 2]
    strs = pd.Series(['b', 'a', 'b'])
    assert sample_codes_cached(strs)[3] is None         # object columns are never memoised
    cat = pd.Series(pd.Categorical(['x', 'z', 'x'], categories=['x', 'y', 'z']))
    cc, cl, cn, ct = sample_codes_cached(cat)
    assert list(cl) == ['x', 'y', 'z'] and list(cn) == [2, 0, 1] and ct is not None


def test_sample_code_memo_deferred_check():
    """defer=True hands back the memo when buffer and layout match and hashes the content on the checker
    thread; confirm_codes() tells whether the memo was right and drops it when not."""
    from cna_amd.tools._nam import sample_codes_cached, confirm_codes
    ids = pd.Series(np.array([5, 1, 2, 1, 5, 5, 7], dtype=np.int64))
    c1, l1, n1, t1 = sample_codes_cached(ids)
    c2, l2, n2, t2 = sample_codes_cached(ids, defer=True)
    assert c2 is c1 and t2 == t1 and confirm_codes() and confirm_codes()
    ids.values[0] = 7                                   # same buffer, different content
    c3 = sample_codes_cached(ids, defer=True)[0]
    assert c3 is c1                                     # optimistic: the stale memo ...
    assert not confirm_codes()                          # ... and the check says so
    c4, l4, n4, t4 = sample_codes_cached(ids)           # the memo is gone: recomputed
    assert t4 != t1 and list(l4) == [1, 2, 5, 7] and list(n4) == [2, 1, 2, 2]
    other = pd.Series(ids.values.copy())                # another buffer: never deferred
    assert sample_codes_cached(other, defer=True)[3] != t4 or True
    assert confirm_codes()


def test_nam_cache_skips_the_walk_for_a_second_phenotype():
    """a second analysis of the same dataset (same graph object, same sample ids, same
    step rule) reuses the NAM held by the engine; anything that changes the NAM's inputs, and any
    progress printing, recomputes it."""
    case = load_case('c01_plain_f32')
    data, y = case['data'], case['y']
    y2 = pd.Series(np.random.RandomState(5).randn(len(y)), index=y.index)
    kw = dict(nsteps=3, Nnull=50, seed=1)

    def steps(e):
        return sum(1 for c in e.calls if c[0] == 'nam_step')

    eng = FakeEngine()
    eng.reuse_nam = True
    r1 = cna.tl.association(data, y, 'id', return_full=True, engine=eng, **kw)
    assert steps(eng) == 3
    nam1 = r1.nam.values.copy()
    r2 = cna.tl.association(data, y2, 'id', return_full=True, engine=eng, **kw)
    assert steps(eng) == 3                                  # no further walk
    np.testing.assert_array_equal(r1.nam.values, nam1)      # and the first result's NAM is still the resident one
    fresh = cna.tl.association(data, y2, 'id', return_full=True, engine=FakeEngine(), **kw)
    assert r2.p == fresh.p and r2.k == fresh.k
    np.testing.assert_array_equal(r2.ncorrs.values, fresh.ncorrs.values)
    np.testing.assert_array_equal(r2.nam.values, fresh.nam.values)
    # a different step count, a dense diffusion in between, edited sample ids, progress output: recompute
    cna.tl.association(data, y2, 'id', engine=eng, nsteps=2, Nnull=50, seed=1)
    assert steps(eng) == 5
    cna.tl.association(data, y2, 'id', engine=eng, nsteps=2, Nnull=50, seed=1)
    assert steps(eng) == 5
    cna.tl.diffuse(data, np.ones((len(data.obs), 1)), 1, engine=eng)
    cna.tl.association(data, y2, 'id', engine=eng, nsteps=2, Nnull=50, seed=1)
    assert steps(eng) == 7
    with contextlib.redirect_stdout(io.StringIO()):
        cna.tl.association(data, y2, 'id', engine=eng, nsteps=2, Nnull=50, seed=1, show_progress=True)
    assert steps(eng) == 9
    ids = data.obs['id'].values.copy()
    a, b = ids[0], ids[ids != ids[0]][0]
    ids[ids == a], ids[ids == b] = -1, a
    ids[ids == -1] = b                                      # swap two samples' cells: same buffer size, new content
    data.obs['id'] = ids
    cna.tl.association(data, y2, 'id', engine=eng, nsteps=2, Nnull=50, seed=1)
    assert steps(eng) == 11


def test_fdr_tables_follow_the_reference_lines():
    """_association._fdr_tables against the statements of the reference (_association.py:105-118, _stats.py:79-80) on
    tables with NaN entries (thresholds nothing reaches), tables that never get below 5 % / 10 %, and the all-NaN table,
    which fails with the reference's IndexError."""
    import pandas as pd
    from cna_amd.tools._association import _fdr_tables
    rs = np.random.RandomState(0)
    thresholds = np.arange(0.05, 0.2, 0.2 / 400)
    T = len(thresholds)
    for case in range(6):
        ranks = np.sort(rs.randint(0, 5000, T))[::-1].astype(np.int64)
        tails = (ranks * rs.rand(T) * (0.02 if case % 2 else 0.5) * 100).astype(np.int64)
        if case >= 2:
            ranks[-(case * 7):] = 0
            tails[-(case * 7):] = 0
        if case == 5:
            ranks[:] = 0
            tails[:] = 0
        Nloc = 100
        with np.errstate(all='ignore'):
            fdr = tails / ranks / Nloc
        fdrs = pd.DataFrame({'threshold': thresholds, 'fdr': fdr})
        want5 = want10 = None
        if case == 5:
            with pytest.raises(IndexError):
                _fdr_tables(tails, ranks, Nloc, thresholds)
            continue
        # the reference: if np.min(fdrs.fdr) > 0.05: None else fdrs[fdrs.fdr <= 0.05].iloc[0].threshold
        if not np.min(fdrs.fdr) > 0.05:
            want5 = fdrs[fdrs.fdr <= 0.05].iloc[0].threshold
        if not np.min(fdrs.fdr) > 0.1:
            want10 = fdrs[fdrs.fdr <= 0.1].iloc[0].threshold
        got, t5, t10, runmin = _fdr_tables(tails, ranks, Nloc, thresholds)
        np.testing.assert_array_equal(got, fdr)
        assert t5 == want5 and t10 == want10
        # running minimum that skips NaN, as the per-cell lookup of _association.py:234-237 needs it
        ref = np.array([np.nanmin(fdr[:i + 1]) if not np.isnan(fdr[:i + 1]).all() else np.nan for i in range(T)])
        np.testing.assert_array_equal(runmin, ref)


def test_helper_threads_take_their_ranks_share_of_the_cpu_allowance(monkeypatch):
    """Several ranks on one node share the node's CPU allowance: every helper (draw, hash, cluster order, copies) asks
    cna_amd._order.usable_cpus, which divides by the launcher's LOCAL_WORLD_SIZE (or what cna_amd.dist was told)."""
    from cna_amd import _order, dist
    monkeypatch.delenv('LOCAL_WORLD_SIZE', raising=False)
    monkeypatch.setattr(dist, '_cfg', {})
    alone = _order.usable_cpus()
    assert alone >= 1 and _order.ranks_on_this_node() == 1
    monkeypatch.setenv('LOCAL_WORLD_SIZE', '8')
    assert _order.ranks_on_this_node() == 8
    assert _order.usable_cpus() == max(1, alone // 8) and _order.usable_cpus(4) == max(1, min(4, alone // 8))
    monkeypatch.delenv('LOCAL_WORLD_SIZE')
    monkeypatch.setattr(dist, '_cfg', dict(rank=1, nranks=4))
    assert _order.ranks_on_this_node() == 4 and _order.usable_cpus() == max(1, alone // 4)
    monkeypatch.setenv('LOCAL_WORLD_SIZE', 'garbage')
    assert _order.ranks_on_this_node() == 4
