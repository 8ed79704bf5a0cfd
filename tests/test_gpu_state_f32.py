"""The opt-in 4-byte diffusion state (cna_set_state_f32 / Engine.set_state_f32).

Default: the scaled state between two steps of a walk is float64 and the NAM is bit-identical to the f64 oracle.  With
the option the state BETWEEN steps is stored in 4 bytes per entry (every sum still runs in float64): the dense step
gathers half the bytes.  What must hold then is the bar of BASELINE.json -- integers exact, floats 1e-5 against the
reference -- not bit-identity: the NAM moves by <= ~1e-7 relative (one rounding to 24 bits of non-negative terms per
stored step), less than the reference's own float32 first step moves it on float32 graphs (2e-7)."""
import warnings

import numpy as np
import pytest

from helpers import relerr, fdr_rows

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng32():
    from cna_amd.engine import Engine
    e = Engine(device=0)
    e.set_state_f32(True)
    yield e
    e.close() if hasattr(e, 'close') else None


def run(engine, data, meta, **kw):
    import cna_amd as cna
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        return cna.tl.association(data, meta['y'], 'id', covs=meta['covs'], batches=meta['batches'], return_full=True,
                                  engine=engine, **kw)


@pytest.mark.parametrize('n,N,extra,nsteps', [(6000, 120, {}, 3), (5000, 200, dict(n_covs=2), 3), (4000, 300, {}, 4),
                                              (3000, 600, dict(n_covs=1, n_batches=3), 3), (2500, 1024, {}, 3),
                                              (4000, 100, {}, None), (5000, 96, {}, 5), (3000, 256, {}, 2)])
def test_walk_on_the_four_byte_state(eng32, n, N, extra, nsteps):
    """Against the default walk on the same inputs and against the f64 oracle: integers equal, the NAM within 3e-7 entry by
    entry -- and NOT bit-identical where a stored step exists (the option was really taken)."""
    from cna_amd import synth
    from cna_amd.engine import get_engine
    from oracle import cna_oracle as orc
    data, meta = synth.make_dataset(n, N, k=15, seed=n + N, **extra)
    kw = dict(nsteps=nsteps, Nnull=130, seed=5)
    a = run(get_engine(), data, meta, **kw)
    coef_a = data.obs['coef'].values.copy()
    b = run(eng32, data, meta, **kw)
    coef_b = data.obs['coef'].values.copy()
    A, B = a.nam.values, b.nam.values
    assert A.shape == B.shape
    err = np.abs(A - B)
    assert (err <= 3e-7 * np.abs(A) + 1e-300).all(), float((err / np.maximum(np.abs(A), 1e-300)).max())
    steps = nsteps if nsteps is not None else 3
    if steps >= 3:
        assert not np.array_equal(A, B)               # (two steps: compressed step -> NAM directly, nothing is stored)
    else:
        assert np.array_equal(A, B)
    assert int(a.k) == int(b.k) and a.p == b.p and np.array_equal(a.kept, b.kept)
    assert relerr(b.namresid.values, a.namresid.values) < 1e-5
    assert relerr(b.ncorrs.values, a.ncorrs.values) < 1e-5 and relerr(coef_b, coef_a) < 1e-5
    np.testing.assert_allclose(b.nullminps, a.nullminps, rtol=1e-5)
    T = min(len(a.fdrs), len(b.fdrs))               # (np.arange's 300 / 301 thresholds: the common prefix)
    assert abs(len(a.fdrs) - len(b.fdrs)) <= 1 and np.array_equal(a.fdrs.num_detected.values[:T], b.fdrs.num_detected.values[:T])
    ref = orc.association(data, meta['y'], 'id', covs=meta['covs'], batches=meta['batches'], mode='f64', **kw)
    assert int(b.k) == ref['k'] and b.p == ref['p'] and np.array_equal(b.kept, ref['kept'])
    assert relerr(b.nam.values.T, ref['nam']) < 3e-7
    assert relerr(b.namresid_svs.values, ref['svs']) < 1e-6
    T = fdr_rows(b.fdrs, ref['fdrs'], ref['ncorrs'])
    assert np.array_equal(b.fdrs.num_detected.values[:T], ref['fdrs']['num_detected'][:T])


def test_option_is_inert_where_it_does_not_apply(eng32):
    """At most 64 samples (two rows per wave) and 65-95 samples (no compressed second step): 8-byte state, same bits."""
    from cna_amd import synth
    from cna_amd.engine import get_engine
    for n, N in ((5000, 40), (4000, 80)):
        data, meta = synth.make_dataset(n, N, k=15, seed=3)
        kw = dict(nsteps=3, Nnull=60, seed=1)
        a = run(get_engine(), data, meta, **kw)
        b = run(eng32, data, meta, **kw)
        assert np.array_equal(a.nam.values, b.nam.values) and a.p == b.p
        assert np.array_equal(a.ncorrs.values, b.ncorrs.values)


def test_switching_the_option_forgets_the_cached_nam():
    from cna_amd import synth
    from cna_amd.engine import Engine
    e = Engine(device=0)
    data, meta = synth.make_dataset(4000, 128, k=15, seed=9)
    kw = dict(nsteps=3, Nnull=60, seed=1)
    a = run(e, data, meta, **kw).nam.values.copy()
    e.set_state_f32(True)
    b = run(e, data, meta, **kw).nam.values.copy()
    e.set_state_f32(False)
    c = run(e, data, meta, **kw).nam.values.copy()
    assert np.array_equal(a, c) and not np.array_equal(a, b)
    assert np.abs(a - b).max() <= 3e-7 * np.abs(a).max()


def test_stepwise_and_tl_nam_on_the_four_byte_state(eng32):
    """cna.tl.nam (QC, batch kurtosis) and the step-by-step generator of the public API."""
    import cna_amd as cna
    from cna_amd import synth
    from cna_amd.engine import get_engine
    data, meta = synth.make_dataset(5000, 110, k=15, seed=21, n_batches=4)
    out = {}
    for name, e in (('f64', get_engine()), ('f32', eng32)):
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            frame, keep = cna.tl.nam(data, 'id', batches=meta['batches'], nsteps=4, engine=e)
        out[name] = (frame.values.copy(), np.asarray(keep).copy())
    assert np.array_equal(out['f64'][1], out['f32'][1])
    assert relerr(out['f32'][0], out['f64'][0]) < 1e-6 and not np.array_equal(out['f32'][0], out['f64'][0])
    # diffuse_stepwise of the public API (dense state): always 8 bytes
    rs = np.random.RandomState(0)
    s0 = rs.rand(len(data.obs), 70)
    x = cna.tl.diffuse(data, s0, 2, engine=get_engine())
    y = cna.tl.diffuse(data, s0, 2, engine=eng32)
    assert np.array_equal(x, y)


def test_config3_on_the_four_byte_state_matches_the_reference_run(eng32):
    """BASELINE.json configs[2] at full size (1M cells x 100 samples) with the option ON against the reference's own
    run: the same assertions as the default path's (tests/test_gpu_config2.py), integers exact."""
    import os
    from helpers import GOLDEN_DIR, load_config2_case
    from test_gpu_config2 import check_config3_against_the_reference
    if not os.path.exists(os.path.join(GOLDEN_DIR, 'd03_config3.npz')):
        pytest.skip('no d03_config3 fixture')
    case3 = load_config2_case('d03_config3')
    check_config3_against_the_reference(case3, eng32)


def test_zero_variance_cells_on_the_four_byte_state(eng32):
    """Cells whose NAM row is constant (a far-away blob whose only sample has no phenotype) are found and dropped exactly
    as on the 8-byte state (_association.py:182-185): exact zeros stay exact zeros in 4 bytes."""
    import pandas as pd
    import scipy.sparse as sp
    import cna_amd as cna
    from cna_amd import synth
    from oracle import cna_oracle as orc
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
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        res = cna.tl.association(d2, y, 'id', return_full=True, engine=eng32, **kw)
    ref = orc.association(d2, y, 'id', mode='f64', **kw)
    assert (~res.kept).sum() == n_iso and np.array_equal(res.kept, ref['kept'])
    assert int(res.k) == ref['k'] and res.p == ref['p']
    assert relerr(res.ncorrs.values, ref['ncorrs']) < 1e-5
    T = fdr_rows(res.fdrs, ref['fdrs'], ref['ncorrs'])
    assert np.array_equal(res.fdrs.num_detected.values[:T], ref['fdrs']['num_detected'][:T])


@pytest.mark.parametrize('N,nsteps', [(100, 3), (128, 4), (97, 3)])
def test_two_edges_per_wave_step_agrees_with_the_wave_per_edge_step(eng32, monkeypatch, N, nsteps):
    """At most 128 columns: k_nam_step32h sums a row's even and odd edges in the two halves of a wave and joins them at
    the end -- another order of the same float64 additions than k_nam_step32's (CNA_STEP32_WIDE=1): 1e-13, not bits."""
    import cna_amd as cna
    from cna_amd import synth
    data, meta = synth.make_dataset(7000, N, k=15, seed=N)
    out = []
    for wide in ('', '1'):
        if wide:
            monkeypatch.setenv('CNA_STEP32_WIDE', wide)
        else:
            monkeypatch.delenv('CNA_STEP32_WIDE', raising=False)
        eng32._nam_sig = None
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            frame, keep = cna.tl.nam(data, 'id', nsteps=nsteps, engine=eng32)
        out.append(frame.values.copy())
    assert relerr(out[0], out[1]) < 1e-13 and np.abs(out[0] - out[1]).max() > 0
    assert (np.abs(out[0] - out[1]) <= 1e-12 * np.abs(out[1])).all()


def test_config4_size_integers_equal_the_default_path(eng32):
    """BASELINE.json configs[3] at full size (2M cells x 200 samples, Nnull 1000): the analysis on the 4-byte state against
    the default one on the same inputs -- p, k, the kept cells and every `num_detected` identical; FDR thresholds,
    coefficients and per-cell FDRs within 1e-5; a 20 000-cell slice of the NAM within 3e-7 entry by entry."""
    import cna_amd as cna
    from cna_amd import synth
    from cna_amd.engine import get_engine
    data, meta = synth.make_dataset(2_000_000, 200, k=30, seed=0)
    kw = dict(nsteps=3, Nnull=1000, seed=0, return_full=True)
    out = {}
    for name, e in (('f64', get_engine()), ('f32', eng32)):
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            res = cna.tl.association(data, meta['y'], 'id', engine=e, **kw)
        sl = np.arange(0, 2_000_000, 100)
        out[name] = dict(p=res.p, k=int(res.k), nd=res.fdrs.num_detected.values.copy(), fdr=res.fdrs.fdr.values.copy(),
                         t5=res.fdr_5p_t, t10=res.fdr_10p_t, coef=data.obs['coef'].values.copy(),
                         coef_fdr=data.obs['coef_fdr'].values.copy(), nam=res.nam.values[:, sl].copy(),
                         nullminps=np.asarray(res.nullminps).copy(), kept=int(res.kept.sum()))
        del res
    a, b = out['f64'], out['f32']
    assert a['p'] == b['p'] and a['k'] == b['k'] and a['kept'] == b['kept']
    assert np.array_equal(a['nd'], b['nd'])
    for key in ('t5', 't10'):                        # (thresholds are fractions of max |coefficient|: floats)
        np.testing.assert_allclose(b[key], a[key], rtol=1e-6, equal_nan=True)
    np.testing.assert_allclose(b['fdr'], a['fdr'], rtol=1e-5, atol=1e-12)
    np.testing.assert_allclose(b['nullminps'], a['nullminps'], rtol=1e-5)
    assert np.abs(b['coef'] - a['coef']).max() <= 1e-5 * np.abs(a['coef']).max()
    np.testing.assert_allclose(b['coef_fdr'], a['coef_fdr'], rtol=1e-4, atol=1e-12, equal_nan=True)
    err = np.abs(a['nam'] - b['nam'])
    assert (err <= 3e-7 * np.abs(a['nam']) + 1e-300).all() and err.max() > 0
