"""Pin the CPU oracle (oracle/cna_oracle.py) against golden vectors captured from the
reference (tests/golden/make_golden.py).  CPU only.

Tolerances: integers / masks / step counts exact; floats 1e-5 relative as BASELINE.json
states (we observe ~1e-12 in 'reference' mode and ~1e-7 in 'f64' mode on float32 graphs).
"""
import numpy as np
import pytest
import scipy.stats as st

from helpers import golden_names, messy_names, load_case, relerr, sign_align, assert_odd_row_harmless
from oracle import cna_oracle as orc

RAISING = {'c07_no_local'}
NAMES = [n for n in golden_names()] + messy_names()


def run_oracle(case, mode):
    call = dict(case['call'])
    return orc.association(case['data'], case['y'], case['sid_name'], batches=case['batches'],
                           covs=case['covs'], donorids=case['donorids'], mode=mode, **call)


@pytest.mark.parametrize('mode,tol', [('reference', 1e-9), ('f64', 1e-5)])
@pytest.mark.parametrize('name', NAMES)
def test_association_matches_reference(name, mode, tol):
    case = load_case(name)
    z = case['z']
    if z['raised'].item() and name not in RAISING:
        # (messy inputs on which the reference itself fails -- its sample filter pairs labels by position and lets a sample
        # without cells through: the restatement fails the same way)
        with pytest.raises(Exception) as info:
            run_oracle(case, mode)
        assert info.type.__name__ == z['raised'].item().split(':')[0]
        return
    out = run_oracle(case, mode)
    # data.obs[key] is written before the reference's local_test=False crash
    np.testing.assert_allclose(out['obs_coef'], z['obs_coef'], rtol=0, atol=tol * np.nanmax(np.abs(z['obs_coef'])),
                               equal_nan=True)
    if name in RAISING:
        assert z['raised'].item().startswith('AttributeError')
        return
    # discrete outputs: exact
    assert out['k'] == int(z['k'])
    assert np.array_equal(out['ks'], z['ks'])
    assert out['r'] == int(z['r'])
    assert np.array_equal(out['kept'], z['kept'])
    assert out['p'] == pytest.approx(float(z['p']), rel=1e-12)
    # floats
    assert relerr(out['nam'].T, z['nam']) < tol
    assert relerr(out['namresid'].T, z['namresid']) < tol
    assert relerr(out['M'], z['M']) < tol
    assert relerr(out['ncorrs'], z['ncorrs']) < tol
    assert relerr(out['nullminps'], z['nullminps']) < tol
    assert relerr(out['svs'], z['svs']) < tol
    assert relerr(out['varexp'], z['varexp']) < tol
    assert relerr(out['yresid'], z['yresid']) < tol
    assert out['r2'] == pytest.approx(float(z['r2']), rel=tol)
    assert out['nullr2_mean'] == pytest.approx(float(z['nullr2_mean']), rel=tol)
    assert out['nullr2_std'] == pytest.approx(float(z['nullr2_std']), rel=tol)
    # PCs: up to sign, only well-separated ones
    kk = int(z['k'])
    U, Uref = sign_align(out['U'], z['U'], kk)
    assert relerr(U, Uref) < tol                     # observed: 7e-14 in reference mode, 2e-7 in f64 mode
    assert relerr(np.abs(out['beta']), np.abs(z['beta'])) < tol
    assert relerr(out['r2_perpc'], z['r2_perpc']) < tol
    if 'fdr_fdr' in z:
        f = out['fdrs']
        # np.arange(maxcorr/4, maxcorr, maxcorr/400) (_association.py:102) yields 300 or 301
        # thresholds depending on the last bits of maxcorr; compare the common prefix.
        # (the odd row, whichever side has it, must be the degenerate `threshold = maxcorr` row:
        # helpers.assert_odd_row_harmless; the 5 % / 10 % thresholds and the per-cell column are compared in full)
        T = assert_odd_row_harmless(f['threshold'], f['num_detected'], z['fdr_threshold'], z['fdr_num_detected'],
                                    np.nanmax(np.abs(z['ncorrs'])), name)
        assert relerr(f['threshold'][:T], z['fdr_threshold'][:T]) < tol
        assert np.array_equal(f['num_detected'][:T], z['fdr_num_detected'][:T])
        assert relerr(f['fdr'][:T], z['fdr_fdr'][:T]) < tol * 10
        for key in ('fdr_5p_t', 'fdr_10p_t'):
            ref = float(z[key])
            if np.isnan(ref):
                assert out[key] is None
            else:
                assert out[key] == pytest.approx(ref, rel=tol)
        np.testing.assert_allclose(out['obs_coef_fdr'], z['obs_coef_fdr'], rtol=tol * 10, atol=1e-12)


@pytest.mark.parametrize('name', [n for n in NAMES if 'steps' in np.load(
    __import__('os').path.join(__import__('helpers').GOLDEN_DIR, n + '.npz')).files])
def test_diffusion_steps(name):
    case = load_case(name)
    z = case['z']
    A = orc.get_graph(case['data'])
    codes, labels = orc.sample_codes(case['data'].obs[case['sid_name']])
    steps = z['steps']
    for mode, tol in (('reference', 1e-13), ('f64', 1e-6)):
        info = orc.build_nam(A, codes, len(labels), nsteps=len(steps), mode=mode)
        assert relerr(info['S_last'], steps[-1]) < tol
        np.testing.assert_allclose(info['medkurt'], z['steps_medkurt'], rtol=max(tol * 100, 1e-10))
    # auto-stop: the step count the reference chose is reproduced
    call = case['call']
    if call.get('nsteps') is None:
        info = orc.build_nam(A, codes, len(labels), nsteps=None, mode='f64')
        assert info['nsteps'] == z['stdout'].item().count('median kurtosis') or info['nsteps'] >= 3


def test_dense_diffuse_and_svd():
    case = load_case('c01_plain_f32')
    z = case['z']
    A = orc.get_graph(case['data'])
    assert relerr(orc.diffuse(A, z['diffuse_in'], 2), z['diffuse_out_2']) < 1e-13
    assert relerr(orc.diffuse(A, z['diffuse_in'], 2, self_weight=0.5), z['diffuse_out_2_sw05']) < 1e-13
    assert relerr(orc.diffuse(A, z['diffuse_in'], 2, mode='f64'), z['diffuse_out_2']) < 1e-6
    U, svs, V, _ = orc.svd_nam(z['nam'].T)
    assert relerr(svs, z['svd_svs']) < 1e-9
    Ua, Ub = sign_align(U, z['svd_U'], 5)
    assert relerr(Ua, Ub) < 1e-6
    Va, Vb = sign_align(V, z['svd_V'], 5)
    assert relerr(Va, Vb) < 1e-6


@pytest.mark.parametrize('name', ['c01_plain_f32', 'c03_covs_batches', 'c12_batchy_qc'])
def test_tl_nam(name):
    case = load_case(name)
    z = case['z']
    out = orc.nam(case['data'], case['sid_name'], batches=case['batches'], nsteps=case['call'].get('nsteps'))
    assert np.array_equal(out['keep'], z['tlnam_keep'])
    assert relerr(out['nam'].T, z['tlnam']) < 1e-12


def test_self_weight():
    case = load_case('c14_selfweight_autostop_unsorted')
    out = orc.nam(case['data'], case['sid_name'], nsteps=2, self_weight=2)
    assert relerr(out['nam'].T, case['z']['tlnam_sw2']) < 1e-12


def test_row_kurtosis_is_scipy():
    rs = np.random.RandomState(0)
    x = rs.rand(50, 17)
    np.testing.assert_allclose(orc.row_kurtosis(x), st.kurtosis(x, axis=1), rtol=1e-12)


def test_tail_counts_matches_histogram_definition():
    rs = np.random.RandomState(1)
    zn = rs.randn(400, 7)
    thr = np.arange(0.5, 2.0, 0.05)
    tc = orc.tail_counts(thr, zn)
    edges = thr ** 2 - 1e-8 - 1e-5 * thr ** 2
    ref = np.array([[(zn[:, j] ** 2 >= e).sum() for e in edges] for j in range(7)])
    assert np.array_equal(tc, ref)


@pytest.mark.parametrize('name', ['c01_plain_f32', 'c06_nnull_cap', 'c10_categorical_ids'])
def test_reference_cost_baseline_matches_reference(name):
    """oracle/reference_cost.py (bench.py's cpu_baseline: the reference's own sequence of library calls,
    per-permutation / per-column / per-cell Python loops included) returns what the reference returned."""
    from oracle import reference_cost as rc
    case = load_case(name)
    z = case['z']
    call = case['call']
    out = rc.association(case['data'], case['y'], case['sid_name'], nsteps=call['nsteps'], Nnull=call['Nnull'],
                         seed=call['seed'])
    assert out['k'] == int(z['k'])
    assert out['p'] == pytest.approx(float(z['p']), rel=1e-12)
    assert relerr(out['ncorrs'], z['ncorrs']) < 1e-9
    assert relerr(out['nullminps'], z['nullminps']) < 1e-8
    T = min(len(out['fdr']), len(z['fdr_fdr']))
    assert T >= 300
    assert np.array_equal(out['num_detected'][:T], z['fdr_num_detected'][:T])
    assert relerr(out['fdr'][:T], z['fdr_fdr'][:T]) < 1e-8
    np.testing.assert_allclose(out['coef_fdr'], z['obs_coef_fdr'], rtol=1e-8, atol=1e-12)
    assert set(out['stages']) == {'nam', 'resid_svd', 'global_test', 'local_test', 'percell_apply'}


@pytest.mark.parametrize('mode,tol', [('reference', 1e-9), ('f64', 1e-5)])
def test_demo_like_config1(mode, tol):
    """BASELINE.json configs[0] (demo recipe, 10 000 cells x 50 samples, covs + 5 batches, nsteps=3, Nnull=100)."""
    from helpers import load_demo_case, assert_matches_demo
    case = load_demo_case()
    out = orc.association(case['data'], case['y'], 'id', batches=case['batches'], covs=case['covs'], mode=mode,
                          **case['call'])
    assert_matches_demo(out, case['z'], tol, obs=dict(coef=out['obs_coef'], coef_fdr=out['obs_coef_fdr']))


def test_table_length_deviation_occurs_both_ways_and_is_harmless():
    """np.arange(maxcorr/4, maxcorr, maxcorr/400) (_association.py:101-102): 300 or 301 rows depending on the last bits
    of maxcorr.  In f64 mode (what the kernels compute) six fixtures land on the other side of the reference -- in BOTH
    directions -- and in each the odd row is `threshold = maxcorr` with num_detected 0 (1 in f12: the threshold rounds to
    just below the largest coefficient), the 5 % / 10 % thresholds agree and the per-cell FDR column agrees on every cell."""
    longer, shorter = [], []
    for name in ('c06_nnull_cap', 'c17_low_sample_size', 'c20_unused_category_autostop', 'f10_messy', 'f12_messy', 'f28_messy'):
        case = load_case(name)
        z = case['z']
        out = run_oracle(case, 'f64')
        f = out['fdrs']
        la, lb = len(f['threshold']), len(z['fdr_threshold'])
        assert abs(la - lb) == 1, (name, la, lb)
        (longer if la > lb else shorter).append(name)
        T = assert_odd_row_harmless(f['threshold'], f['num_detected'], z['fdr_threshold'], z['fdr_num_detected'],
                                    np.nanmax(np.abs(z['ncorrs'])), name)
        assert T == 300
        odd_nd = (f['num_detected'] if la > lb else z['fdr_num_detected'])[-1]
        assert odd_nd == (1 if name == 'f12_messy' else 0)
        for key in ('fdr_5p_t', 'fdr_10p_t'):
            ref = float(z[key])
            assert (out[key] is None) if np.isnan(ref) else out[key] == pytest.approx(ref, rel=1e-5)
        np.testing.assert_allclose(out['obs_coef_fdr'], z['obs_coef_fdr'], rtol=1e-4, atol=1e-12)
    assert longer and shorter, (longer, shorter)
