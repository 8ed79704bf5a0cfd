"""BASELINE.json configs[1] at FULL size on the GPU (200 000 cells x 50 samples, k = 30, nsteps = 3, Nnull = 1000, seed 0)
against (a) the REFERENCE's own results on the same inputs (tests/golden/d02_config2.npz) at the golden tolerances and
(b) the float64 oracle run here at full size, as tightly as the small cases are compared with it."""
import numpy as np
import pytest

from helpers import load_config2_case, assert_matches_config2, relerr, fdr_rows

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def case():
    return load_config2_case()


@pytest.fixture(scope='module')
def result(case):
    import warnings
    import cna_amd as cna
    from cna_amd.engine import get_engine
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        res = cna.tl.association(case['data'], case['y'], 'id', return_full=True, engine=get_engine(), **case['call'])
    return res


def test_config2_matches_the_reference_run(case, result):
    """Every stored field of the reference's run: integers exact, floats 1e-5 (empirical FDRs 1e-4), cells-sized fields
    on every 100th cell entry by entry."""
    z, res, data = case['z'], result, case['data']
    out = dict(p=res.p, k=res.k, ks=res.ks, r=res.r, n_kept=int(res.kept.sum()), nullminps=res.nullminps,
               svs=res.namresid_svs.values, U=res.namresid_sampleXpc.values, M=res.M.values, yresid=res.yresid.values,
               yresid_hat=res.yresid_hat, r2=res.r2, r2_perpc=res.r2_perpc, nullr2_mean=res.nullr2_mean, nullr2_std=res.nullr2_std,
               ncorrs=res.ncorrs.values, nam=res.nam.values.T, namresid=res.namresid.values.T,
               fdrs=dict(threshold=res.fdrs.threshold.values, fdr=res.fdrs.fdr.values, num_detected=res.fdrs.num_detected.values),
               fdr_5p_t=res.fdr_5p_t, fdr_10p_t=res.fdr_10p_t)
    # (inputs that did not regenerate bit for bit -- another libm / numpy dispatch -- still agree to ~1e-7; only the
    # integer counts may then move by one or two)
    assert_matches_config2(out, z, 1e-5, obs=dict(coef=data.obs['coef'].values, coef_fdr=data.obs['coef_fdr'].values),
                           exact_counts=case['same_inputs'])
    V = res.namresid_nbhdXpc.values[z['sub']]
    kk = int(z['k'])
    sgn = np.sign((V[:, :kk] * z['V_sub'][:, :kk]).sum(axis=0))
    assert relerr(V[:, :kk] * sgn, z['V_sub'][:, :kk]) < 1e-5
    assert case['same_inputs'], 'results agree, but the regenerated inputs are not bit-identical to the fixture\'s'


def test_config2_matches_the_f64_oracle_at_full_size(case, result):
    """What test_association_matches_f64_oracle_tightly asserts on 3 000 cells, at 200 000."""
    from oracle import cna_oracle as orc
    res = result
    ref = orc.association(case['data'], case['y'], 'id', mode='f64', **case['call'])
    assert int(res.k) == ref['k'] and res.p == ref['p'] and np.array_equal(res.kept, ref['kept'])
    np.testing.assert_array_equal(res.nam.values.T, ref['nam'])           # bit-identical walk
    assert relerr(res.namresid.values.T, ref['namresid']) < 1e-10
    assert relerr(res.ncorrs.values, ref['ncorrs']) < 1e-10
    assert relerr(res.namresid_svs.values, ref['svs']) < 1e-10
    assert relerr(res.nullminps, ref['nullminps']) < 1e-8
    T = fdr_rows(res.fdrs, ref['fdrs'], ref['ncorrs'])
    assert np.array_equal(res.fdrs.num_detected.values[:T], ref['fdrs']['num_detected'][:T])
    np.testing.assert_allclose(res.fdrs.fdr.values[:T], ref['fdrs']['fdr'][:T], rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(case['data'].obs['coef'].values, ref['obs_coef'], rtol=0, atol=1e-10 * np.abs(ref['obs_coef']).max())
    np.testing.assert_allclose(case['data'].obs['coef_fdr'].values, ref['obs_coef_fdr'], rtol=1e-9, atol=1e-13)


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[2] (1M cells x 100 samples, the "HBM roofline run") against the reference's own run at full size
# (tests/golden/d03_config3.npz; a quarter of an hour and ~40 GB of the reference in the build container)
@pytest.fixture(scope='module')
def case3():
    import os
    from helpers import GOLDEN_DIR
    if not os.path.exists(os.path.join(GOLDEN_DIR, 'd03_config3.npz')):
        pytest.skip('no d03_config3 fixture')
    return load_config2_case('d03_config3')


def test_config3_matches_the_reference_run(case3):
    from cna_amd.engine import get_engine
    check_config3_against_the_reference(case3, get_engine())


def check_config3_against_the_reference(case3, engine):
    import warnings
    import cna_amd as cna
    z, data = case3['z'], case3['data']
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        res = cna.tl.association(data, case3['y'], 'id', return_full=True, engine=engine, **case3['call'])
    sub = z['sub']
    out = dict(p=res.p, k=res.k, ks=res.ks, r=res.r, n_kept=int(res.kept.sum()), nullminps=res.nullminps,
               svs=res.namresid_svs.values, U=res.namresid_sampleXpc.values, M=res.M.values, yresid=res.yresid.values,
               yresid_hat=res.yresid_hat, r2=res.r2, r2_perpc=res.r2_perpc, nullr2_mean=res.nullr2_mean, nullr2_std=res.nullr2_std,
               ncorrs=res.ncorrs.values, nam=None, namresid=None,
               fdrs=dict(threshold=res.fdrs.threshold.values, fdr=res.fdrs.fdr.values, num_detected=res.fdrs.num_detected.values),
               fdr_5p_t=res.fdr_5p_t, fdr_10p_t=res.fdr_10p_t)
    assert_matches_config2(out, z, 1e-5, obs=dict(coef=data.obs['coef'].values, coef_fdr=data.obs['coef_fdr'].values),
                           exact_counts=case3['same_inputs'])
    # the cells-sized frames on the fixture's cells only (800 MB each in full)
    from helpers import assert_elementwise
    nam_sub = res.nam.values[:, sub]
    assert_elementwise(nam_sub, z['nam_sub'], 1e-5, 1e-12, 'nam (every 1000th cell)')
    assert_elementwise(res.namresid.values[:, sub], z['namresid_sub'], 1e-5, 2e-7, 'namresid (every 1000th cell)')
    assert case3['same_inputs'], 'results agree, but the regenerated inputs are not bit-identical to the fixture\'s'
    return res
