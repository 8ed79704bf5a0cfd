"""BASELINE.json configs[1] AT FULL SIZE (200 000 cells x 50 samples, k = 30, nsteps = 3, Nnull = 1000, seed 0): the CPU
oracle against what the REFERENCE ITSELF returned on the same inputs (tests/golden/d02_config2.npz, captured by
tests/golden/make_golden.py:run_config2 in the build container; the inputs are regenerated, the fixture holds their
digest).  Pins the oracle at a BASELINE size, not only at the 2 000-10 000 cells of the other fixtures.  ~1 minute."""
import numpy as np
import pytest

from helpers import load_config2_case, assert_matches_config2
from oracle import cna_oracle as orc


@pytest.fixture(scope='module')
def case():
    return load_config2_case()


def as_out(o):
    return dict(p=o['p'], k=o['k'], ks=o['ks'], r=o['r'], n_kept=int(o['kept'].sum()), nullminps=o['nullminps'], svs=o['svs'],
                U=o['U'], M=o['M'], yresid=o['yresid'], yresid_hat=o['yresid_hat'], r2=o['r2'], r2_perpc=o['r2_perpc'],
                nullr2_mean=o['nullr2_mean'], nullr2_std=o['nullr2_std'], ncorrs=o['ncorrs'], nam=o['nam'], namresid=o['namresid'],
                fdrs=o['fdrs'], fdr_5p_t=o['fdr_5p_t'], fdr_10p_t=o['fdr_10p_t'])


def test_inputs_regenerate_bit_for_bit(case):
    z = case['z']
    assert case['same_inputs'], 'the regenerated C2 inputs differ from the ones the reference was run on'
    assert case['data'].obsp['connectivities'].nnz == int(z['nnz'])


def test_oracle_reference_mode_matches_the_reference_at_config2(case):
    """'reference' mode reproduces the reference's float32 column sums and float32 first step: 1e-9."""
    if not case['same_inputs']:
        pytest.skip('inputs differ from the fixture')
    o = orc.association(case['data'], case['y'], 'id', mode='reference', **case['call'])
    assert_matches_config2(as_out(o), case['z'], 1e-9, obs=dict(coef=o['obs_coef'], coef_fdr=o['obs_coef_fdr']),
                           floors=dict(nam=1e-12, ncorrs=1e-12, namresid=1e-12))
