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
