#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE (imported from /root/reference,
this container only) on small synthetic inputs.  Output: tests/golden/*.npz.

    python tests/golden/make_golden.py

Each fixture holds the inputs (CSR arrays, per-cell sample labels, sample-level
y / covs / batches / donorids, call kwargs as JSON) and every output the reference
produced (SURVEY.md §8a a20 result fields, per-step diffusion states, data.obs columns,
progress text).  Fixtures are data only; the reference source is not copied.
Versions used: see the 'versions' entry of each fixture.
"""
import io
import json
import os
import sys
import contextlib
import warnings

import numpy as np
import pandas as pd
import scipy
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

from refshim import load_reference  # noqa: E402
import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location('synth', os.path.join(ROOT, 'cna_amd', 'synth.py'))
synth = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synth)

cna = load_reference()
from cna.tools._nam import _nam as ref__nam  # noqa: E402


def _labels_array(x):
    x = np.asarray(x)
    if x.dtype == object:
        x = x.astype(str)
    return x


def add_isolated_blob(data, meta, n_iso=20, label=None):
    """Append a far-away blob of cells that all belong to one extra sample."""
    A = data.obsp['connectivities'].tocsr()
    n = A.shape[0]
    rs = np.random.RandomState(7)
    B = sp.random(n_iso, n_iso, density=0.6, random_state=rs, format='csr', dtype=np.float64)
    B = B + B.T
    B.setdiag(0)
    B.eliminate_zeros()
    B.data = np.clip(B.data, 0.05, 1.0)
    A2 = sp.block_diag([A, B.astype(A.dtype)], format='csr')
    A2.sort_indices()
    A2.indices = A2.indices.astype(np.int32)
    A2.indptr = A2.indptr.astype(np.int32)
    sidcol = data.obs.columns[0]
    old = data.obs[sidcol]
    lab = label if label is not None else (int(np.max(old)) + 1)
    new_obs = pd.DataFrame({sidcol: np.concatenate([np.asarray(old), np.repeat(lab, n_iso)])},
                           index=pd.Index(['cell_%d' % i for i in range(n + n_iso)], name='cell'))
    d2 = synth.CellData(new_obs, A2)
    return d2, lab


def result_fields(res):
    out = {}
    out['p'] = np.float64(res.p)
    out['k'] = np.int64(res.k)
    out['ks'] = np.asarray(res.ks, dtype=np.int64)
    out['r'] = np.int64(res.r)
    out['nullminps'] = np.asarray(res.nullminps)
    out['ncorrs'] = res.ncorrs.values
    out['kept'] = np.asarray(res.kept, dtype=bool)
    out['M'] = res.M.values
    out['nam'] = res.nam.values
    out['nam_index'] = _labels_array(res.nam.index)
    out['namresid'] = res.namresid.values
    out['U'] = res.namresid_sampleXpc.values
    out['V'] = res.namresid_nbhdXpc.values
    out['svs'] = res.namresid_svs.values
    out['varexp'] = res.namresid_varexp.values
    out['yresid_hat'] = np.asarray(res.yresid_hat)
    out['yresid'] = np.asarray(res.yresid)
    out['beta'] = np.asarray(res.beta)
    out['r2'] = np.float64(res.r2)
    out['r2_perpc'] = np.asarray(res.r2_perpc)
    out['nullr2_mean'] = np.float64(res.nullr2_mean)
    out['nullr2_std'] = np.float64(res.nullr2_std)
    if res.fdrs is not None:
        out['fdr_threshold'] = res.fdrs.threshold.values
        out['fdr_fdr'] = res.fdrs.fdr.values
        out['fdr_num_detected'] = res.fdrs.num_detected.values.astype(np.int64)
        out['fdr_5p_t'] = np.float64(np.nan if res.fdr_5p_t is None else res.fdr_5p_t)
        out['fdr_10p_t'] = np.float64(np.nan if res.fdr_10p_t is None else res.fdr_10p_t)
    return out


def build_cases():
    cases = []

    def base(name, n=1000, N=20, k=15, seed=0, gen=None, call=None, mutate=None, extras=()):
        cases.append(dict(name=name, gen=dict(n_cells=n, n_samples=N, k=k, seed=seed, **(gen or {})),
                          call=dict(call or {}), mutate=mutate, extras=tuple(extras)))

    base('c01_plain_f32', call=dict(nsteps=3, Nnull=200, seed=0), extras=('steps', 'diffuse', 'svd', 'nam', 'progress'))
    base('c02_covs_autostop', seed=1, gen=dict(n_covs=2), call=dict(Nnull=150, seed=1), extras=('steps', 'progress'))
    base('c03_covs_batches', seed=2, N=24, gen=dict(n_covs=1, n_batches=4), call=dict(nsteps=3, Nnull=120, seed=2),
         extras=('progress', 'nam'))
    base('c04_donorids', seed=3, N=30, call=dict(nsteps=2, Nnull=100, seed=3), mutate='donorids')
    base('c05_ks_f64', seed=4, gen=dict(graph_dtype='float64'), call=dict(nsteps=3, Nnull=100, seed=4, ks=[2, 5]),
         extras=('steps',))
    base('c06_nnull_cap', n=600, seed=5, call=dict(nsteps=3, Nnull=1300, seed=5))
    base('c07_no_local', seed=6, call=dict(nsteps=3, Nnull=100, seed=6, local_test=False))
    base('c08_force_permute_all', seed=7, N=24, gen=dict(n_batches=3),
         call=dict(nsteps=3, Nnull=100, seed=7, force_permute_all=True))
    base('c09_y_nan_extra_reordered', seed=8, N=26, gen=dict(n_covs=1), call=dict(nsteps=3, Nnull=100, seed=8),
         mutate='y_messy')
    base('c10_categorical_ids', seed=9, gen=dict(sid_kind='cat'), call=dict(nsteps=3, Nnull=100, seed=9))
    base('c11_string_ids_null_y', seed=10, gen=dict(sid_kind='str', signal=False), call=dict(nsteps=3, Nnull=100, seed=10))
    base('c12_batchy_qc', seed=11, N=40, n=1200, gen=dict(n_batches=10), call=dict(nsteps=3, Nnull=100, seed=11),
         mutate='batchy', extras=('nam', 'progress'))
    base('c13_zero_variance', seed=12, N=22, call=dict(nsteps=3, Nnull=100, seed=12), mutate='isolated')
    base('c14_selfweight_autostop_unsorted', seed=13, gen=dict(cluster_sorted=False), call=dict(Nnull=100, seed=13),
         extras=('steps', 'nam_sw2'))
    base('c15_ridges_custom', seed=14, N=30, gen=dict(n_batches=10, n_covs=1),
         call=dict(nsteps=3, Nnull=100, seed=14, ridges=[10.0, 1.0, 0.0]), mutate='batchy', extras=('progress',))
    base('c16_ridge_loop', seed=15, N=30, gen=dict(n_batches=10), call=dict(nsteps=3, Nnull=100, seed=15),
         mutate='batchy_all', extras=('progress',))
    # fewer than 10 samples (allowed explicitly), so few permutations that the p-value hits its floor
    base('c17_low_sample_size', seed=16, N=8, n=800, call=dict(nsteps=3, Nnull=20, seed=16, allow_low_sample_size=True))
    # covariates with a missing row, wider PC budget, auto-stopped walk, sample ids handed over unsorted
    base('c18_covs_nan_maxfrac', seed=17, N=28, gen=dict(n_covs=3, cluster_sorted=False),
         call=dict(Nnull=100, seed=17, max_frac_pcs=0.3), mutate='covs_nan', extras=('progress',))
    # categorical sample ids with a category that has no cells (what subsetting an AnnData leaves behind): its NAM row is
    # 0/0 = NaN; pandas' mean skips it in the batch means of _batch_kurtosis (_nam.py:78-82), and with the default stop
    # rule the median kurtosis is NaN, so the walk runs to maxnsteps (_nam.py:59-68)
    base('c19_unused_category_batches', seed=18, N=24, gen=dict(sid_kind='cat', n_batches=4),
         call=dict(nsteps=3, Nnull=100, seed=18), mutate='unused_category', extras=('nam', 'progress'))
    base('c20_unused_category_autostop', seed=19, N=22, gen=dict(sid_kind='cat'),
         call=dict(Nnull=100, seed=19), mutate='unused_category', extras=('progress',))
    # a whole batch of categories without cells: its batch mean is the mean of NaN rows only -- NaN, where pandas' mean had
    # skipped single NaN rows -- so every batch kurtosis is NaN and no neighbourhood passes the QC (_nam.py:78-99); the
    # reference goes on with an empty NAM and stops where the thresholds are formed (_association.py:99-102)
    base('c21_unused_batch', seed=20, N=24, gen=dict(sid_kind='cat', n_batches=4),
         call=dict(nsteps=3, Nnull=100, seed=20), mutate='unused_batch')
    # messy sample-level inputs, drawn at random (seeded): every input in an order of its own, NaNs, samples the data does
    # not have, unused categories, donor groups, custom ks / ridges / max_frac_pcs -- what pandas' label alignment makes
    # of them in the reference is pinned here case by case (tools/fuzz_oracle_vs_reference.py found the first of them)
    for i in range(1, 29):
        frs = np.random.RandomState(7000 + i)
        N = int(frs.choice([11, 14, 20, 26]))
        gen = dict(sid_kind=str(frs.choice(['int', 'str', 'cat'])), n_covs=int(frs.choice([0, 1, 2])),
                   n_batches=int(frs.choice([0, 0, 3, 5])), cluster_sorted=bool(frs.rand() < 0.5))
        call = dict(nsteps=[None, 2, 3][int(frs.randint(3))], Nnull=int(frs.choice([20, 50])), seed=100 + i)
        base('f%02d_messy' % i, n=500, N=N, k=10, seed=200 + i, gen=gen, call=call, mutate='fuzz:%d' % (7000 + i))
    # a NaN among the batch labels (garbage in; what the reference makes of it is still what a drop-in must make of it)
    base('f29_nan_batch', n=500, N=20, k=10, seed=231, gen=dict(n_batches=3), call=dict(nsteps=3, Nnull=50, seed=31),
         mutate='nan_batch')
    base('f30_nan_batch_covs_autostop', n=500, N=14, k=10, seed=232, gen=dict(n_batches=5, n_covs=1, sid_kind='str'),
         call=dict(Nnull=20, seed=32), mutate='nan_batch')
    # ... and when ks is also too large for the samples left, the reference's check of ks speaks first (_association.py:29-33)
    base('f32_nan_batch_ks_too_large', n=500, N=11, k=10, seed=234, gen=dict(n_batches=3), call=dict(nsteps=2, Nnull=50, seed=34, ks=[4, 9]),
         mutate='nan_batch')
    # a constant phenotype: 0/0 when it is standardised, every p-value NaN, the reference stops at their argmin
    base('f33_constant_phenotype', n=500, N=14, k=10, seed=235, call=dict(nsteps=2, Nnull=50, seed=35), mutate='constant_y')
    # integer ids, one sample of y has no cells, y in an order of its own and covs in another: the reference's positionally
    # paired filter lets the sample without cells through and the analysis dies in the SVD of a NaN Gram matrix
    base('f31_absent_sample_let_through', n=500, N=20, k=10, seed=233, gen=dict(n_covs=1),
         call=dict(nsteps=3, Nnull=50, seed=33), mutate='absent_misaligned')
    return cases


def fuzz_inputs(seed, data, meta, sid_name, call):
    """Seeded mutations of the sample-level inputs of one case -> (y, covs, batches, donorids); may edit data.obs and call."""
    rs = np.random.RandomState(seed)
    y, covs, batches, donor = meta['y'].copy(), meta['covs'], meta['batches'], None
    N = len(y)
    if rs.rand() < 0.3:
        y.iloc[int(rs.randint(N))] = np.nan
    if covs is not None and rs.rand() < 0.4:
        covs = covs.copy()
        covs.iloc[int(rs.randint(N)), 0] = np.nan
    col = data.obs[sid_name]
    if isinstance(col.dtype, pd.CategoricalDtype) and rs.rand() < 0.5:
        codes = np.asarray(col.cat.codes).copy()
        codes[codes == 2] = 3                                  # category 2 keeps its phenotype, not its cells
        data.obs[sid_name] = pd.Categorical.from_codes(codes, categories=col.cat.categories)
    if rs.rand() < 0.5:                                        # every sample-level input in an order of its own
        y = y.iloc[rs.permutation(len(y))]
        if covs is not None and rs.rand() < 0.6:
            covs = covs.iloc[rs.permutation(len(covs))]
        if batches is not None and rs.rand() < 0.6:
            batches = batches.iloc[rs.permutation(len(batches))]
    if rs.rand() < 0.25 and not isinstance(col.dtype, pd.CategoricalDtype) and np.asarray(col).dtype.kind in 'iu':
        extra = pd.Index([5000, 5001])                         # phenotypes of samples the data does not have
        y = pd.concat([y, pd.Series([0.3, -1.2], index=extra)])
        if covs is not None:
            covs = pd.concat([covs, pd.DataFrame(np.zeros((2, covs.shape[1])), index=extra, columns=covs.columns)])
        if batches is not None:
            batches = pd.concat([batches, pd.Series([0, 1], index=extra)])
    if batches is None and rs.rand() < 0.25 and N >= 14:
        donor = pd.Series(np.arange(len(y)) // 2, index=y.index)
        y = pd.Series(np.repeat(rs.randn((len(y) + 1) // 2), 2)[:len(y)], index=y.index)
    if rs.rand() < 0.2:
        call['force_permute_all'] = True
    if rs.rand() < 0.25:
        call['ks'] = [int(v) for v in sorted(rs.choice([1, 2, 3, 4], size=2, replace=False))]
    if rs.rand() < 0.2:
        call['max_frac_pcs'] = float(rs.choice([0.05, 0.3, 0.5]))
    if batches is not None and rs.rand() < 0.3:
        call['ridges'] = [float(v) for v in rs.choice([1e3, 10.0, 1.0, 0.0], size=2, replace=False)]
    return y, covs, batches, donor


def run_case(case):
    gen = dict(case['gen'])
    if 'graph_dtype' in gen:
        gen['graph_dtype'] = np.dtype(gen['graph_dtype']).type
    data, meta = synth.make_dataset(**gen)
    y, covs, batches = meta['y'], meta['covs'], meta['batches']
    donorids = None
    sid_name = 'id'
    mut = case['mutate']
    if mut == 'donorids':
        N = len(y)
        donor = np.arange(N) // 2
        donorids = pd.Series(donor, index=y.index)
        yv = np.random.RandomState(99).randn(N // 2 + 1)[donor]
        y = pd.Series(yv + 2.0 * meta['props'][:, 0][donor * 2], index=y.index)
        # y must be identical within donor
        y = pd.Series(y.groupby(donorids).transform('first').values, index=y.index)
    elif mut == 'y_messy':
        yv = y.copy()
        yv.iloc[3] = np.nan
        covs = covs.copy()
        covs.iloc[7, 0] = np.nan
        extra = pd.Series([0.5, -0.25], index=pd.Index([1000, 1001]))
        y = pd.concat([yv, extra])
        covs = pd.concat([covs, pd.DataFrame({'cov0': [0.1, 0.2]}, index=extra.index)])
        perm = np.random.RandomState(5).permutation(len(y))
        y = y.iloc[perm]
        covs = covs.iloc[perm]   # same order as y: the reference mis-aligns its sample filter otherwise
    elif mut == 'covs_nan':
        covs = covs.copy()
        covs.iloc[5, 1] = np.nan
        covs.iloc[11, 2] = np.nan
    elif mut == 'batchy':
        # make one cluster's cells come only from batch-0 samples so that batch
        # kurtosis of those neighbourhoods is extreme (_nam.py:85-99)
        cl = meta['cluster']
        b = batches.values
        sid = np.asarray(data.obs[sid_name]).copy()
        b0 = np.flatnonzero(b == 0)
        target = np.flatnonzero(cl == np.bincount(cl).argmax())
        rs = np.random.RandomState(3)
        sid[target] = rs.choice(b0, size=len(target))
        data.obs[sid_name] = sid
    elif mut == 'batchy_all':
        # every cluster is populated by the samples of a single batch, so that the
        # ridge schedule of _nam.py:142-156 has to iterate
        cl = meta['cluster']
        b = batches.values
        sid = np.asarray(data.obs[sid_name]).copy()
        rs = np.random.RandomState(4)
        for c in np.unique(cl):
            members = np.flatnonzero(cl == c)
            pool = np.flatnonzero(b == (c % (b.max() + 1)))
            sid[members] = rs.choice(pool, size=len(members))
        data.obs[sid_name] = sid
    elif mut is not None and mut.startswith('fuzz:'):
        case['call'] = dict(case['call'])
        y, covs, batches, donorids = fuzz_inputs(int(mut.split(':')[1]), data, meta, sid_name, case['call'])
    elif mut == 'absent_misaligned':
        sid = np.asarray(data.obs[sid_name]).copy()
        sid[sid == 5] = 6
        data.obs[sid_name] = sid
        y = y.iloc[np.random.RandomState(3).permutation(len(y))]    # (label 5 lands on position 15: the filter drops sample 15 and keeps 5)
    elif mut == 'constant_y':
        y = pd.Series(np.full(len(y), 2.0), index=y.index)
    elif mut == 'nan_batch':
        batches = batches.astype(float).copy()
        batches.iloc[4] = np.nan
    elif mut == 'unused_category':
        col = data.obs[sid_name]
        codes = np.asarray(col.cat.codes).copy()
        codes[codes == 5] = 6                                  # sample 5 keeps its category and its phenotype, not its cells
        data.obs[sid_name] = pd.Categorical.from_codes(codes, categories=col.cat.categories)
    elif mut == 'unused_batch':
        col = data.obs[sid_name]
        codes = np.asarray(col.cat.codes).copy()
        gone = np.flatnonzero(np.asarray(batches.reindex(col.cat.categories).values) == np.asarray(batches.values).max())
        for s_ in gone:                                        # every sample of the last batch keeps category, phenotype and
            codes[codes == s_] = s_ - 1                        # batch label -- and loses its cells to a sample of another batch
        data.obs[sid_name] = pd.Categorical.from_codes(codes, categories=col.cat.categories)
    elif mut == 'isolated':
        data, lab = add_isolated_blob(data, meta)
        y = pd.concat([y, pd.Series([np.nan], index=pd.Index([lab]))])

    call = dict(case['call'])
    out = {}
    A = data.obsp['connectivities']
    out['in_indptr'] = A.indptr.astype(np.int64)
    out['in_indices'] = A.indices.astype(np.int32)
    out['in_data'] = A.data
    col = data.obs[sid_name]
    if isinstance(col.dtype, pd.CategoricalDtype):
        out['in_sid_codes'] = np.asarray(col.cat.codes)
        out['in_sid_categories'] = _labels_array(col.cat.categories)
    else:
        out['in_sid'] = _labels_array(col.values)
    out['in_y_index'] = _labels_array(y.index)
    out['in_y'] = y.values.astype(np.float64)
    if covs is not None:
        out['in_covs_index'] = _labels_array(covs.index)
        out['in_covs'] = covs.values.astype(np.float64)
        out['in_covs_columns'] = _labels_array(covs.columns)
    if batches is not None:
        out['in_batches_index'] = _labels_array(batches.index)
        out['in_batches'] = batches.values
    if donorids is not None:
        out['in_donorids_index'] = _labels_array(donorids.index)
        out['in_donorids'] = donorids.values

    extras = case['extras']
    # ---- the headline call
    buf = io.StringIO()
    raised = None
    res = None
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter('always')
        with contextlib.redirect_stdout(buf):
            try:
                res = cna.tl.association(data, y, sid_name, batches=batches, covs=covs, donorids=donorids,
                                         return_full=True, show_progress=('progress' in extras), **call)
            except Exception as e:  # reference quirk: local_test=False dies at _association.py:235
                raised = e
    out['stdout'] = np.array(buf.getvalue())
    out['warnings'] = np.array(json.dumps([str(w.message) for w in wlist
                                           if issubclass(w.category, UserWarning)]))
    out['raised'] = np.array('' if raised is None else type(raised).__name__ + ': ' + str(raised))
    if 'coef' in data.obs:
        out['obs_coef'] = data.obs['coef'].values.astype(np.float64)
    if res is not None:
        out.update(result_fields(res))
        if res.fdrs is not None:
            out['obs_coef_fdr'] = data.obs['coef_fdr'].values.astype(np.float64)

    # ---- extras
    if 'steps' in extras:
        # per-step diffusion state of the sample indicators, as _nam.py:51,58 drives it
        S = pd.get_dummies(data.obs[sid_name])
        nst = call.get('nsteps') or 4
        steps = []
        for i, s in enumerate(cna.tl.diffuse_stepwise(data, S, maxnsteps=nst)):
            steps.append(np.asarray(s, dtype=np.float64))
        out['steps'] = np.stack(steps)
        import scipy.stats as st
        C = S.sum(axis=0).values
        out['steps_medkurt'] = np.array([np.median(st.kurtosis(s / C, axis=1)) for s in steps])
    if 'diffuse' in extras:
        rs = np.random.RandomState(11)
        s0 = rs.rand(A.shape[0], 3)
        out['diffuse_in'] = s0
        out['diffuse_out_2'] = np.asarray(cna.tl.diffuse(data, s0, 2))
        out['diffuse_out_2_sw05'] = np.asarray(cna.tl.diffuse(data, s0, 2, self_weight=0.5))
    if 'svd' in extras and res is not None:
        U, svs, V = cna.tl.svd_nam(res.nam)
        out['svd_U'] = U.values
        out['svd_svs'] = svs.values
        out['svd_V'] = V.values
    if 'nam' in extras:
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            NAM, keep = cna.tl.nam(data, sid_name, batches=batches, nsteps=call.get('nsteps'),
                                   show_progress=True)
        out['tlnam'] = NAM.values
        out['tlnam_index'] = _labels_array(NAM.index)
        out['tlnam_keep'] = np.asarray(keep, dtype=bool)
        out['tlnam_stdout'] = np.array(buf.getvalue())
    if 'nam_sw2' in extras:
        NAM, keep = cna.tl.nam(data, sid_name, nsteps=2, self_weight=2)
        out['tlnam_sw2'] = NAM.values

    out['call'] = np.array(json.dumps(call))
    out['sid_name'] = np.array(sid_name)
    out['versions'] = np.array(json.dumps(dict(numpy=np.__version__, scipy=scipy.__version__,
                                               pandas=pd.__version__, python=sys.version.split()[0],
                                               cna='0.2.3 (/root/reference)')))
    return out


def run_demo_like():
    """BASELINE.json configs[0]: the demo recipe (10 000 cells, 50 samples, 5 batches; makedata.ipynb) with
    the demo's analysis (demo.ipynb: association with `case`, covs = male, batches = batch), nsteps=3,
    Nnull=100.  Cells-sized matrices are kept for every 25th cell only (fixture size)."""
    data, samplem = synth.make_demo_like()
    call = dict(nsteps=3, Nnull=100, seed=0)
    y = samplem['case'].astype(float)
    covs = samplem[['male']].astype(float)
    batches = samplem['batch']
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter('always')
        res = cna.tl.association(data, y, 'id', batches=batches, covs=covs, return_full=True, **call)
    A = data.obsp['connectivities']
    out = dict(in_indptr=A.indptr.astype(np.int64), in_indices=A.indices.astype(np.int32), in_data=A.data,
               in_sid=np.asarray(data.obs['id'].values), call=np.array(json.dumps(call)))
    out['warnings'] = np.array(json.dumps([str(w.message) for w in wlist if issubclass(w.category, UserWarning)]))
    f = result_fields(res)
    sub = np.arange(0, A.shape[0], 25)
    kept_pos = np.flatnonzero(f['kept'])
    assert f['kept'].all()
    for key in ('p', 'k', 'ks', 'r', 'nullminps', 'ncorrs', 'kept', 'M', 'svs', 'varexp', 'yresid', 'yresid_hat', 'beta',
                'r2', 'r2_perpc', 'nullr2_mean', 'nullr2_std', 'fdr_threshold', 'fdr_fdr', 'fdr_num_detected',
                'fdr_5p_t', 'fdr_10p_t'):
        out[key] = f[key]
    out['sub'] = sub
    out['nam_sub'] = f['nam'][:, sub]
    out['namresid_sub'] = f['namresid'][:, sub]
    out['U'] = f['U']
    out['V_sub'] = f['V'][sub]
    out['obs_coef'] = data.obs['coef'].values.astype(np.float64)
    out['obs_coef_fdr'] = data.obs['coef_fdr'].values.astype(np.float64)
    out['versions'] = np.array(json.dumps(dict(numpy=np.__version__, scipy=scipy.__version__, pandas=pd.__version__,
                                               python=sys.version.split()[0], cna='0.2.3 (/root/reference)')))
    path = os.path.join(HERE, 'd01_demo_like.npz')
    np.savez_compressed(path, **out)
    print('%-36s p=%.6g k=%d detected@first=%d  %.0f KB' % ('d01_demo_like', out['p'], out['k'],
                                                           out['fdr_num_detected'][0], os.path.getsize(path) / 1024))


CONFIG2 = dict(n_cells=200_000, n_samples=50, k=30, seed=0)            # BASELINE.json configs[1]
CONFIG2_CALL = dict(nsteps=3, Nnull=1000, seed=0)
CONFIG2_EVERY = 100                                                     # cells-sized results: every 100th cell


CONFIG3 = dict(n_cells=1_000_000, n_samples=100, k=30, seed=0)          # BASELINE.json configs[2]
CONFIG3_EVERY = 1000


def run_config3():
    """BASELINE.json configs[2] (1M cells x 100 samples, the "HBM roofline run") at full size through the reference: ~15
    minutes and ~40 GB in the build container.  Same layout as d02 (results only; every 1000th cell of the cells-sized
    fields)."""
    run_config2(CONFIG3, CONFIG3_EVERY, 'd03_config3')


def run_config2(dataset=None, every=None, name='d02_config2'):
    """BASELINE.json configs[1] AT FULL SIZE through the reference itself: 200 000 cells x 50 samples, k = 30,
    nsteps = 3, Nnull = 1000, seed 0 (about two minutes and ~10 GB here).  The inputs are NOT stored: they are
    `synth.make_dataset(**CONFIG2, builder='cpu')`, regenerated wherever the fixture is used, and recognised by the
    digest of the CSR arrays.  Stored: every sample-level result, the FDR table, the integer counts, and the
    cells-sized results (nam, namresid, V, ncorrs, the two data.obs columns) for every 100th cell."""
    import time
    t0 = time.time()
    dataset = dataset or CONFIG2
    every = every or CONFIG2_EVERY
    data, meta = synth.make_dataset(builder='cpu', **dataset)
    A = data.obsp['connectivities']
    t_gen = time.time() - t0
    y = meta['y']
    t0 = time.time()
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter('always')
        res = cna.tl.association(data, y, 'id', return_full=True, **CONFIG2_CALL)
    t_ref = time.time() - t0
    f = result_fields(res)
    sub = np.arange(0, A.shape[0], every)
    assert f['kept'].all()
    out = dict(call=np.array(json.dumps(CONFIG2_CALL)), dataset=np.array(json.dumps(dataset)),
               graph_digest=np.array(synth.graph_digest(A)), nnz=np.int64(A.nnz), in_y=y.values,
               sid_digest=np.array(__import__('hashlib').sha256(np.asarray(data.obs['id'].values, dtype=np.int64).tobytes()).hexdigest()))
    out['warnings'] = np.array(json.dumps([str(w.message) for w in wlist if issubclass(w.category, UserWarning)]))
    for key in ('p', 'k', 'ks', 'r', 'nullminps', 'M', 'svs', 'varexp', 'yresid', 'yresid_hat', 'beta',
                'r2', 'r2_perpc', 'nullr2_mean', 'nullr2_std', 'fdr_threshold', 'fdr_fdr', 'fdr_num_detected',
                'fdr_5p_t', 'fdr_10p_t', 'U'):
        out[key] = f[key]
    out['n_kept'] = np.int64(f['kept'].sum())
    out['sub'] = sub
    out['ncorrs_sub'] = f['ncorrs'][sub]
    out['ncorrs_absmax'] = np.float64(np.abs(f['ncorrs']).max())
    out['ncorrs_sum'] = np.float64(f['ncorrs'].sum())
    out['nam_sub'] = f['nam'][:, sub]
    out['namresid_sub'] = f['namresid'][:, sub]
    out['V_sub'] = f['V'][sub]
    out['obs_coef_sub'] = data.obs['coef'].values.astype(np.float64)[sub]
    out['obs_coef_fdr_sub'] = data.obs['coef_fdr'].values.astype(np.float64)[sub]
    out['obs_coef_fdr_below_1'] = np.int64((data.obs['coef_fdr'].values < 1).sum())
    out['reference_seconds'] = np.float64(t_ref)
    out['versions'] = np.array(json.dumps(dict(numpy=np.__version__, scipy=scipy.__version__, pandas=pd.__version__,
                                               python=sys.version.split()[0], cna='0.2.3 (/root/reference)')))
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print('%-36s p=%.6g k=%d detected@first=%d  %.0f KB  (dataset %.0f s, reference %.0f s, graph %s)' % (
        name, out['p'], out['k'], out['fdr_num_detected'][0], os.path.getsize(path) / 1024, t_gen, t_ref,
        str(out['graph_digest'])[:16]))


def main():
    only = set(sys.argv[1:])
    if 'd02_config2' in only:                      # two minutes and ~10 GB: only on request
        run_config2()
        only.discard('d02_config2')
        if not only:
            return
    if 'd03_config3' in only:                      # a quarter of an hour and ~40 GB: only on request
        run_config3()
        only.discard('d03_config3')
        if not only:
            return
    if not only or 'd01_demo_like' in only:
        run_demo_like()
    if only == {'d01_demo_like'}:
        return
    for case in build_cases():
        if only and case['name'] not in only:
            continue
        out = run_case(case)
        path = os.path.join(HERE, case['name'] + '.npz')
        np.savez_compressed(path, **out)
        if 'p' in out:
            print('%-36s p=%.6g k=%d kept=%d/%d steps=%s  %.0f KB' % (
                case['name'], out['p'], out['k'], out['kept'].sum(), len(out['kept']),
                out['stdout'].item().count('taking step') or '-', os.path.getsize(path) / 1024))
        else:
            print('%-36s raised %s' % (case['name'], out['raised']))


if __name__ == '__main__':
    main()
