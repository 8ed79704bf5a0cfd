"""CPU oracle for the CNA hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A numpy/scipy restatement of the algorithm of immunogenomics/cna 0.2.3 for the path
NAM construction -> residualisation/PCA -> permutation association test -> local FDRs.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module, and only as the checker / the timed CPU baseline.  ``cna_amd`` never
imports it and has no CPU fallback.

Pinned: every function below is checked against golden vectors captured by running the
reference itself (tests/golden/make_golden.py, tests/test_oracle_golden.py).  The
reference ships no tests or known-answer vectors of its own, so the
goldens are the pin.

Each function cites the reference lines (under /root/reference/src/cna/tools/) it restates.
The restatement works on plain arrays: samples are identified by integer codes, matrices
are cells x samples ("cell-major", the layout the HIP path uses) unless said otherwise.

``mode``:
  'reference' reproduces the reference's dtype flow for a float32 graph (float32 column
              sums and a float32 first SpMM, _nam.py:28,33) -- matches goldens to ~1e-13;
  'f64'       does everything in float64 -- what the HIP kernels compute; differs from the
              reference by ~1e-7 relative on a float32 graph, 0 on a float64 graph.
"""
import numpy as np
import pandas as pd
import scipy.sparse as sp
import scipy.stats as st

DEFAULT_RIDGES = [1e5, 1e4, 1e3, 1e2, 1e1, 1e0, 1e-1, 1e-2, 1e-3, 1e-4, 0]


# --------------------------------------------------------------------------- diffusion
def _as_f64(A):
    """The same CSR with float64 values, entry for entry: `A.astype(float64)` would also SORT the indices of a row and SUM
    duplicate entries of a graph that is not in canonical form -- another order of the same sums than the reference's
    `a.dot(...)`, which walks the stored entries as they are (found by tools/fuzz_graphs.py)."""
    if A.data.dtype == np.float64:
        return A
    return sp.csr_matrix((A.data.astype(np.float64), A.indices, A.indptr), shape=A.shape)


def column_sums(A, self_weight=1, mode='reference'):
    """colsums = A.sum(axis=0) + self_weight   (_nam.py:28).

    scipy sums a CSR over axis 0 in the matrix dtype, accumulating each column in
    ascending row order; the python-int self weight does not upcast float32."""
    if mode == 'f64':
        A = _as_f64(A)
    return np.asarray(A.sum(axis=0)).ravel() + self_weight


def diffusion_step(A, s, colsums, self_weight=1, first_onehot=False, mode='reference'):
    """One random-walk step  s <- A.(s/colsums) + w*s/colsums   (_nam.py:33).

    ``first_onehot``: s is the boolean sample-indicator matrix of _nam.py:51; with a
    float32 graph the reference then evaluates the SpMM in float32 and the self term in
    float64 (bool/f32 -> f32, int*bool/f32 -> f64)."""
    if mode == 'f64':
        A64 = _as_f64(A)
        c = colsums.astype(np.float64)[:, None]
        s = s.astype(np.float64)
        return A64.dot(s / c) + self_weight * s / c
    c = colsums[:, None]
    if first_onehot:
        q = s.astype(bool) / c                      # bool / f32 -> f32 ; bool / f64 -> f64
        sw = (self_weight * s.astype(np.int64)) if float(self_weight).is_integer() \
            else (self_weight * s.astype(np.float64))
        return A.dot(q) + sw / c
    return A.dot(s / c) + self_weight * s / c


def diffuse(A, s, nsteps, self_weight=1, mode='reference'):
    """cna.tl.diffuse on a dense cells x m array (_nam.py:36-41)."""
    colsums = column_sums(A, self_weight, mode)
    for _ in range(nsteps):
        s = diffusion_step(A, s, colsums, self_weight, False, mode)
    return s


def row_kurtosis(x):
    """scipy.stats.kurtosis(x, axis=1) -- Fisher, biased (_nam.py:59,80):
    m4/m2^2 - 3 with central moments about the row mean; scipy returns NaN where
    m2 <= (eps*mean)^2 (catastrophic cancellation guard)."""
    x = np.asarray(x, dtype=np.float64)
    mean = x.mean(axis=1, keepdims=True)
    d = x - mean
    d2 = d * d
    m2 = d2.mean(axis=1)
    m4 = (d2 * d2).mean(axis=1)
    with np.errstate(all='ignore'):
        zero = m2 <= (np.finfo(np.float64).eps * mean[:, 0]) ** 2
        out = np.where(zero, np.nan, m4 / m2 ** 2)
    return out - 3.0


def column_r2(a, b):
    """R(A,B)^2 per column (_nam.py:47-49,60): population moments, NaN when a column of
    B is constant (first step: old_s = 0)."""
    with np.errstate(all='ignore'):
        r = ((a - a.mean(axis=0)) * (b - b.mean(axis=0))).mean(axis=0) / a.std(axis=0, ddof=1) / b.std(axis=0, ddof=1)
    return r ** 2


def build_nam(A, codes, n_samples, nsteps=None, maxnsteps=15, self_weight=1, mode='reference',
              want_diagnostics=False):
    """_nam (_nam.py:44-76).  codes[i] in [0, n_samples) is the column of cell i in
    pd.get_dummies(obs[sid]) (sorted labels / category order).  Returns a dict with
    nam (cells x samples = (s/C), i.e. the transpose of the reference's N x cells frame),
    steps taken, medkurt per step, and optionally the R2 diagnostic per step."""
    n = A.shape[0]
    S = np.zeros((n, n_samples), dtype=bool)
    S[np.arange(n), codes] = True
    C = S.sum(axis=0)
    colsums = column_sums(A, self_weight, mode)
    prevmedkurt = np.inf
    s = S
    old = np.zeros(S.shape)
    medkurts, r2p20 = [], []
    taken = 0
    for i in range(maxnsteps):
        s = diffusion_step(A, s, colsums, self_weight, first_onehot=(i == 0), mode=mode)
        taken = i + 1
        with np.errstate(all='ignore'):
            medkurt = np.median(row_kurtosis(s / C))
        medkurts.append(medkurt)
        if want_diagnostics:
            with np.errstate(all='ignore'):
                r2p20.append(np.percentile(column_r2(s, old), 20))
            old = s
        if nsteps is None:
            if prevmedkurt - medkurt < 3 and i + 1 >= 3:
                break
            prevmedkurt = medkurt
        elif i + 1 == nsteps:
            break
    with np.errstate(all='ignore'):
        nam = s / C
    return dict(nam=nam, nsteps=taken, medkurt=np.array(medkurts), r2p20=np.array(r2p20),
                stopped_auto=(nsteps is None), C=C, colsums=colsums, S_last=s)


def batch_kurtosis(nam, batch_codes, n_batches):
    """_batch_kurtosis (_nam.py:78-82): per cell, Pearson kurtosis (Fisher + 3) over the
    per-batch means of the cell's NAM entries.  nam is cells x samples."""
    means = np.empty((nam.shape[0], n_batches))
    for b in range(n_batches):
        part = nam[:, batch_codes == b]
        # the reference takes DataFrame.mean, which skips NaN: a sample without cells (an unused category of a categorical
        # id column: its NAM row is 0/0) does not count in its batch's mean (fixture c19_unused_category_batches)
        ok = ~np.isnan(part)
        with np.errstate(invalid='ignore', divide='ignore'):
            means[:, b] = np.where(ok, part, 0.0).sum(axis=1) / ok.sum(axis=1)
    return row_kurtosis(means) + 3.0


def qc_keep(nam, batch_codes, n_batches):
    """_qc_nam (_nam.py:85-99) -> (keep mask, threshold or None)."""
    if n_batches == 1:
        return np.ones(nam.shape[0], dtype=bool), None
    kurt = batch_kurtosis(nam, batch_codes, n_batches)
    threshold = max(6, 2 * np.median(kurt))
    with np.errstate(invalid='ignore'):
        keep = kurt < threshold
    return keep, threshold


# ------------------------------------------------------------------ residualise + PCA
def _std1(x, axis):
    return x.std(axis=axis, ddof=1)


def svd_nam(X):
    """svd_nam (_nam.py:102-115) on a cells x samples matrix: re-centre and re-standardise
    each cell over samples (ddof=1), SVD of the samples x samples Gram, V = X U / sqrt(svs)."""
    X = X - X.mean(axis=1, keepdims=True)
    with np.errstate(all='ignore'):
        X = X / _std1(X, 1)[:, None]
    G = X.T.dot(X)
    U, svs, _ = np.linalg.svd(G)
    with np.errstate(all='ignore'):
        V = X.dot(U) / np.sqrt(svs)
    return U, svs, V, G


def resid_nam(X, covs=None, batch_codes=None, n_batches=1, ridges=None, npcs=None):
    """_resid_nam (_nam.py:118-177).  X: cells x samples NAM.  covs: samples x c or None.
    Returns dict(M, r, namresid (cells x samples), U, svs (first npcs), svs_all, V, varexp,
    ridge_log [(ridge, median batch kurtosis)])."""
    n, N = X.shape
    X = X - X.mean(axis=1, keepdims=True)
    if covs is None:
        covs = np.ones((N, 0))
    else:
        covs = np.asarray(covs, dtype=np.float64)
        covs = (covs - covs.mean(axis=0)) / _std1(covs, 0)
    log = []
    if batch_codes is None or n_batches == 1:
        C = covs
        if C.shape[1] == 0:
            M = np.eye(N)
        else:
            M = np.eye(N) - C.dot(np.linalg.solve(C.T.dot(C), C.T))
        X = X.dot(M.T)
    else:
        B = np.zeros((N, n_batches))
        B[np.arange(N), batch_codes] = 1
        B = (B - B.mean(axis=0)) / _std1(B, 0)
        C = np.concatenate([B, covs], axis=1)
        if ridges is None:
            ridges = DEFAULT_RIDGES
        for ridge in ridges:
            L = np.diag([1] * n_batches + [0] * (C.shape[1] - n_batches))
            M = np.eye(N) - C.dot(np.linalg.solve(C.T.dot(C) + ridge * N * L, C.T))
            X = X.dot(M.T)
            med = np.median(batch_kurtosis(X, batch_codes, n_batches))
            log.append((ridge, med))
            if med <= 6:
                break
    with np.errstate(all='ignore'):
        X = X / _std1(X, 1)[:, None]
    U, svs, V, G = svd_nam(X)
    if npcs is None:
        npcs = N
    return dict(M=M, r=C.shape[1], namresid=X, U=U, svs=svs[:npcs], svs_all=svs, V=V,
                varexp=svs / N / n, ridge_log=log, gram=G)


# ------------------------------------------------------------------------ permutations
def conditional_permutation(B, Y, num):
    """_stats.py:4-18.  Uses numpy's global legacy RNG exactly like the reference."""
    batchind = [np.where(B == b)[0] for b in np.unique(B)]
    ix = np.concatenate([bi[np.argsort(np.random.randn(len(bi), num), axis=0)] for bi in batchind])
    bix = np.zeros((len(Y), num)).astype(int)
    bix[np.concatenate(batchind)] = ix
    return Y[bix]


def grouplevel_permutation(G, Y, num):
    """_stats.py:20-32."""
    Gu = np.unique(G)
    Yg = np.array([Y[G == g][0] for g in Gu])
    Gind = np.array([np.where(Gu == g)[0][0] for g in G])
    if (Yg[Gind] != Y).any():
        print('ERROR: the value of Y is not identical within each group of samples')
        return None
    ix = np.argsort(np.random.randn(len(Yg), num), axis=0)
    return Yg[ix][Gind]


def default_ks(n):
    """_association.py:25-28."""
    incr = max(int(0.02 * n), 1)
    maxnpcs = max(min(4 * incr, int(n / 5)), 1)
    return np.arange(incr, maxnpcs + 1, incr)


def minp_stats(Z, M, U, ks, r):
    """_reg/_stats/_minp_stats (_association.py:35-61) for every column of Z (samples x P)
    at once.  Returns (k index chosen, p, r2) per column.  Algebra: with orthonormal U,
    ssefull = ||z||^2 - sum_{j<k} (U_j.z)^2."""
    n = Z.shape[0]
    Zc = M.dot(Z)
    Zc = Zc / _std1(Zc, 0)
    ssered = (Zc * Zc).sum(axis=0)
    ps = np.empty((len(ks), Z.shape[1]))
    r2s = np.empty_like(ps)
    for a, k in enumerate(ks):
        beta = U[:, :k].T.dot(Zc)
        resid = U[:, :k].dot(beta) - Zc
        ssefull = (resid * resid).sum(axis=0)
        with np.errstate(all='ignore'):
            f = ((ssered - ssefull) / k) / (ssefull / n)
            ps[a] = st.f.sf(f, k, n - (1 + r + k))
            r2s[a] = 1 - ssefull / ssered
    kix = np.nanargmin(ps, axis=0)
    cols = np.arange(Z.shape[1])
    return kix, ps[kix, cols], r2s[kix, cols]


# ------------------------------------------------------------------------- local test
def tail_counts(thresholds, znull, atol=1e-8, rtol=1e-5):
    """tail_counts(z=thresholds, znull) of _stats.py:34-62 for ascending thresholds:
    tails[j, t] = #{i : znull[i, j]^2 >= thr_t^2 - atol - rtol*thr_t^2}."""
    znull = np.asarray(znull)
    if znull.ndim == 1:
        znull = znull.reshape(-1, 1)
    z2 = np.asarray(thresholds) ** 2
    edges = z2 - atol - rtol * z2
    out = np.empty((znull.shape[1], len(edges)), dtype=np.int64)
    for j in range(znull.shape[1]):
        v = np.sort(znull[:, j] ** 2)
        v = v[~np.isnan(v)]
        out[j] = len(v) - np.searchsorted(v, edges, side='left')
    return out


def empirical_fdrs(z, znull, thresholds):
    """_stats.py:64-83."""
    tails = tail_counts(thresholds, znull)
    ranks = tail_counts(thresholds, z)
    with np.errstate(all='ignore'):
        fdp = tails / ranks
    return fdp.mean(axis=0)


def percell_fdr(coef, thresholds, fdr):
    """association epilogue (_association.py:234-237): min{fdr_t : thr_t <= |coef|},
    1 if no threshold qualifies (or coef is NaN)."""
    coef = np.asarray(coef, dtype=np.float64)
    with np.errstate(invalid='ignore'):
        run = np.fmin.accumulate(np.asarray(fdr, dtype=np.float64))
    idx = np.searchsorted(thresholds, np.abs(coef), side='right') - 1
    out = np.ones(len(coef))
    ok = (idx >= 0) & ~np.isnan(coef)
    out[ok] = run[idx[ok]]
    return out


def association_core(namresid, M, r, U, y, batches, donorids=None, ks=None, Nnull=1000,
                     force_permute_all=False, local_test=True, seed=None):
    """_association (_association.py:10-129).  namresid is cells x samples."""
    if seed is not None:
        np.random.seed(seed)
    if force_permute_all:
        batches = np.ones(len(y))
    y = (y - y.mean()) / y.std()
    n = len(y)
    if ks is None:
        ks = default_ks(n)
    ks = np.asarray(ks)
    if max(ks) + r >= n:
        raise ValueError('Maximum number of PCs plus number of covariates must be less than n-1.')
    kix, p, r2 = minp_stats(y[:, None], M, U, ks, r)
    k, p, r2 = int(ks[kix[0]]), p[0], r2[0]
    ycond = M.dot(y)
    ycond = ycond / ycond.std(ddof=1)
    beta = U[:, :k].T.dot(ycond)
    yhat = U[:, :k].dot(beta)
    r2_perpc = (beta / np.sqrt(ycond.dot(ycond))) ** 2
    ncorrs = (namresid * y[None, :]).sum(axis=1) / n
    if donorids is not None:
        y_ = grouplevel_permutation(donorids, y, Nnull)
    else:
        y_ = conditional_permutation(batches, y, Nnull)
    _, nullminps, nullr2s = minp_stats(y_, M, U, ks, r)
    hits = int((nullminps <= p + 1e-8).sum())
    pfinal = (hits + 1) / (Nnull + 1)
    res = dict(p=pfinal, p_obs=p, nullminps=nullminps, k=k, ncorrs=ncorrs, ks=ks, beta=beta, r2=r2,
               r2_perpc=r2_perpc, yresid=ycond, yresid_hat=yhat, nullr2_mean=nullr2s.mean(),
               nullr2_std=nullr2s.std(), at_floor=(hits == 0), k_is_max=(k == max(ks)),
               fdrs=None, fdr_5p_t=None, fdr_10p_t=None, y_perm=y_)
    if local_test:
        P = min(1000, Nnull)
        yc = M.dot(y_[:, :P])
        yc = yc / _std1(yc, 0)
        nullncorrs = np.abs(namresid.dot(yc) / n)
        maxcorr = max(np.abs(ncorrs).max(), 0.001)
        thr = np.arange(maxcorr / 4, maxcorr, maxcorr / 400)
        fdr = empirical_fdrs(ncorrs, nullncorrs, thr)
        num = np.array([(np.abs(ncorrs) > t).sum() for t in thr], dtype=np.int64)
        res['fdrs'] = dict(threshold=thr, fdr=fdr, num_detected=num)
        res['tails'] = tail_counts(thr, nullncorrs)
        res['ycond_null'] = yc
        with np.errstate(invalid='ignore'):
            if not np.min(fdr) > 0.05:
                res['fdr_5p_t'] = thr[fdr <= 0.05][0] if (fdr <= 0.05).any() else None
            if not np.min(fdr) > 0.1:
                res['fdr_10p_t'] = thr[fdr <= 0.1][0] if (fdr <= 0.1).any() else None
    return res


# ------------------------------------------------------------------ pandas-facing shell
def sample_codes(obs_col):
    """Column order of pd.get_dummies(obs[sid]) (_nam.py:51): categories for a
    categorical column (unused ones included), else sorted unique labels."""
    if isinstance(obs_col.dtype, pd.CategoricalDtype):
        return np.asarray(obs_col.cat.codes, dtype=np.int64), pd.Index(obs_col.cat.categories)
    codes, uniques = pd.factorize(obs_col, sort=True)
    return codes.astype(np.int64), pd.Index(uniques)


def get_graph(data):
    A = data.obsp['connectivities']
    return sp.csr_matrix(A)


def nam(data, sid_name, batches=None, nsteps=None, self_weight=1, mode='reference'):
    """cna.tl.nam (_nam.py:179-193) -> dict(nam cells_kept x samples, keep, labels, info)."""
    A = get_graph(data)
    codes, labels = sample_codes(data.obs[sid_name])
    info = build_nam(A, codes, len(labels), nsteps=nsteps, self_weight=self_weight, mode=mode)
    if batches is None:
        keep, thr = np.ones(A.shape[0], dtype=bool), None
    else:
        b = batches.reindex(labels).values
        if pd.isna(b).any() and len(np.unique(batches)) > 1:
            # a NaN label is a level of np.unique that `batches == b` never matches: the mean of its (no) members is NaN,
            # every batch kurtosis NaN, no neighbourhood below the threshold (_nam.py:78-99; fixtures f29 / f30)
            keep, thr = np.zeros(A.shape[0], dtype=bool), 6
        else:
            ub = np.unique(b)
            bc = np.searchsorted(ub, b)
            keep, thr = qc_keep(info['nam'], bc, len(ub))
    return dict(nam=info['nam'][keep], keep=keep, labels=labels, info=info, qc_threshold=thr)


def association(data, y, sid_name, batches=None, covs=None, donorids=None, ks=None,
                max_frac_pcs=0.15, nsteps=None, ridges=None, mode='reference', allow_low_sample_size=False,
                **kwargs):
    """cna.tl.association (_association.py:193-242) on pandas inputs; returns a dict of arrays
    (cells x samples orientation for nam / namresid)."""
    obs_sid = data.obs[sid_name]
    if batches is not None and donorids is not None:
        raise ValueError('batches and donorids are mutually exclusive')
    if not set(obs_sid).issubset(set(y.index)):
        raise ValueError("'data[sid_name]' contains values not present in the index of 'y'.")
    if batches is None:
        batches = pd.Series(np.ones(len(y)), index=y.index)
    present = y.index.isin(obs_sid.unique())
    if covs is not None:
        filt = ~(y.isna() | covs.isna().any(axis=1)) & present
    else:
        filt = ~np.isnan(y) & present
    if filt.sum() < 10 and not allow_low_sample_size:       # _association.py:163-172
        raise ValueError('You are supplying phenotype information on fewer than 10 samples.')
    nm = nam(data, sid_name, batches=batches, nsteps=nsteps, mode=mode)
    labels, kept = nm['labels'], nm['keep'].copy()
    if not kept.any() and kwargs.get('local_test', True):
        # no neighbourhood left: the reference goes on with a NAM of no columns until the thresholds are formed from the
        # largest of no coefficients (_association.py:99-102)
        raise ValueError('arange: cannot compute length')
    # The filter above is what the reference forms (_association.py:153-160): when y and covs come in different orders,
    # `y.isna() | covs.isna().any(axis=1)` is indexed by the SORTED UNION of the two indices, while `present` is an array
    # in y's order -- the `&` pairs them by position.  From here on the reference uses that Series as a boolean indexer
    # of frames indexed by y.index (NAM.reindex(y.index)[filter_samples], y[filter_samples], ...): pandas aligns it by
    # LABEL (and refuses it when a label of the frame is missing in it).  Fixtures f02 / f03 / f24.
    if isinstance(filt, pd.Series) and not filt.index.equals(y.index):
        aligned = filt.reindex(y.index)
        if aligned.isna().any():
            raise pd.errors.IndexingError('Unalignable boolean Series provided as indexer (index of the boolean Series and of '
                                          'the indexed object do not match).')
        filt = aligned.astype(bool)
    # NAM.reindex(y.index)[filter_samples]: sample axis follows y.index order
    pos = labels.get_indexer(y.index[filt.values])
    X = nm['nam'][:, pos]
    if (pos < 0).any():
        X = X.copy()
        X[:, pos < 0] = np.nan                                # a label the NAM does not have: reindex gives a NaN row
    sd = _std1(X, 1)
    zero = np.flatnonzero(sd == 0)
    nz = np.flatnonzero(kept)
    kept[nz[zero]] = False
    X = np.delete(X, zero, axis=0)
    batches = batches.reindex(y.index)
    covs = covs.reindex(y.index) if covs is not None else None
    donorids = donorids.reindex(y.index) if donorids is not None else None
    filt = filt.reindex(y.index)
    N = int(filt.sum())
    npcs = min(N, max([10] + [int(max_frac_pcs * N)] + [ks if ks is not None else []][0]))
    bvals = batches[filt].values
    ub = np.unique(bvals)
    rr = resid_nam(X, covs[filt].values if covs is not None else None,
                   np.searchsorted(ub, bvals), len(ub), ridges=ridges, npcs=npcs)
    core = association_core(rr['namresid'], rr['M'], rr['r'], rr['U'], y[filt].values.astype(np.float64),
                            bvals, donorids[filt].values if donorids is not None else None,
                            ks=ks, **kwargs)
    out = dict(core)
    out.update(M=rr['M'], r=rr['r'], namresid=rr['namresid'], U=rr['U'], svs=rr['svs'], V=rr['V'],
               varexp=rr['varexp'], ridge_log=rr['ridge_log'], nam=X, kept=kept, nsteps=nm['info']['nsteps'],
               medkurt=nm['info']['medkurt'], sample_labels=np.asarray(y.index[filt.values]))
    coef = np.full(len(kept), np.nan)
    coef[kept] = core['ncorrs']
    out['obs_coef'] = coef
    if core['fdrs'] is not None:
        out['obs_coef_fdr'] = percell_fdr(coef, core['fdrs']['threshold'], core['fdrs']['fdr'])
    return out
