"""Reference-COST CPU baseline -- TEST / BENCH INFRASTRUCTURE, NOT PRODUCT CODE.

``oracle/cna_oracle.py`` restates WHAT immunogenomics/cna 0.2.3 computes, vectorised; it is
30x faster than the reference itself and therefore a flattering-to-nobody but unrepresentative
CPU baseline.  This module restates HOW the reference spends its time: the same library calls
on the same kinds of objects in the same order --

  * pandas frames for the walk state, `scipy.sparse` `csr.dot` per step, `scipy.stats.kurtosis`
    and the R^2 diagnostic after EVERY step             (/root/reference/src/cna/tools/_nam.py:44-76)
  * pandas `DataFrame.dot` / `.std` for the residualisation and the Gram SVD   (_nam.py:102-177)
  * a Python loop over the Nnull permuted phenotypes, each with a pandas dot, len(ks) small
    regressions and len(ks) calls of `scipy.stats.f.sf`                   (_association.py:35-61,84)
  * the materialised cells x Nnull null-correlation frame and one `np.histogram` per null column
                                                            (_association.py:94-99, _stats.py:34-62)
  * the per-cell `Series.apply` that looks the FDR up in the threshold table  (_association.py:234-237)

Only the case the benchmark uses is covered: no batches, no donor ids, optional covariates, every
sample of `y` present.  `tests/test_oracle_golden.py` checks that it returns the same numbers as
`cna_oracle.association(mode='reference')` (and hence as the goldens captured from the reference);
BASELINE.md §2 holds the reference's own stage timings, which `association()` below reproduces to
within the run-to-run spread when run in the same container.

Imported only by bench.py's cpu_baseline leg and by tests/."""
import time

import numpy as np
import pandas as pd
import scipy.sparse as sp
import scipy.stats as st


class _Clock:
    def __init__(self):
        self.t, self.stages = time.perf_counter(), {}

    def lap(self, name):
        now = time.perf_counter()
        self.stages[name] = self.stages.get(name, 0.0) + now - self.t
        self.t = now


def _walk(a, onehot, nsteps, self_weight=1):
    """_nam.py:21-34,44-76 with a fixed step count: frames in, frame out, diagnostics every step."""
    cells_per_sample = onehot.sum(axis=0)
    colsums = np.array(a.sum(axis=0)).flatten() + self_weight
    s, previous = onehot, np.zeros(onehot.shape)
    for _ in range(nsteps):
        scaled = s / colsums[:, None]
        s = a.dot(scaled) + self_weight * s / colsums[:, None]
        np.median(st.kurtosis(s / cells_per_sample, axis=1))            # computed and (fixed nsteps) unused: _nam.py:59
        centred = (s - s.mean(axis=0)) * (previous - previous.mean(axis=0))
        with np.errstate(all='ignore'):
            np.percentile((centred.mean(axis=0) / s.std(axis=0) / previous.std(axis=0)) ** 2, 20)   # _nam.py:47-49,60-63
        previous = s
    nam = (s / cells_per_sample).T
    return pd.DataFrame(nam, index=nam.index, columns=nam.columns, dtype=float)   # _nam.py:193


def _residualise(nam, covs):
    """_nam.py:118-135,158-163 (single batch) followed by svd_nam (_nam.py:102-115)."""
    n_samples = len(nam)
    x = nam - nam.mean(axis=0)
    if covs is None:
        m = pd.DataFrame(np.eye(n_samples), columns=x.index, index=x.index)
        r = 0
    else:
        c = (covs - covs.mean(axis=0)) / covs.std(axis=0)
        m = np.eye(n_samples) - c.dot(np.linalg.solve(c.T.dot(c), c.T))
        m.columns = m.index
        r = c.shape[1]
    x = m.dot(x)
    x = x / x.std(axis=0)
    x = pd.DataFrame(x, index=nam.index, columns=nam.columns)
    z = x - x.mean(axis=0)
    z = z / z.std(axis=0)
    u, svs, _ = np.linalg.svd(z.dot(z.T))
    v = z.T.dot(u) / np.sqrt(svs)                                      # cells x samples, unused by the test
    return m, r, x, u, svs, v


def association(data, y, sid_name, covs=None, nsteps=3, Nnull=1000, seed=0, ks=None):
    """One `cna.tl.association(data, y, sid_name, covs=covs, nsteps=nsteps, Nnull=Nnull, seed=seed)` at
    the reference's cost.  Returns dict(p, k, ncorrs, fdr (T), coef_fdr (per cell), stages {name: s})."""
    clk = _Clock()
    a = data.obsp['connectivities']
    onehot = pd.get_dummies(data.obs[sid_name])                        # _nam.py:51
    nam = _walk(a, onehot, nsteps).reindex(y.index)                    # _association.py:177-178
    std0 = nam.std(axis=0)                                             # zero-variance columns, _association.py:182
    assert not (std0 == 0).any(), 'benchmark inputs have no zero-variance cell'
    clk.lap('nam')

    m, r, x, u, svs, v = _residualise(nam, covs)
    clk.lap('resid_svd')

    np.random.seed(seed)                                               # _association.py:15-16
    yv = y.values.astype(float)
    yv = (yv - yv.mean()) / yv.std()
    n = len(yv)
    if ks is None:                                                     # _association.py:25-28
        incr = max(int(0.02 * n), 1)
        ks = np.arange(incr, max(min(4 * incr, int(n / 5)), 1) + 1, incr)

    def minp(z):                                                       # _association.py:35-61, one phenotype at a time
        zc = m.dot(z)
        zc = zc / zc.std()
        ps, r2s = [], []
        for k in ks:
            basis = u[:, :k]
            fitted = basis.dot(basis.T.dot(zc))
            sse_full = (fitted - zc).dot(fitted - zc)
            sse_red = zc.dot(zc)
            f = ((sse_red - sse_full) / k) / (sse_full / n)
            ps.append(st.f.sf(f, k, n - (1 + r + k)))
            r2s.append(1 - sse_full / sse_red)
        best = int(np.nanargmin(ps))
        return ks[best], ps[best], r2s[best]

    k, p_obs, _ = minp(yv)
    ncorrs = (yv[:, None] * x).mean(axis=0)                            # _association.py:77
    order = np.argsort(np.random.randn(n, Nnull), axis=0)              # conditional_permutation, one batch: _stats.py:4-18
    y_null = yv[order]
    nullminps = np.array([minp(col)[1] for col in y_null.T])           # the Python loop of _association.py:84
    p = ((nullminps <= p_obs + 1e-8).sum() + 1) / (Nnull + 1)
    clk.lap('global_test')

    n_local = min(1000, Nnull)                                         # _association.py:94-99
    yc = m.dot(y_null[:, :n_local])
    yc /= yc.std(axis=0)
    null_ncorrs = np.abs(x.T.dot(yc) / n)                              # cells x Nnull, materialised
    maxcorr = max(np.abs(ncorrs).max(), 0.001)
    thr = np.arange(maxcorr / 4, maxcorr, maxcorr / 400)
    edges = np.concatenate([thr ** 2 - 1e-8 - 1e-5 * thr ** 2, [np.inf]])     # _stats.py:47-50 (thresholds ascend)

    def tails(values):                                                 # one np.histogram per column, _stats.py:52-59
        hist = np.array([np.histogram(col, bins=edges)[0] for col in values.T ** 2])
        return np.flip(np.cumsum(np.flip(hist, axis=1), axis=1), axis=1)

    null_tails = tails(null_ncorrs.values)
    ranks = tails(ncorrs.values.reshape(-1, 1))
    with np.errstate(all='ignore'):
        fdr = (null_tails / ranks).mean(axis=0)
    table = pd.DataFrame({'threshold': thr, 'fdr': fdr,
                          'num_detected': [(np.abs(ncorrs) > t).sum() for t in thr]})   # _association.py:105-108
    clk.lap('local_test')

    coef = pd.Series(np.nan, index=data.obs.index)
    coef[:] = ncorrs.values

    def lookup(c):                                                     # _association.py:234-237, once per cell
        hit = table.loc[table.threshold <= abs(c)].fdr
        return hit.min() if not hit.empty else 1

    coef_fdr = coef.apply(lookup)
    clk.lap('percell_apply')
    return dict(p=p, k=int(k), ncorrs=ncorrs.values, nullminps=nullminps, fdr=fdr, threshold=thr,
                num_detected=table.num_detected.values, coef_fdr=coef_fdr.values, stages=clk.stages)
