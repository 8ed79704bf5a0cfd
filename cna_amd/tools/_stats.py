"""Sample-space statistics that stay on the host.

Permutation indices must be bit-identical to the reference, which draws them from numpy's
legacy global RNG (``np.random.seed`` / ``np.random.randn`` + ``argsort``; _stats.py:4-32,
_association.py:15-16).  They are O(samples x Nnull) and cost microseconds, so they are
generated here with exactly those calls, in exactly that order, and only the resulting
phenotype matrix goes to the device.  The cells-sized work of _stats.py:34-83 (tail counts,
empirical FDRs) lives in the HIP kernels (cna_null_local / cna_obs_counts).
"""
import numpy as np
import scipy.special as sc


def conditional_permutation(B, Y, num):
    """Permute Y within the levels of B, ``num`` times (reference _stats.py:4-18).

    RNG consumption: one ``randn(len(level), num)`` block per level, levels in
    ``np.unique`` order."""
    members = [np.flatnonzero(B == b) for b in np.unique(B)]
    shuffled = [m[np.argsort(np.random.randn(len(m), num), axis=0)] for m in members]
    src = np.zeros((len(Y), num), dtype=int)
    src[np.concatenate(members)] = np.concatenate(shuffled)
    return Y[src]


def grouplevel_permutation(G, Y, num):
    """Permute whole groups (donors): samples sharing a value of G keep a common Y
    (reference _stats.py:20-32)."""
    groups = np.unique(G)
    per_group = np.array([Y[G == g][0] for g in groups])
    which = np.array([np.where(groups == g)[0][0] for g in G])
    if (per_group[which] != Y).any():
        print('ERROR: the value of Y is not identical within each group of samples')
        return
    order = np.argsort(np.random.randn(len(per_group), num), axis=0)
    return per_group[order][which]


def default_ks(n):
    """PC counts tried by the global test when ``ks`` is not given (_association.py:25-28)."""
    incr = max(int(0.02 * n), 1)
    maxnpcs = max(min(4 * incr, int(n / 5)), 1)
    return np.arange(incr, maxnpcs + 1, incr)


def minp_stats(Z, M, U, ks, r):
    """Global F-test of every column of Z (samples x P) at once.

    Restates _reg/_stats/_minp_stats (_association.py:35-61): condition on covariates
    (M.z), scale by the ddof=1 std, regress on the first k sample-PCs for each k in ks,
    F-test against the null model, keep the k with the smallest p.
    Returns (index into ks, p, r2) per column.
    """
    n = Z.shape[0]
    Zc = M.dot(Z)
    Zc = Zc / Zc.std(axis=0, ddof=1)
    ssered = np.einsum('ij,ij->j', Zc, Zc)
    kmax = int(max(ks))
    Bt = U[:, :kmax].T.dot(Zc)                      # kmax x P projections
    # ||Uk Uk^T z - z||^2 = ||z||^2 - sum_{j<k} (U_j.z)^2 for orthonormal U: one cumulative sum
    # serves every k (agrees with forming the fitted values to ~1e-16/(1-r2))
    fitted = np.cumsum(Bt * Bt, axis=0)
    ssefull = ssered[None, :] - fitted[np.asarray(ks, dtype=int) - 1]
    kcol = np.asarray(ks, dtype=np.float64)[:, None]
    with np.errstate(all='ignore'):
        f = ((ssered - ssefull) / kcol) / (ssefull / n)
        # scipy.stats.f.sf(f, k, dfd) is special.fdtrc(k, dfd, f) inside the support, 1 at or
        # below it and NaN for non-positive degrees of freedom; one ufunc call instead of
        # len(ks) trips through the rv_continuous argument machinery.
        dfd = n - (1 + r + kcol)
        ps = np.where(f <= 0, 1.0, sc.fdtrc(kcol, dfd, f))
        ps = np.where((dfd > 0) & ~np.isnan(f), ps, np.nan)
        r2s = 1 - ssefull / ssered
    best = np.nanargmin(ps, axis=0)
    cols = np.arange(Z.shape[1])
    return best, ps[best, cols], r2s[best, cols]
