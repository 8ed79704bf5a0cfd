"""Permutation-based association test between a sample-level phenotype and the NAM.

Same call surface, result fields, warnings and error behaviour as the reference's
``cna/tools/_association.py``.  Division of labour:

  host (numpy, O(samples^2 x Nnull))      input checks, sample filter, M, LAPACK SVD of the
                                          N x N Gram, permutation indices from numpy's legacy
                                          RNG, batched global F-tests (_association.py:35-88)
  device (HIP, O(cells x ...))            NAM, QC, residualisation, Gram, neighbourhood
                                          coefficients (:77), local null correlations fused with
                                          the tail counting of _stats.py:34-62 (:96-103),
                                          observed ranks / num_detected (:105-108),
                                          data.obs columns incl. the per-cell FDR lookup (:230-237)
"""
import os
import warnings

import numpy as np
import pandas as pd
import scipy.sparse as sp

from .. import _ffi
from ..engine import get_engine
from ._nam import (LazyNamespace, _nam_device, _qc_device, _resid_plan, _resid_run, sample_codes_cached, confirm_codes,
                   shard_of, global_samples, get_connectivity,
                   _small_svd, _defer_pcs, host_blas_threads, _top_pcs, GramPCs)
from . import _nam as _nam_mod
from . import _fast
from ._out import select_output
from ._stats import conditional_permutation, grouplevel_permutation, default_ks, minp_stats, native_draw_start


_pool = None


def _background():
    """One helper thread: lets a blocking device call (ctypes drops the GIL) run while the
    host does sample-space numpy work.  It shares the engine's context with the main thread in
    exactly one place: after the permutation draw it may call engine.condition() -- sample space
    only, the second stream, buffers nobody else touches (zc / gt) -- while the main thread issues
    set_samples / nam_step / select.  The library serialises its allocator for that
    (cna_ctx::alloc_mu) and keeps its last-error text per thread."""
    global _pool
    if _pool is None:
        from concurrent.futures import ThreadPoolExecutor
        _pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix='cna-device')
    return _pool


_EARLY_WALK = True       # the walk queued before the (pandas) validation of the sample-level inputs
_EARLY_COEF = True       # (module constants, not environment switches: each was measured against its alternative --
_EARLY_FDR = True        #  DESIGN.md 6 -- and tests that need the other side patch the attribute)
_DRAW_THREAD = True
_PREFETCH_GRAPH = True    # a new graph's upload beside the factorisation of the sample ids
_PREFETCH_CELLS = 100_000
_NATIVE_DRAW = True       # the draw on the library's host thread (False: the interpreter's helper thread; tests patch it)
_SWITCH_INTERVAL = 5e-5   # GIL hand-over between the helper thread and this one: 2.08 -> 1.86 ms per call at 200k cells (default interval: 5 ms)


import threading as _threading

_switch_lock = _threading.Lock()
_switch_users = 0
_switch_saved = None


def _switch_interval_enter():
    """The interpreter's switch interval is process-wide: calls running at the same time share one change (the
    first sets it, the last puts the caller's value back), so that overlapping calls cannot leave it shortened."""
    global _switch_users, _switch_saved
    if _SWITCH_INTERVAL <= 0:
        return
    import sys
    with _switch_lock:
        if _switch_users == 0:
            _switch_saved = sys.getswitchinterval()
            sys.setswitchinterval(_SWITCH_INTERVAL)
        _switch_users += 1


def _switch_interval_exit():
    global _switch_users, _switch_saved
    if _SWITCH_INTERVAL <= 0:
        return
    import sys
    with _switch_lock:
        _switch_users -= 1
        if _switch_users == 0 and _switch_saved is not None:
            sys.setswitchinterval(_switch_saved)
            _switch_saved = None


class _InlineJob:
    """Future-like wrapper of a job that runs on the calling thread, at most once."""

    def __init__(self, fn):
        self._fn, self._done, self._val = fn, False, None

    def run(self):
        if not self._done:
            self._val, self._done = self._fn(), True

    def result(self):
        self.run()
        return self._val

    def cancel(self):
        self._done = True
        return True

    def exception(self):
        return None
_FUSE = True

_TRACE = None      # list of (label, perf_counter) when tools/host_trace.py switches tracing on


_COEF_FIRST_CELLS = 500000
# From this many cells on (and three or more steps) the walk's last step is queued after validation and planning, which
# run under the first steps; it then knows what the selection pass will be asked for and does it on its way out
# (compute_nam_and_reindex).  Below, the first steps are too short to hide the host work.
# (150 000: at 250 000 cells x 200 samples -- a rank's block of the 2M problem on eight GPUs -- the by-product saves the
# selection pass, 4.23-4.48 -> 4.01-4.29 ms; at 200 000 x 50 it is neutral, 1.60 -> 1.58; it was 300 000 before round 4's
# last day.)  Sharded inputs: the size of the largest block, the same number on every rank.
_DEFER_LAST_CELLS = int(os.environ.get('CNA_DEFER_LAST_CELLS', '150000'))


def _rule_cells(data, engine):
    shard = getattr(data, 'uns', {}).get('cna_shard') if hasattr(data, 'uns') else None
    if shard:
        return -(-int(shard['n_global']) // max(1, int(getattr(engine, 'nranks', 1))))
    return len(data.obs)


def _host_copy(dst, src):
    """dst[:] = src for two contiguous float64 arrays, on several threads when the library is there."""
    try:
        from .. import _ffi
        from .._order import usable_cpus
        lib = _ffi.load()
        if lib.cna_host_copy(dst.ctypes.data, src.ctypes.data, dst.nbytes, min(4, usable_cpus(4))) == 0:   # 4: best on the box (thread start-up beyond)
            return
    except Exception:
        pass
    np.copyto(dst, src)


def _mark(label):
    if _TRACE is not None:
        import time
        _TRACE.append((label, time.perf_counter()))


def _draw_null(y, batches, donorids, Nnull=1000, force_permute_all=False, seed=None):
    """Host-only head of the reference's ``_association`` (_association.py:15-22,79-83):
    seed numpy's global RNG, standardise y (ddof=0) and draw the permuted phenotypes.
    It needs nothing from the device, so the caller runs it while the diffusion kernels
    execute; the RNG is consumed by nothing else in between, so the draws are the same."""
    if seed is not None:
        np.random.seed(seed)
    clean = seed is not None           # freshly seeded: no cached normal pending (RandomState._reset_gauss)
    if force_permute_all:
        batches = np.ones(len(y))
    y = (y - y.mean()) / y.std()
    if donorids is not None:
        y_ = grouplevel_permutation(donorids, y, Nnull, clean=clean)
    else:
        y_ = conditional_permutation(batches, y, Nnull, clean=clean)
    return y, y_


def _forget_pools():
    """In a forked child the worker threads of these pools do not exist; the next user makes new ones."""
    global _pool, _EIG_POOL
    _pool = _EIG_POOL = None


if hasattr(os, 'register_at_fork'):
    os.register_at_fork(after_in_child=_forget_pools)

_TAIL_FIRST_SAMPLES = 128
_TAIL_FIRST = True        # (measured against the sequential order at 200 000 x 50 and 250 000 x 200: DESIGN.md 7 e)
_EIG_POOL = None


def _eig_pool():
    """One worker for LAPACK beside the per-cell pass of a small problem (_nam._top_pcs calls it without the GIL)."""
    global _EIG_POOL
    if _EIG_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _EIG_POOL = ThreadPoolExecutor(max_workers=1, thread_name_prefix='cna-eig')
    return _EIG_POOL


def _fdr_tables(tail_sums, ranks, Nloc, thresholds):
    """(fdr per threshold, first threshold at 5 %, at 10 %, running minimum) from the local null's sums
    (_association.py:105-118, _stats.py:79-80)."""
    fdr_5p_t = fdr_10p_t = None
    with np.errstate(all='ignore'):
        # mean over permutations of tails/ranks (_stats.py:79-80) from the per-threshold sums
        fdr_vals = tail_sums / ranks / Nloc
    # the reference takes np.min of a pandas Series (_association.py:111-118), which skips NaN (0/0 at
    # thresholds nothing reaches); a table without a single finite entry fails there with an
    # IndexError, and so does this
    # (np.nanmin without its all-NaN RuntimeWarning -- and without warnings.catch_warnings(), which copies the process's
    # filter list and voids every module's warning registry: 30 us on the tail of every call)
    with np.errstate(invalid='ignore'):
        finite = fdr_vals[~np.isnan(fdr_vals)]
        fdr_min = finite.min() if finite.size else np.nan
        if not fdr_min > 0.05:
            fdr_5p_t = thresholds[np.flatnonzero(fdr_vals <= 0.05)[0]]
        if not fdr_min > 0.1:
            fdr_10p_t = thresholds[np.flatnonzero(fdr_vals <= 0.1)[0]]
    with np.errstate(invalid='ignore'):
        runmin = np.fmin.accumulate(fdr_vals)
    return fdr_vals, fdr_5p_t, fdr_10p_t, runmin


def _association(engine, res, y, y_, ks=None, Nnull=1000, local_test=True, show_progress=False,
                 npcs=None, n_cells=None, conditioned=False, null_source=None, maxabs=None, on_coef=None,
                 coef_first=False, coef_launched=False, full=False, on_fdr=None):
    """Body of the reference's ``_association`` (_association.py:24-129) against the
    residualised NAM held by ``engine`` (cells x samples), whose Gram-matrix kernels have been
    queued.  ``res`` is the namespace from the residualisation (M, r), ``y`` / ``y_`` the
    standardised phenotype and its permutations (``null_source``: a callable delivering ``y_``
    when it is first needed, so that everything that only needs ``y`` runs while the permutations
    are still being drawn).

    Ordering: the local-null kernel is started first and runs on the GPU while the leading
    eigenvectors of G are taken here (`_top_pcs`) and the global F-tests run on the second stream; the
    sign-defining LAPACK SVD of G (_nam.py:105) runs on a worker thread when the caller wants the full
    result (``full``) and otherwise only if a field that shows PC signs is read.  The values, warnings and
    progress text are those of the reference's sequential order."""
    out = select_output(show_progress)
    M, r = res.M, res.r
    n = len(y)
    if ks is None:
        ks = default_ks(n)
    if isinstance(M, pd.DataFrame) and M.values.dtype == np.float64 and np.isnan(M.values.sum()):
        # a projector with NaNs (a constant covariate: 0/0 when it is standardised, _nam.py:125): the reference has
        # already failed, in the SVD of the residualised NAM (_nam.py:105), before it looks at ks
        _small_svd(engine.gram_fetch())
    if max(ks) + r >= n:
        raise ValueError(
            'Maximum number of PCs plus number of covariates must be less than n-1. ' +
            f'Currently it is {max(ks)+r} while n is {n}. Either reduce the number of covariates ' +
            'or reduce the number of PCs to consider using the optional argument ks=[...].')
    ks_arr = np.asarray(ks)
    Mv = M.values if isinstance(M, pd.DataFrame) and M.values.dtype == np.float64 else np.asarray(M, dtype=np.float64)

    # neighbourhood coefficients -> thresholds (needs y only; already taken with the selection pass
    # when nothing had to be regressed out)
    if maxabs is None:
        _, maxabs = engine.ncorrs(y, fetch=False)
    if maxabs != maxabs:
        # NaN coefficients: the residualised NAM holds NaNs (e.g. a sample without cells that the reference's
        # positionally paired sample filter let through, fixtures f03 / f24) -- the reference has failed before it gets
        # here, in the SVD of the Gram matrix (_nam.py:105, "SVD did not converge"): the same routine on the same matrix
        _small_svd(engine.gram_fetch())
        # a finite Gram matrix after all: then it is the phenotype that is NaN (a constant y: 0/0 when it is standardised,
        # _association.py:22), and the reference's next stop is the argmin over its all-NaN p-values (_association.py:55)
        raise ValueError('All-NaN slice encountered')
    pending = False
    coef_early = False
    thresholds = edges = None
    if local_test:
        Nloc = min(1000, Nnull)
        maxcorr = max(maxabs, 0.001)
        thresholds = np.arange(maxcorr / 4, maxcorr, maxcorr / 400)
        z2 = thresholds ** 2
        edges = z2 - 1e-8 - 1e-5 * z2                        # tail_counts' bin edges (_stats.py:47)
        # the half of the local-null launch that needs the thresholds only (exact cuts, the threshold
        # counts of the observed coefficients) goes out now
        engine.null_local_prepare(Nloc, edges, thresholds)
        # the coefficient column does not depend on the null: queued in front of the null kernel it is
        # on the host while that kernel runs (consumed below, between the two halves of the F-tests)
        coef_early = (on_coef is not None and _EARLY_COEF and getattr(engine, 'percell_coef_launch', None) is not None
                      and getattr(engine, 'global_test_launch', None) is not None
                      and (coef_launched or engine.percell_coef_launch()))

    if null_source is not None:
        y_, conditioned = null_source()
    if y_ is None:   # grouplevel_permutation refused (it printed why); upstream dies on y_.T
        raise AttributeError("'NoneType' object has no attribute 'T'")
    # phenotypes -> device: Zc = M.[y, y_] / std (ddof=1), resident for both tests (already there
    # when the caller could issue it under the diffusion kernels)
    if not conditioned:
        engine.condition(Mv, np.column_stack([y, y_]))
    if local_test:
        # start the local null (device, asynchronous): tail counts from columns 1..Nloc of Zc, summed
        # over permutations on the device; neither cells x Nloc nor Nloc x T ever reaches the host.
        engine.null_local_launch(1, Nloc, None)                   # returns at once
        pending = True

    tail_sums = ranks = num_detected = None
    # Few cells and many samples (a rank's block of a sharded run): the local null is over before LAPACK is, and
    # what follows it on the host -- the FDR table, the per-cell columns -- does not need the eigenvectors.  Then LAPACK
    # runs on a thread of its own and this one takes the null's results and the per-cell pass meanwhile; what the caller
    # sees (values, warnings, progress text, which exception wins) keeps the reference's order.
    # (the size of a rank's block, the same number on every rank)
    block_cells = -(-int(getattr(engine, 'n_global', 0)) // max(1, int(getattr(engine, 'nranks', 1))))
    # (and LAPACK must be long enough to be worth it: 1.25 ms at 200 samples, 0.08 at 50, where the F-tests are better
    # off under the local null as before)
    tail_first = (local_test and coef_early and not coef_first and block_cells < _COEF_FIRST_CELLS and n >= _TAIL_FIRST_SAMPLES
                  and _TAIL_FIRST)
    early_tail = None
    try:
        # PCA of the NAM: LAPACK SVD of the Gram matrix (_nam.py:105), and the global F-tests of the
        # observed phenotype and every permutation (second stream), all under the local-null kernel
        # the coefficient column (already on its way to the host) goes into data.obs under a kernel: with
        # many cells under the Gram kernel, i.e. before this thread waits for it (2M cells: 0.7 ms of column
        # copy against 2.2 ms of Gram; afterwards the host would be the critical path, the integer local null
        # being as short as LAPACK + F-tests); with few cells between the two halves of the F-test call
        coef_first = coef_early and coef_first
        if coef_first:
            on_coef(engine.percell_coef_wait())
        kmax = int(ks_arr.max())
        can_launch = getattr(engine, 'global_test_launch', None) is not None
        # Gram matrix -> leading eigenpairs -> F-tests queued as ONE library call (engine.gram_pcs_tests: the library's
        # own checked eigen-solver, csrc/host_eig.c) when the caller does not want the sign-defining SVD started at once;
        # where the solver steps aside (degenerate leading spectrum, ks beyond a quarter of the samples) LAPACK follows
        one_call = (can_launch and not full and _nam_mod._EIG_NATIVE and getattr(engine, 'gram_pcs_tests', None) is not None)
        G = pcs = None
        if not one_call:
            G = engine.gram_fetch()
            _mark('gram fetched')
            pcs = GramPCs(G)
            if full:
                pcs.start()                                  # LAPACK's SVD beside the F-tests and the local null

        def pcs_then_tests():
            """-> (G, leading eigenvectors or None, F-tests queued)"""
            if one_call:
                G_, U_, queued = engine.gram_pcs_tests(ks_arr, r)
                _nam_mod.eig_stats['native' if queued else 'lapack'] += 1
                if queued:
                    return G_, U_, True
                U_ = _top_pcs(G_, kmax, native=False)
            else:
                G_ = G
                U_ = _top_pcs(G_, kmax)
            if U_ is None or not can_launch:
                return G_, U_, False
            engine.global_test_launch(U_, ks_arr, r)         # second stream
            return G_, U_, True

        if tail_first:
            # on a thread of their own: the F-tests then run under the local null instead of behind the per-cell pass
            # (second stream, their own buffers: nothing the null's tail or the per-cell pass touches)
            fut = _eig_pool().submit(pcs_then_tests)
            try:
                on_coef(engine.percell_coef_wait())
                pending = False                              # (the library clears its flag before it can fail)
                tail_sums, ranks, num_detected = engine.null_local_fetch()
                _mark('null fetched')
                try:
                    early_tail = ('ok', _fdr_tables(tail_sums, ranks, Nloc, thresholds))
                    early_tail += (engine.percell(thresholds, early_tail[1][3]),)
                    if on_fdr is not None:                   # the FDR column into the frame while the eigenvectors are on
                        on_fdr(early_tail[2][1])             # their way (put back if the test then fails)
                except Exception as exc:                     # raised where the sequential order meets it
                    early_tail = ('error', exc)
                _mark('percell done')
            finally:
                G, Uk, tests_queued = fut.result()
        else:
            G, Uk, tests_queued = pcs_then_tests()
        if pcs is None:
            _mark('gram fetched')
            pcs = GramPCs(G)
        if Uk is None:
            Uk = pcs.U[:, :kmax]
        _mark('pcs')
        if not can_launch:
            best, pv, r2v = engine.global_test(Uk, ks_arr, r)
        else:
            if not tests_queued:
                engine.global_test_launch(Uk, ks_arr, r)     # second stream
            try:
                if coef_early and not tail_first and not coef_first:
                    on_coef(engine.percell_coef_wait())
            finally:
                best, pv, r2v = engine.global_test_fetch()
    except BaseException:
        if pending:                     # never leave a pass pending behind an exception -- and never hide that exception
            try:
                engine.null_local_fetch()
            except Exception:           # noqa: BLE001
                pass
        if getattr(engine, 'global_test_discard', None) is not None:
            try:
                engine.global_test_discard()  # (the F-tests may have been queued already, by this thread or the eigenvector thread)
            except Exception:                 # noqa: BLE001 - an engine with other error types: the caller gets the error that brought us here
                pass
        raise
    if pending:
        tail_sums, ranks, num_detected = engine.null_local_fetch()

    if not tail_first:
        _mark('null fetched')
    if (best < 0).any():
        raise ValueError('All-NaN slice encountered')        # np.nanargmin in _minp_stats
    k, p, r2 = ks[best[0]], pv[0], r2v[0]
    if k == max(ks):
        warnings.warn(('data supported use of {} NAM PCs, which is the maximum considered. ' +
                       'Consider allowing more PCs by using the "ks" argument.').format(k))

    # coefficients and r2 of the chosen model
    ycond_v = Mv.dot(y)
    ycond_v = ycond_v / ycond_v.std(ddof=1)
    beta_k = Uk[:, :k].T.dot(ycond_v)                   # up to the sign of every PC
    yhat = Uk[:, :k].dot(beta_k)                         # sign free
    r2_perpc = (beta_k / np.sqrt(ycond_v.dot(ycond_v))) ** 2

    # sample-space frames of the result are only built when somebody reads them; what shows the PC signs
    # (U, beta) comes from LAPACK's SVD of G like upstream's (_nam.py:105, _association.py:70-72)
    nU = len(G)

    def _names():
        return ['PC' + str(i) for i in range(1, nU + 1)]
    res._defer('namresid_sampleXpc', lambda: pd.DataFrame(pcs.U, index=M.index, columns=_names()))
    res._defer('namresid_svs', lambda: pd.Series(pcs.svs, index=_names())[:npcs if npcs is not None else nU])
    res._defer('namresid_varexp', lambda: pd.Series(pcs.svs, index=_names()) / nU / (n_cells() if callable(n_cells) else n_cells))
    res._defer('yresid', lambda: pd.Series(ycond_v, index=getattr(M, 'index', None)))
    res._defer('beta', lambda: pcs.U[:, :k].T.dot(ycond_v))

    nullminps, nullr2s = pv[1:], r2v[1:]
    hits = (nullminps <= p + 1e-8).sum()
    pfinal = (hits + 1) / (Nnull + 1)
    if hits == 0:
        warnings.warn('global association p-value attained minimal possible value. ' +
                      'Consider increasing Nnull')

    fdr_vals = None
    fdr_5p_t = fdr_10p_t = None
    if local_test:
        print('computing neighborhood-level FDRs', file=out)
        if early_tail is not None and early_tail[0] == 'error':
            raise early_tail[1]
        fdr_vals, fdr_5p_t, fdr_10p_t, runmin = early_tail[1] if early_tail is not None else \
            _fdr_tables(tail_sums, ranks, Nloc, thresholds)
        res._defer('fdrs', lambda: pd.DataFrame({'threshold': thresholds, 'fdr': fdr_vals,
                                                 'num_detected': num_detected}))
    else:
        res.fdrs = None

    # data.obs columns (all cells) and, from them, the coefficients of the kept cells
    if early_tail is not None:
        coef_all, fdr_all = early_tail[2]
    elif fdr_vals is not None:
        _mark('pre percell')
        coef_all, fdr_all = engine.percell(thresholds, runmin)
    else:
        coef_all, fdr_all = engine.percell(None, None)

    if early_tail is None:
        _mark('percell done')
    res.__dict__.update({'p': pfinal, 'nullminps': nullminps, 'k': k, 'fdr_5p_t': fdr_5p_t,
                         'fdr_10p_t': fdr_10p_t, 'yresid_hat': yhat, 'ks': ks,
                         'r2': r2, 'r2_perpc': r2_perpc, 'nullr2_mean': nullr2s.mean(),
                         'nullr2_std': nullr2s.std()})
    return coef_all, fdr_all, pcs


def check_types(y, batches, covs, donorids):
    """The type checks that open the reference's check_inputs (_association.py:132-139)."""
    for name, val, kind, label in (('y', y, pd.Series, 'Series'), ('batches', batches, pd.Series, 'Series'),
                                   ('covs', covs, pd.DataFrame, 'DataFrame'),
                                   ('donorids', donorids, pd.Series, 'Series')):
        if (name == 'y' or val is not None) and not isinstance(val, kind):
            raise TypeError(f"'{name}' must be a pandas {label}, but got {type(val)}")


def check_inputs(data, y, sid_name, batches, covs, donorids, allow_low_sample_size, sids_present=None):
    """Input validation and the sample filter (reference _association.py:131-173): same
    exception types, messages and printed warnings.  ``sids_present``: the distinct sample ids
    of the cells if the caller already has them (one hash pass over the cells instead of three)."""
    check_types(y, batches, covs, donorids)
    if sids_present is None:
        sids_present = pd.unique(data.obs[sid_name])
    if isinstance(sids_present, pd.Index) and len(sids_present) == len(y.index) and sids_present.equals(y.index):
        # the usual case -- y is indexed by exactly the samples of data, in canonical order -- needs
        # neither of the two hash joins below (this runs on the critical path of small problems)
        present = np.ones(len(y), dtype=bool)
    else:
        present = y.index.isin(sids_present)
        if not present.all():
            print("WARNING: index of 'y' contains values not present in 'data[sid_name]'. These samples will be ignored.")
        if not pd.Index(sids_present).isin(y.index).all():
            raise ValueError("'data[sid_name]' contains values not present in the index of 'y'.")
    if batches is not None and donorids is not None:
        raise ValueError('We do not currently support conditioning on batch ' +
                         'while also accounting for multiple samples per donor')
    if batches is None:
        batches = pd.Series(np.ones(len(y)), index=y.index)

    if covs is not None:
        filter_samples = ~(y.isna() | covs.isna().any(axis=1)) & present
        if donorids is not None:
            print('WARNING: We currently do not account for multiple samples per donor ' +
                  'when conditioning on covariates. This conditioning may therefore account ' +
                  'only incompletely for the covariates of interest. We expect this to make ' +
                  'only minor differences in most cases, but we have not investigated it formally')
    elif y.dtype.kind == 'f':
        filter_samples = pd.Series(~np.isnan(y.values) & present, index=y.index, name=y.name)
    else:
        filter_samples = ~np.isnan(y) & present

    fvals = getattr(filter_samples, 'values', None)
    N = int(np.count_nonzero(fvals)) if fvals is not None and fvals.dtype == bool else filter_samples.sum()
    if N < 10 and not allow_low_sample_size:
        raise ValueError(
            'You are supplying phenotype information on fewer than 10 samples. This may lead to ' +
            'poor power at low sample sizes because our null distribution is one in which each ' +
            'sample\'s single-cell profile is unchanged but the sample labels are randomly ' +
            'assigned. If you want to run an analysis at this sample size despite the possibility of low ' +
            'power, you can do so by invoking the association(...) function with the argument ' +
            'allow_low_sample_size=True.')
    return batches, filter_samples


def compute_nam_and_reindex(engine, data, y, sid_name, batches, covs, donorids, filter_samples, nsteps,
                            show_progress, codes_labels=None, overlap=None, nam_queued=None, y_std=None,
                            fuse_null=0, null_ready=None, local_test=True, ks=None, **kwargs):
    """Reference compute_nam_and_reindex (_association.py:175-191) on the device: build the NAM,
    QC it, put the sample axis in ``y.index`` order restricted to ``filter_samples``, drop the
    cells whose remaining entries have zero variance.  Leaves the selected NAM in the engine's
    working matrix and returns the bookkeeping the caller needs.  ``overlap``: a host-only
    callable run after the diffusion kernels have been queued and before their first result is
    needed; its return value is passed through.  ``nam_queued``: result of a _nam_device() call the
    caller has already issued for exactly these arguments (its kernels are in flight).  ``y_std``:
    the standardised phenotype of the filtered samples; when nothing has to be regressed out the
    neighbourhood coefficients are then taken in the selection pass (plan.maxabs)."""
    out = select_output(show_progress)
    nam_kwargs = {k: v for k, v in kwargs.items() if k in ('self_weight',)}
    print('computing NAM', file=out)
    finish_walk = None
    if nam_queued is not None:
        labels = nam_queued[0]
        finish_walk = nam_queued[2] if len(nam_queued) > 2 else None        # the last step is still to be queued
    else:
        labels, _ = _nam_device(engine, data, sid_name, nsteps=nsteps, show_progress=show_progress,
                                codes_labels=codes_labels, **nam_kwargs)
    batches_qc = batches
    # NAM.reindex(y.index)[filter_samples]: boolean-Series indexing aligns on the index
    positions = pd.Series(np.arange(len(y)), index=y.index)[filter_samples].values
    sample_index = y.index[positions]
    colmap = labels.get_indexer(sample_index)
    absent_selected = bool((colmap < 0).any())            # (dealt with after the QC, where the reference meets it)
    batches = batches.reindex(y.index)
    covs = covs.reindex(y.index) if covs is not None else None
    donorids = donorids.reindex(y.index) if donorids is not None else None
    filter_samples = filter_samples.reindex(y.index)
    sample_index = pd.Index(sample_index, name=sid_name)
    # the host-only planning (projector, one-hot batches: pandas work) first: it needs nothing from the device and
    # runs while the walk does; the QC below waits for the walk (its batch-kurtosis pass and one read-back)
    extra = overlap(sample_index, batches, covs, donorids, filter_samples) if overlap is not None else None
    plan = extra if hasattr(extra, 'kind') else None
    from ._nam import _lowrank_ok
    single = (plan is not None and plan.kind == 'single' and hasattr(engine, 'set_resid_factors') and _lowrank_ok(engine, plan)
              )
    if finish_walk is not None:
        # The walk's last step is queued here, with the selection pass's arguments when they are "every cell, the
        # samples in place, nothing to regress out" -- what the call below then asks for: that step leaves X, its
        # coefficients and the zero-variance count itself (diffuse.hip:select_tail).  (With covariates the projector
        # would ride along in factored form: measured at 2M x 200 with five of them, 10.4 ms against 7.7 + 2.4 for the
        # two kernels -- the factors through LDS, the projections' wave sums overlapped -- so that case keeps its pass.)
        hint = None
        if (plan is not None and plan.kind == 'identity' and y_std is not None and len(y_std) == len(colmap)
                and len(colmap) == len(labels) and np.array_equal(colmap, np.arange(len(colmap)))
                and len(np.unique(batches_qc)) == 1):
            engine.clear_resid_factors()
            hint = y_std
        finish_walk(hint)
    # Covariates AND batches (the demo's call, demo/demo.ipynb:149) with every sample of the data analysed in place and at
    # most seven batches: QC, selection and the first ridge -- three passes over the cells -- as ONE
    # (engine.select_resid_bk).  It answers what the three would have answered; anything but "no cell fails the QC, no
    # cell of zero variance, the schedule ends at the first ridge" (always so on real data with so few batches) and
    # the three passes run as before, from the untouched NAM.
    onepass = None
    if (plan is not None and plan.kind == 'ridge' and not show_progress and getattr(engine, 'select_resid_bk', None) is not None
            and y_std is not None and len(y_std) == len(colmap) == len(labels) and not absent_selected
            and np.array_equal(colmap, np.arange(len(colmap))) and getattr(plan, 'first_ridge', None) is not None
            and len(plan.ridges) and 2 <= plan.nb <= 7 and _lowrank_ok(engine, plan)
            and os.environ.get('CNA_RIDGE_ONEPASS', '1') not in ('0', 'off', 'no')):
        from ._nam import _batch_codes
        qc_codes, qc_nb = _batch_codes(batches_qc, labels)
        held = getattr(engine, '_sample_counts', None)
        if (qc_nb == plan.nb and (qc_codes >= 0).all() and np.array_equal(qc_codes, plan.bcodes)
                and held is not None and held[1] is labels and (np.asarray(held[0]) > 0).all()):
            got = engine.select_resid_bk(np.asarray(plan.C.values, dtype=np.float64), np.asarray(plan.first_ridge[0], dtype=np.float64),
                                         y_std, plan.bcodes, plan.nb)
            if got is not None and got[0] == 0 and got[1] == 0 and got[3] <= 6:
                onepass = got
    if onepass is not None:
        plan.onepass = onepass                  # (_resid_run: X is final, coefficients and their maximum are there)
        plan.maxabs = onepass[2]
        kept = np.repeat(True, engine.n)
        return (kept, sample_index, colmap, batches, covs, donorids, filter_samples, extra)
    kept = _qc_device(engine, labels, batches_qc, show_progress=show_progress)
    if not kept.any():
        # Every neighbourhood failed the QC (a NaN among the batch labels is enough: it is a level without members, its
        # mean and with it every batch kurtosis NaN, _nam.py:78-99).  The reference goes on with a NAM of no columns
        # and stops where the thresholds are formed from the largest of no coefficients (_association.py:99-102;
        # fixtures f29 / f30): the same error from here, before anything is selected or written.
        # (... after the check of ks against the sample count, which comes first there: _association.py:29-33)
        n_f = int(filter_samples.sum())
        f_ = filter_samples.values.astype(bool)
        r_f = _resid_plan(sample_index, covs[f_] if covs is not None else None, batches[f_] if batches is not None else None).r
        ks_f = ks if ks is not None else default_ks(n_f)
        if max(ks_f) + r_f >= n_f:
            raise ValueError(
                'Maximum number of PCs plus number of covariates must be less than n-1. ' +
                f'Currently it is {max(ks_f)+r_f} while n is {n_f}. Either reduce the number of covariates ' +
                'or reduce the number of PCs to consider using the optional argument ks=[...].')
        # ... and after the global F-test of the observed phenotype (_association.py:64), which the reference runs on the
        # PCs of the empty NAM before it looks at the thresholds: p-values that are all NaN stop it in np.nanargmin
        # (_association.py:60) -- a projector with NaNs (a constant covariate: 0/0 when it is standardised, _nam.py:125), no
        # degrees of freedom left for any k (scipy's F survival function of a non-positive dfd), a constant phenotype
        covs_f = covs[f_] if covs is not None else None
        with np.errstate(all='ignore'):
            nan_projector = covs_f is not None and len(covs_f.T) and bool(
                np.isnan(np.asarray((covs_f - covs_f.mean(axis=0)) / covs_f.std(axis=0), dtype=np.float64)).any())
        no_dof = all(n_f - (1 + r_f + int(k_)) <= 0 for k_ in ks_f)
        nan_y = y_std is not None and len(y_std) == n_f and bool(np.isnan(np.asarray(y_std, dtype=np.float64)).any())
        if nan_projector or no_dof or nan_y:
            raise ValueError('All-NaN slice encountered')
        if local_test:
            raise ValueError('arange: cannot compute length')
        # (without the local test the reference gets as far as its epilogue, which reads the FDR table that was never made:
        # _association.py:233-236, as in every call with local_test=False)
        raise AttributeError("'NoneType' object has no attribute 'loc'")
    if absent_selected:
        # The filter selects a sample the data has no cells of -- the reference's filter pairs `y.isna() | covs.isna()`
        # (indexed by the sorted union of the two indices) with `y.index.isin(...)` (in y's order) BY POSITION
        # (_association.py:153-160), so inputs in different orders can let such a sample through.  Its row of
        # NAM.reindex(y.index) is NaN, the residualised NAM is NaN throughout, and the reference stops in the SVD of the
        # Gram matrix (_nam.py:105) with numpy's message -- the same error here (cause: y, covs, batches and donorids that
        # do not share one index order).
        raise np.linalg.LinAlgError('SVD did not converge')

    nzero = -1
    if (plan is not None and plan.kind == 'identity' and finish_walk is None and y_std is not None
            and len(y_std) == len(colmap) == len(labels) and np.array_equal(colmap, np.arange(len(colmap))) and kept.all()
            and getattr(engine, 'reuse_nam', False) and getattr(engine, 'reuse_x', False)
            and hasattr(engine, 'x_identity_resident') and engine.x_identity_resident()):
        # A further phenotype on the resident dataset (the walk was skipped: NAM cache): the standardised NAM of this
        # very selection is still on the device from the previous analysis -- it does not depend on the phenotype -- so
        # only the coefficients are new.  Opt-in (engine.reuse_x): the coefficients then come from
        # another kernel than in a from-scratch call (same X, another order of the row sums: last-bit differences), and
        # what a call returns should not depend on which calls came before it unless the caller says so.
        engine._fused = None
        _, maxabs = engine.ncorrs(y_std)
        nzero = 0
        plan.maxabs = maxabs
        plan.standardized = True
    elif plan is not None and (plan.kind == 'identity' or single):
        # nothing to regress out, or a projector that the selection pass applies in factored form
        # (covariates without batches): select + centre [+ M] + /std in one pass over the NAM
        if single:
            engine.set_resid_factors(np.asarray(plan.C.values, dtype=np.float64), plan.W)
        if y_std is not None and len(y_std) == len(colmap):
            nzero, maxabs = engine.select_standardized(None if kept.all() else kept, colmap, y=y_std,
                                                       fuse_null=fuse_null, **({'null_ready': null_ready} if null_ready is not None else {}))
            plan.maxabs = maxabs if nzero == 0 else None
        else:
            nzero = engine.select_standardized(None if kept.all() else kept, colmap)
        plan.standardized = nzero == 0
    if nzero != 0 and hasattr(engine, 'select_checked'):
        # something is regressed out: plain selection, with the zero-variance count taken in the same pass
        nzero = engine.select_checked(None if kept.all() else kept, colmap)
        if nzero == 0 and plan is not None:
            # (the one-pass first ridge of _resid_run overwrites X optimistically; this puts it back)
            plan.reselect = lambda: engine.select_checked(None if kept.all() else kept, colmap)
    if nzero != 0:
        zero_var, nzero = engine.zero_variance(colmap)
        if nzero:
            kept = kept & ~zero_var
        engine.select(None if kept.all() else kept, colmap)
    return (kept, sample_index, colmap, batches, covs, donorids, filter_samples, extra)


def association(data, y, sid_name, batches=None, covs=None, donorids=None, ks=None, key_added='coef',
                max_frac_pcs=0.15, nsteps=None, show_progress=False, allow_low_sample_size=False,
                return_full=False, ridges=None, engine=None, **kwargs):
    """cna.tl.association (reference _association.py:193-242).

    Returns the global p-value, or with ``return_full=True`` the full result namespace
    (same field names and types as upstream; the three cells x samples sized frames --
    ``nam``, ``namresid``, ``namresid_nbhdXpc`` -- are copied off the GPU when first read).
    Writes ``data.obs[key_added]`` and ``data.obs[key_added + '_fdr']``.

    Limits the reference does not have (the library answers CNA_EINVAL beyond them): at most 1024 samples, 256 batches
    and 512 FDR thresholds (the reference's own call uses 300 or 301), fewer than 2**31 cells.  One result field may
    differ from the reference's in SHAPE by one row: ``res.fdrs`` ("Result fields whose shape may
    differ by one row")."""
    with host_blas_threads(1):
        eng = engine or get_engine()
        # the fixed-shape call (nsteps given, one batch, a seed, ...) in two library calls (tools/_fast.py); everything
        # else -- and whatever that path meets and does not handle -- goes on below as if it did not exist
        out = _fast.association(data, y, sid_name, batches, covs, donorids, ks, key_added, max_frac_pcs, nsteps, show_progress,
                                allow_low_sample_size, return_full, ridges, eng, kwargs)
        if out is not _fast.NOT_TAKEN:
            return out
        # The resident graph is validated by a hash of its full content (engine.ensure_graph).  On one GPU
        # that hash runs on a helper thread while the kernels are already working on the resident copy
        # (optimistic); before anything leaves this call -- the first data.obs write, the return value --
        # the outcome is collected, and if the matrix was edited in place the call starts over on a fresh
        # upload, with numpy's global RNG put back where it was.
        rng_state = np.random.get_state() if kwargs.get('seed') is None else None
        _switch_interval_enter()                          # helper thread and this one hand the GIL over promptly
        try:
            return _association_attempts(eng, rng_state, data, y, sid_name, batches, covs, donorids, ks, key_added,
                                         max_frac_pcs, nsteps, show_progress, allow_low_sample_size, return_full, ridges,
                                         kwargs)
        finally:
            _switch_interval_exit()


def _association_attempts(eng, rng_state, data, y, sid_name, batches, covs, donorids, ks, key_added, max_frac_pcs, nsteps,
                          show_progress, allow_low_sample_size, return_full, ridges, kwargs):
    for _once in (0,):
        for attempt in (0, 1):
            eng._defer_graph_check = attempt == 0 and hasattr(eng, 'confirm_graph')
            try:
                return _association_call(data, y, sid_name, batches, covs, donorids, ks, key_added, max_frac_pcs, nsteps,
                                         show_progress, allow_low_sample_size, return_full, ridges, eng, **kwargs)
            except _StaleGraph:
                if rng_state is not None:
                    np.random.set_state(rng_state)
            except Exception:
                # an error of the optimistic attempt counts only if its inputs were what the memos said
                stale = not confirm_codes()
                stale = (hasattr(eng, 'confirm_graph') and not eng.confirm_graph()) or stale
                if attempt == 1 or not stale:
                    raise
                if rng_state is not None:
                    np.random.set_state(rng_state)
            finally:
                eng._defer_graph_check = False
        raise RuntimeError('the connectivities matrix keeps changing while it is being analysed')


class _StaleGraph(Exception):
    """The deferred content check found the resident graph out of date (engine.confirm_graph)."""


def _prefetch_graph(engine, data):
    try:
        from ._nam import _prepare_graph
        _prepare_graph(engine, data, 1)
    except Exception:                          # noqa: BLE001 - the regular call meets the same problem and reports it
        pass


def _association_call(data, y, sid_name, batches, covs, donorids, ks, key_added, max_frac_pcs, nsteps,
                      show_progress, allow_low_sample_size, return_full, ridges, engine, **kwargs):
    out = select_output(show_progress)
    engine = engine or get_engine()
    extra = set(kwargs) - {'Nnull', 'force_permute_all', 'local_test', 'seed', 'self_weight'}
    if extra or 'self_weight' in kwargs:
        # upstream forwards **kwargs to _association(), which rejects anything else
        bad = sorted(extra | ({'self_weight'} & set(kwargs)))[0]
        raise TypeError(f"_association() got an unexpected keyword argument '{bad}'")
    Nnull = kwargs.get('Nnull', 1000)

    # factorise the per-cell sample ids once; validation and NAM construction share the result
    _mark('enter')
    sharded = shard_of(data) is not None
    # A graph that is new to the device: its upload (PCIe) and column sums start on a helper thread now, beside the
    # factorisation of the sample ids on this one (2M cells: 26 + 14 ms beside 11 ms) -- _nam_device below then finds
    # the graph resident.  Whatever goes wrong there is left for that call to find and report where it always did.
    prefetch = None
    if _PREFETCH_GRAPH and not sharded and not show_progress and getattr(engine, 'nranks', 1) == 1 and hasattr(engine, 'graph_resident'):
        try:
            A0 = get_connectivity(data)
            if (sp.isspmatrix_csr(A0) or isinstance(A0, getattr(sp, 'csr_array', ()))) and A0.shape[0] >= _PREFETCH_CELLS \
                    and not engine.graph_resident(A0):
                prefetch = _background().submit(_prefetch_graph, engine, data)
        except Exception:                      # noqa: BLE001
            prefetch = None
    # one GPU: the content hash of the ids is checked on a helper thread, like the graph's (confirm_graph below)
    defer_ids = (getattr(engine, '_defer_graph_check', False) and not sharded and getattr(engine, 'nranks', 1) == 1
                 and not getattr(engine, '_has_comm', False))
    try:
        codes, labels, counts, token = sample_codes_cached(data.obs[sid_name], defer=defer_ids)
    finally:
        if prefetch is not None:               # (whatever happened here: nobody else touches the engine while the helper does)
            prefetch.result()
            _mark('graph prefetched')
    if sharded:      # this rank's cells only: agree with the other ranks on the samples and their sizes
        codes, labels, counts, token = global_samples(engine, codes, labels, counts, token)
    _mark('codes')
    # The walk needs the graph and the sample ids only: it is queued before the (pandas) validation of the
    # sample-level inputs, which then runs under it.  An error of the walk is held back until validation has
    # had its say -- the order in which the reference would have raised; with progress output the reference's
    # order of lines is kept instead (validation messages, then 'computing NAM').
    import threading
    walk_queued = threading.Event()
    nam_queued = nam_error = None
    # (the checks that need no look at the data come first: a call with an argument of the wrong type neither starts a
    # walk nor replaces the NAM an earlier result still reads lazily)
    check_types(y, batches, covs, donorids)
    if not show_progress and _EARLY_WALK:
        engine._on_walk_queued = walk_queued.set
        try:
            nam_queued = _nam_device(engine, data, sid_name, nsteps=nsteps, show_progress=False,
                                     codes_labels=(codes, labels, counts, token),
                                     defer_last=(nsteps is not None and nsteps >= 3 and kwargs.get('local_test', True)
                                                 and _rule_cells(data, engine) >= _DEFER_LAST_CELLS))
        except Exception as exc:             # noqa: BLE001 - re-raised below, after validation
            nam_error = exc
        finally:
            engine._on_walk_queued = None
            walk_queued.set()
    used = counts > 0
    batches, filter_samples = check_inputs(data, y, sid_name, batches, covs, donorids, allow_low_sample_size,
                                           sids_present=labels[used] if isinstance(y, pd.Series) or sharded else None)
    if nam_error is not None:
        raise nam_error

    # the permutation draw (numpy RNG + argsort, both outside the GIL) needs only sample-level
    # inputs: it starts on the helper thread right away and is collected just before the
    # phenotypes go to the device.  Nothing else touches numpy's global RNG in between.
    # y[f] with f = filter.reindex(y.index), etc.; when the Series already share y's index object
    # (check_inputs built them on it) that is plain numpy masking
    if filter_samples.index is y.index and batches.index is y.index and donorids is None:
        fv = filter_samples.values
        yv, bv, dv = y.values[fv], batches.values[fv], None
    else:
        f = filter_samples.reindex(y.index)
        b_ = batches.reindex(y.index)
        d_ = donorids.reindex(y.index) if donorids is not None else None
        yv, bv, dv = y[f].values, b_[f].values, (d_[f].values if d_ is not None else None)

    # the standardised phenotype (numpy ddof=0, _association.py:22) needs neither the device nor the draw
    with np.errstate(all='ignore'):
        y_std = (yv - yv.mean()) / yv.std() if len(yv) else yv

    early = {}

    # many cells: the walk is long enough to hide the draw wherever it starts, and a helper thread that starts
    # at once takes the interpreter from this thread's launches (0.2-0.3 ms later first kernel at 2M cells);
    # few cells: the draw is on the critical path and starts at once (holding it back: 2.46 -> 3.0 ms at 200k)
    hold_draw = _DRAW_THREAD and len(data.obs) >= _COEF_FIRST_CELLS

    # The draw itself -- numpy's legacy normal stream, an argsort per permutation (_stats.py:4-18) -- runs on the
    # library's own host thread from here on when the call has the common shape (a seed, no donor groups, an even
    # number of permutations): it needs no interpreter, so it neither waits for this thread's Python nor slows it
    # (a Python helper thread got to it 0.35 ms late and took 0.54 ms for 0.39 ms of work at 200k cells x 50 samples)
    native = None
    # (few cells only: from 500k cells on the walk hides the draw wherever it runs, and a draw that starts while this
    # thread is still queueing the walk costs it CPU time -- measured 6.62 -> 6.73 ms at 1M x 100, 20.1 -> 20.7 at
    # 2M x 200 with 10 000 permutations, against 1.94 -> 1.51 ms at 200k x 50)
    if _NATIVE_DRAW and dv is None and len(yv) and kwargs.get('seed') is not None and len(data.obs) < _COEF_FIRST_CELLS:
        try:
            native = native_draw_start(np.ones(len(yv)) if kwargs.get('force_permute_all', False) else bv, y_std, Nnull,
                                       kwargs.get('seed'))
        except Exception:                     # noqa: BLE001 - the usual draw below reports whatever is wrong
            native = None
        if native is not None:
            _mark('native draw started')

    def null_job():
        if native is not None:
            table = native.wait()
            _mark('draw done')
            out_ = (table[:, 0], table[:, 1:])
            M_ = early.get('M')
            if native.conditioned:
                engine._zc_cols = table.shape[1]          # (what engine.condition notes)
                early['conditioned'] = True
                _mark('conditioned (library thread)')
            elif M_ is not None and len(table) == len(M_):
                try:
                    engine.condition(M_, table)
                    early['conditioned'] = True
                    _mark('conditioned (helper)')
                except Exception as exc:      # the main thread conditions again and reports what it finds
                    early['condition_error'] = exc
            return out_
        if hold_draw:
            walk_queued.wait(0.004)
        _mark('draw starts')
        out_ = _draw_null(yv, bv, dv,
                          Nnull=Nnull, force_permute_all=kwargs.get('force_permute_all', False),
                          seed=kwargs.get('seed'))
        _mark('draw done')
        # If the projector is already known (no ridge schedule; the main thread builds it under the
        # diffusion kernels), condition the phenotypes from here: sample space only, second stream,
        # its own buffers -- the main thread is busy with the selection meanwhile.  Anything unusual
        # is left to the main thread, which does it (and reports errors) where the reference would.
        M_ = early.get('M')
        if M_ is not None and out_[1] is not None and len(out_[0]) == len(M_):
            try:
                engine.condition(M_, np.column_stack([out_[0], out_[1]]))
                early['conditioned'] = True
                _mark('conditioned (helper)')
            except Exception as exc:          # the main thread conditions again and reports what it finds
                early['condition_error'] = exc
        return out_
    _mark('checked')
    # Where the draw runs.  Default: the helper thread, from now on, overlapping this thread's own host work
    # (it shares the GIL, so the 0.4 ms draw takes ~0.9 ms in context -- still the better schedule at 200k
    # cells: 1.96-1.99 ms per call against 2.2-2.4 ms with the draw on this thread after the walk is queued,
    # _DRAW_THREAD = False; no difference at 1M / 2M cells, where the walk hides either).
    null_future = _background().submit(null_job) if _DRAW_THREAD else _InlineJob(null_job)
    _mark('submitted')

    def host_side(sample_index_, batches_, covs_, donorids_, filter_):
        # host-only sample-space work: runs while the diffusion kernels are executing
        plan = _resid_plan(sample_index_, covs_[filter_] if covs_ is not None else covs_,
                           batches_[filter_] if batches_ is not None else batches_, ridges=ridges)
        if len(y_std) == len(sample_index_):
            plan.y_std = y_std                        # lets the residualisation pass take the coefficients on its way out
            plan.coef_first = _EARLY_COEF and len(data.obs) >= _COEF_FIRST_CELLS and kwargs.get('local_test', True)
        if plan.M is not None:
            early['M'] = np.asarray(plan.M, dtype=np.float64)     # the draw conditions with it
            if native is not None and hasattr(engine, 'h') and len(early['M']) == len(native.table):
                # ... on the library's own thread, the moment it has the permutations
                try:
                    early['native_cond'] = native.then_condition(engine, early['M'])
                except Exception:             # noqa: BLE001 - the helper below conditions instead
                    early['native_cond'] = False
        if not _DRAW_THREAD:
            null_future.run()
        return plan

    engine._on_walk_queued = walk_queued.set
    try:
        kept, sample_index, colmap, batches, covs, donorids, filter_samples, plan = \
            compute_nam_and_reindex(engine, data, y, sid_name, batches, covs, donorids, filter_samples, nsteps,
                                    show_progress, codes_labels=(codes, labels, counts, token), overlap=host_side,
                                    nam_queued=nam_queued, y_std=y_std, local_test=kwargs.get('local_test', True), ks=ks,
                                    fuse_null=min(1000, Nnull) if kwargs.get('local_test', True) and _FUSE else 0,
                                    # (the conditioned phenotypes of THIS call on the device before the selection is
                                    # asked for: that call then launches the local null itself)
                                    null_ready=((native.flag if native is not None else (lambda: bool(early.get('conditioned'))))
                                                if (hasattr(engine, 'h') and getattr(engine, 'nranks', 1) == 1
                                                    and not getattr(engine, '_has_comm', False)) else None))
    except BaseException:
        walk_queued.set()
        null_future.cancel() or null_future.exception()     # do not leave the helper thread running
        if native is not None:
            native.abandon()                                # ... nor the library's draw uncollected (the reference fails before
        raise                                               #     it seeds, _association.py:15-16: numpy's generator stays as it was)
    finally:
        engine._on_walk_queued = None
        walk_queued.set()

    def cell_index():
        # names of the kept cells: only needed for the frames of a full result
        return data.obs.index if kept.all() else data.obs.index[kept]
    nam_epoch = engine.nam_epoch

    N = filter_samples.sum()
    npcs = min(N, max([10] + [int(max_frac_pcs * N)] + [ks if ks is not None else []][0]))
    try:
        res = _resid_run(engine, plan, cell_index, show_progress=show_progress)
    except BaseException:
        null_future.cancel() or null_future.exception()     # do not leave the helper thread running
        if native is not None:
            native.abandon()                                # ... nor the library's draw uncollected (the reference fails before
        raise                                               #     it seeds, _association.py:15-16: numpy's generator stays as it was)

    _mark('resid queued')
    print('performing association test', file=out)
    # the permutations are collected inside _association, after everything that only needs y
    def drawn():
        y_null = null_future.result()[1]
        _mark('null drawn')
        return y_null, early.get('conditioned', False)
    # data.obs[key_added] is written early, while the local-null kernel runs (between the two halves
    # of the F-test call); should the test still fail, the column is put back as it was, so that --
    # like upstream -- an exception leaves data.obs alone
    early_coef = {}
    had_key = key_added in data.obs
    previous = data.obs[key_added] if had_key else None

    def confirm_graph():
        ok = confirm_codes()
        if (hasattr(engine, 'confirm_graph') and not engine.confirm_graph()) or not ok:
            raise _StaleGraph()

    fdr_key = f'{key_added}_fdr'
    had_fdr = fdr_key in data.obs
    previous_fdr = data.obs[fdr_key] if had_fdr else None
    big = len(data.obs) >= _COEF_FIRST_CELLS

    def fdr_copied_early():
        job = early_coef.pop('fdr_job', None)
        if job is None:
            return False
        try:
            return bool(job.result()) and engine.percell_fdr_copied_early()
        except Exception:
            return False

    def roll_back():
        # put data.obs back as it was: nothing written by this call -- early or late -- outlives an exception
        # (or a stale attempt); like upstream, whose two assignments are the last statements that can fail
        fdr_copied_early()                                # the helper is done with the column's storage
        if early_coef.pop('written', False):
            if had_key:
                data.obs[key_added] = previous
            elif key_added in data.obs:
                del data.obs[key_added]
            if 'fdr_view' in early_coef or early_coef.get('fdr_touched') or (big and fdr_key in data.obs and not had_fdr):
                if had_fdr:
                    data.obs[fdr_key] = previous_fdr
                elif fdr_key in data.obs:
                    del data.obs[fdr_key]
            early_coef.pop('fdr_view', None)
            early_coef.pop('fdr_touched', None)
            early_coef.pop('fdr_done', None)
            early_coef.pop('values', None)

    def write_coef_early(coef):
        confirm_graph()                                   # nothing reaches data.obs from a stale graph
        early_coef['written'] = True
        data.obs[key_added] = coef
        early_coef['values'] = data.obs[key_added].values
        if big:
            # the FDR column's storage as well (the frame takes a private copy of whatever it is given: 16 MB at 2M
            # cells): made now, under a kernel, and filled in place at the end by a threaded copy
            data.obs[fdr_key] = np.empty(len(data.obs))
            view = data.obs[fdr_key].values
            if view.dtype == np.float64 and view.flags.c_contiguous and view.flags.writeable:
                early_coef['fdr_view'] = view
                if _EARLY_FDR and hasattr(engine, 'percell_fdr_copy_early'):
                    # the column follows the local null on the device and the host still has the SVD and the
                    # F-tests in front of it: the helper thread waits for the column and fills the storage
                    from .._order import usable_cpus
                    early_coef['fdr_job'] = _background().submit(engine.percell_fdr_copy_early, view,
                                                                  min(8, usable_cpus(8)))
        _mark('coef column written')

    def write_fdr_early(fdr):
        # (small-block schedule: the per-cell pass is done before the eigenvectors are; same rules as the coefficient column)
        if 'values' not in early_coef or early_coef.get('fdr_view') is not None or fdr is None:
            return
        confirm_graph()
        early_coef['written'] = early_coef['fdr_touched'] = True
        data.obs[fdr_key] = fdr
        early_coef['fdr_done'] = True

    try:
        coef_all, fdr_all, pcs = _association(engine, res, y_std, None, ks=ks, Nnull=Nnull, full=return_full,
                                                 on_fdr=write_fdr_early,
                                                 local_test=kwargs.get('local_test', True),
                                                 show_progress=show_progress, npcs=npcs, n_cells=engine.x_rows_global,
                                                 null_source=drawn,
                                                 maxabs=getattr(plan, 'maxabs', None), on_coef=write_coef_early,
                                                 coef_first=len(data.obs) >= _COEF_FIRST_CELLS,
                                                 coef_launched=getattr(plan, 'coef_launched', False))
        _mark('_association returned')
        confirm_graph()                               # (stale: the retry starts from the frame as the caller left it)
        _defer_pcs(res, engine, pcs, cell_index)
        res.kept = kept

        def fetch_nam():
            if engine.nam_epoch != nam_epoch:
                raise RuntimeError('res.nam lives on the GPU and a later cna_amd call has replaced it; '
                                   'read it (or call res.materialize()) before running the next analysis')
            return pd.DataFrame(engine.nam_full(keep=kept, cols=colmap, transposed=True), index=sample_index,
                                columns=cell_index(), copy=False)

        res._defer('nam', fetch_nam)
        _mark('lazies set')

        if had_key:
            warnings.warn(f"Key '{key_added}' already exists in data.obs. Overwriting.")
        _mark('warned')
        # coef_all / fdr_all may be views of the engine's pinned buffers: the DataFrame stores its own
        # copy, and that copy (not the view) is what res.ncorrs is built from when somebody reads it
        if 'values' in early_coef:
            coef_kept = early_coef['values']                 # written while the null kernel was running
        else:
            early_coef['written'] = True                  # (roll_back undoes a late write as well)
            data.obs[key_added] = coef_all
            coef_kept = data.obs[key_added].values
        if np.may_share_memory(coef_kept, coef_all):
            coef_kept = np.array(coef_all)
        res._defer('ncorrs', lambda: pd.Series(coef_kept if kept.all() else coef_kept[kept], index=cell_index()))
        _mark('coef written')
        view = early_coef.get('fdr_view')
        early = fdr_copied_early()
        if fdr_all is None:
            pass                                              # local_test=False: see below
        elif early_coef.pop('fdr_done', False) and fdr_key in data.obs:
            pass                                              # written while the eigenvectors were computed
        elif early and view is not None and fdr_key in data.obs and np.shares_memory(data.obs[fdr_key].values, view):
            pass                                              # filled by the helper thread under the SVD
        elif (view is not None and isinstance(fdr_all, np.ndarray) and fdr_all.dtype == np.float64 and fdr_all.flags.c_contiguous
                and fdr_all.shape == view.shape and fdr_key in data.obs and np.shares_memory(data.obs[fdr_key].values, view)):
            _host_copy(view, fdr_all)
        else:
            early_coef['written'] = early_coef['fdr_touched'] = True
            data.obs[fdr_key] = fdr_all
        _mark('obs written')
    except BaseException:
        roll_back()
        # the fused selection call may have launched the local null before whatever raised (a `ks` too large for the
        # cohort, a failed draw, an interrupt): collect and drop it, or every later call on this engine finds it pending
        try:
            engine.null_local_discard()
        except Exception:               # noqa: BLE001 - the caller gets the error that brought us here
            pass
        raise
    if fdr_all is None:
        # upstream has written data.obs[key_added] and then dereferences res.fdrs, which is None when
        # local_test=False (_association.py:231,235): same state of data.obs, same exception
        raise AttributeError("'NoneType' object has no attribute 'loc'")

    if return_full:
        # everything but the three cells x samples frames is materialised now, like upstream
        for name in ('ncorrs', 'fdrs', 'namresid_sampleXpc', 'namresid_svs', 'namresid_varexp', 'yresid', 'beta'):
            getattr(res, name)
        return res
    return res.p
