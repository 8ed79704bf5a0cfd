"""The fixed-shape analysis: cna.tl.association (reference _association.py:193-242) in two library calls.

The reference's function is one straight line; for the call shape that needs no decision of the host between its stages
-- ``nsteps`` given, one batch (``batches=None``), no donor groups, covariates allowed, a ``seed``, the local test on, one
GPU, a graph that is already resident -- the stages are queued by ``cna_assoc_begin`` / ``cna_assoc_finish``
(include/cna_hip.h, csrc/assoc.hip) without the interpreter in between:

    this module (numpy)      check_inputs for that shape (_association.py:131-173), the standardised phenotype (:22), the
                             projector of _nam.py:128-135 (under the walk), the storage of the two data.obs columns (under
                             the walk), p-value / warnings / result namespace (:64-129, :223-242)
    library thread           the permutation draw (_stats.py:4-18) and the conditioning of the phenotypes (:51-52,96-97)
    cna_assoc_begin          the walk (_nam.py:57-70)
    cna_assoc_finish         selection + residualisation + standardisation + coefficients (:175-191, _nam.py:118-159, :77),
                             Gram matrix, leading eigenpairs, global F-tests (_nam.py:105, :35-88), local null (:91-103),
                             FDR table (_stats.py:64-83), per-cell columns written into the frame's own storage (:230-237)

Every stage is the same entry point, in the same order, as the general path of tools/_association.py issues it: same
bits (tests/test_gpu_fast.py compares the two paths field by field).  Anything outside the shape -- and anything unusual
met on the way (a cell of zero variance, NaN coefficients, an eigen-solver that steps aside is handled here; a stale
graph or id column is not) -- returns NOT_TAKEN and the general path runs as if this module did not exist: it is the one
that reproduces the reference's errors, messages and progress text.
"""
import os
import warnings

import numpy as np
import pandas as pd

from .. import _ffi
from ._nam import (LazyNamespace, GramPCs, _NO_NAM, _defer_pcs, _lowrank_ok, _prepare_graph, _resid_plan, _still_resident,
                   _top_pcs, _walk_start, confirm_codes, eig_stats, get_connectivity, global_samples, host_blas_threads,
                   sample_codes_cached, shard_of)
from . import _nam as _nam_mod
from ._stats import default_ks, seeded_draw

ENABLED = os.environ.get('CNA_ONE_CALL', '1') not in ('0', 'off', 'no')     # 0: every call takes the general path (A/B runs, tests)
NOT_TAKEN = object()
stats = {'taken': 0, 'not_eligible': 0, 'general': 0, 'need_pcs': 0, 'stale': 0}
_BIG_CELLS = 500000          # from here on: the coefficient column is copied before the Gram matrix is waited for, on several threads
_eyes = {}
_cpus = None
_TRACE = None      # list of (label, perf_counter) when tools/host_trace.py switches tracing on


def _mark(label):
    if _TRACE is not None:
        import time
        _TRACE.append((label, time.perf_counter()))


def _copy_threads(n):
    global _cpus
    if n < _BIG_CELLS:
        return 1
    if _cpus is None:
        from .._order import usable_cpus
        _cpus = usable_cpus(8)
    return min(8, _cpus)


def _defer_last_cells():
    from ._association import _DEFER_LAST_CELLS
    return _DEFER_LAST_CELLS


def _rule_cells(data, engine):
    from ._association import _rule_cells as rule
    return rule(data, engine)


def _draw_threads(normals):
    """(what tools/_stats.py:native_draw_start gives a draw of this size)"""
    global _cpus
    if normals < 80_000:
        return 1
    if _cpus is None:
        from .._order import usable_cpus
        _cpus = usable_cpus(8)
    return min(4, _cpus)


def _eye(N):
    M = _eyes.get(N)
    if M is None:
        if len(_eyes) > 8:
            _eyes.clear()
        M = _eyes[N] = np.eye(N)
        M.setflags(write=False)
    return M


def _eligible(data, y, batches, covs, donorids, ks, nsteps, show_progress, engine, kwargs):
    """The cheap part of the shape test (no look at the data)."""
    if not ENABLED or show_progress or batches is not None or donorids is not None:
        return False
    # (nsteps=None: the reference's stop rule, _nam.py:59-68 -- evaluated on the device, cna_nam_auto_launch)
    if nsteps is not None and (isinstance(nsteps, bool) or not isinstance(nsteps, (int, np.integer)) or not 1 <= nsteps <= 15):
        return False
    if nsteps is None and getattr(engine, 'nam_auto_launch', None) is None:
        return False
    if getattr(engine, 'assoc_finish', None) is None or getattr(engine, 'h', None) is None:
        return False
    sharded = hasattr(data, 'obs') and shard_of(data) is not None       # one rank's block of cells (cna_amd.dist.shard)
    if ((engine.nranks != 1 or engine._has_comm) and not sharded) or getattr(engine, 'reuse_x', False):
        return False                            # (replicated inputs on several ranks: per-cell outputs are assembled by the general path)
    if not set(kwargs) <= {'Nnull', 'seed', 'local_test', 'force_permute_all'}:
        return False
    Nnull = kwargs.get('Nnull', 1000)
    if kwargs.get('seed') is None or not kwargs.get('local_test', True):
        return False
    if isinstance(Nnull, bool) or not isinstance(Nnull, (int, np.integer)) or Nnull < 2 or Nnull % 2 or Nnull > 1_000_000:
        return False
    if not isinstance(y, pd.Series) or y.dtype != np.float64:
        return False
    if covs is not None:
        if not isinstance(covs, pd.DataFrame) or covs.shape[1] < 1 or len(covs) != len(y):
            return False
        if not (covs.index is y.index or covs.index.equals(y.index)):
            return False
        if not all(dt == np.float64 for dt in covs.dtypes):
            return False
    if not hasattr(data, 'obs'):
        return False
    return True


def association(data, y, sid_name, batches, covs, donorids, ks, key_added, max_frac_pcs, nsteps, show_progress,
                allow_low_sample_size, return_full, ridges, engine, kwargs):
    """-> NOT_TAKEN, or what cna.tl.association returns (p, or the result namespace)."""
    if not _eligible(data, y, batches, covs, donorids, ks, nsteps, show_progress, engine, kwargs):
        stats['not_eligible'] += 1
        return NOT_TAKEN
    _mark('eligible')
    try:
        A = get_connectivity(data)
    except Exception:                          # noqa: BLE001 - the general path reports it
        return NOT_TAKEN
    # a graph that is new to the device goes up by the general path (it overlaps the upload with the factorisation of the ids)
    if engine._graph_ref is None or engine._graph_ref() is not A or engine._graph_key is None:
        stats['not_eligible'] += 1
        return NOT_TAKEN
    Nnull = int(kwargs.get('Nnull', 1000))
    seed = kwargs['seed']

    # ---- check_inputs (_association.py:131-173) for this shape: y is indexed by exactly the samples of the data
    sharded = shard_of(data) is not None
    # (one GPU: the content checks of ids and graph run inside cna_assoc_finish while the device works; several ranks
    # check before anything is queued -- a rank that found its copy stale on its own would leave the others in a collective)
    optimistic = not sharded and engine.nranks == 1 and not engine._has_comm
    codes, labels, counts, token = sample_codes_cached(data.obs[sid_name], defer='caller' if optimistic else False)
    if sharded:      # this rank's cells only: agree with the other ranks on the samples and their sizes
        codes, labels, counts, token = global_samples(engine, codes, labels, counts, token)
    _mark('codes')
    N_all = len(labels)
    if len(y) != N_all or N_all < 2 or not (counts > 0).all() or not labels.equals(y.index):
        stats['not_eligible'] += 1
        return NOT_TAKEN
    n = len(data.obs)
    r = 0 if covs is None else covs.shape[1]
    verify = []
    W_ = {}                                     # what queue_walk() leaves for the rest of the call

    def queue_walk():
        # The walk needs the graph and the sample codes only: every step that does not wait for the phenotype is queued
        # before the phenotype is even looked at (the last one waits when it may take the selection pass along: it needs y)
        engine._defer_graph_check = 'caller' if optimistic else False
        try:
            pend = _nam_mod.take_pending_codes()
            if pend is not None:
                verify.append(pend)
            _prepare_graph(engine, data, 1)
            pend = engine.take_pending_graph()
            if pend is not None:
                verify.extend(pend)
            _mark('graph')
            sig, held = _walk_start(engine, codes, labels, counts, token, nsteps, 15, 1, False)
            walk = held is _NO_NAM
            # (the last step leaves the selection pass's results on its way out under the general path's own rule -- wide
            # sample axis, a block of 150 000 cells or more, _association.py:_DEFER_LAST_CELLS -- so that both paths run the
            # same kernels)
            if nsteps is None:
                # the reference's default: walk until the median kurtosis stops falling (_nam.py:64-68); medians and rule on
                # the device, the first steps queued ahead of the verdict, which the selection pass collects
                may_hint, early = False, 0
                if walk:
                    engine.nam_auto_launch(15)
                    engine._nam_sig = (sig, engine.nam_epoch, None) if sig is not None else None
                W_.update(sig=sig, walk=False, may_hint=False, early=0)
                return
            may_hint = walk and covs is None and N_all > 64 and nsteps >= 3 and _rule_cells(data, engine) >= _defer_last_cells()
            early = (nsteps - 1 if may_hint else nsteps) if walk else 0
            if early:
                engine.assoc_begin_part(0, early, nsteps)
                if early == nsteps:            # the whole walk: whoever analyses this dataset next may keep it (NAM cache)
                    engine._nam_sig = (sig, engine.nam_epoch, nsteps) if sig is not None else None
            W_.update(sig=sig, walk=walk, may_hint=may_hint, early=early)
        finally:
            engine._defer_graph_check = False
        _mark('walk queued')

    def validate_and_draw():
        """-> None (the general path has to take this shape after all: it raises what the reference raises), or
        (N, whole, fv, y_std, colmap, ks_, ks_arr, kmax, the draw)"""
        yv = y.values
        fv = ~np.isnan(yv)
        if covs is not None:
            fv &= ~np.isnan(covs.values).any(axis=1)
        N = int(np.count_nonzero(fv))
        whole = N == N_all
        ysel = yv if whole else yv[fv]
        ks_ = default_ks(N) if ks is None else ks
        try:
            ks_arr = np.asarray(ks_)
            bad_ks = (ks_arr.ndim != 1 or len(ks_arr) < 1 or ks_arr.dtype.kind not in 'iu' or ks_arr.min() < 1
                      or ks_arr.max() + r >= N)
        except Exception:                      # noqa: BLE001
            bad_ks = True
        if (N < 10 and not allow_low_sample_size) or N < 2 or N > 1024 or bad_ks:
            return None
        with np.errstate(all='ignore'):
            y_std = (ysel - ysel.mean()) / ysel.std()      # numpy ddof=0, _association.py:22
        if not np.isfinite(y_std).all():
            return None
        colmap = None if whole else np.flatnonzero(fv).astype(np.int32)   # NAM.reindex(y.index)[filter]: labels == y.index
        _mark('validated')
        # the draw (_stats.py:4-18; one level: batches = ones, _association.py:146-147): replayed from the memo of this seed's
        # permutations when the process has drawn them before (they do not depend on the phenotype), else on the library's
        # thread from now on
        native = seeded_draw(y_std, Nnull, seed)
        if native is None:
            return None
        _mark('draw started')
        return N, whole, fv, y_std, colmap, ks_, ks_arr, int(ks_arr.max()), native

    # Which goes first: whichever chain is longer -- the draw (~10 ns per normal on the library's threads, plus the sorts)
    # or the walk (~11 ps per cell, sample and step on one MI355X); the other starts ~0.1 ms later, under it
    if N_all * Nnull * 1e-2 / _draw_threads(N_all * Nnull) > _rule_cells(data, engine) * N_all * (nsteps or 3) * 1.1e-5:
        got = validate_and_draw()
        if got is None:
            stats['not_eligible'] += 1
            return NOT_TAKEN
        try:
            queue_walk()
        except BaseException:
            got[-1].abandon()
            raise
    else:
        queue_walk()
        got = validate_and_draw()
        if got is None:
            # (the steps queued above are the ones the general path would queue itself: it restarts the walk -- or, with the
            # NAM cache on, finds this one -- and checks graph and ids on its own)
            stats['not_eligible'] += 1
            return NOT_TAKEN
    N, whole, fv, y_std, colmap, ks_, ks_arr, kmax, native = got
    sig, walk, may_hint, early = W_['sig'], W_['walk'], W_['may_hint'], W_['early']

    fdr_key = f'{key_added}_fdr'
    had_key, had_fdr = key_added in data.obs, fdr_key in data.obs
    previous = data.obs[key_added] if had_key else None
    previous_fdr = data.obs[fdr_key] if had_fdr else None
    touched = [False]

    def roll_back():
        # nothing written by this call outlives an exception or a detour through the general path
        if touched[0]:
            touched[0] = False
            for key, had, prev in ((key_added, had_key, previous), (fdr_key, had_fdr, previous_fdr)):
                if had:
                    data.obs[key] = prev
                elif key in data.obs:
                    del data.obs[key]

    def give_up(kind, stale=False):
        roll_back()
        native.abandon()                       # numpy's generator stays as it was found
        if stale:                              # some input is no longer what went to the device: both memos go
            _nam_mod.drop_codes_memo()
            engine.drop_graph()
        stats[kind] += 1
        return NOT_TAKEN

    try:
        if walk and early < nsteps:            # the last step, with the phenotype it may need
            engine.assoc_begin_part(early, nsteps - early, nsteps, y_std if (may_hint and whole) else None)
        if walk:
            engine._nam_sig = (sig, engine.nam_epoch, nsteps) if sig is not None else None
        nam_epoch = engine.nam_epoch
        _mark('walk queued')

        # ---- under the walk: the projector (_nam.py:118-135) and the storage of the two columns
        sample_index = pd.Index(y.index if whole else y.index[fv], name=sid_name)
        if covs is None:
            Mv, Cm, W, M_frame = _eye(N), None, None, None
        else:
            # (the same pandas statements as the reference and the general path -- covs.reindex(y.index)[filter],
            # _association.py:189, :201 -- so that the frame the projector is made from has the same memory layout: BLAS
            # sums in another order on a transposed operand)
            plan = _resid_plan(sample_index, covs.reindex(y.index)[pd.Series(fv, index=y.index)], None)
            M_frame = plan.M
            Mv = np.asarray(M_frame, dtype=np.float64)
            Cm, W = np.asarray(plan.C.values, dtype=np.float64), plan.W
            if plan.kind != 'single' or plan.r != r or not _lowrank_ok(engine, plan) or not np.isfinite(Mv).all():
                return give_up('general')      # (a constant covariate: the general path raises the reference's error)
        _mark('plan')
        touched[0] = True
        data.obs[key_added] = np.empty(n)
        data.obs[fdr_key] = np.empty(n)
        coef_view, fdr_view = data.obs[key_added].values, data.obs[fdr_key].values
        in_place = all(v.dtype == np.float64 and v.flags.c_contiguous and v.flags.writeable and v.shape == (n,)
                       for v in (coef_view, fdr_view))

        _mark('storage')
        threads = _copy_threads(n)
        out = engine.assoc_finish(y_std, Mv, ks_arr, Nnull, native.table, colmap=colmap, Cmat=Cm, W=W, draw_pending=native.pending,
                                  coef_dst=coef_view if in_place else None, fdr_dst=fdr_view if in_place else None,
                                  copy_threads=threads, native_eig=_nam_mod._EIG_NATIVE, resid_tol=_nam_mod._EIG_RESID,
                                  gap_tol=_nam_mod._EIG_GAP, verify=verify, verify_threads=_verify_threads(verify))
        verify = None
        _mark('finish returned')
        if out['status'] == _ffi.ASSOC_STALE:
            return give_up('stale', stale=True)      # (results of a stale copy never get out: nothing was written)
        if out['status'] == _ffi.ASSOC_GENERAL:
            return give_up('general')
        native.wait()
        _mark('rng')                           # numpy's generator: seeded and advanced as the reference leaves it

        if out['status'] == _ffi.ASSOC_NEED_PCS:
            # the library's eigen-solver stepped aside (a degenerate leading spectrum, ks beyond a quarter of the samples)
            stats['need_pcs'] += 1
            eig_stats['lapack'] += 1
            pcs = GramPCs(out['G'])
            with host_blas_threads(1):
                Uk = _top_pcs(out['G'], kmax, native=False)
                if Uk is None:
                    Uk = pcs.U[:, :kmax]
            best, pv, r2v = engine.global_test(Uk, ks_arr, r)
        else:
            eig_stats['native'] += 1
            pcs = None
            Uk, best, pv, r2v = out['U'], out['kidx'], out['minp'], out['r2']
        if (best < 0).any():
            raise ValueError('All-NaN slice encountered')        # np.nanargmin in _minp_stats
        k, p = ks_[best[0]], pv[0]
        if k == max(ks_):
            warnings.warn(('data supported use of {} NAM PCs, which is the maximum considered. ' +
                           'Consider allowing more PCs by using the "ks" argument.').format(k))
        nullminps = pv[1:]
        hits = (nullminps <= p + 1e-8).sum()
        pfinal = (hits + 1) / (Nnull + 1)
        if hits == 0:
            warnings.warn('global association p-value attained minimal possible value. ' +
                          'Consider increasing Nnull')
        fdr_vals = out['fdr']
        if return_full or np.isnan(fdr_vals).all():
            # (a table without a single finite entry: the reference fails in its np.min -- and so does _fdr_tables)
            from ._association import _fdr_tables
            fdr_vals, fdr_5p_t, fdr_10p_t, _ = _fdr_tables(out['tail_sums'], out['ranks'], min(1000, Nnull), out['thr'].copy())
        _mark('p')
        if had_key:
            warnings.warn(f"Key '{key_added}' already exists in data.obs. Overwriting.")
        # data.obs[key], data.obs[key + '_fdr'] (_association.py:230-237): filled in place by the library, unless the
        # frame's storage could not be written that way
        if not (out['coef_in_dst'] and np.shares_memory(data.obs[key_added].values, coef_view)):
            data.obs[key_added] = np.array(out['coef'])
        if not (out['fdr_in_dst'] and np.shares_memory(data.obs[fdr_key].values, fdr_view)):
            data.obs[fdr_key] = np.array(out['fdr_col'])
        touched[0] = False
        _mark('obs checked')
        if return_full:
            res = _full_result(engine, data, out, pcs, Uk, best, pv, r2v, k, pfinal, ks, ks_, r, N, n, Mv, M_frame, y_std,
                               sample_index, colmap, key_added, max_frac_pcs, nam_epoch, fdr_vals, fdr_5p_t, fdr_10p_t)
    except BaseException:
        roll_back()
        native.abandon()
        for undo in (engine.null_local_discard, engine.global_test_discard):
            try:
                undo()
            except Exception:                  # noqa: BLE001 - the caller gets the error that brought us here
                pass
        raise
    finally:
        engine._defer_graph_check = False
    stats['taken'] += 1
    return res if return_full else pfinal


def _verify_threads(verify):
    """Threads of the content check inside cna_assoc_finish: one for an id column of a few MB, the host's share for a graph."""
    global _cpus
    if not verify or sum(a.nbytes for a, _ in verify) < (8 << 20):
        return 1
    if _cpus is None:
        from .._order import usable_cpus
        _cpus = usable_cpus(8)
    return min(4, _cpus)                       # (beside the draw's threads and the eigenpairs: see tools/_stats.py)


def _full_result(engine, data, out, pcs, Uk, best, pv, r2v, k, pfinal, ks, ks_, r, N, n, Mv, M_frame, y_std, sample_index,
                 colmap, key_added, max_frac_pcs, nam_epoch, fdr_vals, fdr_5p_t, fdr_10p_t):
    """The result namespace (_association.py:64-129, _nam.py:168-175): cells-sized and sign-defining fields
    on demand, everything else materialised like upstream."""
    if pcs is None:
        pcs = GramPCs(out['G'])
    r2 = r2v[0]
    nullminps, nullr2s = pv[1:], r2v[1:]
    # coefficients and r2 of the chosen model (_association.py:69-74)
    ycond_v = Mv.dot(y_std)
    ycond_v = ycond_v / ycond_v.std(ddof=1)
    beta_k = Uk[:, :k].T.dot(ycond_v)
    yhat = Uk[:, :k].dot(beta_k)
    r2_perpc = (beta_k / np.sqrt(ycond_v.dot(ycond_v))) ** 2
    thresholds, num_detected = out['thr'].copy(), out['num_detected'].copy()     # (views of the library call's output block)
    res = LazyNamespace()
    res.M = M_frame if M_frame is not None else pd.DataFrame(np.eye(N), columns=sample_index, index=sample_index)
    res.r = r
    x_epoch = engine.x_epoch

    def cell_index():
        return data.obs.index

    def fetch_namresid():
        _still_resident(engine, x_epoch)
        return pd.DataFrame(engine.x_full(transposed=True), index=sample_index, columns=cell_index())
    res._defer('namresid', fetch_namresid)

    def names():
        return ['PC' + str(i) for i in range(1, N + 1)]
    npcs = min(N, max([10] + [int(max_frac_pcs * N)] + [ks if ks is not None else []][0]))
    res._defer('namresid_sampleXpc', lambda: pd.DataFrame(pcs.U, index=sample_index, columns=names()))
    res._defer('namresid_svs', lambda: pd.Series(pcs.svs, index=names())[:npcs])
    res._defer('namresid_varexp', lambda: pd.Series(pcs.svs, index=names()) / N / engine.x_rows_global())   # (all ranks' cells)
    res._defer('yresid', lambda: pd.Series(ycond_v, index=sample_index))
    res._defer('beta', lambda: pcs.U[:, :k].T.dot(ycond_v))
    res._defer('fdrs', lambda: pd.DataFrame({'threshold': thresholds, 'fdr': fdr_vals, 'num_detected': num_detected}))
    res.__dict__.update({'p': pfinal, 'nullminps': nullminps, 'k': k, 'fdr_5p_t': fdr_5p_t, 'fdr_10p_t': fdr_10p_t,
                         'yresid_hat': yhat, 'ks': ks_, 'r2': r2, 'r2_perpc': r2_perpc, 'nullr2_mean': nullr2s.mean(),
                         'nullr2_std': nullr2s.std()})
    _defer_pcs(res, engine, pcs, cell_index)
    kept = np.repeat(True, n)
    res.kept = kept

    def fetch_nam():
        if engine.nam_epoch != nam_epoch:
            raise RuntimeError('res.nam lives on the GPU and a later cna_amd call has replaced it; '
                               'read it (or call res.materialize()) before running the next analysis')
        return pd.DataFrame(engine.nam_full(keep=kept, cols=colmap, transposed=True), index=sample_index,
                            columns=cell_index(), copy=False)
    res._defer('nam', fetch_nam)
    coef_kept = data.obs[key_added].values
    res._defer('ncorrs', lambda: pd.Series(coef_kept, index=cell_index()))
    # everything but the three cells x samples frames is materialised now, like upstream
    for name in ('ncorrs', 'fdrs', 'namresid_sampleXpc', 'namresid_svs', 'namresid_varexp', 'yresid', 'beta'):
        getattr(res, name)
    return res
