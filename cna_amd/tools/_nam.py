"""NAM construction, QC and residualisation/PCA -- host orchestration over the HIP engine.

Mirrors the public surface of the reference's ``cna/tools/_nam.py`` (names, arguments,
return types, progress text); the cells-sized arithmetic is done by libcna_hip.so:

  reference step (file:line)                               device entry point
  colsums = A.sum(axis=0)+w            _nam.py:28          cna_colsums
  s <- A.(s/colsums)+w*s/colsums       _nam.py:33          cna_nam_step / cna_dense_step
  median_i kurtosis_j(s/C)             _nam.py:59          cna_nam_step(want_kurt) + host median
  _batch_kurtosis                      _nam.py:78-82       cna_batch_kurtosis
  NAM_ = M.(NAM - mean)                _nam.py:122-148     cna_resid_apply
  NAM_ /= std                          _nam.py:159         cna_standardize
  NAM.dot(NAM.T)                       _nam.py:105         cna_gram     (N x N SVD stays in LAPACK)
  V = NAM^T U / sqrt(svs)              _nam.py:106         cna_project  (lazy)

Device matrices are cells x samples; DataFrames handed back to the caller are transposed to
the reference's samples x cells on the way out.
"""
import os
import warnings
from argparse import Namespace

import numpy as np
import pandas as pd
import scipy.sparse as sp

from .. import _ffi
from ..engine import get_engine
from ._out import select_output

DEFAULT_RIDGES = [1e5, 1e4, 1e3, 1e2, 1e1, 1e0, 1e-1, 1e-2, 1e-3, 1e-4, 0]

try:   # optional: keeps LAPACK from fanning a 50 x 50 SVD out over every core of a big host
    from threadpoolctl import ThreadpoolController as _TPC
    _blas_pool = _TPC()
except Exception:   # pragma: no cover - threadpoolctl not installed
    _blas_pool = None


import contextlib


def host_blas_threads(limit=1):
    """Context manager: run the host's sample-space linear algebra (N x N, N x Nnull) on `limit`
    BLAS threads.  These matrices are tiny; a threaded BLAS fans them out over every core it
    sees, its idle workers spin, and under a container CPU quota that spinning gets the whole
    process throttled (measured on the GPU box: 25 ms for a 50 x 50 SVD, and 50 ms stalls at
    random places, with 64 OpenBLAS threads under a 16-CPU cgroup quota)."""
    if _blas_pool is None:
        return contextlib.nullcontext()
    return _blas_pool.limit(limits=limit, user_api='blas')


def _small_svd(G):
    """np.linalg.svd of the samples x samples Gram (_nam.py:105): the same LAPACK routine as the
    reference, so PC signs agree."""
    return np.linalg.svd(G)


_LAPACKE_DSYEVR = None


def _lapacke_dsyevr():
    """LAPACKE_dsyevr of the OpenBLAS that scipy itself loads, through ctypes: the same routine as
    scipy.linalg.lapack.dsyevr -- whose f2py wrapper keeps the GIL for the whole call -- without the GIL, so that the
    interpreter goes on beside it (`_association`: the per-cell pass of a small problem under LAPACK).  False when that
    library is not to be found."""
    global _LAPACKE_DSYEVR
    if _LAPACKE_DSYEVR is None:
        _LAPACKE_DSYEVR = False
        try:
            import ctypes as C
            from scipy.linalg import lapack          # noqa: F401  (maps the library)
            path = None
            with open('/proc/self/maps') as f:
                for line in f:
                    if 'libscipy_openblas' in line and '64_' not in os.path.basename(line.split()[-1]):
                        path = line.split()[-1]
                        break
            if path:
                fn = C.CDLL(path).scipy_LAPACKE_dsyevr
                fn.restype = C.c_int
                fn.argtypes = [C.c_int, C.c_char, C.c_char, C.c_char, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_double,
                               C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
                _LAPACKE_DSYEVR = fn
        except Exception:
            _LAPACKE_DSYEVR = False
    return _LAPACKE_DSYEVR


# acceptance of the library's own eigen-solver (csrc/host_eig.c): residual and orthogonality at rounding level, and every
# gap of the leading spectrum wide enough for the individual vectors to be defined to ~1e-10 (two backward-stable solvers
# agree to ~eps ||G|| / gap); anything else is LAPACK's to decide
_EIG_NATIVE = True       # (False: always LAPACK's dsyevr; tests patch it)
_EIG_RESID = 1e-12
_EIG_GAP = 1e-6
eig_stats = {'native': 0, 'lapack': 0}


def _top_pcs_native(G, kmax):
    """csrc/host_eig.c:cna_host_top_eig, accepted only with its evidence (see above); None -> the caller asks LAPACK."""
    n = len(G)
    if not _EIG_NATIVE or n < 8 or 4 * kmax > n or kmax + 1 > 256:
        return None
    try:
        import ctypes as C
        from .. import _ffi
        fn = _ffi.load().cna_host_top_eig
    except Exception:                           # noqa: BLE001
        return None
    Gc = np.ascontiguousarray(G, dtype=np.float64)
    U = np.empty((n, kmax))
    lam = np.empty(kmax + 1)
    resid, ortho = C.c_double(0.0), C.c_double(0.0)
    if fn(Gc.ctypes.data, n, kmax, U.ctypes.data, lam.ctypes.data, C.byref(resid), C.byref(ortho)) != 0:
        return None
    top = lam[0]
    if not (top > 0 and np.isfinite(lam).all()):
        return None
    if not (resid.value <= _EIG_RESID * top and ortho.value <= _EIG_RESID):
        return None
    if not ((lam[:-1] - lam[1:]) > _EIG_GAP * top).all():
        return None                             # a (near-)degenerate leading spectrum: individual vectors are LAPACK's choice
    return U


def _top_pcs(G, kmax, native=True):
    """The kmax leading eigenvectors of the samples x samples Gram matrix (columns, leading first), for the global
    F-tests.  Every statistic of _association.py:35-48 depends on the PCs only through squared projections -- not on
    their signs -- so the tests need not wait for the sign-defining LAPACK SVD of _nam.py:105, which is left to
    `GramPCs` (a worker thread, or whoever first reads a field that shows the signs).  First choice: the library's own
    checked solver for the leading pairs (csrc/host_eig.c; no LAPACK, no interpreter: 2-3x shorter than dsyevr at 200
    samples); else LAPACK's dsyevr with an index range.  Same subspaces to ~1e-14 (tests/test_host_eig.py)."""
    n = len(G)
    if kmax >= n or not np.isfinite(G).all():
        return None                                 # degenerate input: the caller takes the SVD (and its errors)
    if native:
        U = _top_pcs_native(G, kmax)
        if U is not None:
            eig_stats['native'] += 1
            return U
        eig_stats['lapack'] += 1
    return _top_pcs_lapack(G, kmax)


def _top_pcs_lapack(G, kmax):
    """LAPACK's dsyevr with an index range (tridiagonalisation + the wanted eigenpairs)."""
    n = len(G)
    # LAPACK's wrapper directly: scipy.linalg.eigh spends as long again on argument checks and a workspace query
    # (264 -> 208 us at 50 samples; the call sits on the critical path of a small analysis)
    fn = _lapacke_dsyevr()
    if fn:
        import ctypes as C
        a = np.array(G, dtype=np.float64, order='F')            # (destroyed by the routine)
        w = np.empty(n)
        z = np.empty((n, kmax), order='F')
        isuppz = np.empty(2 * max(kmax, 1), dtype=np.int32)
        mm = C.c_int(0)
        info = fn(102, b'V', b'I', b'L', n, a.ctypes.data, n, 0.0, 0.0, n - kmax + 1, n, 0.0, C.addressof(mm),
                  w.ctypes.data, z.ctypes.data, n, isuppz.ctypes.data)          # 102: column major
        if info == 0 and mm.value == kmax:
            return np.ascontiguousarray(z[:, ::-1])
    from scipy.linalg import lapack
    _, z, m, _, info = lapack.dsyevr(G, compute_v=1, range='I', il=n - kmax + 1, iu=n, lower=1, overwrite_a=0)
    if info != 0 or m != kmax:
        from scipy.linalg import eigh
        _, v = eigh(G, subset_by_index=[n - kmax, n - 1], driver='evr', check_finite=False, overwrite_a=False)
        return np.ascontiguousarray(v[:, ::-1])
    return np.ascontiguousarray(z[:, :m][:, ::-1])


_svd_workers = None


def _forget_pools():
    """In a forked child the worker thread of this pool does not exist; the next user makes a new one."""
    global _svd_workers
    _svd_workers = None


if hasattr(os, 'register_at_fork'):
    os.register_at_fork(after_in_child=_forget_pools)


class GramPCs:
    """np.linalg.svd of the samples x samples Gram matrix (_nam.py:105; the LAPACK routine that defines the PC
    signs of the reference), computed when `U` / `svs` are first read -- or from start() on a worker thread,
    beside the local-null kernel."""

    def __init__(self, G):
        self._G, self._out, self._fut = G, None, None

    def _run(self, limit):
        if limit:
            with host_blas_threads(1):
                return _small_svd(self._G)
        return _small_svd(self._G)

    def start(self):
        """Begin now on a worker thread (the caller holds host_blas_threads(1) until it has read the result)."""
        global _svd_workers
        if self._out is None and self._fut is None:
            if _svd_workers is None:
                from concurrent.futures import ThreadPoolExecutor
                _svd_workers = ThreadPoolExecutor(max_workers=1, thread_name_prefix='cna-svd')
            self._fut = _svd_workers.submit(self._run, False)
        return self

    def _get(self):
        if self._out is None:
            U, svs, _ = self._fut.result() if self._fut is not None else self._run(True)
            self._out, self._fut, self._G = (U, svs), None, None
        return self._out

    @property
    def U(self):
        return self._get()[0]

    @property
    def svs(self):
        return self._get()[1]

    def __len__(self):
        return len(self._G) if self._out is None else len(self._out[0])


class LazyNamespace(Namespace):
    """Result namespace whose cells-sized fields are fetched from the device on first access
    (SURVEY.md §8f.1).  ``isinstance(res, argparse.Namespace)`` holds; a lazily provided
    attribute appears in ``vars(res)`` once it has been read."""

    def _defer(self, name, thunk):
        self.__dict__.setdefault('_lazy', {})[name] = thunk

    def __getattr__(self, name):
        lazy = self.__dict__.get('_lazy', {})
        if name in lazy:
            value = lazy.pop(name)()
            setattr(self, name, value)
            return value
        raise AttributeError(name)

    def materialize(self):
        for name in list(self.__dict__.get('_lazy', {})):
            getattr(self, name)
        return self


def get_connectivity(data):
    """The kNN graph of an AnnData-like object (reference _nam.py:12-19).  anndata >= 0.7.2
    keeps it in ``obsp``; objects from older versions only have ``uns['neighbors']``."""
    obsp = getattr(data, 'obsp', None)
    if obsp is not None and 'connectivities' in obsp:
        return obsp['connectivities']
    return data.uns['neighbors']['connectivities']


def _as_csr(a):
    if sp.isspmatrix_csr(a) or isinstance(a, getattr(sp, 'csr_array', ())):
        return a
    return sp.csr_matrix(a)


def sample_codes(col):
    """Column order of ``pd.get_dummies(obs[sid])`` (_nam.py:51): category order for a
    categorical column (unused categories included), sorted unique labels otherwise."""
    if isinstance(col.dtype, pd.CategoricalDtype):
        return np.asarray(col.cat.codes, dtype=np.int32), pd.Index(col.cat.categories)
    # factorize in order of appearance, then rank the (few) labels: pandas' sort=True re-maps the
    # per-cell codes through a generic take that is several times slower than the hashing itself
    codes, uniques = pd.factorize(col)
    uniques = pd.Index(uniques)
    order = uniques.argsort()
    rank = np.empty(len(order) + 1, dtype=np.int32)
    rank[order] = np.arange(len(order), dtype=np.int32)
    rank[-1] = -1                                   # NaN ids (code -1) stay -1
    return rank[codes], uniques[order]


_codes_cache = {}


try:
    from xxhash import xxh3_128_intdigest as _content_hash
except Exception:   # pragma: no cover - xxhash not installed
    import hashlib

    def _content_hash(buf):
        return int.from_bytes(hashlib.blake2b(buf, digest_size=16).digest(), 'little')


def _content_hash_threaded(arr):
    """Large id columns (millions of cells): the library's threaded 64-bit content hash
    (csrc/host_graph.c) -- 16 MB in ~0.1 ms on 8 threads, where one thread of xxh3 takes 0.4 ms of a
    call whose whole GPU part is 35 ms; small columns stay on xxh3 (no thread start-up)."""
    # (the library's hash at every size -- one thread below 4 MB -- so that the library itself can repeat the check:
    # cna_assoc_finish's verify list, tools/_fast.py)
    from .._order import usable_cpus
    threads = 1 if arr.nbytes < (4 << 20) else min(usable_cpus(8), max(1, arr.nbytes >> 21))
    return int(_ffi.load().cna_host_hash64(_ffi.ptr(arr), arr.nbytes, threads))


def _fingerprint(arr):
    """Identity + content check of a numeric per-cell id array: where it lives, its layout, and a
    128-bit hash of its bytes (xxh3: one pass at memory speed, ~10x cheaper than factorising the
    column again, and -- unlike a sum or xor -- sensitive to the order of the values, so an in-place
    shuffle of the ids is seen too)."""
    return (arr.__array_interface__['data'][0], arr.shape, arr.strides, arr.dtype.str,
            _content_hash_threaded(arr))


_codes_pending = None


def sample_codes_cached(col, defer=False):
    """sample_codes() with a one-entry-per-column memo for numeric and categorical id columns.

    Canonicalising the ids (hash 200k-2M values, rank the labels, count cells per sample) is pure
    input preparation and identical for every phenotype tested on a dataset.  The memo is only
    reused when the column's buffer is the same AND the hash of its full content matches, so an
    in-place edit of the ids (a shuffle included) is seen.  Returns (codes, labels, counts, token); the token lets the
    engine keep the codes resident on the device.

    defer=True: when buffer, shape and dtype match the memo, return it at once and hash the content on the
    checker thread (0.35 ms for 2M ids, otherwise in front of the first kernel); the caller MUST call
    confirm_codes() before it lets any result out and start over, without defer, if that returns False."""
    global _codes_pending
    _codes_pending = None
    if isinstance(col.dtype, pd.CategoricalDtype):
        arr = np.asarray(col.cat.codes)
        extra = tuple(col.cat.categories)
    else:
        arr = col.values if isinstance(col.values, np.ndarray) else None
        extra = ()
    if arr is None or arr.dtype.kind not in 'iuf' or not arr.flags.c_contiguous or arr.dtype.itemsize not in (1, 2, 4, 8):
        codes, labels = sample_codes(col)
        counts = np.bincount(codes if codes.min(initial=0) >= 0 else codes[codes >= 0], minlength=len(labels))
        return codes, labels, counts, None
    hit = _codes_cache.get('last')
    where = (arr.__array_interface__['data'][0], arr.shape, arr.strides, arr.dtype.str)
    if defer and hit is not None and hit[0][:4] == where and hit[0][5:] == (extra,) and not isinstance(col.dtype, pd.CategoricalDtype):
        if defer == 'caller':
            # the caller has the content hashed itself (take_pending_codes: the library does it inside cna_assoc_finish,
            # while the device works); confirm_codes() still settles it if the caller never takes it
            _codes_pending = ('caller', arr, hit[0][4])
            return hit[1]
        from ..engine import _checker
        _codes_pending = (_checker().submit(_content_hash_threaded, arr), hit[0][4])
        return hit[1]
    fp = _fingerprint(arr) + (extra,)
    if hit is not None and hit[0] == fp:
        return hit[1]
    codes, labels = sample_codes(col)
    counts = np.bincount(codes if codes.min(initial=0) >= 0 else codes[codes >= 0], minlength=len(labels))
    out = (codes, labels, counts, fp)
    _codes_cache['last'] = (fp, out)
    return out


def confirm_codes():
    """Outcome of the deferred content check of sample_codes_cached(defer=True): True = the memo was the
    column's content (or nothing was deferred).  False: the ids were edited in place; the memo is dropped."""
    global _codes_pending
    pend, _codes_pending = _codes_pending, None
    if pend is None:
        return True
    if pend[0] == 'caller':
        if _content_hash_threaded(pend[1]) == pend[2]:
            return True
    elif pend[0].result() == pend[1]:
        return True
    _codes_cache.pop('last', None)
    return False


def take_pending_codes():
    """The deferred check of sample_codes_cached(defer='caller') as (array, the 64-bit hash its content must have), for a
    caller that has it verified elsewhere (and drops the memo itself -- drop_codes_memo -- when that fails); else None."""
    global _codes_pending
    pend = _codes_pending
    if pend is None or pend[0] != 'caller':
        return None
    _codes_pending = None
    return pend[1], pend[2]


def drop_codes_memo():
    _codes_cache.pop('last', None)


def _column_r2(a, b):
    # R(A,B)**2 of _nam.py:47-49 (diagnostic print only).  The reference's operands are DataFrames,
    # so the covariance is a population moment but both std() calls are pandas' ddof=1.
    with np.errstate(all='ignore'):
        r = ((a - a.mean(axis=0)) * (b - b.mean(axis=0))).mean(axis=0) / a.std(axis=0, ddof=1) / b.std(axis=0, ddof=1)
    return r ** 2


def shard_of(data):
    """(row0, n_global) when `data` holds one rank's block of cells (cna_amd.dist.shard), else None."""
    info = getattr(data, 'uns', None)
    info = info.get('cna_shard') if hasattr(info, 'get') else None
    return None if info is None else (int(info['row0']), int(info['n_global']))


_global_codes_cache = {}


def global_samples(engine, codes, labels, counts, token):
    """Sharded callers: a rank sees the sample ids of its own cells only.  Returns what
    sample_codes_cached() would have returned on the whole dataset, restricted to this rank's cells:
    codes into the sorted union of every rank's labels, the union, cells per sample over all ranks,
    and a token all ranks agree on (None as soon as one rank cannot vouch for its ids)."""
    local = -1 if token is None else (hash(token) & 0x3fffffffffffffff)
    sigs = engine.allgather_fixed([local])[:, 0]
    gtoken = None if (sigs < 0).any() else ('sharded',) + tuple(int(v) for v in sigs)
    hit = _global_codes_cache.get('last')
    if gtoken is not None and hit is not None and hit[0] == gtoken:
        return hit[1]
    everyone = engine.allgather_objects(np.asarray(labels))
    if all(len(l) == len(everyone[0]) and np.array_equal(l, everyone[0]) for l in everyone):
        glabels, gcodes = labels, np.asarray(codes)          # e.g. one categorical dtype shared by all ranks
    else:
        glabels = pd.Index(np.unique(np.concatenate([np.asarray(l, dtype=object) for l in everyone])))
        try:
            glabels = pd.Index(np.asarray(glabels, dtype=np.result_type(*[l.dtype for l in everyone])))
        except TypeError:
            pass
        remap = np.append(glabels.get_indexer(pd.Index(labels)), -1).astype(np.int32)
        gcodes = remap[np.asarray(codes)]
    own = np.bincount(gcodes[gcodes >= 0], minlength=len(glabels)).astype(np.int64)
    gcounts = engine.allgather_fixed(own).sum(axis=0)
    out = (gcodes, glabels, gcounts, gtoken)
    if gtoken is not None:
        _global_codes_cache['last'] = (gtoken, out)
    return out


def _prepare_graph(engine, data, self_weight):
    A = _as_csr(get_connectivity(data))
    if engine.ensure_graph(A, shard=shard_of(data), defer=getattr(engine, '_defer_graph_check', False)):
        engine._nam_sig = None          # new graph: whatever NAM the device holds is stale
    engine.colsums(self_weight)
    return A


# --------------------------------------------------------------------------- diffusion API
def diffuse_stepwise(data, s, maxnsteps=15, show_progress=False, self_weight=1, engine=None):
    """Generator over random-walk steps of a dense cells x m state (reference _nam.py:21-34).
    Yields an ndarray (or a DataFrame when ``s`` is one) after every step."""
    out = select_output(show_progress)
    engine = engine or get_engine()
    _prepare_graph(engine, data, self_weight)
    frame = s if isinstance(s, pd.DataFrame) else None
    arr = np.asarray(s, dtype=np.float64)
    if arr.ndim != 2:
        raise ValueError('s must be 2-dimensional (cells x columns)')
    engine.dense_begin(arr)
    for i in range(maxnsteps):
        print('\ttaking step', i + 1, file=out)
        engine.dense_step()
        cur = engine.dense_state()
        yield pd.DataFrame(cur, index=frame.index, columns=frame.columns) if frame is not None else cur


def diffuse(data, s, nsteps, show_progress=False, self_weight=1, engine=None):
    """State after ``nsteps`` steps (reference _nam.py:36-41)."""
    for s in diffuse_stepwise(data, s, maxnsteps=nsteps, show_progress=show_progress,
                              self_weight=self_weight, engine=engine):
        pass
    return s


# --------------------------------------------------------------------------- NAM on device
_NO_NAM = object()


def _walk_start(engine, codes, labels, counts, token, nsteps, maxnsteps, self_weight, show_progress):
    """What precedes the first step of a walk on a prepared graph (_nam.py:51-54): the sample codes and sizes on the
    device -- unless the device still holds the NAM of exactly these inputs.  Returns (signature of the inputs or
    None, steps of the held NAM or _NO_NAM: the caller walks)."""
    # (cells per sample, aligned with labels: _qc_device leaves samples without cells -- NaN rows of the NAM -- out of
    # the batch means, as pandas' mean does in the reference)
    # (keyed on the labels object itself -- the one _nam_device hands back to its caller, who passes it on to
    # _qc_device: a stash left by another dataset with as many samples can never be mistaken for this one's)
    engine._sample_counts = (np.asarray(counts), labels)
    # NAM cache: the NAM is a function of the graph, the per-cell sample ids, the
    # step rule and the self weight only -- not of the phenotype.  When the device still holds the
    # NAM of exactly these inputs (same resident graph, same id fingerprint, no walk started since),
    # a further analysis on the same dataset skips the diffusion.  Off while progress is printed
    # (the per-step diagnostics are part of the output) and when engine.reuse_nam is False.
    sig = None
    if token is not None and not show_progress and getattr(engine, 'reuse_nam', False):
        sig = (token, nsteps, maxnsteps, float(self_weight))
        held = getattr(engine, '_nam_sig', None)
        if held is not None and held[0] == sig and held[1] == engine.nam_epoch:
            return sig, held[2]
    engine.set_samples(codes, len(labels), counts.astype(np.float64), token=token)
    return sig, _NO_NAM


def _nam_device(engine, data, sid_name, nsteps=None, maxnsteps=15, self_weight=1, show_progress=False,
                codes_labels=None, defer_last=False):
    """Reference ``_nam`` (_nam.py:44-76) with the state resident on the GPU.  On return the
    engine holds NAM = (s/C) (cells x samples); returns (labels, steps taken)."""
    out = select_output(show_progress)

    def walk_queued():
        # lets a caller hold back host work that competes for the interpreter (the permutation draw on its helper
        # thread) until the first kernels are on the device
        cb = getattr(engine, '_on_walk_queued', None)
        if cb is not None:
            engine._on_walk_queued = None
            cb()
    _prepare_graph(engine, data, self_weight)
    token = None
    if codes_labels is not None and len(codes_labels) == 4:
        codes, labels, counts, token = codes_labels
    elif codes_labels is not None and len(codes_labels) == 3:
        codes, labels, counts = codes_labels
    else:
        codes, labels = codes_labels if codes_labels is not None else sample_codes(data.obs[sid_name])
        counts = np.bincount(codes if codes.min(initial=0) >= 0 else codes[codes >= 0], minlength=len(labels))
        if codes_labels is None and engine.view_local:
            codes, labels, counts, token = global_samples(engine, codes, labels, counts, None)
    N = len(labels)
    sig, held_steps = _walk_start(engine, codes, labels, counts, token, nsteps, maxnsteps, self_weight, show_progress)
    if held_steps is not _NO_NAM:
        walk_queued()
        return labels, held_steps
    n = engine.n

    need_kurt = (nsteps is None) or show_progress
    prevmedkurt = np.inf
    old = None
    taken = 0
    if not need_kurt and 1 <= nsteps <= maxnsteps:
        if defer_last and nsteps >= 2 and N > 64 and hasattr(engine, 'nam_select_hint'):
            # All steps but the last are queued now; the caller queues the last one -- finish(y_std or None) -- once it
            # knows what the selection pass will be asked for (validation and the residualisation plan run while the
            # first steps do): with a phenotype the last step does that pass on its way out (cna_nam_select_hint).
            for _ in range(nsteps - 1):
                engine.nam_step(False, True, False)
            walk_queued()
            engine._nam_sig = None

            def finish(y_hint=None):
                if y_hint is not None:
                    engine.nam_select_hint(y_hint)
                engine.nam_step(False, False, True)
                engine._nam_sig = (sig, engine.nam_epoch, nsteps) if sig is not None else None
            return labels, nsteps, finish
        engine.nam_steps(nsteps)                 # nothing to decide between steps: one call queues them all
        walk_queued()
        engine._nam_sig = (sig, engine.nam_epoch, nsteps) if sig is not None else None
        return labels, nsteps
    if (nsteps is None and not show_progress and 1 <= maxnsteps <= 16 and hasattr(engine, 'nam_auto_launch')
            and os.environ.get('CNA_AUTO_HOST', '0') in ('0', '', 'off', 'no')):
        # the reference's default: walk until the median kurtosis stops falling (_nam.py:64-68).  Medians and rule are
        # evaluated on the device and the steps are queued ahead of the verdict: no host round trip per step
        # (the call returns once the first steps are queued; whoever reads the NAM next collects the verdict)
        engine.nam_auto_launch(maxnsteps)
        walk_queued()
        engine._nam_sig = (sig, engine.nam_epoch, None) if sig is not None else None
        return labels, None
    for i in range(maxnsteps):
        last_for_sure = (nsteps is not None and i + 1 == nsteps) or (i + 1 == maxnsteps)
        may_stop = last_for_sure or show_progress or (nsteps is None and i + 1 >= 3)
        engine.nam_step(need_kurt, not last_for_sure, may_stop)
        walk_queued()
        taken = i + 1
        if need_kurt:
            medkurt = engine.stat_median()
            if show_progress:
                # R2(t, t-1) is scale free per column, so NAM = s/C serves as s (_nam.py:60)
                cur = engine.nam_full()
                R2 = _column_r2(cur, old if old is not None else np.zeros_like(cur))
                old = cur
                print('\tmedian kurtosis:', medkurt + 3, file=out)
                with np.errstate(all='ignore'):
                    print('\t20th percentile R2(t,t-1):', np.percentile(R2, 20), file=out)
        if nsteps is None:
            if prevmedkurt - medkurt < 3 and i + 1 >= 3:
                print('stopping after', i + 1, 'steps', file=out)
                break
            prevmedkurt = medkurt
        elif i + 1 == nsteps:
            break
    engine._nam_sig = (sig, engine.nam_epoch, taken) if sig is not None else None
    return labels, taken


def _batch_codes(batches, index):
    """Integer code per sample of ``index`` in ``np.unique(batches)`` order (what the
    reference iterates over in _batch_kurtosis, _nam.py:79-82); NaN -> -1."""
    b = batches.reindex(index) if isinstance(batches, pd.Series) else pd.Series(np.asarray(batches), index=index)
    vals = b.values
    try:
        isnan = pd.isna(vals)
    except TypeError:
        isnan = np.zeros(len(vals), dtype=bool)
    uniq = np.unique(vals[~isnan])
    codes = np.full(len(vals), -1, dtype=np.int32)
    codes[~isnan] = np.searchsorted(uniq, vals[~isnan])
    return codes, len(uniq)


def _qc_device(engine, labels, batches, show_progress=False):
    """Reference ``_qc_nam`` (_nam.py:85-99) -> bool keep mask over all cells."""
    out = select_output(show_progress)
    if len(np.unique(batches)) == 1:
        return np.repeat(True, engine.n)
    codes, nb = _batch_codes(batches, labels)
    if (codes < 0).any():
        # a NaN among the batch labels of the NAM's samples: np.unique makes it a level, `batches == nan` selects nobody,
        # the mean of nobody is NaN and so is every neighbourhood's batch kurtosis (_nam.py:78-99) -- none is kept
        print('throwing out neighborhoods with batch kurtosis >=', 6, file=out)
        print('keeping', 0, 'neighborhoods', file=out)
        return np.repeat(False, engine.n)
    # A sample without cells (an unused category of a categorical id column) has a NaN row in the NAM (0/0, _nam.py:73);
    # the reference's batch means are DataFrame.mean, which skips NaN (_nam.py:78-82): such a sample belongs to no batch
    # here (fixture c19_unused_category_batches; found by differential fuzzing)
    held = getattr(engine, '_sample_counts', None)
    if held is not None and held[1] is labels and len(held[0]) == len(codes):
        codes = np.where(held[0] > 0, codes, -1).astype(np.int32)
    engine.batch_kurtosis(_ffi.MAT_NAM, codes, nb)
    if not show_progress and hasattr(engine, 'stat_qc'):
        # median, threshold and the count of failing cells are formed on the device: when nobody fails (always, with
        # up to seven batches: the kurtosis of so few batch means cannot reach 6) the vector stays where it is
        _, _, n_dropped = engine.stat_qc()
        if n_dropped == 0:
            return np.repeat(True, engine.n)
    threshold = max(6, 2 * engine.stat_median())
    kurtoses = engine.cell_stat(engine.n)
    print('throwing out neighborhoods with batch kurtosis >=', threshold, file=out)
    with np.errstate(invalid='ignore'):
        keep = kurtoses < threshold
    print('keeping', keep.sum(), 'neighborhoods', file=out)
    return keep


def nam(data, sid_name, batches=None, nsteps=None, self_weight=1, max_frac_pcs=0.15, suffix='', ks=None,
        show_progress=False, engine=None, **kwargs):
    """Neighborhood abundance matrix and QC mask (reference _nam.py:179-193).

    Returns ``(DataFrame samples x kept cells, bool keep[n_cells])``.  ``max_frac_pcs``,
    ``suffix``, ``ks`` and extra keywords are accepted and ignored exactly as upstream.

    Limits the reference does not have: at most 1024 samples and 256 batches, fewer than 2**31 cells."""
    out = select_output(show_progress)
    engine = engine or get_engine()
    if batches is None:
        u = data.obs[sid_name].unique()          # (a shard's own samples: one batch either way)
        batches = pd.Series(np.ones(len(u)), index=u)
    print('computing NAM', file=out)
    labels, _ = _nam_device(engine, data, sid_name, nsteps=nsteps, self_weight=self_weight,
                            show_progress=show_progress)
    keep = _qc_device(engine, labels, batches, show_progress=show_progress)
    kept_t = engine.nam_full(keep=keep, transposed=True)       # samples x kept cells, caller's cell order
    index = pd.Index(labels, name=sid_name)
    frame = pd.DataFrame(kept_t, index=index, columns=data.obs.index[keep], dtype=float, copy=False)
    return frame, keep


# ------------------------------------------------------------------ residualise + PCA
def _pc_names(n):
    return ['PC' + str(i) for i in range(1, n + 1)]


def svd_nam(NAM, engine=None):
    """PCA of a samples x cells NAM through its Gram matrix (reference _nam.py:102-115):
    returns ``(U DataFrame, svs Series, V DataFrame)``."""
    engine = engine or get_engine()
    X = np.ascontiguousarray(np.asarray(NAM, dtype=np.float64).T)      # cells x samples
    engine.upload_x(X)
    engine.standardize(center=True)
    G = engine.gram()
    with host_blas_threads(1):
        U, svs, _ = _small_svd(G)
    with np.errstate(all='ignore'):
        V = engine.project_full(U / np.sqrt(svs))
    names = _pc_names(U.shape[1])
    index = NAM.index if isinstance(NAM, pd.DataFrame) else None
    columns = NAM.columns if isinstance(NAM, pd.DataFrame) else None
    return (pd.DataFrame(U, index=index, columns=names),
            pd.Series(svs, index=names),
            pd.DataFrame(V, index=columns, columns=names))


def _names(index_or_thunk):
    return index_or_thunk() if callable(index_or_thunk) else index_or_thunk


def _resid_plan(sample_index, covs, batches, ridges=None):
    """Host-only half of the reference's ``_resid_nam`` (_nam.py:118-146): standardise the
    covariates, one-hot the batches, and -- when no ridge schedule is involved -- form the
    projector M.  Needs nothing from the device, so the caller runs it while the diffusion
    kernels execute."""
    N = len(sample_index)
    plan = Namespace(N=N, sample_index=sample_index, M=None, ridges=None, standardized=False)
    if covs is None:
        covs = pd.DataFrame(np.ones((N, 0)), index=sample_index)
    else:
        covs = (covs - covs.mean(axis=0)) / covs.std(axis=0)
    if batches is None or len(np.unique(batches)) == 1:
        plan.C = covs
        if len(covs.T) == 0:
            plan.kind = 'identity'
            plan.M = pd.DataFrame(np.eye(N), columns=sample_index, index=sample_index)
        else:
            plan.kind = 'single'
            W = np.linalg.solve(covs.T.dot(covs), covs.T)
            M = np.eye(N) - covs.dot(W)
            M.columns = M.index
            plan.M = M
            plan.W = np.asarray(W, dtype=np.float64)          # M = I - C.W: the device applies it in factored form
    else:
        plan.kind = 'ridge'
        B = pd.get_dummies(batches)
        plan.B = (B - B.mean(axis=0)) / B.std(axis=0)
        plan.C = pd.concat([plan.B, covs], axis=1)
        plan.ridges = DEFAULT_RIDGES if ridges is None else ridges
        plan.bcodes, plan.nb = _batch_codes(batches, sample_index)
        # the first ridge's factors now (under the walk): with up to seven batches the schedule always ends there
        if len(plan.ridges):
            plan.first_ridge = _ridge_factors(plan.C, len(plan.B.T), plan.ridges[0], N)
    plan.r = len(plan.C.T)
    return plan


def _ridge_factors(C, n_batch_cols, ridge, N):
    """W = (C^T C + ridge N L)^-1 C^T and M = I - C W of one ridge of the schedule (_nam.py:143-146)."""
    L = np.diag([1] * n_batch_cols + [0] * (len(C.T) - n_batch_cols))
    W = np.linalg.solve(C.T.dot(C) + ridge * len(C) * L, C.T)
    M = np.eye(N) - C.dot(W)
    M.columns = M.index
    return W, M


def _lowrank_ok(engine, plan):
    """Apply M = I - C.W in factored form (engine.resid_lowrank)?  When the engine has it and both factors
    fit the kernel's LDS budget (2 r N doubles <= 128 KB)."""
    return hasattr(engine, 'resid_lowrank') and 0 < plan.r and 2 * plan.r * plan.N * 8 <= 128 * 1024


def _resid_run(engine, plan, cell_index, show_progress=False):
    """Device half of ``_resid_nam`` (_nam.py:122,135,148-159,163): centre, apply M (or the ridge
    schedule, which needs the batch kurtosis of the partially residualised matrix after every
    ridge), divide by the per-cell std, and queue the Gram-matrix kernels.  The engine's working
    matrix X is residualised in place; the Gram matrix is collected with engine.gram_fetch()."""
    out = select_output(show_progress)
    N, C, sample_index = plan.N, plan.C, plan.sample_index
    if plan.kind == 'identity':
        M = plan.M
        if not plan.standardized:              # M = I: centring + division by the std is one row-local pass
            engine.standardize(center=True)
    elif plan.kind == 'single':
        M = plan.M
        if plan.standardized:                  # the selection pass already applied M (engine.set_resid_factors)
            pass
        elif _lowrank_ok(engine, plan):
            # x.M^T = x - (x.W^T).C^T row by row, fused with centring, /std and (y known) the coefficients
            y_std = getattr(plan, 'y_std', None)
            m = engine.resid_lowrank(np.asarray(C.values, dtype=np.float64), plan.W, center=True, standardize=True, y=y_std)
            if y_std is not None:
                plan.maxabs = m
        else:
            engine.resid_apply(M.values, center=True)
            engine.standardize(center=False)
    else:
        B = plan.B
        first = True
        M = None
        y_std = getattr(plan, 'y_std', None)
        done = False
        if getattr(plan, 'onepass', None) is not None:
            # QC, selection and this first ridge were ONE pass over the NAM (tools/_association.py: engine.select_resid_bk)
            W, M = plan.first_ridge
            print('\twith ridge', plan.ridges[0], 'median batch kurtosis = ', plan.onepass[3], file=out)
            first = False
            done = True
        elif (len(plan.ridges) and y_std is not None and _lowrank_ok(engine, plan) and hasattr(engine, 'resid_lowrank_bk')
                and getattr(plan, 'reselect', None) is not None and os.environ.get('CNA_RIDGE_ONEPASS', '1') not in ('0', 'off', 'no')):
            # The first ridge in ONE pass, optimistically: residualise, batch kurtosis (median taken on the device),
            # and -- assuming the schedule ends here, as it always does with up to seven batches -- the division by the
            # std and the coefficients.  Should the median say otherwise, X is selected again and the schedule runs
            # ridge by ridge as below.
            W, M = plan.first_ridge
            m, med = engine.resid_lowrank_bk(np.asarray(C.values, dtype=np.float64), np.asarray(W, dtype=np.float64), y_std,
                                             plan.bcodes, plan.nb)
            if med <= 6:
                print('\twith ridge', plan.ridges[0], 'median batch kurtosis = ', med, file=out)
                plan.maxabs = m
                first = False
                done = True
            else:
                plan.reselect()
        for i, ridge in enumerate(() if done else plan.ridges):
            W, M = plan.first_ridge if i == 0 and hasattr(plan, 'first_ridge') else _ridge_factors(C, len(B.T), ridge, N)
            if _lowrank_ok(engine, plan):
                engine.resid_lowrank(np.asarray(C.values, dtype=np.float64), np.asarray(W, dtype=np.float64), center=first)
            else:
                engine.resid_apply(np.asarray(M.values, dtype=np.float64), center=first)
            first = False
            engine.batch_kurtosis(_ffi.MAT_X, plan.bcodes, plan.nb)
            med = engine.stat_median()
            print('\twith ridge', ridge, 'median batch kurtosis = ', med, file=out)
            if med <= 6:
                break
        if first:   # empty ridge list: only centring applies
            engine.resid_apply(None, center=True)
        if done:
            pass
        elif y_std is not None and not first and _lowrank_ok(engine, plan):
            # division by the std (_nam.py:159) and the coefficients X.y/N (_association.py:77) in one row-local pass
            # (the projector part of that kernel with no factors)
            plan.maxabs = engine.resid_lowrank(np.zeros((N, 0)), np.zeros((0, N)), center=False, standardize=True, y=y_std)
        else:
            engine.standardize(center=False)

    # the caller's coefficient column, when the coefficients came with the residualisation pass: queued in front
    # of the Gram kernels so that the host can write it into the frame while those run
    if (getattr(plan, 'coef_first', False) and getattr(plan, 'maxabs', None) is not None
            and getattr(engine, 'percell_coef_launch', None) is not None):
        plan.coef_launched = bool(engine.percell_coef_launch())
    # svd_nam re-centres / re-standardises (_nam.py:103-104); X is already standardised, so
    # that is an identity up to 1 ulp and the Gram matrix is taken of X directly.
    engine.gram_launch()
    res = LazyNamespace()
    res.M = M
    res.r = plan.r
    epoch = engine.x_epoch
    n_cells = engine.x_rows_total

    def fetch_namresid():
        _still_resident(engine, epoch)
        full_t = engine.x_full(transposed=True)           # samples x cells
        return pd.DataFrame(full_t, index=sample_index, columns=_names(cell_index))

    res._defer('namresid', fetch_namresid)
    return res


def _still_resident(engine, epoch):
    if engine.x_epoch != epoch:
        raise RuntimeError('this result field lives on the GPU and a later cna_amd call has replaced it; '
                           'read it (or call res.materialize()) before running the next analysis')


def _defer_pcs(res, engine, pcs, cell_index):
    """namresid_nbhdXpc: V = NAM^T U / sqrt(svs) (_nam.py:106), computed on the device when read.
    pcs: GramPCs (U and svs are taken when the field is read)."""
    epoch = engine.x_epoch
    n_cells = engine.x_rows_total

    def fetch_V():
        _still_resident(engine, epoch)
        U, svs = pcs.U, pcs.svs
        with np.errstate(all='ignore'):
            V = engine.project_full(U / np.sqrt(svs))
        return pd.DataFrame(V, index=_names(cell_index), columns=_pc_names(len(U)))

    res._defer('namresid_nbhdXpc', fetch_V)
