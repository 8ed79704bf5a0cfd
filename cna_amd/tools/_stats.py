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


_randn_buf = {}


def legacy_randn(m, num, clean=False):
    """``np.random.randn(m, num)`` -- the same values, bit for bit, from numpy's global legacy
    generator, and the generator left exactly where numpy would leave it -- produced by the
    library's vectorised restatement of numpy's MT19937 / polar Box-Muller stream
    (csrc/host_rng.c; 2-2.5x faster, outside the GIL).  The draw is the longest host-side item on
    the critical path of a small analysis.  Falls back to numpy itself whenever the global generator
    is not a plain MT19937 RandomState.  ``clean``: the caller has just seeded the generator (no
    cached second value pending); an even number of draws then runs directly on numpy's own state
    memory instead of a get_state / set_state round trip (2 x 25-50 us)."""
    import ctypes as C
    n = int(m) * int(num)
    rs = getattr(np.random.mtrand, '_rand', None)
    bg = getattr(rs, '_bit_generator', None)
    if n == 0 or bg is None or type(bg).__name__ != 'MT19937':
        return np.random.randn(m, num)
    try:
        from .. import _ffi
        fn = _ffi.load().cna_host_legacy_randn
    except Exception:
        return np.random.randn(m, num)
    out = _randn_buf.get(n)
    if out is None:
        _randn_buf.clear()
        out = _randn_buf[n] = np.empty(n)
    with bg.lock:
        if clean and n % 2 == 0:
            # struct mt19937_state { uint32_t key[624]; int pos; } (numpy/random/src/mt19937/mt19937.h)
            addr = bg.ctypes.state_address
            ch, cg = C.c_int(0), C.c_double(0.0)
            if fn(addr, C.cast(addr + 624 * 4, C.POINTER(C.c_int)), C.byref(ch), C.byref(cg), n, out.ctypes.data) == 0:
                return out.reshape(m, num)
            return np.random.randn(m, num)
        name, key, pos, has_gauss, cached = rs.get_state(legacy=True)
        key = np.ascontiguousarray(key, dtype=np.uint32)
        cp, ch, cg = C.c_int(int(pos)), C.c_int(int(has_gauss)), C.c_double(float(cached))
        if fn(key.ctypes.data, C.byref(cp), C.byref(ch), C.byref(cg), n, out.ctypes.data) != 0:
            return np.random.randn(m, num)
        rs.set_state((name, key, cp.value, ch.value, cg.value))
    return out.reshape(m, num)          # a view of the reused buffer: consume before the next draw


_MID_DRAW = 80_000            # 100 samples x 1000 permutations: 0.91 -> 0.60 ms with four threads on the box (50 x 1000: no gain, 0.33 either way)
_BIG_DRAW = 400_000          # draws x rows from which the threaded host helpers pay (200 samples x 10 000 permutations: 2M)
_threads_set = False


def _host_threads():
    """Let csrc/host_rng.c use the CPUs this process may really use for large draws (set once)."""
    global _threads_set
    if not _threads_set:
        try:
            from .. import _ffi
            from .._order import usable_cpus
            _ffi.load().cna_host_set_threads(usable_cpus(16))
        except Exception:
            pass
        _threads_set = True


def _argsort_gather(R, Yv, out, rows=None):
    """out[rows] = Yv[argsort(R, axis=0)] (R: len(Yv) x num).  Large draws: one threaded C call outside the
    GIL (cna_host_argsort_gather; numpy's argsort along axis 0 of a 200 x 10 000 matrix takes ~24 ms on one
    thread); small ones: numpy (its vectorised sort wins there)."""
    if R.size >= _BIG_DRAW and Yv.dtype == np.float64 and out.dtype == np.float64:
        try:
            from .. import _ffi
            _host_threads()
            Rc = np.ascontiguousarray(R)
            Yc = np.ascontiguousarray(Yv)
            rr = None if rows is None else np.ascontiguousarray(rows, dtype=np.int64)
            if _ffi.load().cna_host_argsort_gather(_ffi.ptr(Rc), Rc.shape[0], Rc.shape[1], _ffi.ptr(Yc), _ffi.ptr(out),
                                                   out.shape[1], _ffi.ptr(rr)) == 0:
                return
        except Exception:
            pass
    if rows is None:
        out[:] = Yv[np.argsort(R, axis=0)]
    else:
        out[rows] = Yv[np.argsort(R, axis=0)]


def conditional_permutation(B, Y, num, clean=False):
    """Permute Y within the levels of B, ``num`` times (reference _stats.py:4-18).

    RNG consumption: one ``randn(len(level), num)`` block per level, levels in
    ``np.unique`` order.  ``clean``: see legacy_randn."""
    levels = np.unique(B)
    if len(Y) * num >= _BIG_DRAW:
        _host_threads()
    if len(levels) == 1 and len(B) == len(Y):
        # one level: members = arange(n), so src is the argsort itself
        R = legacy_randn(len(Y), num, clean)
        if R.size < _BIG_DRAW:
            return Y[np.argsort(R, axis=0)]
        out = np.empty((len(Y), num), dtype=Y.dtype)
        _argsort_gather(R, Y, out)
        return out
    # several levels: out[m] = Y[m[argsort]] = (Y[m])[argsort], level by level, without forming src
    out = np.empty((len(Y), num), dtype=Y.dtype)
    if len(Y):
        out[:] = Y[0]                         # rows of no level (NaN batch labels): upstream's src stays 0 there
    for b in levels:
        m = np.flatnonzero(B == b)
        _argsort_gather(legacy_randn(len(m), num, clean), Y[m], out, rows=m)
        clean = clean and (len(m) * num) % 2 == 0
    return out


class NativeDraw:
    """conditional_permutation(B, Y, num) of a freshly seeded generator, started on the library's own host thread
    (csrc/host_rng.c:cna_host_draw_start): no interpreter involved, so the draw runs while this thread validates,
    plans and launches kernels.  ``wait()`` returns the samples x (1 + num) matrix [Y | permutations] -- the matrix
    the phenotypes are conditioned from -- whose columns 1.. equal the reference's draw bit for bit, numpy's global
    generator left where the reference leaves it."""

    pending = True                               # (the library's thread may still be filling `table`: cna_assoc_finish joins it)
    _memo_key = None
    _idx = None

    def __init__(self, lib, keep, table):
        self._lib, self._keep, self.table, self._done = lib, keep, table, False
        self._state = None                           # (bit generator, address of its state, the worker's copy)
        self.flag = np.zeros(1, dtype=np.int32)      # 1: the worker has conditioned the phenotypes itself (then_condition)
        self._m = None

    def then_condition(self, engine, M):
        """Ask the worker to condition the phenotypes on the device as soon as the draw is there (no interpreter in
        between): engine.condition(M, table).  False when it is too late for that (already collected)."""
        if self._done or self._m is not None:
            return False
        M = np.ascontiguousarray(M, dtype=np.float64)
        if M.shape != (len(self.table), len(self.table)):
            return False
        if self._lib.cna_host_draw_then_condition(engine.h, M.ctypes.data, self.table.ctypes.data, len(self.table),
                                                  self.table.shape[1], self.flag.ctypes.data) != 0:
            return False
        self._m = M
        return True

    @property
    def conditioned(self):
        return int(self.flag[0]) == 1

    def wait(self):
        if not self._done:
            rc = self._lib.cna_host_draw_wait()
            self._done = True
            if rc == 0 and self._memo_key is not None and self._idx is not None and self._keep is not None:
                # the source rows of this seeded draw and where it leaves the generator: the next phenotype replays them
                import ctypes as C
                st = self._keep[4]
                while len(_draw_memo) >= _DRAW_MEMO_ENTRIES:
                    _draw_memo.pop(next(iter(_draw_memo)))
                _draw_memo[self._memo_key] = (self._idx, C.string_at(C.addressof(st), _MT_STATE_BYTES))
            if rc == 0 and self._state is not None:
                # the worker advanced a COPY of a freshly seeded state; numpy's own generator is seeded and written here,
                # once, whole, under its lock (np.random.seed(seed) + the draws: where the reference leaves it)
                import ctypes as C
                bg, addr, st, seed = self._state
                np.random.seed(seed)                  # (also drops a cached second normal, as the reference's call does)
                with bg.lock:
                    C.memmove(addr, st, _MT_STATE_BYTES)
            self._state = None
            self._keep = None
            if rc != 0:
                raise MemoryError('cna_host_draw_wait: the permutation draw failed (%d)' % rc)
        return self.table

    def abandon(self):
        """Collect the worker and forget what it drew: numpy's generator is left exactly as it was found (a draw that
        was started before the inputs were validated, on inputs that turned out not to be the ones to permute)."""
        self._state = None
        try:
            self.wait()
        except Exception:                      # noqa: BLE001
            pass

    def __del__(self):                         # the worker writes into buffers this object keeps alive
        self._state = None                     # (abandon(): a draw nobody collected must not reseed numpy's generator now)
        try:
            self.wait()
        except Exception:                      # noqa: BLE001
            pass


_draw_memo = {}          # (seed, samples, permutations) -> (source rows int32[m x num], generator state after the draw)
DRAW_MEMO = True         # False: every call draws again (bench.py: the timed steps repeat ONE phenotype -- with the memo on they
                         # would skip the draw the reference makes on every call, as they would skip the walk with the NAM cache)
_DRAW_MEMO_ENTRIES = 4
_DRAW_MEMO_BYTES = 64 << 20


class ReplayedDraw:
    """The permuted phenotypes of a seeded one-level draw that this process has made before -- `table[:, 1 + p] =
    Y[source rows of permutation p]` (conditional_permutation, _stats.py:4-18: the rows depend on seed, sample count and
    permutation count only) -- with NativeDraw's interface: wait() leaves numpy's generator where the draw leaves it."""
    flag = None
    conditioned = False
    pending = False

    def __init__(self, table, seed, state):
        self.table, self._seed, self._state = table, seed, state

    def wait(self):
        if self._state is not None:
            import ctypes as C
            st, self._state = self._state, None
            rs = getattr(np.random.mtrand, '_rand', None)
            bg = getattr(rs, '_bit_generator', None)
            np.random.seed(self._seed)
            with bg.lock:
                C.memmove(bg.ctypes.state_address, st, _MT_STATE_BYTES)
        return self.table

    def abandon(self):
        self._state = None

    def then_condition(self, engine, M):
        return False


def seeded_draw(Y, num, seed, threads=None):
    """conditional_permutation(ones, Y, num) of a generator seeded with `seed` (one level: batches=None), as an object with
    .table / .pending / .wait() / .abandon(): replayed from this process's memo of (seed, len(Y), num) when there is one
    (a gather, ~20 us at 50 x 1000 instead of 0.3-0.7 ms of normals and sorts), else drawn on the library's thread
    (native_draw_start) with the source rows recorded for the next phenotype.  None: shape not covered."""
    Y = np.asarray(Y)
    key = None
    if (DRAW_MEMO and isinstance(seed, (int, np.integer)) and not isinstance(seed, bool) and Y.dtype == np.float64
            and len(Y) * num * 4 <= _DRAW_MEMO_BYTES):
        key = (int(seed), len(Y), int(num))
        hit = _draw_memo.get(key)
        if hit is not None:
            rs = getattr(np.random.mtrand, '_rand', None)
            bg = getattr(rs, '_bit_generator', None)
            if bg is not None and type(bg).__name__ == 'MT19937' and _mt_layout_ok(bg, bg.ctypes.state_address):
                from .. import _ffi
                Yc = np.ascontiguousarray(Y)
                table = np.empty((len(Y), num + 1))
                table[:, 0] = Yc
                if threads is None:
                    from .._order import usable_cpus
                    threads = usable_cpus(8) if len(Y) * num >= _BIG_DRAW else 1
                if _ffi.load().cna_host_gather_rows(Yc.ctypes.data, hit[0].ctypes.data, len(Y), int(num), table.ctypes.data + 8,
                                                    num + 1, int(threads)) == 0:
                    return ReplayedDraw(table, seed, hit[1])
    d = native_draw_start(None, Y, num, seed, threads=threads, single_level=True, record=key is not None)
    if d is not None and key is not None:
        d._memo_key = key
    return d


def native_draw_start(B, Y, num, seed, threads=None, single_level=False, record=False):
    """Start conditional_permutation(B, Y, num) of a generator seeded with `seed` (np.random.seed(seed),
    _association.py:15-16) on the library's host thread; None when the shape is not covered (the caller then draws as
    before).  numpy's global generator is not touched before wait(): the worker runs on a copy of the state a PRIVATE
    RandomState has after seed(seed) -- so a draw may be started before the inputs are validated and dropped again
    (NativeDraw.abandon) without a trace."""
    import ctypes as C
    if seed is None or num < 2 or num % 2 or len(Y) < 1 or (not single_level and len(B) != len(Y)):
        return None
    Y = np.asarray(Y)
    if Y.dtype != np.float64:
        return None
    rs = getattr(np.random.mtrand, '_rand', None)
    bg = getattr(rs, '_bit_generator', None)
    if bg is None or type(bg).__name__ != 'MT19937':
        return None
    try:
        from .. import _ffi
        lib = _ffi.load()
        lib.cna_host_draw_start
    except Exception:
        return None
    levels = None if single_level else np.unique(B)       # (single_level: the caller knows B is one level, e.g. batches=None)
    if single_level or (len(levels) == 1 and bool((np.asarray(B) == levels[0]).all())):
        members = np.arange(len(Y), dtype=np.int64)
        off = np.array([0, len(Y)], dtype=np.int64)
    else:
        parts = [np.flatnonzero(B == b).astype(np.int64) for b in levels]
        members = np.concatenate(parts) if parts else np.zeros(0, dtype=np.int64)
        off = np.zeros(len(parts) + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(q) for q in parts])
    if len(members) == 0:
        return None
    Yc = np.ascontiguousarray(Y)
    table = np.empty((len(Y), num + 1))
    table[:, 0] = Yc
    if len(members) < len(Y):
        table[:, 1:] = Yc[0]                  # rows of no level (NaN batch labels): upstream's src stays 0 there
    if threads is None:                        # large draws (10 000 permutations of 200 samples): normals and sorts on several threads
        from .._order import usable_cpus
        if len(Y) * num >= _BIG_DRAW:
            threads = usable_cpus(16)
        elif len(Y) * num >= _MID_DRAW:        # 200 samples x 1000 permutations: 2.1 ms on one thread, as long as the walk of
            threads = usable_cpus(4)           # 250 000 cells -- a rank's block of the 2M problem on eight GPUs
                                               # (eight threads: one step in ten then takes 5 ms -- under a container's CPU quota
                                               # the draw, the eigenpairs, the content check and the runtime's own threads add up)
        else:
            threads = 1
    addr = bg.ctypes.state_address             # struct mt19937_state { uint32_t key[624]; int pos; }
    if not _mt_layout_ok(bg, addr):
        return None
    # The worker advances a copy of a freshly seeded state in memory this object owns; wait() seeds numpy's generator
    # and writes the copy into it under its lock.  Whoever touches np.random in between sees a consistent generator
    # (never a torn one).
    st = (C.c_uint32 * (_MT_STATE_BYTES // 4))()
    try:
        with _private_lock:
            _private_rs.seed(seed)             # (legacy seeding, the same routine np.random.seed runs; raises on a bad seed)
            pbg = _private_rs._bit_generator
            with pbg.lock:
                C.memmove(st, pbg.ctypes.state_address, _MT_STATE_BYTES)
    except (TypeError, ValueError):
        return None                            # np.random.seed(seed) in the caller's own draw reports it
    base = C.addressof(st)
    idx = np.empty((len(Y), int(num)), dtype=np.int32) if record else None
    rc = lib.cna_host_draw_start_idx(base, C.cast(base + 624 * 4, C.POINTER(C.c_int)), Yc.ctypes.data, len(Y), int(num),
                                     len(off) - 1, off.ctypes.data, members.ctypes.data, table.ctypes.data + 8, num + 1,
                                     int(threads), None if idx is None else idx.ctypes.data)
    if rc != 0:
        return None                            # (nothing drawn, numpy's generator untouched: the caller's own draw seeds it)
    d = NativeDraw(lib, (Yc, off, members, bg, st, idx), table)
    d._state = (bg, addr, st, seed)
    d._idx = idx
    return d


_MT_STATE_BYTES = 624 * 4 + 4
_mt_layout = None
import threading as _threading
_private_lock = _threading.Lock()
_private_rs = np.random.RandomState(0)         # seeded per draw under _private_lock; never handed out


def _mt_layout_ok(bg, addr):
    """Once per process: numpy's `struct mt19937_state { uint32_t key[624]; int pos; }` is where this module reads it
    (the public `state` dict against the raw memory)."""
    global _mt_layout
    if _mt_layout is None:
        import ctypes as C
        try:
            with bg.lock:
                raw = (C.c_uint32 * 625).from_address(addr)
                key = np.frombuffer(raw, dtype=np.uint32, count=624).copy()
                pos = C.c_int.from_address(addr + 624 * 4).value
            pub = bg.state['state']
            _mt_layout = bool(int(pub['pos']) == pos and np.array_equal(np.asarray(pub['key'], dtype=np.uint32), key))
        except Exception:                      # noqa: BLE001
            _mt_layout = False
    return _mt_layout


def grouplevel_permutation(G, Y, num, clean=False):
    """Permute whole groups (donors): samples sharing a value of G keep a common Y
    (reference _stats.py:20-32)."""
    groups = np.unique(G)
    per_group = np.array([Y[G == g][0] for g in groups])
    which = np.array([np.where(groups == g)[0][0] for g in G])
    if (per_group[which] != Y).any():
        print('ERROR: the value of Y is not identical within each group of samples')
        return
    order = np.argsort(legacy_randn(len(per_group), num, clean), axis=0)
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
