"""Device engine: thin, typed Python wrapper around one ``cna_ctx`` of libcna_hip.so.

One Engine per process/GPU.  It owns the device-resident graph, diffusion state, NAM and
working matrix; the host code in ``cna_amd.tools`` drives it step by step the way the
reference drives numpy/scipy.  Nothing here computes on the CPU.
"""
import ctypes as C
import os
import weakref

import numpy as np
import scipy.sparse as sp

from . import _ffi, _order
from ._ffi import check, ptr, MAT_NAM, MAT_X


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Engine(_order.CellOrder):
    def __init__(self, device=None, rank=0, nranks=1, unique_id=None, shm=None):
        """unique_id: RCCL id from rank 0 (one GPU per rank).  shm: (segment name, slot bytes) selects
        the host-staged test communicator instead, which lets several ranks share one GPU."""
        self.lib = _ffi.load()
        if device is None:
            device = int(os.environ.get('LOCAL_RANK', '0')) if nranks > 1 else 0
        h = _ffi.c_ctx()
        check(self.lib.cna_ctx_create(int(device), C.byref(h)), 'cna_ctx_create')
        self.h = h
        self.device = int(device)
        self.rank, self.nranks = int(rank), int(nranks)
        if nranks > 1 and unique_id is None and shm is None:
            raise ValueError('nranks > 1 needs the RCCL unique id created by rank 0')
        self._has_comm = unique_id is not None or shm is not None
        self.halo_comm = False
        if shm is not None:
            check(self.lib.cna_comm_init_shm(self.h, self.rank, self.nranks, str(shm[0]).encode(), int(shm[1])),
                  'cna_comm_init_shm')
        elif unique_id is not None:   # also with one rank: every collective then really goes through RCCL
            buf = (C.c_char * 128).from_buffer_copy(bytes(unique_id))
            check(self.lib.cna_comm_init(self.h, self.rank, self.nranks, C.cast(buf, C.c_void_p)), 'cna_comm_init')
            # a rank that cannot reach a peer should say so now, not hang in the first collective of an analysis
            ok = C.c_int(0)
            check(self.lib.cna_comm_selftest(self.h, float(os.environ.get('CNA_COMM_TIMEOUT', '120')), C.byref(ok)),
                  'cna_comm_selftest')
            self.halo_comm = bool(ok.value)     # the halo exchange has a communicator of its own (overlaps the walk step)
        self._graph_key = None
        self._graph_hash = None
        self._pending_check = None
        self._defer_graph_check = False
        self._graph_ref = None
        self._pinned = None
        self._host_threads = _order.usable_cpus(8)
        self._colsum_w = None
        self._codes_token = self._codes_graph = None
        self._assoc_out = None
        self._zc_cols = 0
        self._fused = None
        self.n = 0            # cells in the caller's view: all of them, or (view_local) this rank's block
        self.n_global = 0
        self.row0 = 0
        self.n_local = 0
        self.N = 0
        self.x_rows_total = 0
        self.x_epoch = 0      # bumped whenever the working matrix X is replaced
        self.nam_epoch = 0    # bumped whenever a new NAM is started
        # keep the NAM across analyses of one dataset (tools._nam._nam_device); engine.reuse_nam = False recomputes it
        # every call (what bench.py measures)
        self.reuse_nam = True
        # ... and, opt-in, the standardised NAM of an analysis without covariates and batches: a further phenotype then
        # only takes new coefficients (tools._association.compute_nam_and_reindex)
        self.reuse_x = False

    # ---------------------------------------------------------------- lifetime
    def close(self):
        if getattr(self, 'h', None):
            self.lib.cna_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def comm_info(self):
        """(backend, nranks) of the communicator: 'none' | 'rccl' | 'shm', and the rank count it reports."""
        b, n = C.c_int(0), C.c_int(0)
        check(self.lib.cna_comm_info(self.h, C.byref(b), C.byref(n)), 'cna_comm_info')
        return ('none', 'rccl', 'shm')[b.value], n.value

    def sync(self):
        check(self.lib.cna_ctx_sync(self.h), 'cna_ctx_sync')

    def set_state_f32(self, on=True):
        """Opt-in: keep the diffusion state between two steps of a walk in 4 bytes per entry (one rank, more than 64
        samples; sums, NAM and everything downstream stay float64).  Halves what the dense step gathers; the NAM then
        agrees with the default walk to ~1e-7 relative instead of bit for bit (reference: float64 `s` between the
        iterations of `_nam.py:31-34`).  A NAM held on the device from the other format is forgotten."""
        on = bool(on)
        if on != getattr(self, '_state_f32', None):
            check(self.lib.cna_set_state_f32(self.h, int(on)), 'cna_set_state_f32')
            self._state_f32 = on
            self._nam_sig = None
        return self

    def device_bytes(self):
        b = C.c_int64(0)
        check(self.lib.cna_ctx_device_bytes(self.h, C.byref(b)), 'cna_ctx_device_bytes')
        return b.value

    @staticmethod
    def new_unique_id():
        lib = _ffi.load()
        buf = (C.c_char * 128)()
        check(lib.cna_comm_unique_id(C.cast(buf, C.c_void_p)), 'cna_comm_unique_id')
        return bytes(buf)

    # ---------------------------------------------------------------- graph
    def block(self, n):
        """Rows [r0, r1) of an n-cell problem owned by this rank (ceil split)."""
        rpr = -(-n // self.nranks)
        r0 = min(self.rank * rpr, n)
        return r0, min(r0 + rpr, n)

    def _quick_key(self, A):
        """Cheap identity of a graph: the scipy object, its buffers, sizes and dtypes, and a hash of three
        64 KB windows (head, middle, tail) of values and indices (~20 us).  Necessary, not sufficient."""
        ident = self._ident(A)
        w = 16384
        mid = max(0, A.nnz // 2 - w // 2)
        parts = [slice(0, w), slice(mid, mid + w), slice(max(0, A.nnz - w), A.nnz)]
        probe = tuple(self._hash(arr[p_], 1) for arr in (A.data, A.indices) for p_ in parts)
        return ident + probe

    def _ident(self, A):
        return (A.shape, int(A.nnz), str(A.data.dtype), str(A.indices.dtype), str(A.indptr.dtype), id(A)) + self._buffers(A)

    def _full_hash(self, A):
        """64-bit hashes of the FULL content of data, indices and indptr (csrc/host_graph.c, several
        threads: ~0.5 ms for the 66 MB of a 200k-cell graph, 4.6 ms at 2M cells) -- an in-place edit of
        any entry changes it, and the graph then goes to the device again, as the reference re-reads the
        matrix on every call (_nam.py:25-28)."""
        return tuple(self._hash(arr) for arr in (A.data, A.indices, A.indptr))

    @staticmethod
    def _buffers(A):
        return (A.data.ctypes.data, A.indices.ctypes.data, A.indptr.ctypes.data)

    def _hash(self, arr, threads=None):
        arr = np.ascontiguousarray(arr)
        return int(self.lib.cna_host_hash64(ptr(arr), arr.nbytes, self._host_threads if threads is None else threads))

    def graph_resident(self, A):
        """Cheap and conservative: True when A is, by object and buffers, the matrix whose copy is on the device (whether
        its content still is what went up is ensure_graph's business)."""
        try:
            return (self._graph_key is not None and self._graph_ref is not None and self._graph_ref() is A
                    and self._graph_key[:len(self._ident(A))] == self._ident(A))
        except Exception:                      # noqa: BLE001
            return False

    def pin_graph(self, A):
        """Promise that the connectivities matrix A will not be edited in place while it is resident:
        later calls then recognise it without hashing all of it (what a loop over many phenotypes of
        one dataset wants; bench.py does this and says so).  `unpin_graph()` withdraws the promise."""
        if not sp.isspmatrix_csr(A) and not isinstance(A, sp.csr_array):
            raise TypeError('pin_graph needs the CSR matrix that is passed to the analysis')
        self._pinned = (weakref.ref(A), self._buffers(A))

    def unpin_graph(self):
        self._pinned = None

    # ---- the device cell order of a large graph, computed beside its first analyses
    def _start_reorder(self, A):
        self._drop_reorder()
        box = {'ref': weakref.ref(A), 'buffers': self._buffers(A), 'ident': self._ident(A)}

        def job():
            perm = _order.locality_order(A)
            if perm is None:
                return None
            # (the content the order was made from: adopted only if the matrix still hashes to it.  The renumbered rows
            # themselves are made on the device from the resident copy: cna_graph_reorder)
            return np.ascontiguousarray(perm, dtype=np.int64), self._full_hash(A)
        box['future'] = _reorder_pool().submit(job)
        self._reorder = box

    def _drop_reorder(self):
        self._reorder = None

    def _take_reorder(self, A):
        """The finished order for this very matrix (same object, same buffers), else None; never waits."""
        box = getattr(self, '_reorder', None)
        if box is None or box['ref']() is not A or box['buffers'] != self._buffers(A) or not box['future'].done():
            return None
        self._reorder = None
        try:
            out = box['future'].result()
        except Exception:                      # noqa: BLE001 - the caller's order stays
            return None
        if out is None or out[1] != self._graph_hash:
            return None                        # edited in place since: the next content check uploads it afresh
        pinned = self._pinned is not None and self._pinned[0]() is A and self._pinned[1] == self._buffers(A)
        if not pinned and self._full_hash(A) != out[1]:
            return None                        # ... or edited after the order was made (an unpinned matrix is hashed in full, as always)
        return out

    def reorder_pending(self):
        """True while the cell order of the resident graph is still being computed beside the analyses (the
        graph is then resident in the caller's order: same results, a slower random walk)."""
        box = getattr(self, '_reorder', None)
        return box is not None

    def wait_reorder(self):
        """Block until that computation has finished (benchmarks: the next call adopts it)."""
        box = getattr(self, '_reorder', None)
        if box is not None:
            try:
                box['future'].result()
            except Exception:                  # noqa: BLE001
                pass

    def ensure_graph(self, A, shard=None, defer=False):
        """Upload the connectivities graph unless this very matrix -- same object, same content -- is
        already resident.  The cells are kept on the device in a locality-preserving order (see
        _order.py); everything that crosses this class's boundary is in the caller's order.

        Content check: the full hash of data / indices / indptr (`_full_hash`), unless the matrix is
        pinned (`pin_graph`).  defer=True (one GPU only): when the cheap identity matches, return at once
        and take the full hash on a helper thread; the caller MUST call `confirm_graph()` before it lets
        any result out and start over if that returns False (tools._association does).

        shard=None: A is the whole n x n graph (on every rank; each keeps its row block).
        shard=(row0, n_global): A holds only the rows [row0, row0 + A.shape[0]) of the graph, with
        global column ids -- the block engine.block(n_global) assigns to this rank.  The engine then
        works in the local view: every per-cell input and output covers this rank's cells only."""
        if not sp.issparse(A):
            raise TypeError('connectivities must be a scipy.sparse matrix')
        if not sp.isspmatrix_csr(A) and not isinstance(A, sp.csr_array):
            A = sp.csr_matrix(A)
        if shard is None and A.shape[0] != A.shape[1]:
            raise ValueError('connectivities must be square')
        self._pending_check = None
        pinned = self._pinned is not None and self._pinned[0]() is A and self._pinned[1] == self._buffers(A)
        shard_key = None if shard is None else tuple(int(v) for v in shard)
        staged = None
        if pinned and self._graph_key is not None and self._graph_ref is not None and self._graph_ref() is A:
            # pinned: the caller's promise replaces the content probes (six 64 KB windows from cold memory: 0.15 ms)
            ident = self._ident(A)
            if self._graph_key[:len(ident)] == ident and self._graph_key[-1] == shard_key:
                staged = self._take_reorder(A)
                if staged is None:
                    return False
        if (staged is None and defer == 'caller' and not pinned and self.nranks == 1 and not self._has_comm
                and self._graph_key is not None and self._graph_ref is not None and self._graph_ref() is A
                and getattr(self, '_reorder', None) is None):
            # the caller has ALL of the matrix hashed while the device works (take_pending_graph: inside
            # cna_assoc_finish): the probes of the quick key (six 64 KB windows: 0.1 ms) would tell it nothing more
            ident = self._ident(A)
            if self._graph_key[:len(ident)] == ident and self._graph_key[-1] == shard_key:
                self._pending_check = ('caller', A)
                return False
        quick = self._quick_key(A) + (shard_key,)
        full = None
        if staged is None and self._graph_key == quick and self._graph_ref is not None and self._graph_ref() is A:
            staged = self._take_reorder(A)
            if staged is not None:
                pass
            elif pinned:
                return False
            elif defer == 'caller' and self.nranks == 1 and not self._has_comm:
                # the caller has the three arrays hashed itself (take_pending_graph: inside cna_assoc_finish, while the
                # device works); confirm_graph() still settles it if the caller never takes it
                self._pending_check = ('caller', A)
                return False
            elif defer and self.nranks == 1 and not self._has_comm:
                self._pending_check = _checker().submit(self._full_hash, A)
                return False
            else:
                full = self._full_hash(A)
                if full == self._graph_hash:
                    return False
        if staged is None:
            self._drop_reorder()
        key = quick
        if staged is not None:
            # the device cell order computed beside the first analyses of this graph (see below) has arrived: the
            # resident copy is renumbered ON THE DEVICE (the rows keep the order of their entries: same bits as an upload
            # of the renumbered graph, without the second trip over PCIe); column sums and sample codes move with it,
            # everything else per cell follows as after any upload
            perm, full = staged
            try:
                check(self.lib.cna_graph_reorder(self.h, ptr(perm)), 'cna_graph_reorder')
            except _ffi.CnaHipError:
                # the device order is an optimisation (a second copy of the graph plus temporaries: it can run out of
                # memory); the library leaves the resident copy in the caller's order intact -- that order stays
                return False
            self.perm = perm
            self._keep_dev = None
            self._kept_order_cache = None
            self._x_is_selection = False
            self._graph_key = key
            self._graph_hash = full
            self._pending_check = None
            return True
        elif shard is None:
            n = A.shape[0]
            r0, r1 = self.block(n)
            # A large graph that is new to the device goes up in the caller's order first: the cluster order takes
            # longer than the analysis (0.3 s against 0.02 s at 2M cells) and results do not depend on the device order
            # -- so it is computed on a host thread meanwhile and adopted by the first later call that finds it done
            # (CNA_REORDER_ASYNC=0: in line, as before)
            lazy = (self.nranks == 1 and not self._has_comm and n >= _REORDER_ASYNC_CELLS and _REORDER_ASYNC
                    and os.environ.get('CNA_REORDER', '1') not in ('0', 'off', 'no'))
            perm = None if lazy else _order.locality_order(A)
            if lazy:
                self._start_reorder(A)
            if perm is None:
                lo, hi = int(A.indptr[r0]), int(A.indptr[r1])
                indptr = np.ascontiguousarray(A.indptr[r0:r1 + 1].astype(np.int64) - lo)
                indices = np.ascontiguousarray(A.indices[lo:hi], dtype=np.int32)
                data = A.data[lo:hi]
            else:
                indptr, indices, data = _order.permuted_rows(A, perm, r0, r1)
            order = None if perm is None else perm[r0:r1]
        else:
            r0, n = int(shard[0]), int(shard[1])
            r1 = r0 + A.shape[0]
            if A.shape[1] != n or (r0, r1) != self.block(n):
                raise ValueError('rank %d of %d owns rows [%d, %d) of %d cells (engine.block); got rows [%d, %d) '
                                 'of a %d-column graph' % ((self.rank, self.nranks) + self.block(n) + (n, r0, r1, A.shape[1])))
            # locality order inside the block (RCM of its diagonal part); the ranks exchange their
            # orders once, so that each can relabel the columns that point into other blocks
            perm = _order.locality_order(A[:, r0:r1]) if r1 > r0 else None
            if perm is None:
                perm = np.arange(r1 - r0, dtype=np.int64)
            col_map = _order.inverse(self._allgather_i64(perm + r0))
            indptr, indices, data = _order.permuted_rows(A, perm, 0, r1 - r0, col_map=col_map)
            order = perm
        if data.dtype == np.float32:
            data, f64 = np.ascontiguousarray(data), 0
        else:
            data, f64 = np.ascontiguousarray(data, dtype=np.float64), 1
        check(self.lib.cna_set_local_view(self.h, int(shard is not None)), 'cna_set_local_view')
        check(self.lib.cna_graph_upload(self.h, n, r0, r1 - r0, ptr(indptr), ptr(indices), ptr(data), f64),
              'cna_graph_upload')
        if order is not None:
            check(self.lib.cna_set_cell_order(self.h, ptr(np.ascontiguousarray(order))), 'cna_set_cell_order')
        self.halo = None
        if self.nranks > 1 or (self._has_comm and os.environ.get('CNA_HALO_SELFTEST')):
            plan = _order.halo_plan(indices, r0, r1 - r0, -(-n // self.nranks), self.rank, self.nranks,
                                    self._allgather_i64,
                                    force_self=int(os.environ.get('CNA_HALO_SELFTEST', '0')))
            if plan is not None and os.environ.get('CNA_HALO', '1') not in ('0', 'off'):
                send_rows, send_counts, recv_rows, recv_counts = plan
                check(self.lib.cna_set_halo(self.h, ptr(send_rows), ptr(send_counts), ptr(recv_rows),
                                            ptr(recv_counts)), 'cna_set_halo')
                self.halo = (int(send_counts.sum()), int(recv_counts.sum()))
        self.perm = perm
        self.view_local = shard is not None
        self._keep_dev = None
        self._kept_order_cache = None
        self._x_is_selection = False
        self.n_global, self.row0, self.n_local = n, r0, r1 - r0
        self.n = self.n_local if self.view_local else n
        self._graph_key = key
        self._graph_hash = full if full is not None else self._full_hash(A)
        self._pending_check = None
        try:
            self._graph_ref = weakref.ref(A)
        except TypeError:
            self._graph_ref = None
        self._colsum_w = None
        self._codes_token = self._codes_graph = None
        return True

    def confirm_graph(self):
        """Outcome of the deferred content check of ensure_graph(defer=True): True = the resident graph is
        the caller's graph (or nothing was deferred).  False: the matrix was edited in place since it went
        to the device; the resident copy is dropped and the caller has to redo its work."""
        fut, self._pending_check = self._pending_check, None
        if fut is None:
            return True
        if isinstance(fut, tuple):
            if self._full_hash(fut[1]) == self._graph_hash:
                return True
        elif fut.result() == self._graph_hash:
            return True
        self.drop_graph()
        return False

    def take_pending_graph(self):
        """The deferred check of ensure_graph(defer='caller') as [(array, 64-bit hash its content must have)] x 3 for a
        caller that has them verified elsewhere (and calls drop_graph() when that fails); else None."""
        fut = self._pending_check
        if not isinstance(fut, tuple) or self._graph_hash is None:
            return None
        self._pending_check = None
        A = fut[1]
        return [(np.ascontiguousarray(arr), h) for arr, h in zip((A.data, A.indices, A.indptr), self._graph_hash)]

    def drop_graph(self):
        """Forget the resident graph (its content is no longer what the caller holds): the next call uploads afresh."""
        self._graph_key = None
        self._graph_hash = None
        self._nam_sig = None

    def colsums(self, self_weight=1):
        w = float(self_weight)
        if self._colsum_w != w:
            check(self.lib.cna_colsums(self.h, w), 'cna_colsums')
            self._colsum_w = w

    def fetch_colsums(self):
        out = np.empty(self.n_global)
        check(self.lib.cna_fetch_colsums(self.h, ptr(out)), 'cna_fetch_colsums')
        return self.cells_to_user(out[self.row0:self.row0 + self.n_local] if self.view_local else out)

    # ---------------------------------------------------------------- NAM
    def set_samples(self, codes, n_samples, counts, token=None):
        """Sample code per cell + cells per sample.  ``token``: an opaque, hashable identity of
        `codes` (see tools._nam.sample_codes_cached); when it equals the token of the codes already
        on the device for this graph, the upload is skipped and only the walk is restarted."""
        if token is not None and token == self._codes_token and self._codes_graph == self._graph_key:
            check(self.lib.cna_restart_nam(self.h), 'cna_restart_nam')
            self.nam_epoch += 1
            return
        if len(codes) != self.n:
            raise ValueError('need one sample code per cell')
        codes = self.cells_to_device(np.asarray(codes))
        if self.view_local:      # every rank needs the sample of every cell (first walk step): blocks in rank order
            codes = self._allgather_i64(codes)
        codes = np.ascontiguousarray(codes, dtype=np.int32)
        counts = _f64(counts)
        check(self.lib.cna_set_samples(self.h, ptr(codes), int(n_samples), ptr(counts)), 'cna_set_samples')
        self.N = int(n_samples)
        self.nam_epoch += 1
        self._codes_token, self._codes_graph = token, self._graph_key

    def nam_step(self, want_kurt, may_continue, may_stop):
        check(self.lib.cna_nam_step(self.h, int(bool(want_kurt)), int(bool(may_continue)), int(bool(may_stop))),
              'cna_nam_step')

    def nam_steps(self, nsteps):
        """nsteps walk steps in one call (fixed step count, nothing for the host to decide in between)."""
        check(self.lib.cna_nam_steps(self.h, int(nsteps)), 'cna_nam_steps')

    def nam_select_hint(self, y_std):
        """Tell the walk in progress which standardised phenotype select_standardized() will be called with (all
        cells, samples in place, nothing regressed out): its last step then does that pass on its way out."""
        yv = None if y_std is None else _f64(y_std)
        check(self.lib.cna_nam_select_hint(self.h, ptr(yv), 0 if yv is None else len(yv)), 'cna_nam_select_hint')
        if yv is not None:
            # the step that follows overwrites the working matrix: results of earlier calls that still read it lazily
            # (res.namresid, res.namresid_nbhdXpc) must see that now, not only once select_standardized() has run
            self.x_epoch += 1

    def nam_auto(self, maxnsteps=15):
        """The walk with the reference's stop rule (nsteps=None, _nam.py:64-68) in one call, medians and rule on
        the device; returns (steps taken, median kurtosis after every step)."""
        taken = C.c_int(0)
        med = np.empty(int(maxnsteps))
        check(self.lib.cna_nam_auto(self.h, int(maxnsteps), C.byref(taken), ptr(med)), 'cna_nam_auto')
        return taken.value, med[:taken.value]

    def nam_auto_launch(self, maxnsteps=15):
        """First half of nam_auto(): queue the walk and return at once.  The verdict is collected by
        nam_auto_finish() or by whatever engine call next reads the NAM."""
        check(self.lib.cna_nam_auto_launch(self.h, int(maxnsteps)), 'cna_nam_auto_launch')
        self._auto_max = int(maxnsteps)

    def nam_auto_finish(self):
        taken = C.c_int(0)
        med = np.empty(self._auto_max)
        check(self.lib.cna_nam_auto_finish(self.h, C.byref(taken), ptr(med)), 'cna_nam_auto_finish')
        return taken.value, med[:taken.value]

    def stat_median(self):
        """np.median of the per-cell statistic of the last kernel that made one, over all cells (or all
        kept cells, all ranks): exact radix select on the device, no cells-sized transfer."""
        m = C.c_double(0.0)
        check(self.lib.cna_stat_median(self.h, C.byref(m)), 'cna_stat_median')
        return m.value

    def stat_qc(self):
        """(median, threshold = max(6, 2 median), cells failing `kurtosis < threshold`) of the NAM's batch kurtosis
        (_nam.py:94-96), decided on the device; a count of zero means every cell is kept."""
        m, t, nd = C.c_double(0.0), C.c_double(0.0), C.c_int64(0)
        check(self.lib.cna_stat_qc(self.h, C.byref(m), C.byref(t), C.byref(nd)), 'cna_stat_qc')
        return m.value, t.value, nd.value

    def cell_stat(self, n_expected, nam_space=True):
        """Per-cell statistic of the last kernel that made one: over all cells in the caller's
        order (nam_space) or over the rows of X in device order (see x_stat())."""
        out = np.empty(int(n_expected))
        check(self.lib.cna_fetch_cell_stat(self.h, ptr(out), int(n_expected)), 'cna_fetch_cell_stat')
        return self.cells_to_user(out) if nam_space else out

    # ---------------------------------------------------------------- dense diffusion
    def dense_load(self, s_local):
        s_local = _f64(s_local)
        if s_local.ndim != 2 or s_local.shape[0] != self.n_local:
            raise ValueError('dense state must be (local cells) x m')
        check(self.lib.cna_dense_load(self.h, ptr(s_local), s_local.shape[1]), 'cna_dense_load')
        self._dense_m = s_local.shape[1]

    def dense_step(self):
        check(self.lib.cna_dense_step(self.h), 'cna_dense_step')

    def dense_fetch(self):
        out = np.empty((self.n_local, self._dense_m))
        check(self.lib.cna_dense_fetch(self.h, ptr(out)), 'cna_dense_fetch')
        return out

    # ---------------------------------------------------------------- QC / selection
    def batch_kurtosis(self, which, batch_codes, n_batches):
        bc = np.ascontiguousarray(batch_codes, dtype=np.int32)
        check(self.lib.cna_batch_kurtosis(self.h, which, ptr(bc), int(n_batches)), 'cna_batch_kurtosis')

    def zero_variance(self, colmap):
        cm = None if colmap is None else np.ascontiguousarray(colmap, dtype=np.int32)
        flags = np.zeros(self.n, dtype=np.uint8)
        nz = C.c_int64(0)
        check(self.lib.cna_zero_variance(self.h, ptr(cm), 0 if cm is None else len(cm), ptr(flags), C.byref(nz)),
              'cna_zero_variance')
        return self.cells_to_user(flags.astype(bool)), nz.value

    def _selection(self, keep_global):
        self._x_is_selection = True
        if keep_global is None:
            self._keep_dev = None
            self._kept_order_cache = None
            self.x_rows_total = self.n
            return None, 0
        idx = self.local_keep(keep_global)
        self.x_rows_total = int(np.count_nonzero(self._keep_dev))
        return idx, len(idx)

    def select(self, keep_global, colmap):
        """keep_global: bool mask over all cells (or None = all); colmap: NAM column per output column."""
        cm = None if colmap is None else np.ascontiguousarray(colmap, dtype=np.int32)
        idx, nk = self._selection(keep_global)
        check(self.lib.cna_select(self.h, ptr(idx), nk, ptr(cm), 0 if cm is None else len(cm)), 'cna_select')
        self.x_epoch += 1

    def set_resid_factors(self, Cmat, W):
        """The next select_standardized() also residualises: M = I - Cmat.W applied between centring and
        the division by the std (one pass over the NAM for selection + residualisation + standardisation)."""
        Cmat, W = _f64(Cmat), _f64(W)
        check(self.lib.cna_set_resid_factors(self.h, ptr(Cmat), ptr(W), int(Cmat.shape[1]), int(Cmat.shape[0])),
              'cna_set_resid_factors')

    def clear_resid_factors(self):
        check(self.lib.cna_set_resid_factors(self.h, None, None, 0, max(int(self.N), 1)), 'cna_set_resid_factors')

    def select_checked(self, keep_global, colmap):
        """select() and, in the same pass, the number of selected cells with zero variance over the
        selected samples (non-zero: redo with zero_variance() + select())."""
        cm = None if colmap is None else np.ascontiguousarray(colmap, dtype=np.int32)
        idx, nk = self._selection(keep_global)
        nz = C.c_int64(0)
        check(self.lib.cna_select_checked(self.h, ptr(idx), nk, ptr(cm), 0 if cm is None else len(cm), C.byref(nz)),
              'cna_select_checked')
        self.x_epoch += 1
        return nz.value

    def select_standardized(self, keep_global, colmap, y=None, fuse_null=0, null_ready=None):
        """select() + centre + divide by std in one pass (M = I); returns the number of selected
        cells with zero variance (non-zero: redo with zero_variance()/select()).  With ``y`` (the
        standardised phenotype in the selected samples' order) the neighbourhood coefficients are
        taken in the same pass and (n_zero, max |ncorrs|) is returned -- ncorrs(y) is then done.
        ``fuse_null`` = P > 0 (with y): when no cell has zero variance the same call also queues what
        would follow from values it returns -- gram_launch(), the thresholds of the local null from
        max|ncorrs|, null_local_prepare(P, ...) and percell_coef_launch(); those methods then find
        their work done (they compare arguments) and return at once.  ``null_ready``: a callable asked right before
        the call whether the conditioned phenotypes of this analysis are on the device already (condition() has
        returned); the prepared local-null pass is then launched in the same call (columns 1 .. P)."""
        cm = None if colmap is None else np.ascontiguousarray(colmap, dtype=np.int32)
        idx, nk = self._selection(keep_global)
        nz = C.c_int64(0)
        m = C.c_double(0.0)
        yv = None if y is None else _f64(y)
        self.null_local_discard()       # (a pass left behind by an analysis that raised after its launch)
        if yv is not None and fuse_null > 0:
            T, gq, cq, nl = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
            thr = np.empty(512)
            # null_ready: a callable (the caller's own word, now) or an int32 array of one element that the library's
            # draw thread sets once its conditioning has returned (read by the C call at the moment it matters)
            flag = null_ready if isinstance(null_ready, np.ndarray) else None
            col0 = 1 if (flag is not None or (null_ready is not None and null_ready())) else -1
            check(self.lib.cna_select_standardized_fused(self.h, ptr(idx), nk, ptr(cm), 0 if cm is None else len(cm),
                                                         C.byref(nz), ptr(yv), C.byref(m), int(fuse_null), C.byref(T),
                                                         ptr(thr), C.byref(gq), C.byref(cq), col0, ptr(flag), C.byref(nl)),
                  'cna_select_standardized_fused')
            self.x_epoch += 1
            self._fused = dict(epoch=self.x_epoch, P=int(fuse_null), thr=thr[:T.value], gram=bool(gq.value),
                               coef=bool(cq.value), prepared=T.value > 0 and not nl.value, null=bool(nl.value))
            if gq.value:
                self._gram_cols = self.N if cm is None else len(cm)
            if T.value:
                self._null_T, self._null_obs = T.value, True
            return nz.value, m.value
        check(self.lib.cna_select_standardized(self.h, ptr(idx), nk, ptr(cm), 0 if cm is None else len(cm),
                                               C.byref(nz), ptr(yv), C.byref(m)), 'cna_select_standardized')
        self.x_epoch += 1
        return nz.value if y is None else (nz.value, m.value)

    def _fused_done(self, what):
        """True once: the fused selection call already issued `what` for the current working matrix."""
        f = getattr(self, '_fused', None)
        if f is None or f['epoch'] != self.x_epoch or not f.get(what):
            return False
        f[what] = False
        return True

    def upload_x(self, x_local):
        x_local = _f64(x_local)
        check(self.lib.cna_upload_x(self.h, ptr(x_local), x_local.shape[0], x_local.shape[1]), 'cna_upload_x')
        self.x_rows_total = x_local.shape[0]   # single-rank use (cna.tl.svd_nam)
        self._x_is_selection = False
        self.x_epoch += 1

    # ---------------------------------------------------------------- residualise + PCA
    def resid_apply(self, M, center):
        M = None if M is None else _f64(M)
        check(self.lib.cna_resid_apply(self.h, ptr(M), int(bool(center))), 'cna_resid_apply')

    def resid_lowrank(self, Cmat, W, center=True, standardize=False, y=None):
        """X <- (X - mean).M^T [/ std] for M = I - Cmat.W (Cmat: N x r, W: r x N) in one row-local pass; with
        `y` also the neighbourhood coefficients, returning max |ncorrs| (else None)."""
        Cmat, W = _f64(Cmat), _f64(W)
        r = Cmat.shape[1]
        if W.shape != (r, Cmat.shape[0]):
            raise ValueError('W must be r x N for an N x r C')
        yv = None if y is None else _f64(y)
        m = C.c_double(0.0)
        check(self.lib.cna_resid_lowrank(self.h, ptr(Cmat), ptr(W), int(r), int(bool(center)), int(bool(standardize)),
                                         ptr(yv), C.byref(m)), 'cna_resid_lowrank')
        return m.value if y is not None else None

    def resid_lowrank_bk(self, Cmat, W, y, batch_codes, n_batches):
        """One ridge in one pass: X <- (X - mean).M^T for M = I - Cmat.W, its batch kurtosis and the median of that
        (on the device), then / std and the coefficients X.y/N.  Returns (max |ncorrs|, median batch kurtosis); when
        the median is > 6 the caller must restore X before going on with the next ridge."""
        Cmat, W, yv = _f64(Cmat), _f64(W), _f64(y)
        r = Cmat.shape[1]
        bc = np.ascontiguousarray(batch_codes, dtype=np.int32)
        m, med = C.c_double(0.0), C.c_double(0.0)
        check(self.lib.cna_resid_lowrank_bk(self.h, ptr(Cmat), ptr(W), int(r), ptr(yv), C.byref(m), ptr(bc), int(n_batches),
                                            C.byref(med)), 'cna_resid_lowrank_bk')
        return m.value, med.value

    def select_resid_bk(self, Cmat, W, y, batch_codes, n_batches):
        """Selection (every cell, samples in place) + QC + the first ridge in ONE pass over the NAM (cna_select_resid_bk).
        -> None (shape not covered, nothing queued) or (rows failing the QC, rows of zero variance, max |ncorrs|, median
        batch kurtosis of the residualised rows); X is final when the first two are 0 and the median is <= 6."""
        Cmat, W, yv = _f64(Cmat), _f64(W), _f64(y)
        bc = np.ascontiguousarray(batch_codes, dtype=np.int32)
        if Cmat.ndim != 2 or W.shape != (Cmat.shape[1], Cmat.shape[0]) or len(yv) != Cmat.shape[0] or len(bc) != len(yv):
            return None
        m, med = C.c_double(0.0), C.c_double(0.0)
        nq, nz, done = C.c_int64(0), C.c_int64(0), C.c_int(0)
        self.null_local_discard()
        check(self.lib.cna_select_resid_bk(self.h, ptr(Cmat), ptr(W), int(Cmat.shape[1]), ptr(yv), C.byref(m), ptr(bc), int(n_batches),
                                           C.byref(med), C.byref(nq), C.byref(nz), C.byref(done)), 'cna_select_resid_bk')
        if not done.value:
            return None
        self._selection(None)
        self.x_epoch += 1
        return nq.value, nz.value, m.value, med.value

    def standardize(self, center=False):
        check(self.lib.cna_standardize(self.h, int(bool(center))), 'cna_standardize')

    def gram(self):
        rows, cols = self.matrix_shape(MAT_X)
        G = np.empty((cols, cols))
        check(self.lib.cna_gram(self.h, ptr(G)), 'cna_gram')
        return G

    def gram_launch(self):
        if self._fused_done('gram'):
            return
        check(self.lib.cna_gram_launch(self.h), 'cna_gram_launch')
        self._gram_cols = self.matrix_shape(MAT_X)[1]

    def gram_fetch(self):
        G = np.empty((self._gram_cols, self._gram_cols))
        check(self.lib.cna_gram_fetch(self.h, ptr(G)), 'cna_gram_fetch')
        return G

    def gram_pcs_tests(self, ks, r, native=True, resid_tol=1e-12, gap_tol=1e-6):
        """gram_fetch(), the leading max(ks) eigenpairs on the host and -- when the library's own solver is accepted
        (tools/_nam.py:_top_pcs_native's rule) -- global_test_launch(), in ONE call without the interpreter in between.
        Returns (G, U or None, queued): queued -> collect with global_test_fetch(); else the caller finds the
        eigenvectors elsewhere and launches the tests itself."""
        ks = np.ascontiguousarray(ks, dtype=np.int32)
        kmax = int(ks.max())
        n = self._gram_cols
        G = np.empty((n, n))
        U = np.empty((n, max(kmax, 1)))
        ok = C.c_int(0)
        check(self.lib.cna_gram_pcs_tests(self.h, kmax, ptr(ks), len(ks), int(r), int(bool(native)), float(resid_tol), float(gap_tol),
                                          ptr(G), ptr(U), C.byref(ok)), 'cna_gram_pcs_tests')
        return G, (U if ok.value else None), bool(ok.value)

    def project(self, W):
        W = _f64(W)
        rows, cols = self.matrix_shape(MAT_X)
        out = np.empty((rows, W.shape[1]))
        check(self.lib.cna_project(self.h, ptr(W), W.shape[1], ptr(out)), 'cna_project')
        return out

    # ---------------------------------------------------------------- association
    def x_identity_resident(self):
        """True when X on the device is the standardised NAM of the resident walk for "every cell, samples in place,
        nothing regressed out": a further analysis of the dataset with that selection keeps it (ncorrs(y) next)."""
        yes = C.c_int(0)
        check(self.lib.cna_x_identity(self.h, C.byref(yes)), 'cna_x_identity')
        return bool(yes.value)

    def ncorrs(self, y, fetch=False):
        y = _f64(y)
        rows, cols = self.matrix_shape(MAT_X)
        out = np.empty(rows) if fetch else None
        m = C.c_double(0.0)
        check(self.lib.cna_ncorrs(self.h, ptr(y), ptr(out), C.byref(m)), 'cna_ncorrs')
        if fetch and (self.nranks == 1 or self.view_local):
            out = self.kept_to_user(out)     # kept cells (of the caller's view) in the caller's order
        return out, m.value

    def null_local(self, Yc, edges):
        Yc = _f64(Yc)
        edges = _f64(edges)
        P, T = Yc.shape[1], len(edges)
        tails = np.empty((P, T), dtype=np.int64)
        check(self.lib.cna_null_local(self.h, ptr(Yc), P, ptr(edges), T, ptr(tails)), 'cna_null_local')
        return tails

    def condition(self, M, Y):
        """Zc = M.Y / std(M.Y, ddof=1) per column, kept on the device (column 0: observed phenotype).
        Sample space only and on the second stream: may be issued while the diffusion is running."""
        M, Y = _f64(M), _f64(Y)
        if M.shape != (Y.shape[0], Y.shape[0]):
            raise ValueError('M must be square with one row per phenotype entry')
        check(self.lib.cna_condition_phenotypes(self.h, ptr(M), ptr(Y), Y.shape[0], Y.shape[1]),
              'cna_condition_phenotypes')
        self._zc_cols = Y.shape[1]

    def null_local_resident(self, col0, P, edges, sums_only=False):
        """Tail counts of the local null on resident columns: P x T matrix, or (sums_only) its sum
        over permutations, which is all the FDR formula needs."""
        edges = _f64(edges)
        T = len(edges)
        tails = None if sums_only else np.empty((int(P), T), dtype=np.int64)
        sums = np.empty(T, dtype=np.int64) if sums_only else None
        check(self.lib.cna_null_local_resident(self.h, int(col0), int(P), ptr(edges), T, ptr(tails), ptr(sums)),
              'cna_null_local_resident')
        return sums if sums_only else tails

    def null_local_i8_stats(self):
        """(integer path used, outputs rechecked in f64, fell back to the f64 kernel) of the last sums-only pass."""
        used, fb = C.c_int(0), C.c_int(0)
        rc = C.c_int64(0)
        check(self.lib.cna_null_local_i8_stats(self.h, C.byref(used), C.byref(rc), C.byref(fb)), 'cna_null_local_i8_stats')
        return bool(used.value), int(rc.value), bool(fb.value)

    def null_local_prepare(self, P, edges, thr=None):
        """First half of null_local_launch: needs the thresholds only (exact cuts, observed counts);
        finish with null_local_launch(col0, P, None)."""
        edges = _f64(edges)
        thr = None if thr is None else _f64(thr)
        f = getattr(self, '_fused', None)
        if (thr is not None and f is not None and f['epoch'] == self.x_epoch and f['prepared'] and f['P'] == int(P)
                and np.array_equal(f['thr'], thr)):
            f['prepared'] = False             # the fused selection call prepared exactly this pass
            return
        if f is not None and f['epoch'] == self.x_epoch and f.get('null'):
            if thr is not None and f['P'] == int(P) and np.array_equal(f['thr'], thr):
                return                        # ... and launched it as well (null_local_launch finds that out)
            # the fused call's own thresholds are not the caller's (never seen; numpy's arange and its C restatement
            # are compared in the tests): collect that pass and start over
            f['null'] = False
            self._null_T, self._null_obs = len(f['thr']), True
            self.null_local_fetch()
        if f is not None:
            f['coef'] = False                 # a fresh prepare: whatever rode along with the fused one is void
        check(self.lib.cna_null_local_prepare(self.h, int(P), ptr(edges), len(edges), 0, ptr(thr)),
              'cna_null_local_prepare')
        self._null_T = len(edges)
        self._null_obs = thr is not None

    def null_local_launch(self, col0, P, edges, thr=None):
        """Queue a local-null pass on resident columns; collect with null_local_fetch().  With `thr`
        the threshold counts of the observed coefficients (obs_counts) are queued in front of it."""
        if edges is None:                       # prepared pass
            if int(col0) == 1 and self._fused_done('null'):
                return                          # the fused selection call launched it already
            check(self.lib.cna_null_local_launch(self.h, int(col0), int(P), None, self._null_T, 0, None),
                  'cna_null_local_launch')
            return
        edges = _f64(edges)
        thr = None if thr is None else _f64(thr)
        check(self.lib.cna_null_local_launch(self.h, int(col0), int(P), ptr(edges), len(edges), 0, ptr(thr)),
              'cna_null_local_launch')
        self._null_T = len(edges)
        self._null_obs = thr is not None

    def null_local_fetch(self):
        """Tail sums of the pending pass, or (tail sums, ranks, num_detected) when it was launched with thr."""
        T = self._null_T
        sums = np.empty(T, dtype=np.int64)
        if not self._null_obs:
            check(self.lib.cna_null_local_fetch(self.h, None, ptr(sums), None, None), 'cna_null_local_fetch')
            return sums
        ranks, numdet = np.empty(T, dtype=np.int64), np.empty(T, dtype=np.int64)
        check(self.lib.cna_null_local_fetch(self.h, None, ptr(sums), ptr(ranks), ptr(numdet)), 'cna_null_local_fetch')
        return sums, ranks, numdet

    def null_local_discard(self):
        """Drop a local-null pass that was launched (here or by the fused selection call) and never fetched -- the
        caller raised in between -- and whatever else the fused call queued for an analysis that will not happen.
        No-op when nothing is pending; called on the error paths of association() and before every new selection."""
        self._fused = None
        check(self.lib.cna_null_local_discard(self.h), 'cna_null_local_discard')

    def global_test(self, U, ks, r):
        """min-p F-test of every resident phenotype column -> (index into ks, p, r2) arrays."""
        ks = np.ascontiguousarray(ks, dtype=np.int32)
        kmax = int(ks.max())
        Uk = _f64(U[:, :kmax])
        P = self._zc_cols
        minp, r2, kidx = np.empty(P), np.empty(P), np.empty(P, dtype=np.int32)
        check(self.lib.cna_global_test(self.h, ptr(Uk), kmax, ptr(ks), len(ks), int(r), ptr(minp), ptr(r2), ptr(kidx)),
              'cna_global_test')
        return kidx, minp, r2

    def global_test_launch(self, U, ks, r):
        """First half of global_test(): queue it and return; collect with global_test_fetch()."""
        ks = np.ascontiguousarray(ks, dtype=np.int32)
        kmax = int(ks.max())
        Uk = _f64(U[:, :kmax])
        check(self.lib.cna_global_test_launch(self.h, ptr(Uk), kmax, ptr(ks), len(ks), int(r)), 'cna_global_test_launch')

    def global_test_fetch(self):
        P = self._zc_cols
        minp, r2, kidx = np.empty(P), np.empty(P), np.empty(P, dtype=np.int32)
        check(self.lib.cna_global_test_fetch(self.h, ptr(minp), ptr(r2), ptr(kidx)), 'cna_global_test_fetch')
        return kidx, minp, r2

    def global_test_discard(self):
        """Collect and drop a queued global test nobody will fetch (error paths); no-op when none is pending."""
        try:
            self.global_test_fetch()
        except _ffi.CnaHipError:
            pass

    # ---------------------------------------------------------------- the fixed-shape analysis in two calls
    def assoc_begin(self, nsteps, y_hint=None):
        """Queue the walk of a fixed-shape analysis (cna_assoc_begin): `nsteps` steps (0: the NAM on the device is
        kept), with `y_hint` the last step also does the selection pass (nam_select_hint).  Returns at once."""
        yv = None if y_hint is None else _f64(y_hint)
        check(self.lib.cna_assoc_begin(self.h, int(nsteps), ptr(yv), 0 if yv is None else len(yv)), 'cna_assoc_begin')
        if yv is not None and nsteps >= 2:
            self.x_epoch += 1             # (that step overwrites the working matrix: see nam_select_hint)

    def assoc_begin_part(self, first, count, total, y_hint=None):
        """Steps first .. first + count - 1 (0-based) of a walk of `total` steps (cna_assoc_begin_part); y_hint goes with
        the part that holds the last step."""
        yv = None if y_hint is None else _f64(y_hint)
        check(self.lib.cna_assoc_begin_part(self.h, int(first), int(count), int(total), ptr(yv), 0 if yv is None else len(yv)),
              'cna_assoc_begin_part')
        if yv is not None and total >= 2 and first + count == total:
            self.x_epoch += 1

    def assoc_finish(self, y, M, ks, Nnull, table, colmap=None, Cmat=None, W=None, draw_pending=False, conditioned=False,
                     coef_dst=None, fdr_dst=None, coef_first=False, copy_threads=1, native_eig=True, resid_tol=1e-12,
                     gap_tol=1e-6, run_steps=None, y_hint=None, verify=(), verify_threads=1):
        """Selection (+ projector) -> Gram -> eigenpairs -> F-tests, conditioned phenotypes -> local null -> FDR table ->
        per-cell columns, in ONE blocking library call (cna_assoc_finish; include/cna_hip.h).  Returns a dict: status,
        n_zero, maxabs, thr, tail_sums, ranks, num_detected, fdr, runmin, G, U (None unless accepted), minp / r2 / kidx
        (None when status is ASSOC_NEED_PCS), coef / fdr (views of pinned buffers, or the caller's own storage).
        run_steps = nsteps: the walk is queued by the same call (cna_assoc_run = assoc_begin(nsteps, y_hint) + this)."""
        a = _ffi.AssocArgs()
        keep = []                                     # arrays the struct points to

        def P(arr):
            keep.append(arr)
            return arr.ctypes.data
        yv, Mv = _f64(y), _f64(M)
        N = len(yv)
        ksv = np.ascontiguousarray(ks, dtype=np.int32)
        kmax = int(ksv.max())
        P1 = int(Nnull) + 1
        if Mv.shape != (N, N) or table.shape != (N, P1) or table.dtype != np.float64 or not table.flags.c_contiguous:
            raise ValueError('assoc_finish: M must be N x N and table a C-contiguous float64 N x (Nnull + 1) matrix')
        a.colmap = None if colmap is None else P(np.ascontiguousarray(colmap, dtype=np.int32))
        a.n_sel, a.y, a.M = N, P(yv), P(Mv)
        if Cmat is not None and W is not None and np.shape(Cmat)[1] > 0:
            Cv, Wv = _f64(Cmat), _f64(W)
            if Cv.shape[0] != N or Wv.shape != (Cv.shape[1], N):
                raise ValueError('assoc_finish: C must be N x r and W r x N')
            a.r, a.resid_C, a.resid_W = Cv.shape[1], P(Cv), P(Wv)
        a.ks, a.K, a.Nnull = P(ksv), len(ksv), int(Nnull)
        a.table = P(table)
        a.draw_pending, a.conditioned = int(bool(draw_pending)), int(bool(conditioned))
        a.use_native_eig, a.coef_first = int(bool(native_eig)), int(bool(coef_first))
        a.resid_tol, a.gap_tol = float(resid_tol), float(gap_tol)
        if coef_dst is not None and fdr_dst is not None:
            a.coef_dst, a.fdr_dst, a.n_dst = coef_dst.ctypes.data, fdr_dst.ctypes.data, len(coef_dst)
            keep += [coef_dst, fdr_dst]
        a.copy_threads = int(copy_threads)
        # verify: [(array, 64-bit content hash it must still have)], checked by the library while the device works
        a.n_verify = len(verify)
        for i, (arr, hv) in enumerate(verify):
            a.verify_ptr[i], a.verify_bytes[i], a.verify_hash[i] = arr.ctypes.data, arr.nbytes, int(hv)
            keep.append(arr)
        a.verify_threads = int(verify_threads)
        G, U = np.empty((N, N)), np.empty((N, max(kmax, 1)))
        minp, r2, kidx = np.empty(P1), np.empty(P1), np.empty(P1, dtype=np.int32)
        a.G, a.U, a.minp, a.r2, a.kidx = P(G), P(U), P(minp), P(r2), P(kidx)
        if self._assoc_out is None:
            o = _ffi.AssocOut()
            raw = np.frombuffer(o, dtype=np.uint8)    # numpy views of the struct's arrays (a ctypes slice makes a list)
            views = {f: raw[getattr(_ffi.AssocOut, f).offset:][:8 * _ffi.ASSOC_MAXT].view(np.float64 if f in ('thr', 'fdr', 'runmin') else np.int64)
                     for f in ('thr', 'fdr', 'runmin', 'tail_sums', 'ranks', 'num_detected')}
            self._assoc_out = (o, views)
        o, ov = self._assoc_out
        # bookkeeping of what the call replaces on the device, before it can fail half-way
        self._fused = None
        self._selection(None)
        self.x_epoch += 1
        self._gram_cols = N
        self._zc_cols = P1
        if run_steps is None:
            check(self.lib.cna_assoc_finish(self.h, C.byref(a), C.byref(o)), 'cna_assoc_finish')
        else:
            hv = None if y_hint is None else _f64(y_hint)
            check(self.lib.cna_assoc_run(self.h, int(run_steps), ptr(hv), 0 if hv is None else len(hv), C.byref(a), C.byref(o)),
                  'cna_assoc_run')
        T = int(o.T)
        self._null_T, self._null_obs = T, True
        status = o.status
        out = dict(status=status, n_zero=o.n_zero, maxabs=o.max_abs, T=T, G=G, null_fused=o.null_fused,
                   U=U if o.eig_accepted else None, coef_in_dst=o.coef_in_dst, fdr_in_dst=o.fdr_in_dst, t_ms=o.t_ms)
        if status in (_ffi.ASSOC_GENERAL, _ffi.ASSOC_STALE):
            return out
        # (views of the call's own output block, valid until the next assoc_finish: a caller that keeps them copies them)
        for f, v in ov.items():
            out[f] = v[:T]
        if status == _ffi.ASSOC_DONE:
            out['minp'], out['r2'], out['kidx'] = minp, r2, kidx
        out['coef'] = coef_dst if o.coef_in_dst else (self._pinned_view(C.c_void_p(o.coef_ptr)) if o.coef_ptr else None)
        out['fdr_col'] = fdr_dst if o.fdr_in_dst else (self._pinned_view(C.c_void_p(o.fdr_ptr)) if o.fdr_ptr else None)
        return out

    @property
    def last_assoc_t_ms(self):
        """Stage times of the last cna_assoc_finish (ms from its entry; include/cna_hip.h: cna_assoc_out.t_ms), or None."""
        return None if self._assoc_out is None else list(self._assoc_out[0].t_ms)

    def obs_counts(self, edges, thr):
        edges, thr = _f64(edges), _f64(thr)
        T = len(thr)
        ranks = np.empty(T, dtype=np.int64)
        numdet = np.empty(T, dtype=np.int64)
        check(self.lib.cna_obs_counts(self.h, ptr(edges), ptr(thr), T, ptr(ranks), ptr(numdet)), 'cna_obs_counts')
        return ranks, numdet

    def percell(self, thr=None, runmin=None):
        """Per-cell coefficient (NaN for dropped cells) and FDR columns over all cells, caller's order.
        The arrays are views of pinned buffers owned by the engine: valid until the next percell()
        -- copy them (assigning to a DataFrame column does) before running another analysis."""
        cp, fp = C.c_void_p(), C.c_void_p()
        if thr is None:
            check(self.lib.cna_percell_fdr_pinned(self.h, None, None, 0, C.byref(cp), None), 'cna_percell_fdr_pinned')
            return self._pinned_view(cp), None
        thr, runmin = _f64(thr), _f64(runmin)
        check(self.lib.cna_percell_fdr_pinned(self.h, ptr(thr), ptr(runmin), len(thr), C.byref(cp), C.byref(fp)),
              'cna_percell_fdr_pinned')
        return self._pinned_view(cp), self._pinned_view(fp)

    def percell_coef_launch(self):
        """Queue the per-cell coefficient column ahead of the local-null kernel (it only needs the
        observed phenotype); False when this engine assembles per-cell outputs across ranks."""
        if (self.nranks > 1 or self._has_comm) and not self.view_local:
            return False
        if self._fused_done('coef'):
            return True
        check(self.lib.cna_percell_coef_launch(self.h), 'cna_percell_coef_launch')
        return True

    def percell_coef_wait(self):
        """The coefficient column queued by percell_coef_launch (pinned view, see percell()); callable
        from a helper thread."""
        cp = C.c_void_p()
        check(self.lib.cna_percell_coef_wait(self.h, C.byref(cp)), 'cna_percell_coef_wait')
        return self._pinned_view(cp)

    def percell_fdr_copy_early(self, dst, threads=4):
        """Copy the FDR column of the pending local-null pass into `dst` (contiguous float64, all cells) once the
        device has delivered it; for a helper thread.  False: not applicable, nothing copied."""
        done = C.c_int(0)
        check(self.lib.cna_percell_fdr_copy_early(self.h, dst.ctypes.data, dst.shape[0], int(threads), C.byref(done)),
              'cna_percell_fdr_copy_early')
        return bool(done.value)

    def percell_fdr_copied_early(self):
        """True when the FDR column percell() last returned is the one percell_fdr_copy_early() copied."""
        yes = C.c_int(0)
        check(self.lib.cna_percell_fdr_copied_early(self.h, C.byref(yes)), 'cna_percell_fdr_copied_early')
        return bool(yes.value)

    def _pinned_view(self, p):
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape=(self.n,))

    # ---------------------------------------------------------------- D2H
    def matrix_shape(self, which):
        r, c = C.c_int64(0), C.c_int(0)
        check(self.lib.cna_matrix_shape(self.h, which, C.byref(r), C.byref(c)), 'cna_matrix_shape')
        return r.value, c.value

    def fetch_matrix(self, which, transposed=False):
        rows, cols = self.matrix_shape(which)
        out = np.empty((cols, rows) if transposed else (rows, cols))
        check(self.lib.cna_fetch_matrix(self.h, which, ptr(out), int(bool(transposed))), 'cna_fetch_matrix')
        return out

    def fetch_rows(self, which, rows=None, cols=None, transposed=False):
        """out[i, j] = M[rows[i], cols[j]] (None = all, in order) picked and ordered on the device;
        transposed=True returns the (columns x rows) layout.  which: MAT_NAM, MAT_X or MAT_PROJ."""
        if which == _ffi.MAT_PROJ:
            have, width = self._proj_shape
        else:
            have, width = self.matrix_shape(which)
        r = None if rows is None else np.ascontiguousarray(rows, dtype=np.int64)
        cm = None if cols is None else np.ascontiguousarray(cols, dtype=np.int32)
        n_out = have if r is None else len(r)
        n_cols = width if cm is None else len(cm)
        out = np.empty((n_cols, n_out) if transposed else (n_out, n_cols))
        check(self.lib.cna_fetch_rows(self.h, which, ptr(r), n_out, ptr(cm), n_cols, ptr(out), int(bool(transposed))),
              'cna_fetch_rows')
        return out

    def project_keep(self, W):
        """X.W left on the device; read it with fetch_rows(MAT_PROJ, ...)."""
        W = _f64(W)
        check(self.lib.cna_project_keep(self.h, ptr(W), W.shape[1]), 'cna_project_keep')
        self._proj_shape = (self.matrix_shape(MAT_X)[0], W.shape[1])

    def x_rows_global(self):
        """Rows of the working matrix over all ranks (a collective in the local view)."""
        if not self.view_local or self.nranks == 1:
            return self.x_rows_total
        return int(self.allgather_fixed([self.x_rows_total]).sum())

    def allgather_fixed(self, values):
        """(nranks x len(values)) int64: the same-length vector of every rank, one collective."""
        a = np.ascontiguousarray(values, dtype=np.int64)
        if self.nranks == 1:
            return a.reshape(1, -1).copy()
        out = np.empty(self.nranks * len(a), dtype=np.int64)
        check(self.lib.cna_allgather_host(self.h, ptr(a), len(a), ptr(out), len(out)), 'cna_allgather_host')
        return out.reshape(self.nranks, len(a))

    def allgather_objects(self, obj):
        """[obj of rank 0, obj of rank 1, ...] for small picklable host objects (sample labels)."""
        import pickle
        if self.nranks == 1:
            return [obj]
        blob = pickle.dumps(obj, protocol=4)
        pad = (-len(blob)) % 8
        words = np.frombuffer(blob + b'\0' * pad, dtype=np.int64)
        sizes = self.allgather_fixed([len(blob), len(words)])
        flat = self._allgather_i64(words)
        out, off = [], 0
        for nbytes, nwords in sizes:
            out.append(pickle.loads(flat[off:off + nwords].tobytes()[:nbytes]))
            off += nwords
        return out

    def gather_rows_host(self, local, n_total):
        """Row blocks of every rank, concatenated in rank order (no-op on one GPU)."""
        local = _f64(local)
        if self.nranks == 1:
            return local
        cols = local.shape[1] if local.ndim == 2 else 1
        out = np.empty((int(n_total), cols) if local.ndim == 2 else int(n_total))
        check(self.lib.cna_allgather_host(self.h, ptr(local), local.size, ptr(out), out.size), 'cna_allgather_host')
        return out

    def _allgather_i64(self, a):
        """Concatenation over ranks of int64 vectors of differing lengths (graph preparation)."""
        a = np.ascontiguousarray(a, dtype=np.int64)
        if self.nranks == 1:
            return a.copy()
        sizes = np.zeros(self.nranks)
        check(self.lib.cna_allgather_host(self.h, ptr(np.array([float(len(a))])), 1, ptr(sizes), self.nranks),
              'cna_allgather_host')
        out = np.empty(int(sizes.sum()), dtype=np.int64)
        check(self.lib.cna_allgather_host(self.h, ptr(a), len(a), ptr(out), len(out)), 'cna_allgather_host')
        return out

    # ---------------------------------------------------------------- synthetic inputs
    def knn_graph(self, X, k):
        """scanpy-like connectivities of the points X (n x d, d <= 64) built on the device
        (cna_knn_graph): CSR float32 / int32, sorted indices.  Input generator for benchmarks."""
        X = np.ascontiguousarray(X, dtype=np.float32)
        n, d = X.shape
        cap = 2 * n * (k - 1)
        indptr = np.empty(n + 1, dtype=np.int64)
        indices = np.empty(cap, dtype=np.int32)
        data = np.empty(cap, dtype=np.float32)
        nnz = C.c_int64(0)
        check(self.lib.cna_knn_graph(self.h, ptr(X), n, d, int(k), ptr(indptr), ptr(indices), ptr(data), C.byref(nnz)),
              'cna_knn_graph')
        m = nnz.value
        A = sp.csr_matrix((data[:m].copy(), indices[:m].copy(), indptr.astype(np.int32 if m < 2 ** 31 else np.int64)),
                          shape=(n, n))
        A.has_sorted_indices = True
        return A

    # ---------------------------------------------------------------- measurement
    def prof_enable(self, on=True, walk_only=False):
        """HIP-event timing of the kernel groups (include/cna_hip.h: cna_prof_enable); walk_only: the walk kernels and the
        communication spans only."""
        check(self.lib.cna_prof_enable(self.h, (2 if walk_only else 1) if on else 0), 'cna_prof_enable')

    def prof_reset(self):
        check(self.lib.cna_prof_reset(self.h), 'cna_prof_reset')

    def prof(self):
        """{kernel name: (total ms, launches)} for kernels launched while profiling was on."""
        out = {}
        for k, name in enumerate(_ffi.KERNELS):
            ms, n = C.c_double(0.0), C.c_int64(0)
            check(self.lib.cna_prof_get(self.h, k, C.byref(ms), C.byref(n)), 'cna_prof_get')
            if n.value:
                out[name] = (ms.value, n.value)
        return out


_check_pool = None


_REORDER_ASYNC = os.environ.get('CNA_REORDER_ASYNC', '1') not in ('0', 'off', 'no')
_REORDER_ASYNC_CELLS = 100000
_reorder_workers = None


def _reorder_pool():
    global _reorder_workers
    if _reorder_workers is None:
        from concurrent.futures import ThreadPoolExecutor
        _reorder_workers = ThreadPoolExecutor(max_workers=1, thread_name_prefix='cna-order')
    return _reorder_workers


def _checker():
    """One thread for deferred graph checks (the hash itself runs on the library's own threads)."""
    global _check_pool
    if _check_pool is None:
        from concurrent.futures import ThreadPoolExecutor
        _check_pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix='cna-graph-check')
    return _check_pool


def _forget_pools():
    """In a forked child the worker threads of these pools do not exist; the next user makes new ones."""
    global _reorder_workers, _check_pool
    _reorder_workers = _check_pool = None


if hasattr(os, 'register_at_fork'):
    os.register_at_fork(after_in_child=_forget_pools)

_default = None


def get_engine():
    """The process-wide engine (created on first use; one GPU per process)."""
    global _default
    if _default is None:
        from . import dist
        cfg = dist.current()
        _default = Engine(device=cfg.get('device'), rank=cfg.get('rank', 0), nranks=cfg.get('nranks', 1),
                          unique_id=cfg.get('unique_id'), shm=cfg.get('shm'))
    return _default


def set_engine(engine):
    global _default
    _default = engine
