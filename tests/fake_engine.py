"""TEST DOUBLE for cna_amd.engine.Engine, built from the CPU oracle.

Lives under tests/ on purpose: it lets the CPU test-suite exercise the *host* logic of
cna_amd.tools (input checks, sample reindexing, ridge schedule, permutation null, result
fields, row-block sharding and the collectives it needs) without a GPU, and lets
world_size-2 gloo tests check that a sharded run equals an unsharded one.  The product never
imports this file and has no CPU path of its own.

Collectives: ``coll`` is None (single rank) or an object with allreduce_sum(ndarray),
allreduce_max(float) and allgather(ndarray rows) -- see GlooColl below.
"""
import numpy as np
import scipy.sparse as sp

from oracle import cna_oracle as orc
from cna_amd import _order

MAT_NAM, MAT_X = 0, 1


class GlooColl:
    """Host collectives over torch.distributed (gloo) for the sharded CPU tests."""

    def __init__(self):
        import torch.distributed as td
        self.td = td
        self.rank, self.nranks = td.get_rank(), td.get_world_size()

    def allreduce_sum(self, a):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(a).copy())
        self.td.all_reduce(t)
        return t.numpy()

    def allreduce_max(self, v):
        import torch
        t = torch.tensor([float(v)], dtype=torch.float64)
        self.td.all_reduce(t, op=self.td.ReduceOp.MAX)
        return float(t[0])

    def allgather(self, a):
        box = [None] * self.nranks
        self.td.all_gather_object(box, np.ascontiguousarray(a))
        return np.concatenate(box, axis=0)

    def allgather_object(self, obj):
        box = [None] * self.nranks
        self.td.all_gather_object(box, obj)
        return box


class FakeEngine(_order.CellOrder):
    """order: None (caller's cell order on the "device"), 'rcm' (what the real engine does), 'random'
    or an explicit permutation -- the last two exercise every order conversion of _order.CellOrder."""

    def __init__(self, coll=None, order=None):
        self.coll = coll
        self.order = order
        self.reuse_nam = False
        self.rank = coll.rank if coll else 0
        self.nranks = coll.nranks if coll else 1
        self.n = self.n_global = self.row0 = self.n_local = self.N = 0
        self.x_rows_total = 0
        self.x_epoch = self.nam_epoch = 0
        self.calls = []

    # -- collectives
    def _sum(self, a):
        return self.coll.allreduce_sum(a) if self.coll else a

    def _gather(self, a):
        return self.coll.allgather(a) if self.coll else a

    def _exchange(self, new_local):
        """State rows after a step, as the next step sees them: everything (all-gather) or, with a
        halo plan, only the own block and the rows the plan delivers -- the rest is NaN, so a plan
        that misses a needed row poisons the result."""
        if self.halo is None:
            return self._gather(new_local)
        send_rows, send_counts, recv_rows, recv_counts = self.halo
        so = np.concatenate([[0], np.cumsum(send_counts)])
        ro = np.concatenate([[0], np.cumsum(recv_counts)])
        parcels = {p: new_local[send_rows[so[p]:so[p + 1]]] for p in range(self.nranks)}
        box = self.coll.allgather_object(parcels)
        out = np.full((self.n_global, new_local.shape[1]), np.nan)
        out[self.row0:self.row0 + self.n_local] = new_local
        for p in range(self.nranks):
            out[recv_rows[ro[p]:ro[p + 1]]] = box[p][self.rank]
        return out

    def _allgather_i64(self, a):
        return self._gather(np.asarray(a, dtype=np.int64))

    def allgather_fixed(self, values):
        a = np.asarray(values, dtype=np.int64)
        return self._gather(a.reshape(1, -1)) if self.coll else a.reshape(1, -1)

    def allgather_objects(self, obj):
        return self.coll.allgather_object(obj) if self.coll else [obj]

    def x_rows_global(self):
        if not self.view_local or not self.coll:
            return self.x_rows_total
        return int(self.allgather_fixed([self.x_rows_total]).sum())

    def gather_rows_host(self, local, n_total):
        out = self._gather(np.asarray(local))
        assert out.shape[0] == n_total
        return out

    def block(self, n):
        rpr = -(-n // self.nranks)
        r0 = min(self.rank * rpr, n)
        return r0, min(r0 + rpr, n)

    # -- graph
    def ensure_graph(self, A, shard=None, defer=False):
        if getattr(self, '_graph_obj', None) is A:
            return False
        self._graph_obj = A
        A = sp.csr_matrix(A)
        if shard is None:
            self.n_global = A.shape[0]
            r0, r1 = self.block(self.n_global)
            n_order, diag = self.n_global, A
        else:
            r0, self.n_global = int(shard[0]), int(shard[1])
            r1 = r0 + A.shape[0]
            assert (r0, r1) == self.block(self.n_global) and A.shape[1] == self.n_global
            n_order, diag = r1 - r0, A[:, r0:r1]
        self.row0, self.n_local = r0, r1 - r0
        if self.order is None:
            self.perm = None
        elif self.order == 'random':
            self.perm = np.random.RandomState(7 + (self.rank if shard is not None else 0)).permutation(n_order).astype(np.int64)
        elif self.order == 'rcm':
            self.perm = _order.locality_order(diag)
        else:
            self.perm = np.asarray(self.order, dtype=np.int64)
        if shard is not None:
            if self.perm is None:
                self.perm = np.arange(n_order, dtype=np.int64)
            col_map = _order.inverse(self._allgather_i64(self.perm + r0))
            indptr, indices, data = _order.permuted_rows(A, self.perm, 0, r1 - r0, col_map=col_map)
            self.A_local = sp.csr_matrix((data.astype(np.float64), indices, indptr), shape=(r1 - r0, self.n_global))
        elif self.perm is None:
            self.A_local = A[r0:r1].astype(np.float64)
        else:
            indptr, indices, data = _order.permuted_rows(A, self.perm, r0, r1)
            self.A_local = sp.csr_matrix((data.astype(np.float64), indices, indptr), shape=(r1 - r0, self.n_global))
        self.view_local = shard is not None
        self.n = self.n_local if self.view_local else self.n_global
        self._keep_dev = None
        self._kept_order_cache = None
        self._x_is_selection = False
        self.halo = None
        if self.coll:       # same plan the real engine hands to cna_set_halo
            self.halo = _order.halo_plan(self.A_local.indices, r0, r1 - r0, -(-self.n_global // self.nranks), self.rank,
                                         self.nranks, lambda a: self.coll.allgather(np.asarray(a, dtype=np.int64)))
        self._w = None
        return True

    def colsums(self, self_weight=1):
        part = np.asarray(self.A_local.sum(axis=0)).ravel()
        self.colsum = self._sum(part) + self_weight
        self.w = self_weight

    def fetch_colsums(self):
        own = self.colsum[self.row0:self.row0 + self.n_local] if self.view_local else self.colsum
        return self.cells_to_user(own.copy())

    # -- NAM
    def set_samples(self, codes, n_samples, counts, token=None):
        self.codes = self.cells_to_device(np.asarray(codes))
        if self.view_local:
            self.codes = self._gather(self.codes)
        self.N = int(n_samples)
        self.counts = np.asarray(counts, dtype=np.float64)
        S = np.zeros((self.n_global, self.N))
        S[np.arange(self.n_global), self.codes] = 1.0
        self.S = S
        self.steps = 0
        self.nam_epoch += 1

    def _step(self, S):
        c = self.colsum[:, None]
        T = S / c
        loc = slice(self.row0, self.row0 + self.n_local)
        new_local = self.A_local.dot(T) + self.w * S[loc] / c[loc]
        return new_local

    def nam_step(self, want_kurt, may_continue, may_stop):
        self.calls.append(('nam_step', bool(want_kurt), bool(may_continue), bool(may_stop)))
        new_local = self._step(self.S)
        self.S = self._exchange(new_local)
        self.steps += 1
        with np.errstate(all='ignore'):
            if may_stop:
                self.nam = new_local / self.counts
            if want_kurt:
                self.stat_local = orc.row_kurtosis(new_local / self.counts)
                self.stat = self._gather(self.stat_local)

    def nam_steps(self, nsteps):
        for i in range(nsteps):
            self.nam_step(False, i + 1 < nsteps, i + 1 == nsteps)

    def nam_select_hint(self, y_std):
        # (the device engine lets the walk's last step do the selection pass; this double only records the hint -- the
        # host-side schedule that produces it is what the CPU tests exercise)
        self.calls.append(('nam_select_hint', None if y_std is None else np.array(y_std, dtype=np.float64)))

    def clear_resid_factors(self):
        self.calls.append(('clear_resid_factors',))

    def stat_median(self):
        with np.errstate(all='ignore'):
            return float(np.median(self.stat))

    def cell_stat(self, n_expected, nam_space=True):
        stat = self.stat_local if self.view_local else self.stat
        assert len(stat) == n_expected
        return self.cells_to_user(stat.copy()) if nam_space else stat.copy()

    # -- dense diffusion
    def dense_load(self, s_local):
        self.D = self._exchange(np.asarray(s_local, dtype=np.float64))

    def dense_step(self):
        self.D_local = self._step(self.D)
        self.D = self._exchange(self.D_local)

    def dense_fetch(self):
        return self.D_local.copy()

    # -- QC / selection
    def batch_kurtosis(self, which, batch_codes, n_batches):
        mat = self.nam if which == MAT_NAM else self.X
        bc = np.asarray(batch_codes)
        with np.errstate(all='ignore'):
            local = orc.batch_kurtosis(mat, bc, n_batches)
        self.stat_local = local
        self.stat = self._gather(local)

    def zero_variance(self, colmap):
        sub = self.nam if colmap is None else self.nam[:, np.asarray(colmap)]
        with np.errstate(all='ignore'):
            flags = sub.std(axis=1, ddof=1) == 0
        if self.view_local:
            total = int(self.coll.allreduce_sum(np.array([int(flags.sum())]))[0]) if self.coll else int(flags.sum())
            return self.cells_to_user(flags), total
        flags = self._gather(flags.astype(np.uint8)).astype(bool)
        return self.cells_to_user(flags), int(flags.sum())

    def select(self, keep_global, colmap):
        self._x_is_selection = True
        if keep_global is None:
            self._keep_dev, self._kept_order_cache = None, None
            rows, self.keep_local = self.nam, None
        else:
            idx = self.local_keep(keep_global)
            rows = self.nam[idx]
            self.keep_local = np.zeros(self.n_local, dtype=bool)
            self.keep_local[idx] = True
        self.X = rows if colmap is None else rows[:, np.asarray(colmap)]
        self.X = np.array(self.X)
        self.x_rows_total = self.n if keep_global is None else int(np.sum(keep_global))
        self.x_epoch += 1

    def select_standardized(self, keep_global, colmap, y=None, fuse_null=0):
        self.select(keep_global, colmap)
        with np.errstate(all='ignore'):
            nz = int((self.X.std(axis=1, ddof=1) == 0).sum())
        if self.coll:
            nz = int(self.coll.allreduce_sum(np.array([nz]))[0])
        self.standardize(center=True)
        if y is None:
            return nz
        with np.errstate(all='ignore'):
            return nz, self.ncorrs(y)[1]

    def upload_x(self, x_local):
        self.X = np.array(x_local, dtype=np.float64)
        self.x_rows_total = self.X.shape[0]
        self._x_is_selection = False
        self.x_epoch += 1

    # -- residualise + PCA
    def resid_apply(self, M, center):
        if center:
            self.X = self.X - self.X.mean(axis=1, keepdims=True)
        if M is not None:
            self.X = self.X.dot(np.asarray(M).T)

    def standardize(self, center=False):
        if center:
            self.X = self.X - self.X.mean(axis=1, keepdims=True)
        with np.errstate(all='ignore'):
            self.X = self.X / self.X.std(axis=1, ddof=1)[:, None]

    def gram(self):
        return self._sum(self.X.T.dot(self.X))

    def gram_launch(self):
        self._G = self.gram()

    def gram_fetch(self):
        return self._G

    def project(self, W):
        return self.X.dot(np.asarray(W))

    # -- association
    def ncorrs(self, y, fetch=False):
        self.nc = (self.X * np.asarray(y)[None, :]).sum(axis=1) / self.X.shape[1]
        m = np.abs(self.nc).max() if len(self.nc) else 0.0
        if self.coll:
            m = self.coll.allreduce_max(m)
        out = None
        if fetch:
            out = self.kept_to_user(self.nc.copy()) if self.nranks == 1 or self.view_local else self.nc.copy()
        return out, float(m)

    def null_local(self, Yc, edges):
        z2 = (np.abs(self.X.dot(Yc)) / self.X.shape[1]) ** 2
        tails = np.array([[(z2[:, p] >= e).sum() for e in edges] for p in range(Yc.shape[1])], dtype=np.int64)
        return self._sum(tails)

    def condition(self, M, Y):
        Zc = np.asarray(M).dot(np.asarray(Y))
        self.Zc = Zc / Zc.std(axis=0, ddof=1)
        self._M = np.asarray(M)
        self._Y = np.asarray(Y)

    def null_local_resident(self, col0, P, edges, sums_only=False):
        tails = self.null_local(self.Zc[:, col0:col0 + P], edges)
        return tails.sum(axis=0) if sums_only else tails

    def null_local_prepare(self, P, edges, thr=None):
        self._prepared = (np.asarray(edges), None if thr is None else np.asarray(thr))

    def null_local_launch(self, col0, P, edges, thr=None):
        if edges is None:
            edges, thr = self._prepared
        self._pending = self.null_local_resident(col0, P, edges, sums_only=True)
        if thr is not None:
            self._pending = (self._pending,) + tuple(self.obs_counts(edges, thr))

    def null_local_fetch(self):
        out, self._pending = self._pending, None
        return out

    def null_local_discard(self):
        self._pending = None

    def global_test(self, U, ks, r):
        kix, p, r2 = orc.minp_stats(self._Y, self._M, np.asarray(U), np.asarray(ks), r)
        return kix.astype(np.int32), p, r2

    def obs_counts(self, edges, thr):
        z = np.abs(self.nc)
        ranks = np.array([(self.nc ** 2 >= e).sum() for e in edges], dtype=np.int64)
        numdet = np.array([(z > t).sum() for t in thr], dtype=np.int64)
        return self._sum(ranks), self._sum(numdet)

    def percell(self, thr=None, runmin=None):
        coef = np.full(self.n_local, np.nan)
        if self.keep_local is None:
            coef[:] = self.nc
        else:
            coef[self.keep_local] = self.nc
        coef = self.cells_to_user(coef if self.view_local else self._gather(coef))
        if thr is None:
            return coef, None
        idx = np.searchsorted(thr, np.abs(coef), side='right') - 1
        fdr = np.ones(len(coef))
        ok = (idx >= 0) & ~np.isnan(coef)
        fdr[ok] = np.asarray(runmin)[idx[ok]]
        return coef, fdr

    # -- D2H
    def matrix_shape(self, which):
        m = self.nam if which == MAT_NAM else self.X
        return m.shape

    def fetch_matrix(self, which, transposed=False):
        m = self.nam if which == MAT_NAM else self.X
        return np.array(m.T if transposed else m)
