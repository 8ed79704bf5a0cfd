"""Cell order on the device.

The random walk gathers, for every cell, the state rows of its ~30 graph neighbours.  How many of
those rows are already in L2 / Infinity Cache depends only on how the cells are numbered, and the
caller's numbering is arbitrary (the reference never looks at it).  So the engine is free to keep
the cells in a locality-preserving order of its own -- clusters of cells that share neighbours
(round 1: reverse Cuthill-McKee of the kNN graph) -- as long as nothing order-dependent leaks out.

Invariants that keep results bit-identical to the caller's order:
  * only rows are renumbered; inside a row the neighbours stay in the caller's CSR order, so every
    per-cell sum adds the same numbers in the same sequence;
  * every per-cell quantity crossing the engine boundary is converted here (`CellOrder`), so
    `cna_amd.tools` and the tests only ever see the caller's order.

`perm[i]` is the caller's index of device row i.  Multi-GPU: every rank computes the same `perm`
from the same graph (the ordering is deterministic) and owns a contiguous block of *device* rows; with a
banded adjacency most neighbours of a block live in the block itself.
"""
import os

import numpy as np
import scipy.sparse as sp


_allowance = None


def usable_cpus(limit=None, share=True):
    """CPUs this process may really use: the cgroup v2 quota if there is one, else the affinity mask -- divided by the
    ranks of the job on this node (share=False: the whole allowance, e.g. for the CPU baseline that rank 0 runs alone)."""
    global _allowance
    if _allowance is None or _allowance[0] != os.getpid():
        n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
        try:
            with open('/sys/fs/cgroup/cpu.max') as f:
                quota, period = f.read().split()
            if quota != 'max':
                n = max(1, min(n, int(int(quota) / int(period))))
        except Exception:
            pass
        _allowance = (os.getpid(), n)          # (read once per process: this sits on the path of every analysis)
    n = _allowance[1]
    # several ranks on one node (one process per GPU) share that allowance: every rank's helper threads -- the draw,
    # the content hash, the cluster order, the column copies -- take their share, not all of it (eight ranks x four
    # draw threads on a 16-CPU allowance made the draw, identical on every rank, the longest item of a rank's step)
    if share:
        n = max(1, n // ranks_on_this_node())
    return n if limit is None else max(1, min(n, int(limit)))


def ranks_on_this_node():
    """Processes of this job on this node: what cna_amd.dist was told (`local_ranks`: the launcher's LOCAL_WORLD_SIZE, else
    every rank of a one-node job); without a job description the launcher's LOCAL_WORLD_SIZE, else 1."""
    try:
        from . import dist
        cfg = dist.current()
        if cfg.get('nranks'):
            return max(1, int(cfg.get('local_ranks') or cfg['nranks']))
    except Exception:                          # noqa: BLE001
        pass
    try:
        k = int(os.environ.get('LOCAL_WORLD_SIZE', '0'))
        return k if k >= 1 else 1
    except ValueError:
        return 1


DEFAULT_CLUSTER = 512


def locality_order(A):
    """Device order of the cells, or None when the reordering is switched off (CNA_REORDER=0) or
    pointless.  Default: clusters of DEFAULT_CLUSTER cells grown greedily by "most edges into the
    cluster" (csrc/host_graph.c), laid out in breadth-first order of the clusters; each XCD then walks
    one cluster at a time (diffuse.hip: xcd_chunk) and finds most of its neighbours' rows in its own
    4 MB L2.  Measured on MI355X, us per dense walk step (tools/kbench_order.py,
    profiles/r02_kbench_order.txt): 2M x 200: 10877 (reverse Cuthill-McKee, one contiguous eighth of
    the rows per XCD) -> 7757; 1M x 100: 2160 -> 1980; 200k x 50: 202 -> 203.  It is also cheaper to
    compute than scipy's RCM (0.15 s against 0.6 s at 1M cells).  CNA_ORDER=rcm selects RCM,
    CNA_ORDER=cluster:B another cluster size."""
    if os.environ.get('CNA_REORDER', '1') in ('0', 'off', 'no') or A.shape[0] < 2:
        return None
    kind = os.environ.get('CNA_ORDER', 'cluster')
    if kind.startswith('cluster'):
        B = int(kind.split(':')[1]) if ':' in kind else DEFAULT_CLUSTER
        return cluster_order(sp.csr_matrix(A), B)
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    perm = reverse_cuthill_mckee(sp.csr_matrix(A), symmetric_mode=False)
    return np.ascontiguousarray(perm, dtype=np.int64)


def cluster_graph(A, order, B):
    """Symmetric weight matrix (scipy CSR, nc x nc) of the clusters of a cell order: entry (c, d) = number of graph
    edges between the cells order[c*B:(c+1)*B] and order[d*B:(d+1)*B] (csrc/host_graph.c:cna_host_cluster_graph)."""
    from . import _ffi
    lib = _ffi.load()
    A = sp.csr_matrix(A)
    n = A.shape[0]
    nc = -(-n // B)
    indptr = np.ascontiguousarray(A.indptr, dtype=np.int64)
    indices = np.ascontiguousarray(A.indices, dtype=np.int32)
    order = np.ascontiguousarray(order, dtype=np.int64)
    ptr = np.zeros(nc + 1, dtype=np.int64)
    tot = lib.cna_host_cluster_graph(n, _ffi.ptr(indptr), _ffi.ptr(indices), _ffi.ptr(order), int(B), _ffi.ptr(ptr), None, None)
    if tot < 0:
        raise MemoryError('cna_host_cluster_graph')
    col = np.zeros(max(tot, 1), dtype=np.int32)
    cnt = np.zeros(max(tot, 1), dtype=np.int64)
    if lib.cna_host_cluster_graph(n, _ffi.ptr(indptr), _ffi.ptr(indices), _ffi.ptr(order), int(B), _ffi.ptr(ptr), _ffi.ptr(col),
                                  _ffi.ptr(cnt)) != tot:
        raise RuntimeError('cna_host_cluster_graph')
    W = sp.csr_matrix((cnt[:tot].astype(np.float64), col[:tot], ptr), shape=(nc, nc))
    return ((W + W.T) * 0.5).tocsr()                  # (a directed input graph: both directions count)


def block_traffic(A, order, nparts):
    """Rows every block of a cell order sends between diffusion steps, peers counted: out[g] = number of distinct
    (row of block g, other block that has a neighbour of it) pairs -- what `cna_graph_upload`'s halo plan comes to for the
    contiguous blocks of ceil(n / nparts) cells of `order` (None: the caller's order)."""
    A = sp.csr_matrix(A)
    n = A.shape[0]
    cap = -(-n // nparts)
    if order is None:
        blk = (np.arange(n, dtype=np.int64) // cap).astype(np.int32)
    else:
        blk = np.empty(n, dtype=np.int32)
        blk[np.asarray(order, dtype=np.int64)] = (np.arange(n, dtype=np.int64) // cap).astype(np.int32)
    wants = np.repeat(blk, np.diff(A.indptr))                       # block of the row that reads column j
    cut = np.flatnonzero(wants != blk[A.indices])
    pairs = np.unique(A.indices[cut].astype(np.int64) * nparts + wants[cut])
    return np.bincount(blk[pairs // nparts], minlength=nparts)


def _swap_refine(W, blk, movable, nparts, max_swaps=100000):
    """Kernighan-Lin hill climbing on the cluster graph: while a pair of equal-sized clusters in two blocks exists whose
    exchange lowers the weight between the blocks, exchange the best such pair (block pairs in turn, until a whole round
    finds none).  blk: block of every cluster (>= nparts: not assigned), edited in place."""
    nc = W.shape[0]
    indptr, indices, data = W.indptr, W.indices, W.data
    hot = sp.csr_matrix((np.ones(nc), (np.arange(nc), np.minimum(blk, nparts))), shape=(nc, nparts + 1))
    M = np.asarray((W @ hot).todense())                             # M[c, g]: weight between cluster c and block g
    swaps = 0
    again = True
    while again and swaps < max_swaps:
        again = False
        for a in range(nparts):
            for b in range(a + 1, nparts):
                while swaps < max_swaps:
                    ca = np.flatnonzero((blk == a) & movable)
                    cb = np.flatnonzero((blk == b) & movable)
                    if len(ca) == 0 or len(cb) == 0:
                        break
                    Da = M[ca, b] - M[ca, a]
                    Db = M[cb, a] - M[cb, b]
                    # the best pair among the few best of either side (the pair's own link counts against it twice)
                    ta = ca[np.argsort(-Da, kind='stable')[:8]]
                    tb = cb[np.argsort(-Db, kind='stable')[:8]]
                    link = np.asarray(W[ta][:, tb].todense())
                    gain = (M[ta, b] - M[ta, a])[:, None] + (M[tb, a] - M[tb, b])[None, :] - 2.0 * link
                    i, j = np.unravel_index(np.argmax(gain), gain.shape)
                    if gain[i, j] <= 1e-9:
                        break
                    c, d = int(ta[i]), int(tb[j])
                    for x, src, dst in ((c, a, b), (d, b, a)):
                        nb = indices[indptr[x]:indptr[x + 1]]
                        w = data[indptr[x]:indptr[x + 1]]
                        M[nb, src] -= w
                        M[nb, dst] += w
                        blk[x] = dst
                    swaps += 1
                    again = True
    return swaps


def partition_order(A, nparts, B=None, refine=True, compare=True):
    """A cell order whose `nparts` contiguous blocks of ceil(n / nparts) cells make good row blocks for a sharded run
    (the state rows of foreign neighbours are what the ranks exchange between diffusion steps):
    order[i] = caller's index of the i-th cell.

    The clusters of the library's cluster order (512 cells that share neighbours) are merged into communities of at most
    one block -- heaviest normalised link first, Kruskal with a size cap, so that what is tightly linked ends up
    together and populations are not cut while anything lighter can be -- and the communities are packed whole into the
    blocks, largest first, each into the block it has the most edges to among those with room (none: best fit); what
    fits nowhere is poured into the room that is left, in cluster order.  `refine`: pairs of clusters are then exchanged
    between blocks while that lowers the number of edges between blocks (`_swap_refine`).  `compare`: the result is kept
    only if its busiest block sends fewer rows than the busiest block of the caller's order (`block_traffic`) -- a
    dataset that arrives sorted by population is best cut where it is -- otherwise the caller's order is returned.
    On the benchmark's generator (20 populations, cells sorted by population), rows sent by the busiest block / by all
    blocks (`block_traffic`): 2M cells, 8 blocks: 192k / 966k -> 76k / 392k (edges cut 20 % -> 3.3 %); 4 blocks: 118k /
    373k -> 90k / 199k; 1M cells, 8 blocks: 97k / 483k -> 67k / 431k; 1M cells in 4 blocks and 200k cells: the caller's
    order is kept.  A dataset whose cells come in random order: every row is wanted elsewhere before, 10-35 % after.
    Host work, once per dataset, before `dist.shard`; deterministic (every rank computes the same order)."""
    A = sp.csr_matrix(A)
    n = A.shape[0]
    if nparts < 2 or n < 2 * nparts:
        return cluster_order(A, B or DEFAULT_CLUSTER)
    cap = -(-n // nparts)
    if B is None:                                     # at least ~16 clusters per block to pack with
        B = DEFAULT_CLUSTER
        while B > 16 and cap < 16 * B:
            B //= 2
    base = cluster_order(A, B)
    nc = -(-n // B)
    csize = np.full(nc, B, dtype=np.int64)
    csize[-1] = n - B * (nc - 1)
    Wfull = cluster_graph(A, base, B)
    W = sp.triu(Wfull, k=1).tocoo()
    score = W.data / np.sqrt(csize[W.row].astype(np.float64) * csize[W.col])
    parent = np.arange(nc, dtype=np.int64)
    size = csize.copy()

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x
    rows, cols = W.row.tolist(), W.col.tolist()
    for e in np.argsort(-score, kind='stable').tolist():
        a, b = find(rows[e]), find(cols[e])
        if a != b and size[a] + size[b] <= cap:
            if size[a] < size[b]:
                a, b = b, a
            parent[b] = a
            size[a] += size[b]
    lab_c = np.unique(np.array([find(x) for x in range(nc)]), return_inverse=True)[1]
    ncomp = int(lab_c.max()) + 1
    sizes = np.bincount(lab_c, weights=csize, minlength=ncomp).astype(np.int64)
    # edges between communities, for the choice of the block
    hot = sp.csr_matrix((np.ones(nc), (np.arange(nc), lab_c)), shape=(nc, ncomp))
    Wc = (hot.T @ Wfull @ hot).tocsr()
    room = np.full(nparts, cap, dtype=np.int64)
    room[-1] = n - cap * (nparts - 1)
    home = np.full(ncomp, -1, dtype=np.int64)
    for c in np.argsort(-sizes, kind='stable'):
        fits = np.flatnonzero(room >= sizes[c])
        if len(fits):
            nb = Wc.indices[Wc.indptr[c]:Wc.indptr[c + 1]]
            wt = Wc.data[Wc.indptr[c]:Wc.indptr[c + 1]]
            placed = home[nb] >= 0
            pull = np.bincount(home[nb[placed]], weights=wt[placed], minlength=nparts)[fits]
            if pull.max() > 0:
                g = fits[np.argmax(pull)]            # the block it shares the most edges with
            else:
                g = fits[np.argmin(room[fits])]      # best fit: the fullest block that still takes it
            home[c] = g
            room[g] -= sizes[c]
    cl_home = home[lab_c]
    blk = np.where(cl_home >= 0, cl_home, nparts).astype(np.int64)
    if refine:
        _swap_refine(Wfull.tocsr(), blk, (blk < nparts) & (csize == B), nparts)
    # the sequence of clusters: block by block its clusters (community by community, each in cluster order), the others
    # poured into what room is left, in cluster order
    cl_seq = np.lexsort((np.arange(nc), lab_c, blk))
    whole = cl_seq[blk[cl_seq] < nparts]
    split = cl_seq[blk[cl_seq] == nparts]
    # cells: clusters may straddle the fill line of a block, so the pouring is done cell by cell
    def cells_of(clusters):
        if len(clusters) == 0:
            return np.zeros(0, dtype=np.int64)
        return np.concatenate([base[c * B:(c + 1) * B] for c in clusters.tolist()])
    out = np.empty(n, dtype=np.int64)
    pour = cells_of(split)
    o = take = 0
    wh_home = blk[whole]
    for g in range(nparts):
        mine = cells_of(whole[wh_home == g])
        size_g = min(cap, n - g * cap)
        out[o:o + len(mine)] = mine
        o += len(mine)
        need = size_g - len(mine)
        out[o:o + need] = pour[take:take + need]
        o += need
        take += need
    assert o == n and take == len(pour)
    if compare and block_traffic(A, None, nparts).max() <= block_traffic(A, out, nparts).max():
        return np.arange(n, dtype=np.int64)
    return out


def cluster_order(A, B):
    """Order in which clusters of B cells that share neighbours are consecutive (csrc/host_graph.c:
    cna_host_cluster_order): order[i] = caller's index of device row i.  Integer work on the host, a
    few tenths of a second per million cells."""
    from . import _ffi
    n = A.shape[0]
    order = np.empty(n, dtype=np.int64)
    if n == 0:
        return order
    indptr = np.ascontiguousarray(A.indptr, dtype=np.int64)
    indices = np.ascontiguousarray(A.indices, dtype=np.int32)
    # on several threads from 131072 cells up (regions by a multi-source BFS, ordered independently): the result
    # depends on the graph only, not on the thread count
    threads = usable_cpus(16)
    if threads >= 1:
        got = _ffi.load().cna_host_cluster_order_mt(n, _ffi.ptr(indptr), _ffi.ptr(indices), int(B), threads, _ffi.ptr(order))
    else:
        got = _ffi.load().cna_host_cluster_order(n, _ffi.ptr(indptr), _ffi.ptr(indices), int(B), _ffi.ptr(order))
    if got < 0:
        raise MemoryError('cna_host_cluster_order')
    return order


def inverse(perm):
    inv = np.empty(len(perm), dtype=np.int64)
    inv[perm] = np.arange(len(perm), dtype=np.int64)
    return inv


def permuted_rows(A, perm, r0, r1, col_map=None):
    """CSR pieces (indptr int64, indices int32, data) of device rows [r0, r1) of P A P^T: row i
    is the caller's row perm[i] with its columns relabelled and left in their original order.
    col_map (sharded callers, whose A holds the rows of one block only): device column of every
    global caller column, instead of the inverse of `perm`.  Done by the library's threaded host
    helper (cna_host_permute_rows: 0.1 s for the 80M edges of a 2M-cell graph; numpy fancy indexing
    took 0.8 s)."""
    from . import _ffi
    lib = _ffi.load()
    inv = np.ascontiguousarray(inverse(perm) if col_map is None else col_map, dtype=np.int64)
    perm = np.ascontiguousarray(perm, dtype=np.int64)
    ptr = np.ascontiguousarray(A.indptr, dtype=np.int64)
    idx = np.ascontiguousarray(A.indices, dtype=np.int32)
    dat = np.ascontiguousarray(A.data)
    if dat.dtype.itemsize not in (4, 8):
        dat = dat.astype(np.float64)
    nnz = int(lib.cna_host_permuted_nnz(_ffi.ptr(perm), int(r0), int(r1), _ffi.ptr(ptr)))
    indptr = np.empty(r1 - r0 + 1, dtype=np.int64)
    indices = np.empty(max(nnz, 1), dtype=np.int32)
    data = np.empty(max(nnz, 1), dtype=dat.dtype)
    rc = lib.cna_host_permute_rows(_ffi.ptr(perm), int(r0), int(r1), _ffi.ptr(ptr), _ffi.ptr(idx), _ffi.ptr(dat),
                                   dat.dtype.itemsize, _ffi.ptr(inv), _ffi.ptr(indptr), _ffi.ptr(indices), _ffi.ptr(data),
                                   usable_cpus(16))
    if rc != 0:
        raise ValueError('cna_host_permute_rows: bad arguments')
    return indptr, indices[:nnz], data[:nnz]


def halo_plan(indices, row0, n_local, rows_per_rank, rank, nranks, allgather_i64, force_self=0):
    """Which state rows must travel between diffusion steps when the cells are sharded by row blocks.

    indices: global column ids of this rank's CSR block.  allgather_i64(a): concatenation, in rank
    order, of every rank's int64 vector a (lengths may differ).  Returns None when exchanging whole
    blocks (the all-gather) moves about as little, else
      (send_rows, send_counts, recv_rows, recv_counts)
    send_rows: local rows other ranks reference, grouped by destination; recv_rows: global rows
    this rank references outside its block, grouped by owner.  Every rank takes the same decision.
    force_self > 0 (tests on one GPU): also "exchange" that many of the rank's own rows with itself."""
    cols = np.unique(indices)
    remote = cols[(cols < row0) | (cols >= row0 + n_local)].astype(np.int64)
    if force_self:
        own = np.arange(row0, row0 + n_local, max(1, n_local // force_self), dtype=np.int64)
        remote = np.sort(np.concatenate([remote, own]))
    owner = remote // rows_per_rank
    recv_counts = np.bincount(owner, minlength=nranks).astype(np.int64)
    want = allgather_i64(recv_counts).reshape(nranks, nranks)        # want[r, p]: rows r needs from p
    n_global_remote = want.sum() - np.trace(want)
    full = float(nranks) * (nranks - 1) * rows_per_rank               # rows an all-gather delivers
    if not force_self and n_global_remote > 0.6 * full:
        return None
    lists = allgather_i64(remote)
    starts = np.concatenate([[0], np.cumsum(want.sum(axis=1))])
    send, send_counts = [], np.zeros(nranks, dtype=np.int64)
    for r in range(nranks):
        off = starts[r] + want[r, :rank].sum()
        send.append(lists[off:off + want[r, rank]] - row0)
        send_counts[r] = want[r, rank]
    send_rows = np.concatenate(send).astype(np.int64) if send else np.zeros(0, dtype=np.int64)
    return (np.ascontiguousarray(send_rows), send_counts, np.ascontiguousarray(remote), recv_counts)


class CellOrder:
    """Mixin for engines: conversions between the caller's cell order and the device's.

    The engine provides: perm (or None), n, row0, n_local, nranks, block(), gather_rows_host(),
    fetch_matrix(), project(), dense_load(), dense_fetch(), and records in `_keep_dev` the keep mask
    (device order, None = all cells) of the last select and in `_x_is_selection` whether the
    working matrix X came from the NAM (True) or from upload_x (False: already caller order).

    Two views.  Replicated (default): the caller holds all n cells on every rank, per-cell vectors
    are global and `perm` is a global permutation.  Local (`view_local`, sharded callers): the
    caller holds the cells of this rank's row block only; n == n_local, per-cell vectors and `perm`
    are local to the block and nothing cells-sized is gathered."""
    perm = None
    view_local = False
    _nam_sig = None          # (inputs signature, nam_epoch, steps taken) of the NAM held by the device
    _keep_dev = None
    _x_is_selection = False
    _kept_order_cache = None

    # per-cell vectors / row blocks over ALL cells
    def cells_to_device(self, v):
        return v if self.perm is None or v is None else np.asarray(v)[self.perm]

    def cells_to_user(self, v):
        if self.perm is None:
            return v
        out = np.empty_like(v)
        out[self.perm] = v
        return out

    def local_keep(self, keep_global):
        """Caller's keep mask -> indices of kept cells inside this rank's block (device order)."""
        keep_dev = self.cells_to_device(np.asarray(keep_global, dtype=bool))
        self._keep_dev = keep_dev
        self._kept_order_cache = None
        mine = keep_dev if self.view_local else keep_dev[self.row0:self.row0 + self.n_local]
        return np.ascontiguousarray(np.flatnonzero(mine), dtype=np.int64)

    # rows of X (kept cells of all ranks, device order) -> caller's order
    def _kept_order(self):
        """X row (device order, kept cells of all ranks) of every kept cell in the caller's order --
        two gathers through the inverse permutation instead of a sort."""
        if self._kept_order_cache is None:
            inv = inverse(self.perm)
            if self._keep_dev is None:
                self._kept_order_cache = inv
            else:
                pos = np.cumsum(self._keep_dev) - 1                 # device row -> X row
                kept_user = np.zeros(len(self.perm), dtype=bool)
                kept_user[self.perm[self._keep_dev]] = True
                self._kept_order_cache = pos[inv[np.flatnonzero(kept_user)]]
        return self._kept_order_cache

    def kept_to_user(self, m):
        if self.perm is None or not self._x_is_selection:
            return m
        return m[self._kept_order()]

    def _all_rows(self, local, n_total):
        return local if self.nranks == 1 or self.view_local else self.gather_rows_host(local, n_total)

    # whole matrices in the caller's order.  On one GPU the rows (and columns) are picked, ordered and
    # -- if asked -- transposed by a gather kernel before the copy (engine.fetch_rows); sharded runs
    # gather the row blocks on the host and reorder there.
    def _on_device(self):
        return (self.nranks == 1 or self.view_local) and hasattr(self, 'fetch_rows')

    def _nam_rows(self, keep):
        """device row of every (kept) cell in the caller's order, or None for "all, in order" """
        if keep is not None and np.all(keep):
            keep = None
        if self.perm is None:
            return None if keep is None else np.flatnonzero(keep).astype(np.int64)
        inv = inverse(self.perm)
        return inv if keep is None else inv[np.flatnonzero(keep)]

    def nam_full(self, keep=None, cols=None, transposed=False):
        """NAM (cells x samples) of all cells, or of the cells of a boolean mask `keep` and the samples
        `cols`, in the caller's order; transposed=True gives samples x cells."""
        from ._ffi import MAT_NAM
        if self._on_device():
            return self.fetch_rows(MAT_NAM, self._nam_rows(keep), cols, transposed)
        m = self.cells_to_user(self._all_rows(self.fetch_matrix(MAT_NAM), self.n))
        if keep is not None and not np.all(keep):
            m = m[np.asarray(keep, dtype=bool)]
        if cols is not None:
            m = m[:, np.asarray(cols)]
        return np.ascontiguousarray(m.T) if transposed else m

    def x_full(self, transposed=False):
        """Working matrix X over the kept cells (cells x samples, or its transpose)."""
        from ._ffi import MAT_X
        if self._on_device():
            rows = self._kept_order() if (self.perm is not None and self._x_is_selection) else None
            return self.fetch_rows(MAT_X, rows, None, transposed)
        m = self.kept_to_user(self._all_rows(self.fetch_matrix(MAT_X), self.x_rows_total))
        return np.ascontiguousarray(m.T) if transposed else m

    def project_full(self, W):
        from ._ffi import MAT_PROJ
        if self._on_device() and hasattr(self, 'project_keep'):
            self.project_keep(W)
            rows = self._kept_order() if (self.perm is not None and self._x_is_selection) else None
            return self.fetch_rows(MAT_PROJ, rows, None, False)
        return self.kept_to_user(self._all_rows(self.project(W), self.x_rows_total))

    def x_stat(self, ordered=True):
        """Per-row statistic of the last X-space kernel over the kept cells: in the caller's order,
        or (ordered=False, e.g. for a median) in whatever order the device holds them."""
        v = self.cell_stat(self.x_rows_total, nam_space=False)
        return self.kept_to_user(v) if ordered else v

    def dense_begin(self, arr):
        self._nam_sig = None                  # the dense walk reuses the state buffers of the NAM
        self.nam_epoch += 1
        r0, r1 = (0, arr.shape[0]) if self.view_local else self.block(arr.shape[0])
        self.dense_load(arr[r0:r1] if self.perm is None else arr[self.perm[r0:r1]])

    def dense_state(self):
        return self.cells_to_user(self._all_rows(self.dense_fetch(), self.n))
