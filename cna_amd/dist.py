"""Multi-GPU bootstrap: one process per GPU, cells sharded by contiguous row blocks.

The data path uses RCCL directly from libcna_hip.so (cna_comm_init).  RCCL needs one
out-of-band exchange -- rank 0's 128-byte unique id -- which any host channel can carry.
``init_from_env`` carries it over an abstract Unix-domain socket between the ranks of a node (what
bench.py does under `torch.distributed.run` -- the launcher only provides RANK / WORLD_SIZE /
MASTER_PORT; torch.distributed itself is not used); ``init_from_torch`` uses an already-initialised
``torch.distributed`` group for that single broadcast; ``init`` takes the id directly.
"""
import os

_cfg = {}


def init(rank, nranks, unique_id=None, device=None, shm=None, local_ranks=None):
    """Describe this process' place in the job before the first engine is created.  ``shm``:
    (segment name, slot bytes) selects the host-staged test communicator (ranks may share a GPU).
    ``local_ranks``: processes of this job on THIS node (they share the host's CPUs: _order.usable_cpus); default: the
    launcher's LOCAL_WORLD_SIZE, else all `nranks` (one node)."""
    global _cfg
    if nranks > 1 and unique_id is None and shm is None:
        raise ValueError('unique_id required when nranks > 1 (create it on rank 0 with new_unique_id())')
    if local_ranks is None:
        try:
            local_ranks = int(os.environ.get('LOCAL_WORLD_SIZE', '0')) or None
        except ValueError:
            local_ranks = None
    # with one rank a unique_id is optional: when given, collectives still go through RCCL
    _cfg = dict(rank=int(rank), nranks=int(nranks), unique_id=unique_id, shm=shm,
                device=int(device) if device is not None else None,
                local_ranks=max(1, min(int(local_ranks), int(nranks))) if local_ranks else int(nranks))
    from . import engine
    engine.set_engine(None)


def current():
    return dict(_cfg)


def new_unique_id():
    from .engine import Engine
    return Engine.new_unique_id()


def init_from_torch(device=None, always_comm=False):
    """Broadcast the RCCL id over an existing torch.distributed process group."""
    import torch.distributed as td
    if not td.is_initialized():
        raise RuntimeError('torch.distributed is not initialised')
    rank, nranks = td.get_rank(), td.get_world_size()
    if device is None:
        device = int(os.environ.get('LOCAL_RANK', rank))
    uid = None
    if nranks > 1 or always_comm:
        box = [new_unique_id() if rank == 0 else None]
        td.broadcast_object_list(box, src=0)
        uid = box[0]
    init(rank, nranks, uid, device)
    return rank, nranks


def _exchange_id(rank, nranks, key, make, timeout=600.0):
    """Rank 0's 128-byte RCCL id to every rank of this node, without torch: rank 0 listens on an abstract Unix-domain
    socket named after the job (`key`: MASTER_PORT of the launcher, unique per job on a node; no file is left
    behind and a stale name cannot exist -- the kernel drops it with the last descriptor) and hands the id to the
    nranks - 1 processes that connect."""
    import socket
    import struct
    import time
    name = b'\0cna_amd_rdzv_' + str(key).encode()
    if rank == 0:
        uid = make()
        srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        srv.bind(name)
        srv.listen(max(nranks, 1))
        deadline = time.time() + timeout
        served = 0
        try:
            # only processes of this user are served (SO_PEERCRED), and a stranger's connection does not use up one
            # of the nranks - 1 hand-overs
            while served < nranks - 1:
                srv.settimeout(max(0.1, deadline - time.time()))
                conn, _ = srv.accept()
                with conn:
                    try:
                        cred = conn.getsockopt(socket.SOL_SOCKET, socket.SO_PEERCRED, struct.calcsize('3i'))
                        uid_peer = struct.unpack('3i', cred)[1]
                    except OSError:
                        uid_peer = -1
                    if uid_peer != os.getuid():
                        continue
                    conn.sendall(uid)
                    served += 1
        finally:
            srv.close()
        return uid
    deadline = time.time() + timeout
    while True:
        cli = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        try:
            cli.connect(name)
            buf = b''
            while len(buf) < 128:
                part = cli.recv(128 - len(buf))
                if not part:
                    raise ConnectionError('rank 0 closed the rendezvous socket early')
                buf += part
            return buf
        except (FileNotFoundError, ConnectionRefusedError):
            if time.time() > deadline:
                raise TimeoutError('no RCCL id from rank 0 (rendezvous %r)' % name)
            time.sleep(0.02)
        finally:
            cli.close()


def init_from_env(always_comm=False, shm=None):
    """One process per GPU started by a launcher that sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT
    (`python -m torch.distributed.run`, `torchrun`, bench.py's own spawner): rank 0 makes the RCCL id and the
    other ranks of the node fetch it through `_exchange_id` -- torch.distributed is neither imported nor
    initialised, so the job holds ONE RCCL communicator, the library's.  shm=(name, slot bytes): the host-staged
    test communicator instead (ranks may share a GPU; no id needed)."""
    rank = int(os.environ.get('RANK', '0'))
    nranks = int(os.environ.get('WORLD_SIZE', '1'))
    device = int(os.environ.get('LOCAL_RANK', rank))
    if shm is not None:
        init(rank, nranks, None, 0, shm=tuple(shm))          # the test communicator: every rank on GPU 0
        return rank, nranks
    uid = None
    if nranks > 1 or always_comm:
        # the id travels over a Unix-domain socket: the ranks of ONE node (a multi-node launcher must hand the id over
        # itself: init(rank, nranks, unique_id))
        local = os.environ.get('LOCAL_WORLD_SIZE')
        nnodes = os.environ.get('GROUP_WORLD_SIZE') or os.environ.get('NNODES')
        if (local is not None and int(local) != nranks) or (nnodes is not None and nnodes.isdigit() and int(nnodes) > 1):
            raise RuntimeError('cna_amd.dist.init_from_env serves the ranks of one node (WORLD_SIZE=%d, LOCAL_WORLD_SIZE=%s); '
                               'pass the RCCL id yourself with init(rank, nranks, unique_id)' % (nranks, local))
        key = os.environ.get('MASTER_PORT', '29533') + '_' + os.environ.get('TORCHELASTIC_RUN_ID', 'none')
        uid = _exchange_id(rank, nranks, key, new_unique_id) if nranks > 1 else new_unique_id()
    init(rank, nranks, uid, device)
    return rank, nranks


def barrier():
    """All ranks of the job (a small collective through the library's own communicator)."""
    from .engine import get_engine
    get_engine().allgather_fixed([0])


def max_over_ranks(x):
    """max of a host float over the ranks (timings): gathered as integers, exact to a nanosecond for seconds."""
    from .engine import get_engine
    got = get_engine().allgather_fixed([int(round(float(x) * 1e9))])
    return float(got.max()) * 1e-9


def block(n_cells, rank, nranks):
    """Rows [r0, r1) of an n_cells problem that rank `rank` of `nranks` owns (ceil split; the same
    rule as Engine.block and cna_graph_upload)."""
    rpr = -(-int(n_cells) // int(nranks))
    r0 = min(int(rank) * rpr, int(n_cells))
    return r0, min(r0 + rpr, int(n_cells))


def shard(data, rank=None, nranks=None, partition=None):
    """This rank's block of an AnnData-like dataset, for sharded runs: the rows [r0, r1) of
    ``data.obs`` and of the connectivities graph (all columns, global ids).

    ``partition``: which cells form the blocks.  None / False: the caller's order is cut into contiguous blocks.
    True: the cells are first put into ``cna_amd._order.partition_order(A, nranks)`` -- whole populations of the
    graph packed into the blocks -- so that a block's cells have few neighbours outside it (the rows exchanged
    between diffusion steps) -- unless the busiest block of the caller's own order sends no more rows than
    the busiest of those (a dataset that arrives sorted by population), then the caller's order is kept ('always': never
    kept); every rank computes the same order.  An index array: that order.  The
    block keeps the caller's ``obs`` index, so per-cell results are matched by name as before.  (The analysis is then
    that of the dataset with its cells renumbered: every row still adds its neighbours in the caller's order, the
    column sums add their rows in the new one -- results agree with the unpartitioned run to rounding, ~1e-15.)  Pass the result to
    ``cna.tl.association`` / ``cna.tl.nam`` / ``cna.tl.diffuse`` on every rank: per-cell inputs and
    outputs (``obs`` columns, ``res.ncorrs``, ``res.kept``, NAM columns) then cover this rank's
    cells only and no cells-sized vector is gathered; sample-level results (p-value, PCs, FDR
    thresholds) are global and identical on all ranks.  A loader that never materialises the whole
    dataset can build the same object itself: ``obs`` of the block, an (r1 - r0) x n_cells CSR under
    ``obsp['connectivities']`` and ``uns['cna_shard'] = {'row0': r0, 'n_global': n_cells}``."""
    import scipy.sparse as sp
    from .synth import CellData
    from .tools._nam import get_connectivity
    cfg = current()
    rank = cfg.get('rank', 0) if rank is None else rank
    nranks = cfg.get('nranks', 1) if nranks is None else nranks
    A = sp.csr_matrix(get_connectivity(data))
    n = A.shape[0]
    r0, r1 = block(n, rank, nranks)
    if partition is not None and partition is not False and nranks > 1:
        import numpy as np
        from . import _order
        if partition is True or (isinstance(partition, str) and partition == 'always'):
            order = _order.partition_order(A, nranks, compare=partition is True)
        else:
            order = np.asarray(partition, dtype=np.int64)
        if len(order) != n or len(np.unique(order)) != n:
            raise ValueError('partition must be a permutation of the cells')
        mine = order[r0:r1]
        rows = A[mine]                                        # this block's rows, columns still in the caller's numbering
        inv = _order.inverse(order)
        rows = sp.csr_matrix((rows.data, inv[rows.indices].astype(rows.indices.dtype), rows.indptr), shape=rows.shape)
        rows.has_sorted_indices = False                       # (every row keeps the caller's order of its entries: the order its sums are formed in)
        part = CellData(data.obs.iloc[mine].copy(), rows)
        part.uns['cna_shard'] = {'row0': r0, 'n_global': n, 'order': order}
        return part
    part = CellData(data.obs.iloc[r0:r1].copy(), A[r0:r1])
    part.uns['cna_shard'] = {'row0': r0, 'n_global': n}
    return part
