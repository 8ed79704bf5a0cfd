"""Multi-GPU bootstrap: one process per GPU, cells sharded by contiguous row blocks.

The data path uses RCCL directly from libcna_hip.so (cna_comm_init).  RCCL needs one
out-of-band exchange -- rank 0's 128-byte unique id -- which any host channel can carry.
``init_from_torch`` uses an already-initialised ``torch.distributed`` group for that single
broadcast (this is what bench.py does under torchrun); ``init`` takes the id directly.
"""
import os

_cfg = {}


def init(rank, nranks, unique_id=None, device=None, shm=None):
    """Describe this process' place in the job before the first engine is created.  ``shm``:
    (segment name, slot bytes) selects the host-staged test communicator (ranks may share a GPU)."""
    global _cfg
    if nranks > 1 and unique_id is None and shm is None:
        raise ValueError('unique_id required when nranks > 1 (create it on rank 0 with new_unique_id())')
    # with one rank a unique_id is optional: when given, collectives still go through RCCL
    _cfg = dict(rank=int(rank), nranks=int(nranks), unique_id=unique_id, shm=shm,
                device=int(device) if device is not None else None)
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


def block(n_cells, rank, nranks):
    """Rows [r0, r1) of an n_cells problem that rank `rank` of `nranks` owns (ceil split; the same
    rule as Engine.block and cna_graph_upload)."""
    rpr = -(-int(n_cells) // int(nranks))
    r0 = min(int(rank) * rpr, int(n_cells))
    return r0, min(r0 + rpr, int(n_cells))


def shard(data, rank=None, nranks=None):
    """This rank's block of an AnnData-like dataset, for sharded runs: the rows [r0, r1) of
    ``data.obs`` and of the connectivities graph (all columns, global ids).  Pass the result to
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
    part = CellData(data.obs.iloc[r0:r1].copy(), A[r0:r1])
    part.uns['cna_shard'] = {'row0': r0, 'n_global': n}
    return part
