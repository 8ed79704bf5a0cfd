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
