"""Launcher shim for boxes with ONE GPU: under `python -m torch.distributed.run --nproc-per-node N tools/one_gpu_rank.py
bench.py ...` every rank gets device 0 and a host id of its own (see tools/rccl_ranks_one_gpu.sh), then runs the given
script as __main__ with the remaining arguments -- the launcher's own environment (RANK, WORLD_SIZE, MASTER_PORT,
TORCHELASTIC_RUN_ID, ...) otherwise untouched, which is the point: the rendezvous of cna_amd.dist.init_from_env under
the launcher the driver uses."""
import os
import runpy
import sys

rank = os.environ.get('RANK', '0')
os.environ.update(LOCAL_RANK='0', NCCL_HOSTID='cna_one_gpu_host_' + rank, NCCL_SOCKET_IFNAME='lo', NCCL_IB_DISABLE='1',
                  HSA_ENABLE_IPC_MODE_LEGACY='0')
script = sys.argv[1]
sys.argv = sys.argv[1:]
sys.path.insert(0, os.path.dirname(os.path.abspath(script)))
runpy.run_path(script, run_name='__main__')
