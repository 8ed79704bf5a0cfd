#!/usr/bin/env python3
"""Headline benchmark: cells*permutations / second of an end-to-end ``cna.tl.association``
(BASELINE.json metric) on synthetic data, HIP path, graph resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C2|C3|C4w]

A *step* is one full association() call -- NAM diffusion (3 steps) -> QC/selection ->
residualisation -> Gram/SVD -> global permutation test -> fused local null + FDRs ->
data.obs write-back -- on one synthetic dataset.  At N=1 the workload is BASELINE.json
configs[1] ("C2": 200k cells, 50 samples, k=30, nsteps=3, Nnull=1000).  With N>1 (launched
by torch.distributed.run, one rank per GPU) the cells axis is sharded by row blocks with a
fixed 200k cells per GPU (weak scaling); RCCL carries the state exchange between diffusion
steps and the small all-reduces.

Prints ONE JSON line on rank 0 (see the repo's bench contract) with two extra objects:
  roofline     for the kernel that dominates the timed region (HIP-event timed in this run)
  cpu_baseline the CPU oracle timed here on a bounded sample of the same workload (N=1 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F64_MFMA_PEAK_TF = 78.6      # AMD MI355X datasheet: FP64 matrix 78.6 TFLOP/s (the guide lists no f64 row)

# HBM-side traffic per launch from rocprofv3 PMC passes of THIS bench command (profiles/r01_pmc_summary.txt,
# tools/pmc_run.sh): bytes = 2 * FETCH_SIZE KB + WRITE_SIZE KB.  The factor 2 on FETCH_SIZE is the guide's
# gfx950 correction, re-calibrated here on k_ncorrs (streams the 83.2 MB matrix once, FETCH_SIZE reads 41.6 MB).
# Counted at the L2's fabric side, i.e. Infinity-Cache hits included.  Only valid for the profiled workload.
PMC_TRAFFIC = {
    ('C2', 'nam_step'): 2 * 195300e3 + 84590e3,
    ('C2', 'nam_first'): 2 * 37780e3 + 81250e3,
    ('C2', 'null_local'): 2 * (2 * 41660e3 + 18750e3),     # two launches per pass
}

WORKLOADS = {
    # name: (cells per GPU, samples, kNN k, nsteps, Nnull)
    'C2': (200_000, 50, 30, 3, 1000),
    'C3': (1_000_000, 100, 30, 3, 1000),
    'C4w': (250_000, 200, 30, 3, 1000),     # 8 GPUs x 250k = BASELINE config 4
    'C4': (2_000_000, 200, 30, 3, 1000),    # BASELINE config 4 on ONE GPU (fits: ~14 GB of 288 GB)
    'C5': (2_000_000, 200, 30, 3, 10000),   # BASELINE config 5 on ONE GPU: + 5 covariates, Nnull = 10000
}


def usable_cpus():
    """CPUs this process may use: cgroup v2 quota if set, else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def algorithmic_work(kernel, n, nnz, N, P, T, wA):
    """Algorithmic bytes / flops of ONE launch (DESIGN.md 'Kernels')."""
    ld = (N + 3) // 4 * 4
    if kernel == 'nam_step':
        return 'hbm', nnz * (4 + wA) + 8 * (n + 1) + 8 * n + 2 * 8 * n * N
    if kernel == 'nam_first':
        return 'hbm', nnz * (4 + wA) + 8 * (n + 1) + n * (4 + 8) + 8 * n * N
    if kernel == 'null_local':
        return 'mfma', 2.0 * n * N * P
    if kernel == 'gram':
        return 'mfma', 2.0 * n * N * N
    if kernel in ('resid_xb', 'project_xb'):
        return 'mfma', 2.0 * n * N * N
    if kernel == 'colsum':
        return 'hbm', nnz * (4 + wA) + 8 * n
    if kernel in ('standardize',):
        return 'hbm', 2 * 8 * n * ld
    if kernel in ('select',):
        return 'hbm', 2 * 8 * n * ld
    if kernel in ('ncorrs', 'zero_variance'):
        return 'hbm', 8 * n * ld + 8 * n
    return 'hbm', 8 * n


def load_or_make_dataset(synth, n, N, k, rank, world, td, n_covs=0):
    """One synthetic dataset for the whole job: rank 0 generates it (the kNN search is the slow,
    CPU-only part) and the other ranks of this node read it from /dev/shm -- every rank holds the
    full `data` object, as with a replicated AnnData."""
    if world == 1:
        return synth.make_dataset(n, N, k=k, seed=0, n_covs=n_covs)
    import pickle
    path = '/dev/shm/cna_bench_%s_%d_%d.pkl' % (os.environ.get('MASTER_PORT', '0'), n, N)
    if rank == 0:
        out = synth.make_dataset(n, N, k=k, seed=0, n_covs=n_covs)
        try:
            with open(path + '.tmp', 'wb') as f:
                pickle.dump(out, f, protocol=pickle.HIGHEST_PROTOCOL)
            os.replace(path + '.tmp', path)
        except OSError:                      # no room in /dev/shm: the other ranks generate their own copy
            pass
    td.barrier()
    if rank != 0:
        if os.path.exists(path):
            with open(path, 'rb') as f:
                out = pickle.load(f)
        else:
            out = synth.make_dataset(n, N, k=k, seed=0, n_covs=n_covs)
    td.barrier()
    if rank == 0 and os.path.exists(path):
        os.remove(path)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=60)
    ap.add_argument('--workload', default='C2', choices=sorted(WORKLOADS))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample-cells', type=int, default=200_000)
    ap.add_argument('--profile-host', default=None, help='write a cProfile of 3 extra steps to this file')
    ap.add_argument('--comm', default='rccl', choices=['rccl', 'shm'],
                    help="shm: plumbing check of the N>1 path on ONE GPU (all ranks on device 0, gloo for the "
                         "rendezvous, the library's shared-memory test communicator instead of RCCL); not a benchmark")
    ap.add_argument('--inputs', default='sharded', choices=['sharded', 'replicated'],
                    help='N>1 only.  sharded (default): every rank is handed its own block of cells '
                         '(cna_amd.dist.shard) and gets per-cell results for that block; replicated: every rank '
                         'holds the whole dataset and the whole result, like a replicated AnnData')
    ap.add_argument('--force-dist', action='store_true',
                    help='take the torch.distributed + RCCL code path even with one rank (plumbing check)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit('bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)' % args.gpus)
    td = None
    if args.comm == 'shm' and world > 1:
        import torch.distributed as td
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        td.init_process_group(backend='gloo', rank=rank, world_size=world)
        from cna_amd import dist
        dist.init(rank, world, device=0, shm=('cna_bench_%s' % os.environ['MASTER_PORT'], 64 << 20))
    elif world > 1 or args.force_dist:
        import torch
        import torch.distributed as td
        torch.cuda.set_device(local_rank)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        td.init_process_group(backend='nccl', device_id=torch.device('cuda', local_rank), rank=rank, world_size=world)
        from cna_amd import dist
        dist.init_from_torch(device=local_rank, always_comm=args.force_dist)

    import warnings
    # every repeated call warns that data.obs['coef'] exists (as the reference does); keep the
    # formatting and the stderr write of that message out of the timed loop
    warnings.filterwarnings('ignore', message="Key '.*' already exists in data.obs")
    import cna_amd as cna
    from cna_amd import synth
    cna.tune_host_allocator()       # host-side: no mmap/munmap churn for per-cell numpy temporaries
    from cna_amd.engine import get_engine
    from cna_amd.tools._nam import get_connectivity

    cells_per_gpu, N, k, nsteps, Nnull = WORKLOADS[args.workload]
    n_covs = 5 if args.workload == 'C5' else 0
    n = cells_per_gpu * world
    t0 = time.time()
    data, meta = load_or_make_dataset(synth, n, N, k, rank, world, td, n_covs=n_covs)
    t_gen = time.time() - t0
    A = get_connectivity(data)
    nnz = int(A.nnz)
    deg_full = np.diff(A.indptr)
    sharded_inputs = (world > 1 or args.force_dist) and args.inputs == 'sharded'
    if sharded_inputs:
        from cna_amd import dist
        data = dist.shard(data, rank, world)     # from here on this rank knows its own cells only
        del A
    y = meta['y']
    eng = get_engine()
    eng.reuse_nam = False           # every timed step recomputes the NAM (no result caching across steps)
    kw = dict(nsteps=nsteps, Nnull=Nnull, seed=0)
    if meta.get('covs') is not None:
        kw['covs'] = meta['covs']

    on_gpu_group = td is not None and args.comm != 'shm'

    def sync():
        eng.sync()
        if on_gpu_group:
            import torch
            torch.cuda.synchronize()
            td.barrier()
            torch.cuda.synchronize()
        elif td is not None:
            td.barrier()

    # graph H2D + first call (also the PCIe-inclusive single-call time, reported separately)
    sync()
    t0 = time.perf_counter()
    p_first = cna.tl.association(data, y, 'id', **kw)
    sync()
    t_cold = time.perf_counter() - t0
    for _ in range(max(args.warmup - 1, 0)):
        cna.tl.association(data, y, 'id', **kw)

    eng.prof_reset()
    eng.prof_enable(True)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        p_last = cna.tl.association(data, y, 'id', **kw)
    sync()
    dt = time.perf_counter() - t0
    eng.prof_enable(False)
    prof = eng.prof()
    if td is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64, device='cuda' if on_gpu_group else 'cpu')
        td.all_reduce(t, op=td.ReduceOp.MAX)
        dt = float(t[0])
    assert p_first == p_last

    if args.profile_host and rank == 0:
        import cProfile
        import io
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(3):
            cna.tl.association(data, y, 'id', **kw)
        pr.disable()
        buf = io.StringIO()
        pstats.Stats(pr, stream=buf).sort_stats('tottime').print_stats(45)
        with open(args.profile_host, 'w') as f:
            f.write(buf.getvalue())

    if td is not None:
        # every rank pushes out what its runtime libraries buffered (RCCL's version banner) before
        # rank 0 goes on to print the result line: nothing from another rank can land after it
        sys.stderr.flush()
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        td.barrier()
    if rank != 0:
        if td is not None:
            td.destroy_process_group()
        return

    ms_per_step = dt / args.steps * 1e3
    value = n * Nnull * args.steps / dt
    wA = get_connectivity(data).data.dtype.itemsize
    T = 300
    n_loc = eng.n_local
    deg = deg_full
    rows_loc = slice(eng.row0, eng.row0 + n_loc)
    if sharded_inputs:
        nnz_loc = int(deg[rows_loc].sum())
    else:
        nnz_loc = int(deg[eng.perm[rows_loc]].sum() if eng.perm is not None else deg[rows_loc].sum())
    kernels = {}
    for name, (ms, cnt) in prof.items():
        bound, work = algorithmic_work(name, n_loc, nnz_loc, N, min(1000, Nnull), T, wA)
        avg_s = ms / cnt * 1e-3
        ach = work / avg_s / (1e9 if bound == 'hbm' else 1e12)
        peak = HBM_PEAK_GBS if bound == 'hbm' else F64_MFMA_PEAK_TF
        kernels[name] = dict(total_ms=round(ms, 4), launches=cnt, avg_us=round(ms / cnt * 1e3, 2), bound=bound,
                             achieved=round(ach, 3), peak=peak, unit='GB/s' if bound == 'hbm' else 'TFLOP/s',
                             frac=round(ach / peak, 4))
    dom = max((k_ for k_ in kernels if k_ != 'rccl'), key=lambda k_: kernels[k_]['total_ms'])
    kd = kernels[dom]
    roofline = dict(kernel=dom, bound=kd['bound'], achieved=kd['achieved'], peak=kd['peak'], unit=kd['unit'],
                    frac=kd['frac'], traffic=PMC_TRAFFIC.get((args.workload, dom)) if world == 1 else None,
                    traffic_source='profiles/r01_pmc_summary.txt (rocprofv3 --pmc, same command)', avg_us=kd['avg_us'], launches_per_step=kd['launches'] // args.steps)
    gpu_ms_per_step = sum(v['total_ms'] for v in kernels.values()) / args.steps

    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        from oracle import cna_oracle as orc
        ns = min(args.cpu_sample_cells, n)
        sdata, smeta = (data, meta) if ns == n else synth.make_dataset(ns, N, k=k, seed=0)
        # give the CPU path the cores this container may actually use (cgroup quota), not the
        # host's core count: oversubscribed BLAS threads only get the process throttled
        threads = usable_cpus()
        try:
            from threadpoolctl import threadpool_limits
            limiter = threadpool_limits(limits=threads, user_api='blas')
        except Exception:
            import contextlib
            limiter = contextlib.nullcontext()
        with limiter:
            orc.association(sdata, smeta['y'], 'id', mode='reference', **dict(kw, Nnull=50))   # warm caches
            t0 = time.perf_counter()
            ref = orc.association(sdata, smeta['y'], 'id', mode='reference', **kw)
            t_cpu = time.perf_counter() - t0
        cpu = dict(value=round(ns * Nnull / t_cpu, 1), unit='cell*perm/s', cores=int(threads), kind='port',
                   seconds=round(t_cpu, 2), host_cpus=os.cpu_count(),
                   p_value=float(ref['p']) if isinstance(ref, dict) and 'p' in ref else None,
                   sample='oracle/cna_oracle.py association(mode=reference) on %d cells x %d samples, k=%d, '
                          'nsteps=%d, Nnull=%d (same generator, seed 0); numpy/scipy vectorised port, '
                          'BLAS threads as listed' % (ns, N, k, nsteps, Nnull))

    out = {
        'metric': 'cells*permutations/sec end-to-end cna.tl.association',
        'value': round(value, 1), 'unit': 'cell*perm/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': '%s: %d cells (%d per GPU) x %d samples, k=%d kNN (%.1f nnz/row, float32 CSR), '
                               'nsteps=%d, Nnull=%d, local FDR pass on, NAM cache off' % (args.workload, n, cells_per_gpu, N, k,
                                                                          nnz / n, nsteps, Nnull),
                   'parallelism': 'cells sharded in %d row block(s)%s%s%s' % (
                       world, '' if world == 1 else (', every rank holds its block of cells only' if sharded_inputs
                                                     else ', dataset and per-cell results replicated on every rank'),
                       '' if eng.halo is None else ', halo exchange %d/%d rows out/in on rank 0' % eng.halo,
                       ' [--comm shm: ranks share one GPU, plumbing check only]' if args.comm == 'shm' and world > 1 else ''), 'p_value': p_last},
        'roofline': roofline,
        'cpu_baseline': cpu,
        'gpu_kernel_ms_per_step': round(gpu_ms_per_step, 3),
        'host_ms_per_step': round(ms_per_step - gpu_ms_per_step, 3),
        'first_call_ms_incl_graph_h2d': round(t_cold * 1e3, 1),
        'dataset_gen_s': round(t_gen, 1),
        'kernels': kernels,
    }
    # the JSON line is the last thing this process writes: tear the communicators down first and
    # push out whatever the libraries (RCCL prints a version banner) still hold in C stdio buffers
    try:
        eng.close()
    except Exception:
        pass
    if td is not None:
        td.destroy_process_group()
    sys.stderr.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.write(json.dumps(out) + '\n')
    sys.stdout.flush()              # (a normal exit follows: profilers attached to this process write at exit)


if __name__ == '__main__':
    main()
