#!/usr/bin/env python3
"""Headline benchmark: cells*permutations / second of an end-to-end ``cna.tl.association``
(BASELINE.json metric) on synthetic data, HIP path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C4|C3|C2|C5] [--scaling strong|weak]

A *step* is one full association() call -- NAM diffusion (3 steps) -> QC/selection ->
residualisation -> Gram/SVD -> global permutation test -> fused local null + FDRs ->
data.obs write-back -- on one synthetic dataset, with the graph, its device cell order and the
factorised sample ids resident on the GPU (the steady state of analysing several phenotypes of one
dataset; `engine.pin_graph`), and the walk and the permutation draw recomputed every step (NAM cache and the memo of a
seed's permutations off: the timed steps repeat one phenotype and must not skip what the reference does per call).  The cold first call
(graph preparation, upload over PCIe) is timed separately and reported beside it; it is never `value`.

Workload: BASELINE.json configs[3] ("C4": 2M cells x 200 samples, k=30, nsteps=3, Nnull=1000), the
largest configuration -- it fits one MI355X (~20 GB of 288 GB).  With N > 1 ranks (one per GPU; launched
by torch.distributed.run, or self-spawned when `--gpus N` is given to a plain `python bench.py`) the
SAME 2M x 200 problem is sharded over the ranks by row blocks of cells (strong scaling, BASELINE.json's
"sharded over 8 x MI355X"); `--scaling weak` keeps the per-GPU block fixed instead.  At N = 1 two more
lines ride along in the same JSON object (`other_configs`): configs[4] ("C5": C4 + 5 covariates, Nnull = 10000),
configs[2] ("C3", 1M x 100), configs[1] ("C2", 200k x 50), two reference call shapes at C3 and "C4_block8" (250k x 200: one
rank's share of C4 on eight GPUs as a problem of its own -- the input of DESIGN.md 7's scaling estimate).

Prints ONE JSON line on rank 0 (see the repo's bench contract) with two extra objects:
  roofline     for the kernel that dominates the timed region (HIP-event timed in this run)
  cpu_baseline oracle/reference_cost.py (the reference's own sequence of library calls, Python loops
               included) timed here on a bounded sample of the same workload (N = 1 only)
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F64_MFMA_PEAK_TF = 78.6      # AMD MI355X datasheet: FP64 matrix 78.6 TFLOP/s (the guide lists no f64 row)
I8_MFMA_PEAK_TOPS = 5033.0   # dense i8 matrix: 2x the bf16 rate (guide: >= 4404 TOPS measured with 32x32x32): 1024 SIMDs x 2048 ops/clk x 2.4 GHz

# HBM-side traffic per launch from rocprofv3 PMC passes of THIS bench command (tools/pmc_step.sh ->
# profiles/*_pmc_traffic.json: bytes = 2 * FETCH_SIZE KB + WRITE_SIZE KB; the factor 2 on FETCH_SIZE is the
# guide's gfx950 correction, re-calibrated on k_ncorrs, which streams the matrix once).  Counted at the L2's
# fabric side, i.e. Infinity-Cache hits included.  Only valid for the profiled workload on one GPU.
PMC_FILE = os.path.join(ROOT, "profiles", "r06_pmc_traffic.json")

METRIC = 'cells*permutations/sec end-to-end cna.tl.association'
ARITHMETIC_I8 = ('f64 throughout (diffusion, QC, residualisation, Gram, F-tests); the local-null products '
                 'as exact 24-bit fixed-point digits on the i8 matrix cores with an f64 recheck of every '
                 'output within the error bound of a threshold (same integer counts as the f64 kernel)')
DETAILS_FILE = os.path.join(ROOT, 'bench_details.json')

WORKLOADS = {
    # name: (cells, samples, kNN k, nsteps, Nnull, covariates[, batches])
    'C2': (200_000, 50, 30, 3, 1000, 0),
    'C3': (1_000_000, 100, 30, 3, 1000, 0),
    'C4': (2_000_000, 200, 30, 3, 1000, 0),     # BASELINE config 4
    'C5': (2_000_000, 200, 30, 3, 10000, 5),    # BASELINE config 5: + 5 covariates, Nnull = 10000
    # the reference's own call shapes at the size of configs[2]: its default walk rule (nsteps=None,
    # _association.py:194, _nam.py:64-68) and the demo's call with covariates AND batches (demo/demo.ipynb:149)
    'C3_default_nsteps': (1_000_000, 100, 30, None, 1000, 0),
    'C3_covs_batches': (1_000_000, 100, 30, 3, 1000, 2, 5),
    # one rank's share of C4 on eight GPUs as a problem of its own (no exchange): what the kernels of a block take when
    # the block is all there is -- the input of DESIGN.md 7's predicted timeline
    'C4_block8': (250_000, 200, 30, 3, 1000, 0),
    # C4 / C3 with the OPT-IN 4-byte state between walk steps (Engine.set_state_f32): not the default and never
    # `value` -- the NAM is then within ~1e-7 of the default walk's instead of bit-identical; what exactness of the walk costs
    'C4_state_f32': (2_000_000, 200, 30, 3, 1000, 0),
    'C3_state_f32': (1_000_000, 100, 30, 3, 1000, 0),
    # C4 as a drop-in caller gets it: WITHOUT engine.pin_graph (an API the reference does not have) -- the content of the
    # connectivities matrix and of the id column is hashed in full on every call (inside cna_assoc_finish, while the
    # device works); same dataset object as the C4 run
    'C4_unpinned': (2_000_000, 200, 30, 3, 1000, 0),
    'C2_unpinned': (200_000, 50, 30, 3, 1000, 0),
    # C2 / a rank's block with the permutation memo ON (the library's default for users: the permutations of a seeded draw
    # do not depend on the phenotype, so a second phenotype with the same seed replays them) -- never `value`: the timed
    # steps repeat one phenotype and would skip a draw the reference makes on every call
    'C2_draw_memo': (200_000, 50, 30, 3, 1000, 0),
    'C4_block8_draw_memo': (250_000, 200, 30, 3, 1000, 0),
}
WORKLOAD_OPTS = {'C4_state_f32': dict(state_f32=True), 'C3_state_f32': dict(state_f32=True),
                 'C4_unpinned': dict(pin=False), 'C2_unpinned': dict(pin=False),
                 'C2_draw_memo': dict(draw_memo=True), 'C4_block8_draw_memo': dict(draw_memo=True)}
DEFAULT_STEPS = {'C2_draw_memo': (100, 60), 'C4_block8_draw_memo': (50, 10), 'C4_unpinned': (20, 5), 'C2_unpinned': (100, 60), 'C4_state_f32': (20, 5), 'C3_state_f32': (20, 5), 'C4_block8': (50, 10), 'C2': (100, 60), 'C3': (20, 5), 'C4': (20, 5), 'C5': (20, 5), 'C3_default_nsteps': (10, 3),
                 'C3_covs_batches': (10, 3)}


def usable_cpus():
    """CPUs this process may use: cgroup v2 quota if set, else the affinity mask (the whole allowance: the CPU baseline
    runs on rank 0 alone, whatever LOCAL_WORLD_SIZE says)."""
    from cna_amd._order import usable_cpus as u
    return u(share=False)


def algorithmic_work(kernel, n, nnz, N, P, T, wA):
    """Algorithmic bytes / flops of ONE launch (DESIGN.md 'Kernels')."""
    ld = (N + 3) // 4 * 4
    if kernel in ('nam_step', 'nam_step_sparse'):      # the dense gather / the second step on the compressed state: same algorithmic bytes
        return 'hbm', nnz * (4 + wA) + 8 * (n + 1) + 8 * n + 2 * 8 * n * N
    if kernel == 'nam_first':
        return 'hbm', nnz * (4 + wA) + 8 * (n + 1) + n * (4 + 8) + 8 * n * N
    if kernel == 'null_local':
        return 'mfma', 2.0 * n * N * P
    if kernel == 'gram':
        return 'mfma', 1.0 * n * N * (N + 1)          # X^T X is symmetric: the upper triangle is all the algorithm needs
    if kernel in ('resid_xb', 'project_xb'):
        return 'mfma', 2.0 * n * N * N
    if kernel == 'colsum':
        return 'hbm', nnz * (4 + wA) + 8 * n
    if kernel in ('standardize',):
        return 'hbm', 2 * 8 * n * ld
    if kernel in ('select',):
        return 'hbm', 2 * 8 * n * ld
    if kernel in ('ncorrs', 'zero_variance', 'nam_finish'):
        return 'hbm', 8 * n * ld + 8 * n
    return 'hbm', 8 * n


def load_or_make_dataset(synth, n, N, k, rank, world, n_covs=0, n_batches=0):
    """One synthetic dataset for the whole job: rank 0 generates it (graph construction) and the other ranks of
    this node read it from /dev/shm.  No collective is involved (the ranks that wait must not sit in the
    communicator's set-up meanwhile): rank 0 publishes the file by an atomic rename, the others poll for it; the
    caller removes it after the job's first barrier."""
    if world == 1:
        return synth.make_dataset(n, N, k=k, seed=0, n_covs=n_covs, n_batches=n_batches), None
    import pickle
    path = '/dev/shm/cna_bench_%s_%s_%d_%d.pkl' % (os.environ.get('MASTER_PORT', '0'), os.environ.get('TORCHELASTIC_RUN_ID', 'x'), n, N)
    if rank == 0:
        out = synth.make_dataset(n, N, k=k, seed=0, n_covs=n_covs, n_batches=n_batches)
        with open(path + '.tmp', 'wb') as f:
            pickle.dump(out, f, protocol=pickle.HIGHEST_PROTOCOL)
        os.replace(path + '.tmp', path)
        return out, path
    deadline = time.time() + 1800
    while not os.path.exists(path):
        if time.time() > deadline:
            sys.exit('bench.py: rank %d saw no dataset from rank 0' % rank)
        time.sleep(0.05)
    with open(path, 'rb') as f:
        return pickle.load(f), path


def self_spawn(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one per GPU, the
    environment torch.distributed.run would give them) and pass rank 0's output through."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    # a rank that dies (an exception, a device fault) leaves the others blocked in their next collective: the
    # first non-zero exit ends the job -- the remaining ranks are stopped and its code is passed on
    import time
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                print('bench.py: a rank exited with status %d; stopping the other %d' % (code, len(live)), file=sys.stderr)
                for q in live:
                    q.terminate()
                deadline = time.time() + 10
                for q in live:
                    try:
                        q.wait(max(0.1, deadline - time.time()))
                    except subprocess.TimeoutExpired:
                        q.kill()
        time.sleep(0.05)
    sys.exit(rc)


class stdout_to_stderr:
    """fd 1 -> fd 2 while a library that writes to C stdout starts up (RCCL prints a version banner when the first
    communicator is created): rank 0's stdout carries the JSON line and nothing else."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        os.dup2(self.saved, 1)
        os.close(self.saved)


def time_workload(name, args, rank, world, steps, warmup, want_kernels=True, dataset=None):
    """Generate the dataset of `name`, run the cold call, warm up, time `steps` calls.  Returns a dict of
    raw measurements (rank 0 fills the JSON from it)."""
    import warnings
    import cna_amd as cna
    from cna_amd import synth
    from cna_amd.engine import get_engine
    from cna_amd.tools._nam import get_connectivity

    n_total, N, k, nsteps, Nnull, n_covs = WORKLOADS[name][:6]
    n_batches = WORKLOADS[name][6] if len(WORKLOADS[name]) > 6 else 0
    n = n_total * world if args.scaling == 'weak' else n_total
    t0 = time.time()
    if dataset is not None:                      # (another configuration of the same dataset: C4_unpinned after C4)
        (data, meta), shared_file = dataset, None
    else:
        (data, meta), shared_file = load_or_make_dataset(synth, n, N, k, rank, world, n_covs=n_covs, n_batches=n_batches)
    t_gen = time.time() - t0
    A = get_connectivity(data)
    nnz = int(A.nnz)
    wA = A.data.dtype.itemsize
    deg_full = np.diff(A.indptr)
    sharded_inputs = (world > 1 or args.force_dist) and args.inputs == 'sharded'
    if sharded_inputs:
        from cna_amd import dist
        # from here on this rank knows its own cells only; which cells those are: whole populations of the graph packed
        # into the blocks (cna_amd._order.partition_order, what a loader that cares about the exchange volume does), or
        # --partition caller: contiguous runs of the generator's order
        data = dist.shard(data, rank, world, partition=(args.partition == 'populations'))
        del A
    y = meta['y']
    with stdout_to_stderr():
        eng = get_engine()                       # (with a communicator: collective set-up, RCCL's banner)
    eng.reuse_nam = False           # every timed step recomputes the NAM (no result caching across steps)
    from cna_amd.tools import _stats as _cna_stats
    _cna_stats.DRAW_MEMO = bool(WORKLOAD_OPTS.get(name, {}).get('draw_memo', False))   # ... and draws its permutations again
    eng.set_state_f32(bool(WORKLOAD_OPTS.get(name, {}).get('state_f32')))      # (back to the default for every other workload)
    pinned = bool(WORKLOAD_OPTS.get(name, {}).get('pin', True))
    if pinned:
        eng.pin_graph(get_connectivity(data))    # the bench never edits the graph in place (see module docstring)
    else:
        eng.unpin_graph()                        # ... the `*_unpinned` lines: what a caller who does not know pin_graph gets
    kw = dict(nsteps=nsteps, Nnull=Nnull, seed=0)
    if meta.get('covs') is not None:
        kw['covs'] = meta['covs']
    if meta.get('batches') is not None:
        kw['batches'] = meta['batches']

    from cna_amd import dist

    def sync():
        # device idle on every rank, then a barrier through the library's own communicator, then idle again
        eng.sync()
        if world > 1 or args.force_dist:
            dist.barrier()
            eng.sync()

    sync()
    if shared_file and rank == 0 and os.path.exists(shared_file):
        os.remove(shared_file)                    # every rank has loaded it

    # cold call: graph preparation (cell order, block lists), H2D over PCIe, first analysis
    sync()
    t0 = time.perf_counter()
    p_first = cna.tl.association(data, y, 'id', **kw)
    sync()
    t_cold = time.perf_counter() - t0
    # A large graph is analysed in the caller's cell order first while the device order is computed on a host thread
    # (engine.ensure_graph); the first later call that finds it done re-uploads the graph in that order.  Here: wait
    # for it, time that adopting call on its own, then warm up -- the timed steps run in the steady state.
    t_adopt = None
    if getattr(eng, 'reorder_pending', lambda: False)():
        eng.wait_reorder()
        sync()
        t0 = time.perf_counter()
        cna.tl.association(data, y, 'id', **kw)
        sync()
        t_adopt = time.perf_counter() - t0
    for _ in range(max(warmup - 1, 0)):
        cna.tl.association(data, y, 'id', **kw)
    assert not getattr(eng, 'reorder_pending', lambda: False)()

    # The timed steps carry HIP events around the walk kernels (the dominant kernel of every configuration: `roofline`) and
    # the communication spans only; the per-kernel table comes from a second loop of the same steps with every kernel
    # group timed -- two event records per group are ~0.1 ms of host time per analysis, 8 % of the 200 000-cell one.
    eng.prof_reset()
    eng.prof_enable(True, walk_only=True)
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        p_last = cna.tl.association(data, y, 'id', **kw)
    sync()
    dt = time.perf_counter() - t0
    eng.prof_enable(False)
    prof_timed = eng.prof()
    eng.prof_reset()
    eng.prof_enable(True)
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        cna.tl.association(data, y, 'id', **kw)
    sync()
    dt_all_spans = time.perf_counter() - t0
    eng.prof_enable(False)
    prof = eng.prof()
    prof.update(prof_timed)                        # (the walk and the communication: as measured in the timed steps)
    dt_local = dt
    if world > 1 or args.force_dist:
        dt = dist.max_over_ranks(dt)               # the slowest rank's clock
    assert p_first == p_last
    # what a rank's step is made of, from every rank (N > 1: so that a scaling curve explains itself): HIP-event times of
    # its compute kernels, of the main communicator's collectives, of the halo exchange and of the main stream's wait for
    # it, its own wall clock -- microseconds per step, gathered through the library's communicator
    comm_keys = ('rccl', 'halo_exchange', 'halo_wait')
    side_keys = ('condition', 'global_test')       # second stream, beside the main one
    mine = dict(wall_ms=dt_local / steps * 1e3,
                kernel_ms=sum(ms for k_, (ms, _) in prof.items() if k_ not in comm_keys + side_keys) / steps,
                allreduce_ms=prof.get('rccl', (0.0, 0))[0] / steps,
                halo_exchange_ms=prof.get('halo_exchange', (0.0, 0))[0] / steps,
                halo_wait_ms=prof.get('halo_wait', (0.0, 0))[0] / steps)
    mine['host_ms'] = mine['wall_ms'] - mine['kernel_ms'] - mine['allreduce_ms'] - mine['halo_wait_ms']
    rank_keys = sorted(mine)
    per_rank = eng.allgather_fixed([int(round(mine[k_] * 1e3)) for k_ in rank_keys]).astype(np.float64) * 1e-3

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

    n_loc = eng.n_local
    rows_loc = slice(eng.row0, eng.row0 + n_loc)
    if sharded_inputs:
        nnz_loc = int(deg_full[rows_loc].sum())
    else:
        nnz_loc = int(deg_full[eng.perm[rows_loc]].sum() if eng.perm is not None else deg_full[rows_loc].sum())
    try:
        i8 = eng.null_local_i8_stats()
    except Exception:
        i8 = (False, 0, False)
    return dict(name=name, n=n, N=N, k=k, nsteps=nsteps, Nnull=Nnull, n_covs=n_covs, n_batches=n_batches, nnz=nnz, wA=wA, dt=dt, t_cold=t_cold, i8=i8,
                t_gen=t_gen, prof=prof, dt_all_spans=dt_all_spans, p=p_last, t_adopt=t_adopt, n_loc=n_loc, nnz_loc=nnz_loc, halo=getattr(eng, 'halo', None),
                sharded_inputs=sharded_inputs, data=data, meta=meta, kw=kw, steps=steps, warmup=warmup, comm=eng.comm_info(),
                per_rank={k_: [round(float(v), 3) for v in per_rank[:, i]] for i, k_ in enumerate(rank_keys)}, pinned=pinned,
                two_call_path=_two_call_stats(),
                halo_comm=getattr(eng, 'halo_comm', False), dev_bytes=int(eng.device_bytes()),
                state_f32=bool(WORKLOAD_OPTS.get(name, {}).get('state_f32')))


def _two_call_stats():
    try:
        from cna_amd.tools import _fast
        return dict(_fast.stats)
    except Exception:
        return None


def rccl_transport(rank):
    """Which transports RCCL set its channels up with, from the NCCL_DEBUG=INFO log this rank wrote (main() points
    NCCL_DEBUG_FILE at it): counts of 'via P2P/...' (xGMI / PCIe peer access), 'via SHM/...' (host memory) and 'via NET/...'
    (sockets / NICs) channel lines.  None when there is no log."""
    path = os.environ.get('CNA_BENCH_NCCL_LOG')
    if not path or not os.path.exists(path):
        return None
    out = {}
    try:
        with open(path, errors='replace') as f:
            for line in f:
                i = line.find(' via ')
                if i < 0 or 'Channel' not in line:
                    continue
                kind = line[i + 5:].split()[0].split('/')
                key = '/'.join(kind[:2]) if kind[0] in ('P2P', 'SHM') else kind[0]
                out[key] = out.get(key, 0) + 1
    except OSError:
        return None
    return out or None


def ranks_summary(m, world):
    """{key: [max over ranks, rank 0]} of the per-rank step decomposition, plus what crossed between the ranks."""
    pr = m.get('per_rank') or {}
    out = {k_: [max(v), v[0]] for k_, v in pr.items() if v}
    halo = m.get('halo')
    if halo:
        row_bytes = 8 * m['N']                      # one state row
        out['halo_rows_out_in_rank0'] = list(halo)
        out['halo_mb_per_exchange_out_in_rank0'] = [round(h * row_bytes / 1e6, 3) for h in halo]
    return out


def kernel_table(m, world):
    T = 300
    kernels = {}
    for name, (ms, cnt) in m['prof'].items():
        bound, work = algorithmic_work(name, m['n_loc'], m['nnz_loc'], m['N'], min(1000, m['Nnull']), T, m['wA'])
        if m.get('state_f32') and name in ('nam_step', 'nam_step_sparse') and m['N'] >= 96 and m['nsteps'] == 3:
            # opt-in 4-byte state: the second step writes, the third reads, 4 bytes per entry where 8(d) prices 8
            work -= 4 * m['n_loc'] * m['N']
        step_work = work if name == 'nam_step' else None           # SURVEY 8(d)'s bytes of a diffusion step, whatever the launch also does
        if name == 'nam_first' and m['N'] >= 96 and world == 1 and m['n_loc']:
            # from 96 samples on (one GPU) the first step leaves a row as its non-zeros -- 16-byte {value, sample} records,
            # at most 64, and a count byte -- instead of the dense 8N-byte row 8(d) prices (diffuse.hip:first_tail): priced
            # on the EXPECTED number of distinct samples among a cell and its neighbours, N (1 - (1 - 1/N)^(deg + 1))
            deg = m['nnz_loc'] / m['n_loc']
            pairs = min(64.0, m['N'] * (1.0 - (1.0 - 1.0 / m['N']) ** (deg + 1.0)))
            work = work - 8 * m['n_loc'] * m['N'] + m['n_loc'] * (16.0 * pairs + 1.0)
        if name == 'nam_step' and 'select' not in m['prof'] and m['N'] > 64:
            # the last step also did the selection pass (diffuse.hip:select_tail): it writes X instead of the NAM (same
            # bytes) plus what that pass adds -- three digit planes of 32 ceil(N/32) bytes, coefficient, row scale
            work += m['n_loc'] * (3 * 32 * ((m['N'] + 31) // 32) + 24)
        if name == 'gram' and cnt > m['steps']:
            # the product ran range by range under the walk's last step (c_api.hip:ranged_last_step): a launch covers
            # 1 / ranges of the cells
            work /= cnt / m['steps']
        avg_s = ms / cnt * 1e-3
        ach = work / avg_s / (1e9 if bound == 'hbm' else 1e12)
        peak = HBM_PEAK_GBS if bound == 'hbm' else F64_MFMA_PEAK_TF
        kernels[name] = dict(total_ms=round(ms, 4), launches=cnt, avg_us=round(ms / cnt * 1e3, 2), bound=bound,
                             achieved=round(ach, 3), peak=peak, unit='GB/s' if bound == 'hbm' else 'TFLOP/s',
                             frac=round(ach / peak, 4))
        if step_work is not None:
            kernels[name]['frac_step_bytes'] = round(step_work / avg_s / 1e9 / peak, 4)
        if name == 'null_local' and m.get('i8', (False,))[0]:
            # the pass ran on the integer matrix cores (csrc/null_i8.hip): six exact i8 digit products stand in for
            # one f64 product, so the algorithmic work is 6 x 2nNP' integer operations against the i8 peak; the time
            # covers the whole pass (quantisation of X and Yc, products + binning, f64 recheck, reductions)
            ach = 6.0 * work / avg_s / 1e12
            kernels[name].update(achieved=round(ach, 1), peak=I8_MFMA_PEAK_TOPS, unit='TOP/s (i8)',
                                 frac=round(ach / I8_MFMA_PEAK_TOPS, 4), f64_equivalent_tflops=round(work / avg_s / 1e12, 1),
                                 rechecked_in_f64=m['i8'][1], fell_back_to_f64=m['i8'][2])
    return kernels


def pmc_traffic(workload, kernel):
    try:
        with open(PMC_FILE) as f:
            tab = json.load(f)
        v = tab.get(workload, {}).get(kernel)
        return (float(v), os.path.relpath(PMC_FILE, ROOT) + ' (rocprofv3 --pmc passes of this command)') if v else (None, None)
    except Exception:
        return None, None


def summary(m, world, steps):
    """The per-workload part of the JSON line."""
    kernels = kernel_table(m, world)
    comm_spans = ('rccl', 'halo_exchange', 'halo_wait')      # communication and waiting: reported under `ranks`, not as kernels
    dom = max((k_ for k_ in kernels if k_ not in comm_spans), key=lambda k_: kernels[k_]['total_ms'])
    kd = kernels[dom]
    traffic, src = pmc_traffic(m['name'], dom) if world == 1 else (None, None)
    roofline = dict(kernel=dom, bound=kd['bound'], achieved=kd['achieved'], peak=kd['peak'], unit=kd['unit'],
                    frac=kd['frac'], traffic=traffic, traffic_source=src, avg_us=kd['avg_us'],
                    launches_per_step=kd['launches'] // steps)
    if 'frac_step_bytes' in kd:
        # the same launch priced on SURVEY 8(d)'s step bytes alone (`frac` also counts what the fused selection pass
        # writes: X instead of the NAM, digit planes, coefficients)
        roofline['frac_step_bytes'] = kd['frac_step_bytes']
    if dom == 'nam_step':
        # what the row-gather actually moves, beside the algorithmic bytes the fraction is priced on: every edge fetches
        # its neighbour's 8N-byte state row through the vector L1 (the chip delivers 17-19 TB/s on this
        # pattern), and `traffic` of it comes from behind the L2 (4.8-5.8 TB/s of 128-byte lines at these sizes)
        gathered = float(m['nnz_loc']) * 8.0 * m['N']
        roofline['gathered_bytes_per_launch'] = gathered
        roofline['gathered_TBps'] = round(gathered / (kd['avg_us'] * 1e-6) / 1e12, 2)
        if traffic:
            roofline['behind_l2_TBps'] = round(traffic / (kd['avg_us'] * 1e-6) / 1e12, 2)
    # (the exchange runs beside the walk on its own stream and `halo_wait` is the main stream standing still: neither is kernel
    # time; the collectives of the main communicator are, as before)
    gpu_ms = sum(v['total_ms'] for k_, v in kernels.items() if k_ not in ('halo_exchange', 'halo_wait')) / steps
    ms_per_step = m['dt'] / steps * 1e3
    return dict(value=round(m['n'] * m['Nnull'] * steps / m['dt'], 1), ms_per_step=round(ms_per_step, 3), roofline=roofline,
                kernels=kernels, gpu_kernel_ms_per_step=round(gpu_ms, 3),
                ms_per_step_all_spans=None if m.get('dt_all_spans') is None else round(m['dt_all_spans'] / steps * 1e3, 3),
                host_ms_per_step=round(ms_per_step - gpu_ms, 3), ranks=ranks_summary(m, world),
                graph_pinned=m.get('pinned', True), two_call_path=m.get('two_call_path'),
                cold_first_call=dict(ms=round(m['t_cold'] * 1e3, 1), value=round(m['n'] * m['Nnull'] / m['t_cold'], 1),
                                     call_that_adopts_the_device_order_ms=None if m.get('t_adopt') is None else round(m['t_adopt'] * 1e3, 1),
                                     note='first call: graph H2D over PCIe (in the caller\'s cell order when the graph is large: '
                                          'the device order is computed on a host thread beside it and adopted by a later call, '
                                          'whose time -- the resident graph renumbered on the device -- is listed too), column sums, first-use allocations, one analysis'),
                dataset_gen_s=round(m['t_gen'], 1), p_value=m['p'])


def workload_text(m, world, args):
    return ('%s: %d cells (%d per GPU) x %d samples, k=%d kNN (%.1f nnz/row, float32 CSR), nsteps=%s, Nnull=%d%s%s, '
            'local FDR pass on; graph + device cell order + sample codes resident (%s), '
            'walk and permutation draw recomputed every step (NAM cache and draw memo off%s)' % (
                m['name'], m['n'], -(-m['n'] // world), m['N'], m['k'], m['nnz'] / m['n'],
                'None (the reference\'s stop rule)' if m['nsteps'] is None else str(m['nsteps']), m['Nnull'],
                ', %d covariates' % m['n_covs'] if m['n_covs'] else '', ', %d batches' % m['n_batches'] if m.get('n_batches') else '',
                'graph pinned: engine.pin_graph' if m.get('pinned', True) else 'NOT pinned: graph and ids hashed in full on every call, as a drop-in caller gets it',
                '; the last walk step leaves the standardised NAM and its coefficients directly -- the raw NAM, which this '
                'call does not read, is materialised on demand' if (m['nsteps'] is not None and m['nsteps'] >= 2 and m['N'] > 64
                                                                   and not m['n_covs'] and not m.get('n_batches')) else ''))


def cpu_reference_cost(synth, ns, N, k, nsteps, Nnull, n_covs, extrapolate_to=None):
    """oracle/reference_cost.py -- the reference's own sequence of library calls (pandas frames, scipy csr.dot +
    st.kurtosis per step, a Python loop over the permutations with st.f.sf, the materialised cells x Nnull null
    matrix, np.histogram per null column, per-cell Series.apply; 1.04x the wall time of the real reference on
    50k x 50 x 1000 in the build container) -- timed on `ns` cells of the generator (seed 0) on this box's host cores.
    BLAS gets the cores the cgroup allows; everything else is single-threaded, as in the reference."""
    from oracle import reference_cost as rc
    sdata, smeta = synth.make_dataset(ns, N, k=k, seed=0, n_covs=n_covs)
    # give the CPU path the cores this container may actually use (cgroup quota), not the
    # host's core count: oversubscribed BLAS threads only get the process throttled
    threads = usable_cpus()
    blas = None
    try:
        from threadpoolctl import threadpool_limits, threadpool_info
        limiter = threadpool_limits(limits=threads, user_api='blas')
    except Exception:
        import contextlib
        limiter = contextlib.nullcontext()
        threadpool_info = None
    Pc = min(Nnull, 1000)
    with limiter:
        if threadpool_info is not None:
            blas = max([int(i.get('num_threads', 1)) for i in threadpool_info() if i.get('user_api') == 'blas'] or [1])
        t0 = time.perf_counter()
        ref = rc.association(sdata, smeta['y'], 'id', covs=smeta.get('covs'), nsteps=nsteps, Nnull=Pc, seed=0)
        t_cpu = time.perf_counter() - t0
    st = ref['stages']
    cpu = dict(value=round(ns * Pc / t_cpu, 1), unit='cell*perm/s', cores=int(threads), kind='port',
               mode='reference-cost', seconds=round(t_cpu, 2), host_cpus=os.cpu_count(), blas_threads=blas, p_value=float(ref['p']),
               stages_s={k_: round(v, 2) for k_, v in st.items()},
               value_without_percell_apply=round(ns * Pc / max(t_cpu - st.get('percell_apply', 0.0), 1e-9), 1),
               sample_short='reference-cost port on %d cells x %d samples, k=%d, nsteps=%s, Nnull=%d, same generator seed 0' % (ns, N, k, nsteps, Pc),
               sample='oracle/reference_cost.py (the reference\'s own sequence of library calls: pandas frames, '
                      'scipy csr.dot + st.kurtosis per step, a Python loop over the permutations with st.f.sf, the '
                      'materialised cells x Nnull null matrix, np.histogram per null column, per-cell Series.apply; '
                      '1.04x the wall time of the real reference on 50k x 50 x 1000 in the build container) on %d '
                      'cells x %d samples, k=%d, nsteps=%s, Nnull=%d of the same generator (seed 0); BLAS threads as '
                      'listed, everything else single-threaded as in the reference' % (ns, N, k, nsteps, Pc))
    if extrapolate_to is not None and extrapolate_to != ns:
        # every stage of the reference's cost profile is linear in the cell count at fixed samples and Nnull
        # (the null loop's per-permutation cost does not depend on the cells, the rest is per cell):
        # the sample's stage times scaled to the workload's cells.  EXTRAPOLATED, and labelled so.
        scaled = {k_: (v if k_ == 'global_test' else v * extrapolate_to / ns) for k_, v in st.items()}
        tot = sum(scaled.values())
        cpu['extrapolated_to_workload'] = dict(cells=extrapolate_to, seconds=round(tot, 1),
                                               stages_s={k_: round(v, 1) for k_, v in scaled.items()},
                                               value=round(extrapolate_to * Pc / max(tot, 1e-9), 1))
    return cpu


def workload_short(m, world):
    """<= 200 characters: what the step is, for the contract line (`workload_text` is the long form)."""
    return ('%s: %d cells x %d samples, k=%d kNN (%.1f nnz/row, f32 CSR), nsteps=%s, Nnull=%d%s%s, local FDR on; '
            'graph resident, walk + draw recomputed every step' % (
                m['name'], m['n'], m['N'], m['k'], m['nnz'] / m['n'], m['nsteps'], m['Nnull'],
                ', %d covs' % m['n_covs'] if m['n_covs'] else '', ', %d batches' % m['n_batches'] if m.get('n_batches') else ''))[:200]


def parallelism_text(m, world, args):
    return 'cells sharded in %d row block(s)%s%s%s' % (
        world, '' if world == 1 else ((', every rank holds its block of cells only (blocks: %s)' % (
            'whole populations of the graph packed per block' if args.partition == 'populations' else "contiguous runs of the caller's order")) if m['sharded_inputs']
                                      else ', dataset and per-cell results replicated on every rank'),
        '' if m['halo'] is None else ', halo exchange %d/%d rows out/in on rank 0' % m['halo'],
        ' [--comm shm: ranks share one GPU, plumbing check only]' if args.comm == 'shm' and world > 1 else '')


def _roofline_short(r):
    """The roofline object of the contract line.  `frac` is priced on SURVEY 8(d)'s bytes of the launch ALONE
    (`frac_step_bytes` of the details); what the fused by-product adds is `frac_fused`."""
    if not r:
        return None
    frac = r.get('frac_step_bytes', r.get('frac'))
    out = dict(kernel=r.get('kernel'), bound=r.get('bound'), achieved=round(frac * r['peak'], 3) if frac is not None else None,
               peak=r.get('peak'), unit=r.get('unit'), frac=frac, traffic=r.get('traffic'), avg_us=r.get('avg_us'))
    if 'frac_step_bytes' in r:
        out['frac_fused'] = r.get('frac')
    return out


def _cpu_short(c):
    if not c:
        return None
    out = {k_: c.get(k_) for k_ in ('value', 'unit', 'cores', 'kind', 'mode', 'seconds', 'value_without_percell_apply') if k_ in c}
    if c.get('sample_short'):
        out['sample'] = c['sample_short']
    if c.get('extrapolated_to_workload'):
        e = c['extrapolated_to_workload']
        out['extrapolated'] = dict(cells=e.get('cells'), seconds=e.get('seconds'), value=e.get('value'))
    for k_ in ('stages_s', 'blas_threads', 'gpu_ms_per_step', 'gpu_over_cpu', 'p_value'):
        if k_ in c:
            out[k_] = c[k_]
    return out


def assemble_details(m, main_sum, cpu, cpu_c2, extra, world, steps, warmup, args):
    """Everything this run measured (the side file); `contract_line` condenses it."""
    return {
        'metric': METRIC,
        'value': main_sum['value'], 'unit': 'cell*perm/s', 'n_gpus': world, 'steps': steps,
        'warmup': warmup, 'ms_per_step': main_sum['ms_per_step'], 'higher_is_better': True,
        'scaling': args.scaling, 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': workload_text(m, world, args), 'workload_short': workload_short(m, world),
                   'parallelism': parallelism_text(m, world, args),
                   'communicator': {'backend': m['comm'][0], 'nranks_reported_by_communicator': m['comm'][1],
                                    'halo_rows_out_in_rank0': m['halo'],
                                    'device_bytes_rank0': m.get('dev_bytes'),       # everything the library holds on rank 0's GPU (cna_ctx_device_bytes)
                                    'halo_exchange_overlaps_the_walk_step': bool(m.get('halo_comm')) or (m['comm'][0] == 'shm' and world > 1)},
                   'arithmetic': ARITHMETIC_I8 if m.get('i8', (False,))[0] else 'f64 throughout',
                   'p_value': m['p']},
        'roofline': main_sum['roofline'],
        'cpu_baseline': cpu,
        'cpu_baseline_C2_full': cpu_c2,
        'gpu_kernel_ms_per_step': main_sum['gpu_kernel_ms_per_step'],
        'host_ms_per_step': main_sum['host_ms_per_step'],
        # the same steps again with HIP events around EVERY kernel group (where `kernels` comes from; the timed steps carry
        # them around the walk kernels and the communication spans only)
        'ms_per_step_all_spans': main_sum.get('ms_per_step_all_spans'),
        'ranks': main_sum.get('ranks') if world > 1 or args.force_dist else None,
        'rccl_transport': rccl_transport(0) if world > 1 or args.force_dist else None,
        'two_call_path': main_sum.get('two_call_path'),
        'cold_first_call': main_sum['cold_first_call'],
        'first_call_ms_incl_graph_h2d': main_sum['cold_first_call']['ms'],
        'value_cold': main_sum['cold_first_call']['value'],      # the same metric on the first call (graph preparation + H2D included)
        'dataset_gen_s': main_sum['dataset_gen_s'],
        'kernels': main_sum['kernels'],
        'other_configs': extra,
    }


def contract_line(d, details_file=None):
    """The ONE stdout line of the bench contract, kept small (target <= 6 KB, tests/test_bench_line.py): the contract's
    keys, `roofline`, `cpu_baseline`, and one {ms_per_step, value, roofline} triple per other configuration.  Everything
    else (per-kernel tables of every configuration, notes, cold-call breakdown) is in `bench_details.json` / on stderr."""
    cfg = d['config']
    others = {}
    for name, o in (d.get('other_configs') or {}).items():
        if 'error' in o:
            others[name] = dict(error=str(o['error'])[:120])
            continue
        r = o.get('roofline') or {}
        others[name] = dict(ms_per_step=o.get('ms_per_step'), value=o.get('value'), steps=o.get('steps'),
                            gpu_kernel_ms=o.get('gpu_kernel_ms_per_step'), host_ms=o.get('host_ms_per_step'),
                            first_call_ms=(o.get('cold_first_call') or {}).get('ms'), p_value=o.get('p_value'),
                            **({'scaling': o['scaling'], 'cells': o.get('cells'), 'ranks': o.get('ranks')} if o.get('scaling') else {}),
                            roofline=dict(kernel=r.get('kernel'), frac=r.get('frac_step_bytes', r.get('frac')), avg_us=r.get('avg_us')))
    top = sorted((d.get('kernels') or {}).items(), key=lambda kv: -kv[1]['total_ms'])[:6]
    steps = max(int(d['steps']), 1)
    out = {k_: d[k_] for k_ in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
                                'scaling', 'vs_baseline', 'dtype', 'data')}
    comm = cfg.get('communicator') or {}
    out['config'] = dict(workload=cfg.get('workload_short', cfg['workload'])[:200], parallelism=cfg['parallelism'][:200],
                         arithmetic=('f64; local null as exact i8 digit products on the matrix cores + f64 recheck near cuts'
                                     if cfg['arithmetic'] != 'f64 throughout' else 'f64 throughout')[:120],
                         comm=comm.get('backend'), comm_ranks=comm.get('nranks_reported_by_communicator'),
                         halo_rows_out_in=comm.get('halo_rows_out_in_rank0'),
                         device_gb_rank0=None if comm.get('device_bytes_rank0') is None else round(comm['device_bytes_rank0'] / 1e9, 3),
                         halo_overlaps_step=comm.get('halo_exchange_overlaps_the_walk_step'), p_value=cfg.get('p_value'))
    out['roofline'] = _roofline_short(d.get('roofline'))
    out['cpu_baseline'] = _cpu_short(d.get('cpu_baseline'))
    for key in ('cpu_baseline_C2_full', 'cpu_baseline_C3_full'):
        if d.get(key):
            out[key] = _cpu_short(d[key])
    out['gpu_kernel_ms_per_step'] = d.get('gpu_kernel_ms_per_step')
    out['host_ms_per_step'] = d.get('host_ms_per_step')
    out['first_call_ms_incl_graph_h2d'] = d.get('first_call_ms_incl_graph_h2d')
    out['value_cold'] = d.get('value_cold')
    if d.get('ranks'):
        # N > 1: what a rank's step is made of -- {key: [max over ranks, rank 0]} in ms per step (HIP events on every rank,
        # gathered through the communicator) -- and which transports RCCL set up, so that a scaling curve explains itself
        out['ranks'] = d['ranks']
        out['rccl_transport'] = d.get('rccl_transport')
    # the kernels that make the step, each with its own fraction of its own roof: {name: [ms per step, launches per step, frac]}
    out['kernels_ms_per_step'] = {k_: [round(v['total_ms'] / steps, 3), round(v['launches'] / steps, 2),
                                       v.get('frac_step_bytes', v.get('frac'))] for k_, v in top}
    out['other_configs'] = others
    out['details'] = os.path.basename(details_file or DETAILS_FILE) + ' (and stderr): per-kernel tables of every configuration'
    line = json.dumps(out, separators=(',', ':'))
    assert '\n' not in line
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None)
    ap.add_argument('--warmup', type=int, default=None)
    ap.add_argument('--workload', default='C4', choices=sorted(WORKLOADS))
    ap.add_argument('--scaling', default='strong', choices=['strong', 'weak'],
                    help='N>1: strong = the same total problem sharded over the ranks (default, BASELINE configs[3]); '
                         'weak = the workload size PER GPU')
    ap.add_argument('--no-extra', action='store_true', help='N=1: skip the additional C5, C3 and C2 lines')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample-cells', type=int, default=150_000)
    ap.add_argument('--no-cpu-c2-full', action='store_true',
                    help='N=1: skip the reference-cost CPU run of the FULL C2 workload (~1-2 min of host time)')
    ap.add_argument('--cpu-full', default='auto',
                    help='N=1: BASELINE configurations whose FULL workload the reference-cost CPU path runs (C2, C3).  auto: C2, '
                         'and C3 as well (the "HBM roofline run": ~100 s of host time, ~30 GB) when the host allows it (>= 16 usable cores, >= 64 GB free)')
    ap.add_argument('--details', default=DETAILS_FILE, help='where the full per-kernel tables go (JSON)')
    ap.add_argument('--profile-host', default=None, help='write a cProfile of 3 extra steps to this file')
    ap.add_argument('--comm', default='rccl', choices=['rccl', 'shm'],
                    help="shm: plumbing check of the N>1 path on ONE GPU (all ranks on device 0, gloo for the "
                         "rendezvous, the library's shared-memory test communicator instead of RCCL); not a benchmark")
    ap.add_argument('--inputs', default='sharded', choices=['sharded', 'replicated'],
                    help='N>1 only.  sharded (default): every rank is handed its own block of cells '
                         '(cna_amd.dist.shard) and gets per-cell results for that block; replicated: every rank '
                         'holds the whole dataset and the whole result, like a replicated AnnData')
    ap.add_argument('--partition', default='populations', choices=['populations', 'caller'],
                    help='N>1, sharded inputs: how the cells are dealt to the ranks (see cna_amd.dist.shard)')
    ap.add_argument('--force-dist', action='store_true',
                    help='take the communicator (RCCL) code path even with one rank (plumbing check)')
    args = ap.parse_args()

    if args.gpus > 1 and 'RANK' not in os.environ:
        self_spawn(args)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        sys.exit('bench.py --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if world > 1 or args.force_dist:
        # The launcher (torch.distributed.run, or self_spawn above) only provides RANK / LOCAL_RANK / WORLD_SIZE /
        # MASTER_PORT.  torch.distributed is NOT initialised: the job's one communicator is the library's own RCCL
        # communicator, whose id travels over a Unix-domain socket (cna_amd.dist.init_from_env); barriers and the
        # max over ranks of the timing go through it as well.
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        if args.comm == 'rccl':
            # RCCL says which transport every channel got (P2P over xGMI / PCIe, host shared memory, sockets) at INFO level:
            # into a file per rank, parsed for the result line (rccl_transport) -- a scaling curve that came over sockets
            # should say so itself
            if os.environ.get('NCCL_DEBUG', 'VERSION').upper() in ('VERSION', 'WARN'):     # (the image exports VERSION)
                os.environ['NCCL_DEBUG'] = 'INFO'
            log = '/tmp/cna_bench_nccl_%s_r%d.log' % (os.environ['MASTER_PORT'], rank)
            try:
                if os.path.exists(log):
                    os.remove(log)
            except OSError:
                pass
            os.environ.setdefault('NCCL_DEBUG_FILE', log)
            os.environ['CNA_BENCH_NCCL_LOG'] = os.environ['NCCL_DEBUG_FILE']
        from cna_amd import dist
        if args.comm == 'shm':
            dist.init_from_env(shm=('cna_bench_%s' % os.environ['MASTER_PORT'],
                                    (512 << 20) if WORKLOADS[args.workload][0] > 1_000_000 else (64 << 20)))   # a slot holds one rank's halo rows
        else:
            dist.init_from_env(always_comm=args.force_dist)

    import warnings
    # every repeated call warns that data.obs['coef'] exists (as the reference does); keep the
    # formatting and the stderr write of that message out of the timed loop
    warnings.filterwarnings('ignore', message="Key '.*' already exists in data.obs")
    warnings.filterwarnings('ignore', message='global association p-value attained minimal')
    warnings.filterwarnings('ignore', message='data supported use of')
    import cna_amd as cna
    from cna_amd import synth
    cna.tune_host_allocator()       # host-side: no mmap/munmap churn for per-cell numpy temporaries
    from cna_amd.engine import get_engine

    steps = args.steps if args.steps is not None else DEFAULT_STEPS[args.workload][0]
    warmup = args.warmup if args.warmup is not None else DEFAULT_STEPS[args.workload][1]
    m = time_workload(args.workload, args, rank, world, steps, warmup)
    eng = get_engine()
    weak = None
    if world > 1 and args.scaling == 'strong' and args.workload == 'C4' and not args.no_extra:
        # the second triple of an N > 1 line: WEAK scaling -- 250 000 cells x 200 samples per rank (a rank's block of C4
        # on eight GPUs), so that one run shows both what a fixed problem gains from N GPUs and what a fixed block costs
        # beside N - 1 others (every rank takes part; rank 0 reports)
        import copy
        wargs = copy.copy(args)
        wargs.scaling = 'weak'
        st_, wu_ = DEFAULT_STEPS['C4_block8']
        try:
            weak = time_workload('C4_block8', wargs, rank, world, st_, wu_)
        except Exception as e:                      # noqa: BLE001 - the weak triple never takes the headline down
            weak = dict(error=repr(e))

    if world > 1 or args.force_dist:
        # every rank pushes out what its runtime libraries buffered (RCCL's version banner) before
        # rank 0 goes on to print the result line: nothing from another rank can land after it
        sys.stderr.flush()
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        from cna_amd import dist
        dist.barrier()
    if rank != 0:
        return

    main_sum = summary(m, world, steps)
    extra = {}
    if weak is not None:
        if 'error' in weak:
            extra['C4_block8_weak'] = weak
        else:
            s_ = summary(weak, world, weak['steps'])
            extra['C4_block8_weak'] = dict(workload=workload_text(weak, world, args), steps=weak['steps'], warmup=weak['warmup'],
                                           scaling='weak', cells=weak['n'], **s_)
            del weak['data'], weak['meta']
    if world == 1 and not args.no_extra and args.workload in ('C4', 'C2'):
        # the same steps as a drop-in caller gets them: no engine.pin_graph (same dataset object, resident graph)
        uname = args.workload + '_unpinned'
        st_, wu_ = DEFAULT_STEPS[uname]
        try:
            mm = time_workload(uname, args, rank, world, st_, wu_, dataset=(m['data'], m['meta']))
            extra[uname] = dict(workload=workload_text(mm, world, args), steps=st_, warmup=wu_, **summary(mm, world, st_))
            del mm
        except Exception as e:                      # noqa: BLE001
            extra[uname] = dict(error=repr(e))

    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_reference_cost(synth, min(args.cpu_sample_cells, m['n']), m['N'], m['k'], m['nsteps'], m['Nnull'], m['n_covs'],
                                 extrapolate_to=m['n'])
    del m['data'], m['meta']

    if world == 1 and not args.no_extra and args.workload == 'C4':
        for name in ('C5', 'C3', 'C2', 'C3_default_nsteps', 'C3_covs_batches', 'C4_block8', 'C4_state_f32', 'C3_state_f32',
                     'C2_draw_memo', 'C4_block8_draw_memo'):
            st_, wu_ = DEFAULT_STEPS[name]
            try:
                mm = time_workload(name, args, rank, world, st_, wu_)
                s_ = summary(mm, world, st_)
                extra[name] = dict(workload=workload_text(mm, world, args), steps=st_, warmup=wu_, **s_)
                del mm
            except Exception as e:                      # the extra lines never take the headline down
                extra[name] = dict(error=repr(e))

    cpu_c2 = None
    cpu_full = {}
    if world == 1 and not args.no_cpu_baseline and not args.no_cpu_c2_full:
        # BASELINE.md 3 asks for the CPU path on a BASELINE configuration in full, in the same run: configs[1] (C2:
        # 200k x 50, nsteps 3, Nnull 1000), no sampling, no extrapolation -- beside this run's GPU time of the same
        # configuration.  (--cpu-full C2,C3 adds configs[2]: minutes of host time and ~30 GB, not in the default run.)
        want_full = args.cpu_full
        if want_full == 'auto':
            want_full = 'C2'
            try:
                free_gb = os.sysconf('SC_AVPHYS_PAGES') * os.sysconf('SC_PAGE_SIZE') / 2 ** 30
            except (ValueError, OSError):
                free_gb = 0.0
            if usable_cpus() >= 16 and free_gb >= 64 and args.workload == 'C4':
                want_full = 'C2,C3'
        for wname in [w for w in want_full.split(',') if w]:
            if wname not in ('C2', 'C3') or not (args.workload == 'C4' or args.workload == wname):
                continue
            n2, N2, k2, ns2, P2, c2 = WORKLOADS[wname][:6]
            try:
                one = cpu_reference_cost(synth, n2, N2, k2, ns2, P2, c2)
                g = main_sum if args.workload == wname else extra.get(wname, {})
                if g.get('ms_per_step'):
                    one['gpu_ms_per_step'] = g['ms_per_step']
                    one['gpu_over_cpu'] = round(one['seconds'] * 1e3 / g['ms_per_step'], 1)
                    one['same_p_value_as_gpu'] = bool(abs(one['p_value'] - g.get('p_value', -1)) < 1e-12)
            except Exception as e:
                one = dict(error=repr(e))
            cpu_full[wname] = one
        cpu_c2 = cpu_full.get('C2')

    details = assemble_details(m, main_sum, cpu, cpu_c2, extra, world, steps, warmup, args)
    if cpu_full.get('C3'):
        details['cpu_baseline_C3_full'] = cpu_full['C3']
    line = contract_line(details)
    # the full tables (every kernel of every configuration, notes, stage times) go to a side file and to stderr; the
    # ONE line on stdout stays small enough for any consumer (round 4's 20 KB line was not parsed by the driver)
    try:
        with open(args.details, 'w') as f:
            json.dump(details, f, indent=1)
    except OSError as e:
        print('bench.py: could not write %s: %r' % (args.details, e), file=sys.stderr)
    print('bench.py details: ' + json.dumps(details), file=sys.stderr)
    # the JSON line is the last thing this process writes: tear the communicators down first and
    # push out whatever the libraries (RCCL prints a version banner) still hold in C stdio buffers
    try:
        eng.close()
    except Exception:
        pass
    sys.stderr.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.write(line + '\n')
    sys.stdout.flush()              # (a normal exit follows: profilers attached to this process write at exit)


if __name__ == '__main__':
    main()
