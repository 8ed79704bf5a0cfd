#!/usr/bin/env python3
"""Headline benchmark: cells*permutations / second of an end-to-end ``cna.tl.association``
(BASELINE.json metric) on synthetic data, HIP path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C4|C3|C2|C5] [--scaling strong|weak]

A *step* is one full association() call -- NAM diffusion (3 steps) -> QC/selection ->
residualisation -> Gram/SVD -> global permutation test -> fused local null + FDRs ->
data.obs write-back -- on one synthetic dataset, with the graph, its device cell order and the
factorised sample ids resident on the GPU (the steady state of analysing several phenotypes of one
dataset; `engine.pin_graph`), and the walk recomputed every step (NAM cache off).  The cold first call
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
PMC_FILE = os.path.join(ROOT, "profiles", "r05_pmc_traffic.json")

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
}
WORKLOAD_OPTS = {'C4_state_f32': dict(state_f32=True), 'C3_state_f32': dict(state_f32=True)}
DEFAULT_STEPS = {'C4_state_f32': (20, 5), 'C3_state_f32': (20, 5), 'C4_block8': (50, 10), 'C2': (100, 60), 'C3': (20, 5), 'C4': (20, 5), 'C5': (20, 5), 'C3_default_nsteps': (10, 3),
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


def time_workload(name, args, rank, world, steps, warmup, want_kernels=True):
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
    eng.set_state_f32(bool(WORKLOAD_OPTS.get(name, {}).get('state_f32')))      # (back to the default for every other workload)
    eng.pin_graph(get_connectivity(data))        # the bench never edits the graph in place (see module docstring)
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

    eng.prof_reset()
    eng.prof_enable(True)
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        p_last = cna.tl.association(data, y, 'id', **kw)
    sync()
    dt = time.perf_counter() - t0
    eng.prof_enable(False)
    prof = eng.prof()
    if world > 1 or args.force_dist:
        dt = dist.max_over_ranks(dt)               # the slowest rank's clock
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
                t_gen=t_gen, prof=prof, p=p_last, t_adopt=t_adopt, n_loc=n_loc, nnz_loc=nnz_loc, halo=getattr(eng, 'halo', None),
                sharded_inputs=sharded_inputs, data=data, meta=meta, kw=kw, steps=steps, warmup=warmup, comm=eng.comm_info(),
                halo_comm=getattr(eng, 'halo_comm', False), dev_bytes=int(eng.device_bytes()),
                state_f32=bool(WORKLOAD_OPTS.get(name, {}).get('state_f32')))


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
    dom = max((k_ for k_ in kernels if k_ != 'rccl'), key=lambda k_: kernels[k_]['total_ms'])
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
    gpu_ms = sum(v['total_ms'] for v in kernels.values()) / steps
    ms_per_step = m['dt'] / steps * 1e3
    return dict(value=round(m['n'] * m['Nnull'] * steps / m['dt'], 1), ms_per_step=round(ms_per_step, 3), roofline=roofline,
                kernels=kernels, gpu_kernel_ms_per_step=round(gpu_ms, 3),
                host_ms_per_step=round(ms_per_step - gpu_ms, 3),
                cold_first_call=dict(ms=round(m['t_cold'] * 1e3, 1), value=round(m['n'] * m['Nnull'] / m['t_cold'], 1),
                                     call_that_adopts_the_device_order_ms=None if m.get('t_adopt') is None else round(m['t_adopt'] * 1e3, 1),
                                     note='first call: graph H2D over PCIe (in the caller\'s cell order when the graph is large: '
                                          'the device order is computed on a host thread beside it and adopted by a later call, '
                                          'whose time -- the resident graph renumbered on the device -- is listed too), column sums, first-use allocations, one analysis'),
                dataset_gen_s=round(m['t_gen'], 1), p_value=m['p'])


def workload_text(m, world, args):
    return ('%s: %d cells (%d per GPU) x %d samples, k=%d kNN (%.1f nnz/row, float32 CSR), nsteps=%s, Nnull=%d%s%s, '
            'local FDR pass on; graph + device cell order + sample codes resident (graph pinned: engine.pin_graph), '
            'walk recomputed every step (NAM cache off%s)' % (
                m['name'], m['n'], -(-m['n'] // world), m['N'], m['k'], m['nnz'] / m['n'],
                'None (the reference\'s stop rule)' if m['nsteps'] is None else str(m['nsteps']), m['Nnull'],
                ', %d covariates' % m['n_covs'] if m['n_covs'] else '', ', %d batches' % m['n_batches'] if m.get('n_batches') else '',
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
            'graph resident, walk recomputed every step' % (
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
    ap.add_argument('--cpu-full', default='C2', help='N=1: BASELINE configurations whose FULL workload the reference-cost CPU path runs (C2, C3)')
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

    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_reference_cost(synth, min(args.cpu_sample_cells, m['n']), m['N'], m['k'], m['nsteps'], m['Nnull'], m['n_covs'],
                                 extrapolate_to=m['n'])
    del m['data'], m['meta']

    extra = {}
    if world == 1 and not args.no_extra and args.workload == 'C4':
        for name in ('C5', 'C3', 'C2', 'C3_default_nsteps', 'C3_covs_batches', 'C4_block8', 'C4_state_f32', 'C3_state_f32'):
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
        for wname in [w for w in args.cpu_full.split(',') if w]:
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
