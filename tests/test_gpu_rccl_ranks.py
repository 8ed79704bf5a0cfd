"""More than one RCCL rank, on ONE GPU (-m gpu).

RCCL refuses two ranks of a communicator on the same device ("Duplicate GPU detected") by comparing host hash and bus
id; with NCCL_HOSTID every rank claims a host of its own, the check does not apply and the ranks reach each other
through the library's socket transport over the loopback interface.  Slow (host-staged, one GPU time-shared) and
exactly what is wanted here: the code the shared-memory test communicator of tests/test_gpu_multirank.py stands in for
-- ncclCommInitRank with n > 1, the ncclCommSplit duplicate for the halo stream, the timed start-up self-test, grouped
ncclSend / ncclRecv of halo rows beside the walk of the interior rows, the real all-reduces and all-gathers, and
cna_amd.dist.init_from_env's hand-over of the id -- runs end to end against the golden vectors of the reference."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rank_env(rank, world, port, overlap=True):
    os.environ.update(RANK=str(rank), LOCAL_RANK='0', WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), NCCL_HOSTID='cna_test_host_%d' % rank,
                      NCCL_SOCKET_IFNAME='lo', NCCL_IB_DISABLE='1', HSA_ENABLE_IPC_MODE_LEGACY='0',
                      CNA_COMM_TIMEOUT='60', CNA_HALO_OVERLAP='1' if overlap else '0')
    os.environ.pop('TORCHELASTIC_RUN_ID', None)


def _worker(rank, world, port, name, partition, overlap, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import warnings
    warnings.simplefilter('ignore')
    _rank_env(rank, world, port, overlap)
    try:
        import cna_amd as cna
        from cna_amd import dist
        from cna_amd.engine import get_engine
        from helpers import load_case
        case = load_case(name)
        assert dist.init_from_env() == (rank, world)           # rank 0's id over the Unix-domain socket
        eng = get_engine()
        part = dist.shard(case['data'], rank, world, partition='always' if partition else None)
        res = cna.tl.association(part, case['y'], case['sid_name'], batches=case['batches'], covs=case['covs'],
                                 donorids=case['donorids'], return_full=True, **case['call'])
        out = dict(p=res.p, k=int(res.k), kept=res.kept, num=res.fdrs.num_detected.values, fdr=res.fdrs.fdr.values,
                   cells=list(part.obs.index), coef=part.obs['coef'].values, coef_fdr=part.obs['coef_fdr'].values,
                   ncorrs=res.ncorrs.values, nam=res.nam.values, nam_cells=list(res.nam.columns),
                   namresid=res.namresid.values, varexp=res.namresid_varexp.values, view=eng.view_local,
                   halo=eng.halo, halo_comm=eng.halo_comm, comm=eng.comm_info())
        # a second phenotype on the resident shard: sample memo and NAM cache agreed on by all ranks
        y2 = case['y'].copy()
        y2[:] = np.random.RandomState(5).randn(len(y2))
        res2 = cna.tl.association(part, y2, case['sid_name'], batches=case['batches'], covs=case['covs'],
                                  donorids=case['donorids'], return_full=True, **case['call'])
        from cna_amd.tools import _fast
        out['two_call_path'] = dict(_fast.stats)     # (the second phenotype of a shaped call goes through cna_assoc_begin / _finish)
        out['p2'], out['ncorrs2'], out['cells2'] = res2.p, res2.ncorrs.values, list(res2.ncorrs.index)
        s0 = np.random.RandomState(1).rand(case['data'].obsp['connectivities'].shape[0], 3)
        order = part.uns['cna_shard'].get('order')
        r0 = part.uns['cna_shard']['row0']
        mine = np.arange(r0, r0 + len(part.obs)) if order is None else order[r0:r0 + len(part.obs)]
        out['diffuse'] = cna.tl.diffuse(part, s0[mine], 2)
        dist.barrier()
        eng.close()
        q.put((rank, out))
    except BaseException as e:
        import traceback
        q.put((rank, 'ERROR %r\n%s' % (e, traceback.format_exc())))
        q.close()
        q.join_thread()
        os._exit(1)


def _one_gpu(name, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import warnings
    warnings.simplefilter('ignore')
    import cna_amd as cna
    from cna_amd.engine import Engine
    from helpers import load_case
    case = load_case(name)
    eng = Engine(device=0)
    y2 = case['y'].copy()
    y2[:] = np.random.RandomState(5).randn(len(y2))
    res2 = cna.tl.association(case['data'], y2, case['sid_name'], batches=case['batches'], covs=case['covs'],
                              donorids=case['donorids'], return_full=True, engine=eng, **case['call'])
    q.put((-1, dict(p2=res2.p, ncorrs2=res2.ncorrs.values, cells=list(res2.ncorrs.index))))
    eng.close()


def _skip_if_no_transport(out):
    """The ranks reach each other through RCCL's socket transport over `lo`: a box on which the communicator cannot even
    be created (no loopback interface) cannot run these tests -- anything after that point is a failure."""
    if isinstance(out, str) and 'cna_comm_init failed' in out:
        pytest.skip('RCCL could not create a communicator between ranks on this box: ' + out.splitlines()[0][:200])


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.mark.parametrize('name,world,partition,overlap', [
    ('c01_plain_f32', 2, None, True), ('c12_batchy_qc', 3, True, True), ('c03_covs_batches', 4, True, True),
    ('c13_zero_variance', 2, None, False), ('c11_string_ids_null_y', 2, True, True),
    # the walk's stop rule and the batch-kurtosis loop take medians over the cells of all ranks; a float64 graph
    ('c02_covs_autostop', 2, True, True), ('c05_ks_f64', 3, None, True), ('c16_ridge_loop', 2, True, False),
    ('c14_selfweight_autostop_unsorted', 4, True, True)])
def test_rccl_ranks_sharded_inputs(name, world, partition, overlap):
    """Every rank: its own block of cells (contiguous, or whole populations: partition=True), real RCCL between the ranks.
    Per-cell results matched by cell name and sample-level results are the reference's; every rank reports the same
    sample-level results; a second phenotype on the resident shards equals what one GPU holding everything computes."""
    import multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from helpers import load_case, relerr
    from oracle import cna_oracle as orc
    import scipy.sparse as sp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, partition, overlap, q)) for r in range(world)]
    procs.append(ctx.Process(target=_one_gpu, args=(name, q)))
    for p in procs[:-1]:
        p.start()
    got = {}
    try:
        for _ in range(world):
            r, out = q.get(timeout=300)
            _skip_if_no_transport(out)
            assert not isinstance(out, str), out
            got[r] = out
        procs[-1].start()
        one = q.get(timeout=240)[1]
    finally:
        for p in procs:
            if p.pid is not None:
                p.join(timeout=30)
                if p.is_alive():
                    p.kill()
    case = load_case(name)
    z = case['z']
    names = list(case['data'].obs.index)
    where = {c: i for i, c in enumerate(names)}
    n = len(names)
    a = got[0]
    coef = np.full(n, np.nan)
    cfdr = np.full(n, np.nan)
    kept = np.zeros(n, dtype=bool)
    nam = np.full((z['nam'].shape[0], n), np.nan)
    namresid = np.full((z['namresid'].shape[0], n), np.nan)
    ncorrs = np.full(n, np.nan)
    ncorrs2 = np.full(n, np.nan)
    diffuse = np.full((n, 3), np.nan)
    seen = 0
    for r in range(world):
        g = got[r]
        assert g['view'] and tuple(g['comm']) == ('rccl', world)
        # one batch, a seed (nsteps given or the stop rule): the second call (resident shard) goes through cna_assoc_begin / _finish on every rank
        assert g['two_call_path']['taken'] == (1 if name in ('c01_plain_f32', 'c11_string_ids_null_y', 'c05_ks_f64', 'c13_zero_variance', 'c02_covs_autostop',
                                                          'c14_selfweight_autostop_unsorted') else 0), g['two_call_path']
        assert g['halo_comm'], 'the halo communicator did not pass the start-up self-test'
        assert g['halo'] is not None and g['halo'][0] > 0 and g['halo'][1] > 0
        assert g['p'] == a['p'] and g['k'] == a['k'] and g['p2'] == a['p2']
        for key in ('fdr', 'num', 'varexp'):
            np.testing.assert_array_equal(g[key], a[key])
        idx = np.array([where[c] for c in g['cells']], dtype=np.int64)
        seen += len(idx)
        coef[idx], cfdr[idx], kept[idx] = g['coef'], g['coef_fdr'], g['kept']
        diffuse[idx] = g['diffuse']
        cols = [where[c] for c in g['nam_cells']]
        nam[:, cols] = g['nam']
        namresid[:, cols] = g['namresid']
        ncorrs[cols] = g['ncorrs']
        ncorrs2[[where[c] for c in g['cells2']]] = g['ncorrs2']
    assert seen == n
    assert a['k'] == int(z['k']) and a['p'] == pytest.approx(float(z['p']), rel=1e-12)
    assert np.array_equal(kept, z['kept'])
    T = min(len(a['num']), len(z['fdr_num_detected']))
    assert np.array_equal(a['num'][:T], z['fdr_num_detected'][:T])
    assert relerr(a['fdr'][:T], z['fdr_fdr'][:T]) < 1e-4
    assert np.array_equal(np.isnan(coef), np.isnan(z['obs_coef']))
    assert relerr(coef[~np.isnan(coef)], z['obs_coef'][~np.isnan(coef)]) < 1e-5
    np.testing.assert_allclose(cfdr, z['obs_coef_fdr'], rtol=1e-4, atol=1e-12)
    assert relerr(nam[:, z['kept']], z['nam']) < 1e-5
    assert relerr(namresid[:, z['kept']], z['namresid']) < 1e-5
    assert relerr(ncorrs[z['kept']], z['ncorrs']) < 1e-5
    A = sp.csr_matrix(case['data'].obsp['connectivities'])
    s0 = np.random.RandomState(1).rand(A.shape[0], 3)
    assert relerr(diffuse, orc.diffuse(A, s0, 2, mode='f64')) < 1e-12
    # second phenotype: what one GPU holding everything computes
    assert a['p2'] == pytest.approx(one['p2'], rel=1e-12)
    one_nc = np.full(n, np.nan)
    one_nc[[where[c] for c in one['cells']]] = one['ncorrs2']
    np.testing.assert_allclose(ncorrs2, one_nc, rtol=1e-9, atol=1e-13, equal_nan=True)


def _dead_peer_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    _rank_env(rank, world, port)
    os.environ['CNA_COMM_TIMEOUT'] = '8'
    try:
        from cna_amd import dist
        if rank == 1:
            # takes the id, joins the communicator and then never answers: stopped in front of the self-test
            import signal
            import cna_amd._ffi as F
            lib = F.load()
            real = lib.cna_comm_selftest

            def stall(*args):
                os.kill(os.getpid(), signal.SIGSTOP)
                return real(*args)
            lib.cna_comm_selftest = stall
        dist.init_from_env()
        from cna_amd.engine import get_engine
        get_engine()
        q.put((rank, 'engine created'))
    except BaseException as e:
        q.put((rank, 'ERROR %s: %s' % (type(e).__name__, e)))
        q.close()
        q.join_thread()
        os._exit(1)


def test_rccl_selftest_reports_a_silent_peer():
    """A rank whose peer joined the communicator and then stopped answering gets an error from the start-up self-test
    within CNA_COMM_TIMEOUT seconds -- not a hang in the first collective of an analysis."""
    import multiprocessing as mp
    import signal
    import time
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dead_peer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    t0 = time.time()
    try:
        r, msg = q.get(timeout=120)
        took = time.time() - t0
        assert r == 0 and msg.startswith('ERROR') and 'cna_comm_selftest' in msg, msg
        assert took < 90
    finally:
        for p in procs:
            if p.is_alive():
                try:
                    os.kill(p.pid, signal.SIGCONT)
                except OSError:
                    pass
                p.kill()
            p.join(timeout=10)


def _replicated_worker(rank, world, port, name, halo, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import warnings
    warnings.simplefilter('ignore')
    _rank_env(rank, world, port)
    os.environ['CNA_HALO'] = '1' if halo else '0'
    try:
        import cna_amd as cna
        from cna_amd import dist
        from cna_amd.engine import get_engine
        from helpers import load_case
        case = load_case(name)
        dist.init_from_env()
        eng = get_engine()
        res = cna.tl.association(case['data'], case['y'], case['sid_name'], batches=case['batches'],
                                 covs=case['covs'], donorids=case['donorids'], return_full=True, **case['call'])
        out = dict(p=res.p, k=int(res.k), ncorrs=res.ncorrs.values, kept=res.kept, fdr=res.fdrs.fdr.values,
                   num=res.fdrs.num_detected.values, coef=case['data'].obs['coef'].values,
                   coef_fdr=case['data'].obs['coef_fdr'].values, nam=res.nam.values, namresid=res.namresid.values,
                   V=res.namresid_nbhdXpc.values, rows=(eng.row0, eng.n_local), halo=eng.halo, comm=eng.comm_info())
        s0 = np.random.RandomState(1).rand(case['data'].obsp['connectivities'].shape[0], 3)
        out['diffuse'] = cna.tl.diffuse(case['data'], s0, 2)
        NAM, keep = cna.tl.nam(case['data'], case['sid_name'], batches=case['batches'])
        out['tlnam'], out['tlkeep'] = NAM.values, keep
        dist.barrier()
        eng.close()
        q.put((rank, out))
    except BaseException as e:
        import traceback
        q.put((rank, 'ERROR %r\n%s' % (e, traceback.format_exc())))
        q.close()
        q.join_thread()
        os._exit(1)


@pytest.mark.parametrize('name,world,halo', [('c12_batchy_qc', 2, True), ('c03_covs_batches', 3, False)])
def test_rccl_ranks_replicated_inputs(name, world, halo):
    """The drop-in convention under real RCCL: every rank passes the whole dataset, owns a block of rows on the device
    and returns results for all cells (all-gathers of the per-cell results, the unpermuting all-reduce); with the halo
    exchange and with the all-gather of the state (CNA_HALO=0).  Reference results, the same on every rank."""
    import multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from helpers import load_case, relerr
    from oracle import cna_oracle as orc
    import scipy.sparse as sp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_replicated_worker, args=(r, world, port, name, halo, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    try:
        for _ in range(world):
            r, out = q.get(timeout=300)
            _skip_if_no_transport(out)
            assert not isinstance(out, str), out
            got[r] = out
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    z = load_case(name)['z']
    a = got[0]
    n = len(z['kept'])
    rpr = -(-n // world)
    for r in range(world):
        g = got[r]
        assert tuple(g['comm']) == ('rccl', world)
        assert g['rows'] == (min(r * rpr, n), max(0, min(rpr, n - r * rpr)))
        assert (g['halo'] is not None) == halo
        assert g['p'] == a['p'] and g['k'] == a['k']
        for key in ('ncorrs', 'kept', 'fdr', 'num', 'coef', 'coef_fdr', 'nam', 'namresid', 'V', 'diffuse', 'tlnam', 'tlkeep'):
            np.testing.assert_array_equal(g[key], a[key], err_msg=key)
    assert a['k'] == int(z['k']) and a['p'] == pytest.approx(float(z['p']), rel=1e-12)
    assert np.array_equal(a['kept'], z['kept'])
    assert relerr(a['ncorrs'], z['ncorrs']) < 1e-5
    assert relerr(a['nam'], z['nam']) < 1e-5 and relerr(a['namresid'], z['namresid']) < 1e-5
    T = min(len(a['fdr']), len(z['fdr_fdr']))
    assert np.array_equal(a['num'][:T], z['fdr_num_detected'][:T])
    assert relerr(a['fdr'][:T], z['fdr_fdr'][:T]) < 1e-4
    assert np.array_equal(np.isnan(a['coef']), np.isnan(z['obs_coef']))
    assert relerr(a['coef'][~np.isnan(a['coef'])], z['obs_coef'][~np.isnan(a['coef'])]) < 1e-5
    case = load_case(name)
    A = sp.csr_matrix(case['data'].obsp['connectivities'])
    s0 = np.random.RandomState(1).rand(A.shape[0], 3)
    assert relerr(a['diffuse'], orc.diffuse(A, s0, 2, mode='f64')) < 1e-12
    assert np.array_equal(a['tlkeep'], z['kept'])
