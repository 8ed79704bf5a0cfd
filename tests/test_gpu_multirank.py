"""Multi-rank HIP path on ONE GPU (-m gpu): G processes, each with its own context on device 0,
cells sharded by row blocks, the library's host-staged shared-memory communicator in place of RCCL
(which refuses two ranks per GPU unless told they are on different hosts: tests/test_gpu_rccl_ranks.py does that).  Everything else is the production path: the C ABI's block
offsets, ragged gathers, halo pack / exchange / unpack, the unpermuting all-reduce, and the Python
host code on every rank.  Results must equal the golden vectors of the reference and be identical
on all ranks."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, seg, name, halo, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import warnings
    warnings.simplefilter('ignore')
    os.environ['CNA_HALO'] = '1' if halo else '0'
    try:
        import cna_amd as cna
        from cna_amd.engine import Engine
        from helpers import load_case
        case = load_case(name)
        eng = Engine(device=0, rank=rank, nranks=world, shm=(seg, 8 << 20))
        res = cna.tl.association(case['data'], case['y'], case['sid_name'], batches=case['batches'],
                                 covs=case['covs'], donorids=case['donorids'], return_full=True, engine=eng,
                                 **case['call'])
        out = dict(p=res.p, k=int(res.k), ncorrs=res.ncorrs.values, kept=res.kept, fdr=res.fdrs.fdr.values,
                   num=res.fdrs.num_detected.values, coef=case['data'].obs['coef'].values,
                   coef_fdr=case['data'].obs['coef_fdr'].values, nam=res.nam.values,
                   namresid=res.namresid.values, V=res.namresid_nbhdXpc.values, rows=(eng.row0, eng.n_local),
                   halo=eng.halo, perm=eng.perm is not None)
        import scipy.sparse as sp
        A = sp.csr_matrix(case['data'].obsp['connectivities'])
        s0 = np.random.RandomState(1).rand(A.shape[0], 3)
        out['diffuse'] = cna.tl.diffuse(case['data'], s0, 2, engine=eng)
        NAM, keep = cna.tl.nam(case['data'], case['sid_name'], batches=case['batches'], engine=eng)
        out['tlnam'] = NAM.values
        out['tlkeep'] = keep
        eng.close()
        q.put((rank, out))
    except BaseException as e:     # report instead of leaving the other ranks in a barrier forever
        import traceback
        q.put((rank, 'ERROR %r\n%s' % (e, traceback.format_exc())))
        q.close()
        q.join_thread()                                    # the report must be out before the process is
        os._exit(1)


@pytest.mark.parametrize('name,world,halo', [('c12_batchy_qc', 2, True), ('c03_covs_batches', 3, True),
                                             ('c13_zero_variance', 2, False), ('c01_plain_f32', 4, True),
                                             # messy sample-level inputs (orders of their own, NaNs, an unused category)
                                             ('f02_messy', 2, True), ('f12_messy', 3, True), ('f06_messy', 2, False),
                                             ('c19_unused_category_batches', 2, True)])
def test_sharded_hip_path_on_one_gpu(name, world, halo):
    import multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from helpers import load_case, relerr
    from oracle import cna_oracle as orc
    import scipy.sparse as sp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    seg = 'cna_test_%d_%s' % (os.getpid(), name[:3])
    procs = [ctx.Process(target=_worker, args=(r, world, seg, name, halo, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    try:
        for _ in range(world):
            r, out = q.get(timeout=240)
            assert not isinstance(out, str), out
            got[r] = out
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    case = load_case(name)
    z = case['z']
    n = len(z['kept'])
    rpr = -(-n // world)
    a = got[0]
    assert a['perm']
    for r in range(world):
        assert got[r]['rows'] == (min(r * rpr, n), max(0, min(rpr, n - r * rpr)))
        assert (got[r]['halo'] is not None) == halo
        for key in ('ncorrs', 'kept', 'fdr', 'num', 'coef', 'coef_fdr', 'nam', 'namresid', 'diffuse', 'tlnam', 'tlkeep'):
            np.testing.assert_array_equal(got[r][key], a[key])      # every rank holds the full result
        assert got[r]['p'] == a['p'] and got[r]['k'] == a['k']
    if halo:
        assert sum(got[r]['halo'][0] for r in range(world)) == sum(got[r]['halo'][1] for r in range(world)) > 0
    assert a['k'] == int(z['k']) and a['p'] == pytest.approx(float(z['p']), rel=1e-12)
    assert np.array_equal(a['kept'], z['kept'])
    assert relerr(a['ncorrs'], z['ncorrs']) < 1e-5
    assert relerr(a['nam'], z['nam']) < 1e-5 and relerr(a['namresid'], z['namresid']) < 1e-5
    T = min(len(a['fdr']), len(z['fdr_fdr']))
    assert np.array_equal(a['num'][:T], z['fdr_num_detected'][:T])
    assert relerr(a['fdr'][:T], z['fdr_fdr'][:T]) < 1e-4
    assert np.array_equal(np.isnan(a['coef']), np.isnan(z['obs_coef']))
    ok = ~np.isnan(a['coef'])
    assert relerr(a['coef'][ok], z['obs_coef'][ok]) < 1e-5
    np.testing.assert_allclose(a['coef_fdr'], z['obs_coef_fdr'], rtol=1e-4, atol=1e-12)
    A = sp.csr_matrix(case['data'].obsp['connectivities'])
    s0 = np.random.RandomState(1).rand(A.shape[0], 3)
    assert relerr(a['diffuse'], orc.diffuse(A, s0, 2, mode='f64')) < 1e-13
    # signs of the PCs are LAPACK's on every rank; V only has to be consistent with U there
    assert a['V'].shape == z['V'].shape


def _fuzz_worker(rank, world, seg, seed, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import warnings
    warnings.simplefilter('ignore')
    try:
        import cna_amd as cna
        from cna_amd import synth
        from cna_amd.engine import Engine
        from test_gpu_parity import _fuzz_config
        cfg, call = _fuzz_config(seed)
        data, meta = synth.make_dataset(cfg['n'], cfg['N'], k=cfg['k'], seed=seed, n_covs=cfg['n_covs'],
                                        n_batches=cfg['n_batches'], graph_dtype=np.dtype(cfg['graph_dtype']).type,
                                        sid_kind=cfg['sid_kind'], cluster_sorted=cfg['cluster_sorted'], signal=cfg['signal'])
        eng = Engine(device=0, rank=rank, nranks=world, shm=(seg, 8 << 20)) if world > 1 else Engine(device=0)
        res = cna.tl.association(data, meta['y'], 'id', covs=meta['covs'], batches=meta['batches'], return_full=True,
                                 allow_low_sample_size=True, engine=eng, **call)
        out = dict(p=res.p, k=int(res.k), kept=res.kept, ncorrs=res.ncorrs.values, fdr=res.fdrs.fdr.values,
                   num=res.fdrs.num_detected.values, coef=data.obs['coef'].values, coef_fdr=data.obs['coef_fdr'].values,
                   nam=res.nam.values)
        eng.close()
        q.put((rank, out))
    except BaseException as e:
        import traceback
        q.put((rank, 'ERROR %r\n%s' % (e, traceback.format_exc())))
        q.close()
        q.join_thread()                                    # the report must be out before the process is
        os._exit(1)


@pytest.mark.parametrize('seed,world', [(3, 2), (12, 3), (20, 2), (15, 4)])
def test_sharded_random_configurations_equal_single_gpu(seed, world):
    """Configurations of the random sweep (QC dropping cells, batches with the ridge schedule,
    covariates) run sharded over `world` ranks on one GPU: every rank must hold what one GPU computes
    -- integers and masks exactly, floats to rounding (sums over cells are split differently)."""
    import multiprocessing as mp
    ctx = mp.get_context('spawn')
    results = {}
    for w in (1, world):
        q = ctx.Queue()
        seg = 'cna_fz_%d_%d_%d' % (os.getpid(), seed, w)
        procs = [ctx.Process(target=_fuzz_worker, args=(r, w, seg, seed, q)) for r in range(w)]
        for p in procs:
            p.start()
        got = {}
        try:
            for _ in range(w):
                r, out = q.get(timeout=240)
                assert not isinstance(out, str), out
                got[r] = out
        finally:
            for p in procs:
                p.join(timeout=30)
                if p.is_alive():
                    p.kill()
        results[w] = got
    one = results[1][0]
    for r in range(world):
        g = results[world][r]
        assert g['k'] == one['k'] and np.array_equal(g['kept'], one['kept']) and np.array_equal(g['num'], one['num'])
        assert g['p'] == pytest.approx(one['p'], rel=1e-12)
        # column sums of a float64 graph are added in a different order when sharded (float32 graphs: exact)
        np.testing.assert_allclose(g['nam'], one['nam'], rtol=1e-13, atol=0)
        np.testing.assert_allclose(g['ncorrs'], one['ncorrs'], rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(g['fdr'], one['fdr'], rtol=1e-9, atol=1e-13, equal_nan=True)
        np.testing.assert_allclose(g['coef'], one['coef'], rtol=1e-9, atol=1e-13, equal_nan=True)
        np.testing.assert_allclose(g['coef_fdr'], one['coef_fdr'], rtol=1e-9, atol=1e-13)


def _sparse_worker(rank, world, seg, q):
    sys.path.insert(0, ROOT)
    import warnings
    warnings.simplefilter('ignore')
    try:
        import cna_amd as cna
        from cna_amd import synth
        from cna_amd.engine import Engine
        data, meta = synth.make_dataset(9000, 120, k=15, seed=31)
        eng = Engine(device=0, rank=rank, nranks=world, shm=(seg, 32 << 20)) if world > 1 else Engine(device=0)
        eng.prof_enable(True)
        res = cna.tl.association(data, meta['y'], 'id', nsteps=3, Nnull=100, seed=2, return_full=True, engine=eng)
        eng.sync()
        prof = eng.prof()
        out = dict(p=res.p, k=int(res.k), ncorrs=res.ncorrs.values, halo=getattr(eng, 'halo', None),
                   sparse=prof.get('nam_step_sparse', (0, 0))[1], dense=prof.get('nam_step', (0, 0))[1],
                   select=prof.get('select', (0, 0))[1], fdr=res.fdrs.fdr.values, num=res.fdrs.num_detected.values)
        out['nam'] = res.nam.values           # (last: with the selection by-product the raw NAM is one more dense launch)
        out['bytes'] = eng.device_bytes()
        out['rows'] = (eng.row0, eng.n_local)
        eng.close()
        q.put((rank, out))
    except BaseException as e:
        import traceback
        q.put((rank, 'ERROR %r\n%s' % (e, traceback.format_exc())))
        q.close()
        q.join_thread()                                    # the report must be out before the process is
        os._exit(1)


@pytest.mark.parametrize('world,defer', [(2, False), (3, False), (2, True)])
def test_compressed_second_step_across_ranks(world, defer, monkeypatch):
    """120 samples: the second walk step gathers (sample, value) pairs instead of dense rows (k_nam_step_sparse).
    Sharded, a rank has the pairs of its own rows only; the rows the halo exchange brings arrive dense and are
    marked so -- the step then runs on every rank (one sparse and one dense step, as on one GPU) and the NAM is
    bit for bit the single-GPU NAM.  The sparse step is followed by an exchange, so it walks the rows other ranks have
    asked for first and the rest while they travel (cna_nam_step: two launches)."""
    import multiprocessing as mp
    ctx = mp.get_context('spawn')
    results = {}
    # defer: the schedule of large inputs on every rank (the workers read the variable when they import the package) --
    # the last walk step also does the selection pass, the two counters of that pass cross the ranks in one collective
    monkeypatch.setenv('CNA_DEFER_LAST_CELLS', '0' if defer else '1000000000')
    for w in (1, world):
        q = ctx.Queue()
        seg = 'cna_sp_%d_%d' % (os.getpid(), w)
        procs = [ctx.Process(target=_sparse_worker, args=(r, w, seg, q)) for r in range(w)]
        for p in procs:
            p.start()
        got = {}
        try:
            for _ in range(w):
                r, out = q.get(timeout=240)
                assert not isinstance(out, str), out
                got[r] = out
        finally:
            for p in procs:
                p.join(timeout=30)
                if p.is_alive():
                    p.kill()
        results[w] = got
    one = results[1][0]
    assert one['sparse'] == 1 and one['dense'] == 1
    for r in range(world):
        g = results[world][r]
        assert g['halo'] is not None and g['halo'][1] > 0          # rows do arrive from other ranks
        # (every step that reads a state is two launches: the rows without a foreign neighbour under the exchange that is
        # still in flight, then the rest -- the walk's last step included, since round 5)
        assert g['sparse'] == 2 and g['dense'] == 2, (g['sparse'], g['dense'])
        assert g['select'] == (0 if defer else 1) and one['select'] == (0 if defer else 1)
        np.testing.assert_array_equal(g['nam'], one['nam'])
        np.testing.assert_array_equal(g['num'], one['num'])
        np.testing.assert_allclose(g['fdr'], one['fdr'], rtol=1e-9, atol=1e-13, equal_nan=True)
        assert g['k'] == one['k'] and g['p'] == pytest.approx(one['p'], rel=1e-12)
        np.testing.assert_allclose(g['ncorrs'], one['ncorrs'], rtol=1e-9, atol=1e-13)


@pytest.mark.parametrize('world', [2, 4])
def test_state_for_local_and_halo_rows_only(world, monkeypatch):
    """a rank owns its rows of the graph AND of the state.  With the halo exchange the diffusion state holds
    this rank's rows followed by the rows it receives (csrc/c_api.hip:cna_set_halo, `t_compact`), the walk reads the graph
    through column indices renumbered into that row space, and what arrives lands in the tail directly.  Same bits as the
    global row space (CNA_COMPACT_STATE=0) and as one GPU; and the device memory of a rank shrinks with the block."""
    import multiprocessing as mp
    ctx = mp.get_context('spawn')
    results = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('CNA_COMPACT_STATE', mode)
        q = ctx.Queue()
        seg = 'cna_ct_%d_%s_%d' % (os.getpid(), mode, world)
        procs = [ctx.Process(target=_sparse_worker, args=(r, world, seg, q)) for r in range(world)]
        for p in procs:
            p.start()
        got = {}
        try:
            for _ in range(world):
                r, out = q.get(timeout=240)
                assert not isinstance(out, str), out
                got[r] = out
        finally:
            for p in procs:
                p.join(timeout=30)
                if p.is_alive():
                    p.kill()
        results[mode] = got
    for r in range(world):
        a, b = results['1'][r], results['0'][r]
        assert a['halo'] is not None and a['halo'][1] > 0 and a['halo'] == b['halo']
        np.testing.assert_array_equal(a['nam'], b['nam'])
        np.testing.assert_array_equal(a['num'], b['num'])
        np.testing.assert_array_equal(a['ncorrs'], b['ncorrs'])
        np.testing.assert_array_equal(a['fdr'], b['fdr'])
        assert a['p'] == b['p'] and a['k'] == b['k'] and a['sparse'] == b['sparse'] and a['dense'] == b['dense']
        # the state (two buffers of 8 x 120 bytes per row) and the pairs (1 KB per row) are what a rank holds per cell of
        # the whole problem in the global row space: 9000 rows there, n_local + halo here
        n_state = a['rows'][1] + a['halo'][1]
        saved = b['bytes'] - a['bytes']
        want = (9000 - n_state) * 2 * 8 * 120 + (9000 - a['rows'][1]) * 1024
        assert saved > 0.8 * want, (saved, want, a['bytes'], b['bytes'])


def _shard_worker(rank, world, seg, name, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import warnings
    warnings.simplefilter('ignore')
    try:
        import cna_amd as cna
        from cna_amd import dist
        from cna_amd.engine import Engine
        from helpers import load_case
        case = load_case(name)
        part = dist.shard(case['data'], rank, world)       # this rank's cells only
        eng = Engine(device=0, rank=rank, nranks=world, shm=(seg, 8 << 20))
        res = cna.tl.association(part, case['y'], case['sid_name'], batches=case['batches'], covs=case['covs'],
                                 donorids=case['donorids'], return_full=True, engine=eng, **case['call'])
        out = dict(p=res.p, k=int(res.k), ncorrs=res.ncorrs.values, kept=res.kept, fdr=res.fdrs.fdr.values,
                   num=res.fdrs.num_detected.values, coef=part.obs['coef'].values, coef_fdr=part.obs['coef_fdr'].values,
                   nam=res.nam.values, namresid=res.namresid.values, V=res.namresid_nbhdXpc.values,
                   varexp=res.namresid_varexp.values, n_obs=len(part.obs), view=eng.view_local, halo=eng.halo)
        # a second phenotype on the resident shard (sample memo + NAM cache agreed on by all ranks)
        y2 = case['y'].copy()
        y2[:] = np.random.RandomState(5).randn(len(y2))
        res2 = cna.tl.association(part, y2, case['sid_name'], batches=case['batches'], covs=case['covs'],
                                  donorids=case['donorids'], return_full=True, engine=eng, **case['call'])
        from cna_amd.tools import _fast
        out['two_call_path'] = dict(_fast.stats)     # (the second phenotype of a shaped call goes through cna_assoc_begin / _finish)
        out['p2'], out['ncorrs2'] = res2.p, res2.ncorrs.values
        s0 = np.random.RandomState(1).rand(case['data'].obsp['connectivities'].shape[0], 3)
        r0 = part.uns['cna_shard']['row0']
        out['diffuse'] = cna.tl.diffuse(part, s0[r0:r0 + len(part.obs)], 2, engine=eng)
        NAM, keep = cna.tl.nam(part, case['sid_name'], batches=case['batches'], engine=eng)
        out['tlnam'], out['tlkeep'] = NAM.values, keep
        eng.close()
        q.put((rank, out))
    except BaseException as e:
        import traceback
        q.put((rank, 'ERROR %r\n%s' % (e, traceback.format_exc())))
        q.close()
        q.join_thread()                                    # the report must be out before the process is
        os._exit(1)


def _partition_shard_worker(rank, world, seg, name, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import warnings
    warnings.simplefilter('ignore')
    try:
        import cna_amd as cna
        from cna_amd import dist
        from cna_amd.engine import Engine
        from helpers import load_case
        case = load_case(name)
        part = dist.shard(case['data'], rank, world, partition='always')   # whole populations of the graph per block
        eng = Engine(device=0, rank=rank, nranks=world, shm=(seg, 8 << 20))
        res = cna.tl.association(part, case['y'], case['sid_name'], batches=case['batches'], covs=case['covs'],
                                 donorids=case['donorids'], return_full=True, engine=eng, **case['call'])
        out = dict(p=res.p, k=int(res.k), kept=res.kept, num=res.fdrs.num_detected.values, fdr=res.fdrs.fdr.values,
                   cells=list(part.obs.index), coef=part.obs['coef'].values, coef_fdr=part.obs['coef_fdr'].values,
                   ncorrs=res.ncorrs.values, nam=res.nam.values, nam_cells=list(res.nam.columns),
                   namresid=res.namresid.values, order=part.uns['cna_shard']['order'], view=eng.view_local,
                   halo=eng.halo)
        eng.close()
        q.put((rank, out))
    except BaseException as e:
        import traceback
        q.put((rank, 'ERROR %r\n%s' % (e, traceback.format_exc())))
        q.close()
        q.join_thread()                                    # the report must be out before the process is
        os._exit(1)


def _one_gpu_worker(name, q):
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
    q.put((0, dict(p2=res2.p, ncorrs2=res2.ncorrs.values)))
    eng.close()


@pytest.mark.parametrize('name,world', [('c01_plain_f32', 2), ('c12_batchy_qc', 3), ('c13_zero_variance', 2),
                                        ('c03_covs_batches', 4), ('c11_string_ids_null_y', 2)])
def test_sharded_inputs_on_one_gpu(name, world):
    """Sharded callers (cna_amd.dist.shard): each rank is handed its own block of cells and nothing
    else -- obs rows, graph rows with global column ids -- and gets per-cell results for that block;
    no cells-sized vector crosses ranks (cna_set_local_view).  Stitched in rank order the pieces are
    the reference's results; sample-level results agree on every rank."""
    import multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from helpers import load_case, relerr
    from oracle import cna_oracle as orc
    import scipy.sparse as sp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    seg = 'cna_sh_%d_%s' % (os.getpid(), name[:3])
    procs = [ctx.Process(target=_shard_worker, args=(r, world, seg, name, q)) for r in range(world)]
    procs.append(ctx.Process(target=_one_gpu_worker, args=(name, q)))
    for p in procs[:-1]:
        p.start()
    got = {}
    try:
        for _ in range(world):
            r, out = q.get(timeout=240)
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
    n = len(z['kept'])
    rpr = -(-n // world)
    parts = [got[r] for r in range(world)]
    a = parts[0]
    for r, g in enumerate(parts):
        assert g['view'] and g['n_obs'] == max(0, min(rpr, n - r * rpr)) == len(g['kept']) == len(g['coef'])
        # nsteps given, one batch, a seed: the second call (resident graph) is the two-call path on every rank (c13: the second
        # phenotype has no NaN, so the sample whose absence left cells of zero variance is back)
        assert g['two_call_path']['taken'] == (1 if name in ('c01_plain_f32', 'c11_string_ids_null_y', 'c13_zero_variance') else 0), g['two_call_path']
        assert g['p'] == a['p'] and g['k'] == a['k'] and g['p2'] == a['p2']
        for key in ('fdr', 'num', 'varexp'):
            np.testing.assert_array_equal(g[key], a[key])
    cat = lambda key, axis=0: np.concatenate([g[key] for g in parts], axis=axis)
    assert a['k'] == int(z['k']) and a['p'] == pytest.approx(float(z['p']), rel=1e-12)
    assert np.array_equal(cat('kept'), z['kept'])
    assert relerr(cat('ncorrs'), z['ncorrs']) < 1e-5
    assert relerr(cat('nam', 1), z['nam']) < 1e-5 and relerr(cat('namresid', 1), z['namresid']) < 1e-5
    assert cat('V').shape == z['V'].shape
    T = min(len(a['fdr']), len(z['fdr_fdr']))
    assert np.array_equal(a['num'][:T], z['fdr_num_detected'][:T])
    assert relerr(a['fdr'][:T], z['fdr_fdr'][:T]) < 1e-4
    coef = cat('coef')
    assert np.array_equal(np.isnan(coef), np.isnan(z['obs_coef']))
    assert relerr(coef[~np.isnan(coef)], z['obs_coef'][~np.isnan(coef)]) < 1e-5
    np.testing.assert_allclose(cat('coef_fdr'), z['obs_coef_fdr'], rtol=1e-4, atol=1e-12)
    A = sp.csr_matrix(case['data'].obsp['connectivities'])
    s0 = np.random.RandomState(1).rand(A.shape[0], 3)
    assert relerr(cat('diffuse'), orc.diffuse(A, s0, 2, mode='f64')) < 1e-13
    assert cat('tlkeep').shape == (n,) and cat('tlnam', 1).shape[0] == a['tlnam'].shape[0]
    # second phenotype: equal to what one GPU holding everything computes
    assert a['p2'] == pytest.approx(one['p2'], rel=1e-12)
    np.testing.assert_allclose(cat('ncorrs2'), one['ncorrs2'], rtol=1e-9, atol=1e-13)


@pytest.mark.parametrize('name,world', [('c01_plain_f32', 2), ('c12_batchy_qc', 4), ('c03_covs_batches', 3)])
def test_sharded_inputs_partitioned_by_population_on_one_gpu(name, world):
    """dist.shard(..., partition='always') through the HIP path: each rank's block is made of whole populations of the
    graph (_order.partition_order), the cells renumbered accordingly.  Per-cell results, matched by cell name, and the
    sample-level results are the reference's (integers exact, floats to 1e-5: the column sums add their rows in
    another order); every rank reports the same sample-level results.  The gloo twin of this test is
    tests/test_sharded_gloo.py::test_sharded_inputs_partitioned_by_population."""
    import multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from helpers import load_case, relerr
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    seg = 'cna_pp_%d_%s' % (os.getpid(), name[:3])
    procs = [ctx.Process(target=_partition_shard_worker, args=(r, world, seg, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    try:
        for _ in range(world):
            r, out = q.get(timeout=240)
            assert not isinstance(out, str), out
            got[r] = out
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    case = load_case(name)
    z = case['z']
    names = list(case['data'].obs.index)
    where = {c: i for i, c in enumerate(names)}
    n = len(names)
    a = got[0]
    order = a['order']
    assert sorted(order.tolist()) == list(range(n))
    rpr = -(-n // world)
    coef = np.full(n, np.nan)
    cfdr = np.full(n, np.nan)
    kept = np.zeros(n, dtype=bool)
    nam = np.full((z['nam'].shape[0], n), np.nan)
    namresid = np.full((z['namresid'].shape[0], n), np.nan)
    ncorrs = np.full(n, np.nan)
    for r in range(world):
        g = got[r]
        assert g['view'] and g['cells'] == [names[i] for i in order[r * rpr:(r + 1) * rpr]]
        assert g['p'] == a['p'] and g['k'] == a['k']
        np.testing.assert_array_equal(g['num'], a['num'])
        np.testing.assert_array_equal(g['fdr'], a['fdr'])
        idx = np.array([where[c] for c in g['cells']], dtype=np.int64)
        coef[idx], cfdr[idx], kept[idx] = g['coef'], g['coef_fdr'], g['kept']
        cols = [where[c] for c in g['nam_cells']]
        nam[:, cols] = g['nam']
        namresid[:, cols] = g['namresid']
        ncorrs[cols] = g['ncorrs']
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
