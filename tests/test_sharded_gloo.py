"""N>1 path on CPU: two processes (gloo), cells sharded by row blocks, the real host
orchestration of cna_amd.tools driven through the engine test double with gloo collectives.
Checks that a sharded run equals the unsharded one (the collectives the HIP engine issues
through RCCL are the same ones: column sums, state all-gather between diffusion steps, Gram,
tail histograms, threshold counts, max, per-cell vectors)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import warnings
    warnings.simplefilter('ignore')
    import torch.distributed as td
    td.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    import cna_amd as cna
    from fake_engine import FakeEngine, GlooColl
    from helpers import load_case
    case = load_case(name)
    eng = FakeEngine(GlooColl(), order='rcm' if name != 'c13_zero_variance' else None)
    res = cna.tl.association(case['data'], case['y'], case['sid_name'], batches=case['batches'], covs=case['covs'],
                             donorids=case['donorids'], return_full=True, engine=eng, **case['call'])
    out = dict(p=res.p, k=int(res.k), ncorrs=res.ncorrs.values, kept=res.kept, fdr=res.fdrs.fdr.values,
               num=res.fdrs.num_detected.values, coef=case['data'].obs['coef'].values,
               coef_fdr=case['data'].obs['coef_fdr'].values, nam=res.nam.values, namresid=res.namresid.values,
               V=res.namresid_nbhdXpc.values, rows=(eng.row0, eng.n_local),
               halo=None if eng.halo is None else (int(eng.halo[1].sum()), int(eng.halo[3].sum())))
    q.put((rank, out))
    td.barrier()
    td.destroy_process_group()


@pytest.mark.parametrize('name,world', [('c01_plain_f32', 2), ('c12_batchy_qc', 2), ('c13_zero_variance', 2),
                                        ('c03_covs_batches', 4)])
def test_sharded_equals_reference(name, world):
    import torch.multiprocessing as mp
    from helpers import load_case, relerr
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    z = load_case(name)['z']
    a, b = got[0], got[1]
    n = len(z['kept'])
    print('halo rows (sent, received):', [got[r]['halo'] for r in range(world)])
    rpr = -(-n // world)
    for r in range(world):
        assert got[r]['rows'] == (min(r * rpr, n), max(0, min(rpr, n - r * rpr)))
    if world > 2:   # several peers per rank: what is sent overall is what is received overall
        assert sum(got[r]['halo'][0] for r in range(world)) == sum(got[r]['halo'][1] for r in range(world)) > 0
    for r in range(2, world):
        np.testing.assert_array_equal(got[r]['coef'], a['coef'])
    for key in ('p', 'k'):
        assert a[key] == b[key]
    for key in ('ncorrs', 'kept', 'fdr', 'num', 'coef', 'coef_fdr', 'nam', 'namresid'):
        np.testing.assert_array_equal(a[key], b[key])          # both ranks hold the full result
    assert a['k'] == int(z['k']) and a['p'] == pytest.approx(float(z['p']), rel=1e-12)
    assert np.array_equal(a['kept'], z['kept'])
    assert relerr(a['ncorrs'], z['ncorrs']) < 1e-5
    assert relerr(a['nam'], z['nam']) < 1e-5 and relerr(a['namresid'], z['namresid']) < 1e-5
    T = min(len(a['fdr']), len(z['fdr_fdr']))
    assert np.array_equal(a['num'][:T], z['fdr_num_detected'][:T])
    assert relerr(a['fdr'][:T], z['fdr_fdr'][:T]) < 1e-4
    np.testing.assert_allclose(a['coef_fdr'], z['obs_coef_fdr'], rtol=1e-4, atol=1e-12)
    assert a['V'].shape == z['V'].shape


def _shard_worker(rank, world, port, name, order, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import warnings
    warnings.simplefilter('ignore')
    import torch.distributed as td
    td.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    import cna_amd as cna
    from cna_amd import dist
    from fake_engine import FakeEngine, GlooColl
    from helpers import load_case
    case = load_case(name)
    part = dist.shard(case['data'], rank, world)          # this rank's cells only
    eng = FakeEngine(GlooColl(), order=order)
    res = cna.tl.association(part, case['y'], case['sid_name'], batches=case['batches'], covs=case['covs'],
                             donorids=case['donorids'], return_full=True, engine=eng, **case['call'])
    out = dict(p=res.p, k=int(res.k), ncorrs=res.ncorrs.values, kept=res.kept, fdr=res.fdrs.fdr.values,
               num=res.fdrs.num_detected.values, coef=part.obs['coef'].values, coef_fdr=part.obs['coef_fdr'].values,
               nam=res.nam.values, namresid=res.namresid.values, V=res.namresid_nbhdXpc.values,
               varexp=res.namresid_varexp.values, cells=list(res.nam.columns), n_obs=len(part.obs), view=eng.view_local)
    s0 = np.random.RandomState(1).rand(case['data'].obsp['connectivities'].shape[0], 3)
    r0 = part.uns['cna_shard']['row0']
    out['diffuse'] = cna.tl.diffuse(part, s0[r0:r0 + len(part.obs)], 2, engine=eng)
    NAM, keep = cna.tl.nam(part, case['sid_name'], batches=case['batches'], engine=eng)
    out['tlnam'], out['tlkeep'] = NAM.values, keep
    q.put((rank, out))
    td.barrier()
    td.destroy_process_group()


@pytest.mark.parametrize('name,world,order', [('c01_plain_f32', 2, 'rcm'), ('c12_batchy_qc', 3, 'random'),
                                              ('c13_zero_variance', 2, None), ('c10_categorical_ids', 2, 'rcm'),
                                              ('c11_string_ids_null_y', 2, 'random')])
def test_sharded_inputs_local_view(name, world, order):
    """Every rank is handed ONLY its block of cells (cna_amd.dist.shard): obs rows, graph rows.  The
    per-cell results it gets back cover that block; stitched together in rank order they are the
    reference's, and the sample-level results are the reference's on every rank."""
    import torch.multiprocessing as mp
    import scipy.sparse as sp
    from helpers import load_case, relerr
    from oracle import cna_oracle as orc
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, name, order, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    case = load_case(name)
    z = case['z']
    n = len(z['kept'])
    rpr = -(-n // world)
    parts = [got[r] for r in range(world)]
    for r, g in enumerate(parts):
        assert g['view'] and g['n_obs'] == max(0, min(rpr, n - r * rpr)) == len(g['kept']) == len(g['coef'])
        assert g['p'] == parts[0]['p'] and g['k'] == parts[0]['k']
        np.testing.assert_array_equal(g['fdr'], parts[0]['fdr'])
        np.testing.assert_array_equal(g['num'], parts[0]['num'])
        np.testing.assert_array_equal(g['varexp'], parts[0]['varexp'])
    cat = lambda key, axis=0: np.concatenate([g[key] for g in parts], axis=axis)
    a = parts[0]
    assert a['k'] == int(z['k']) and a['p'] == pytest.approx(float(z['p']), rel=1e-12)
    assert np.array_equal(cat('kept'), z['kept'])
    assert relerr(cat('ncorrs'), z['ncorrs']) < 1e-5
    assert relerr(cat('nam', 1), z['nam']) < 1e-5 and relerr(cat('namresid', 1), z['namresid']) < 1e-5
    assert cat('V').shape == z['V'].shape
    kept_names = np.asarray(case['data'].obs.index)[z['kept']]
    assert [c for g in parts for c in g['cells']] == list(kept_names)
    T = min(len(a['fdr']), len(z['fdr_fdr']))
    assert np.array_equal(a['num'][:T], z['fdr_num_detected'][:T])
    assert relerr(a['fdr'][:T], z['fdr_fdr'][:T]) < 1e-4
    coef = cat('coef')
    assert np.array_equal(np.isnan(coef), np.isnan(z['obs_coef']))
    assert relerr(coef[~np.isnan(coef)], z['obs_coef'][~np.isnan(coef)]) < 1e-5
    np.testing.assert_allclose(cat('coef_fdr'), z['obs_coef_fdr'], rtol=1e-4, atol=1e-12)
    A = sp.csr_matrix(case['data'].obsp['connectivities'])
    s0 = np.random.RandomState(1).rand(A.shape[0], 3)
    assert relerr(cat('diffuse'), orc.diffuse(A, s0, 2, mode='f64')) < 1e-12
    if 'tlnam' in z.files if hasattr(z, 'files') else 'tlnam' in z:
        assert relerr(cat('tlnam', 1), z['tlnam']) < 1e-5
    assert cat('tlkeep').shape == (n,)


def _partition_worker(rank, world, port, name, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import warnings
    warnings.simplefilter('ignore')
    import torch.distributed as td
    td.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    import cna_amd as cna
    from cna_amd import dist
    from fake_engine import FakeEngine, GlooColl
    from helpers import load_case
    case = load_case(name)
    part = dist.shard(case['data'], rank, world, partition='always')     # whole populations of the graph per block
    eng = FakeEngine(GlooColl(), order='rcm')
    res = cna.tl.association(part, case['y'], case['sid_name'], batches=case['batches'], covs=case['covs'],
                             donorids=case['donorids'], return_full=True, engine=eng, **case['call'])
    out = dict(p=res.p, k=int(res.k), kept=res.kept, num=res.fdrs.num_detected.values, fdr=res.fdrs.fdr.values,
               cells=list(part.obs.index), coef=part.obs['coef'].values, coef_fdr=part.obs['coef_fdr'].values,
               nam=res.nam.values, nam_cells=list(res.nam.columns), order=part.uns['cna_shard']['order'],
               halo=None if eng.halo is None else int(eng.halo[3].sum()))
    q.put((rank, out))
    td.barrier()
    td.destroy_process_group()


@pytest.mark.parametrize('name,world', [('c01_plain_f32', 2), ('c12_batchy_qc', 4)])
def test_sharded_inputs_partitioned_by_population(name, world):
    """dist.shard(..., partition='always'): the blocks are made of whole populations of the graph
    (_order.partition_order) instead of contiguous runs of the caller's cells.  The analysis is the reference's on the
    renumbered dataset: per-cell results matched by cell name, sample-level results equal on every rank and equal to
    the golden values (integers exact, floats to 1e-5; the column sums add their rows in another order: rounding)."""
    import torch.multiprocessing as mp
    from helpers import load_case, relerr
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_partition_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
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
    for r in range(world):
        g = got[r]
        assert g['cells'] == [names[i] for i in order[r * rpr:(r + 1) * rpr]]
        assert g['p'] == a['p'] and g['k'] == a['k']
        np.testing.assert_array_equal(g['num'], a['num'])
        idx = np.array([where[c] for c in g['cells']])
        coef[idx], cfdr[idx], kept[idx] = g['coef'], g['coef_fdr'], g['kept']
        nam[:, [where[c] for c in g['nam_cells']]] = g['nam']
    assert a['k'] == int(z['k']) and a['p'] == pytest.approx(float(z['p']), rel=1e-12)
    assert np.array_equal(kept, z['kept'])
    T = min(len(a['num']), len(z['fdr_num_detected']))
    assert np.array_equal(a['num'][:T], z['fdr_num_detected'][:T])
    assert relerr(a['fdr'][:T], z['fdr_fdr'][:T]) < 1e-4
    assert np.array_equal(np.isnan(coef), np.isnan(z['obs_coef']))
    assert relerr(coef[~np.isnan(coef)], z['obs_coef'][~np.isnan(coef)]) < 1e-5
    np.testing.assert_allclose(cfdr, z['obs_coef_fdr'], rtol=1e-4, atol=1e-12)
    assert relerr(nam[:, z['kept']], z['nam']) < 1e-5


def test_partition_order_packs_populations():
    """_order.partition_order on a graph of separated populations whose cells come in random order: a permutation;
    its blocks send far fewer rows than blocks of the caller's order or of the plain cluster order (what a rank
    ships between diffusion steps), and about as few as the generator's own population-sorted order."""
    sys.path.insert(0, ROOT)
    import scipy.sparse as sp
    from cna_amd import synth, _order
    X, _ = synth.mixture_points(24000)
    A0 = synth.fuzzy_knn_graph(X, k=15).tocsr()
    n = A0.shape[0]
    shuffle = np.random.RandomState(0).permutation(n)
    A = A0[shuffle][:, shuffle].tocsr()
    deg = np.diff(A.indptr)

    def boundary(order, G):
        inv = np.empty(n, np.int64)
        inv[order] = np.arange(n)
        rpr = -(-n // G)
        cut = np.repeat(inv // rpr, deg) != inv[A.indices] // rpr
        b = np.zeros(n, bool)
        b[np.repeat(inv, deg)[cut]] = True
        return b.mean()
    for G in (2, 4, 8):
        o = _order.partition_order(A, G)
        assert sorted(o.tolist()) == list(range(n))
        sorted_by_population = boundary(np.argsort(shuffle), G)        # the generator's order, undone
        assert boundary(np.arange(n), G) > 0.9                           # a random order: every row is wanted elsewhere
        assert boundary(o, G) < 0.6 * boundary(_order.cluster_order(A, 512), G)
        assert boundary(o, G) < max(0.35, 1.3 * sorted_by_population), (G, boundary(o, G), sorted_by_population)
        # block_traffic counts what the halo plan sends: distinct (row, peer block) pairs
        tr = _order.block_traffic(A, o, G)
        assert tr.shape == (G,) and boundary(o, G) * n <= tr.sum() <= (G - 1) * boundary(o, G) * n + 1
        assert tr.max() < 0.5 * _order.block_traffic(A, None, G).max()
    assert np.array_equal(_order.partition_order(A, 1), _order.cluster_order(A, 512))
    # a dataset that arrives sorted by population and is cut between populations keeps its own order ...
    two = sp.block_diag([A0[:100][:, :100], A0[:100][:, :100]]).tocsr()
    assert np.array_equal(_order.partition_order(two, 2), np.arange(200))
    # ... unless asked otherwise
    forced = _order.partition_order(two, 2, compare=False)
    assert sorted(forced.tolist()) == list(range(200))
