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
