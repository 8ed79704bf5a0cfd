#!/usr/bin/env python3
"""Differential run against the float64 oracle on random GRAPHS (what scanpy would never emit, and what it might): hub
rows and hub columns (more neighbours than the 64 pairs a compressed row holds, more than one 64-edge batch), empty rows,
a diagonal, duplicates, unsorted indices, weights above 1, float32 / float64 values, int64 indices, CSC / COO containers,
asymmetric matrices.  Every case: NAM bit-identical, k / p / kept / num_detected equal.
    python tools/fuzz_graphs.py [seconds=300] [seed=0]"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter('ignore')
import numpy as np, pandas as pd, scipy.sparse as sp
import cna_amd as cna
from cna_amd.synth import CellData
from cna_amd.engine import get_engine
from oracle import cna_oracle as orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
eng = get_engine()


def relerr(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    d = np.abs(a - b)
    return float(np.nanmax(d) / max(np.nanmax(np.abs(b)), 1e-300)) if d.size else 0.0


def random_graph(n, tag):
    deg = int(rs.choice([4, 12, 30]))
    rows = np.repeat(np.arange(n), deg)
    cols = (rows + rs.randint(1, max(2, n // 20), size=len(rows)) * rs.choice([-1, 1], size=len(rows))) % n     # banded-ish locality
    vals = rs.rand(len(rows)) * (3.0 if rs.rand() < 0.2 else 1.0)
    if rs.rand() < 0.6:                                    # hubs: rows / columns with hundreds of entries
        tag.append('hubs')
        for h in rs.choice(n, size=int(rs.randint(1, 6)), replace=False):
            m = int(rs.randint(70, min(700, n - 1)))
            other = rs.choice(n, size=m, replace=False)
            rows = np.concatenate([rows, np.full(m, h), other]); cols = np.concatenate([cols, other, np.full(m, h)])
            vals = np.concatenate([vals, rs.rand(2 * m)])
    if rs.rand() < 0.3:
        tag.append('diag'); d = rs.choice(n, size=n // 3, replace=False)
        rows = np.concatenate([rows, d]); cols = np.concatenate([cols, d]); vals = np.concatenate([vals, rs.rand(len(d))])
    if rs.rand() < 0.4:                                    # some cells without any neighbour
        tag.append('empty'); dead = rs.choice(n, size=max(1, n // 50), replace=False)
        keep = ~np.isin(rows, dead) & ~np.isin(cols, dead)
        rows, cols, vals = rows[keep], cols[keep], vals[keep]
    dtype = rs.choice([np.float32, np.float64]); tag.append(dtype.__name__)
    order = np.lexsort((rs.rand(len(rows)), rows)) if rs.rand() < 0.5 else np.lexsort((cols, rows))      # unsorted inside rows, or sorted
    rows, cols, vals = rows[order], cols[order], vals[order].astype(dtype)
    indptr = np.zeros(n + 1, dtype=np.int64); np.add.at(indptr, rows + 1, 1); indptr = np.cumsum(indptr)
    A = sp.csr_matrix((vals, cols.astype(np.int32), indptr.astype(np.int32)), shape=(n, n))              # duplicates stay duplicates
    kind = rs.rand()
    if kind < 0.15:
        tag.append('csc'); A = sp.csc_matrix(A)
    elif kind < 0.3:
        tag.append('coo'); A = sp.coo_matrix(A)
    elif kind < 0.45:
        tag.append('i64'); A.indices = A.indices.astype(np.int64); A.indptr = A.indptr.astype(np.int64)
    return A


t_end = time.time() + budget
done = 0
fails = []
kinds = {}
while time.time() < t_end:
    n = int(rs.choice([400, 900, 2000, 3500]))
    N = int(rs.choice([8, 20, 50, 64, 65, 96, 120, 200, 300]))
    tag = ['N%d' % N]
    A = random_graph(n, tag)
    sid = rs.randint(0, N, size=n)
    sid[:N] = np.arange(N)                                  # every sample has a cell
    obs = pd.DataFrame({'id': sid}, index=pd.Index(['c%d' % i for i in range(n)]))
    data = CellData(obs, A)
    y = pd.Series(rs.randn(N) + 0.5 * np.bincount(sid, weights=np.arange(n) / n, minlength=N) / np.bincount(sid, minlength=N), index=np.arange(N))
    kw = dict(nsteps=[None, 1, 2, 3, 5][int(rs.randint(5))], Nnull=int(rs.choice([40, 100])), seed=int(rs.randint(1000)))
    tag.append('steps%s' % kw['nsteps'])
    for t in tag:
        kinds[t] = kinds.get(t, 0) + 1
    ref = res = eref = eres = None
    try:
        ref = orc.association(data, y, 'id', mode='f64', allow_low_sample_size=True, **kw)
    except Exception as e:                       # noqa: BLE001
        eref = e
    try:
        res = cna.tl.association(data, y, 'id', return_full=True, engine=eng, allow_low_sample_size=True, **kw)
    except Exception as e:                       # noqa: BLE001
        eres = e
    done += 1
    try:
        if eref is not None or eres is not None:
            assert eref is not None and eres is not None and type(eref) is type(eres), ('one side raised', repr(eref)[:100], repr(eres)[:100])
            continue
        assert np.array_equal(res.nam.values.T, ref['nam']), ('nam bits', relerr(res.nam.values.T, ref['nam']))
        assert int(res.k) == ref['k'] and res.p == ref['p'], ('k / p', res.k, ref['k'], res.p, ref['p'])
        assert np.array_equal(res.kept, ref['kept']), 'kept'
        assert relerr(res.ncorrs.values, ref['ncorrs']) < 1e-9, 'ncorrs'
        T = min(len(res.fdrs), len(ref['fdrs']['fdr']))
        assert np.array_equal(res.fdrs.num_detected.values[:T], ref['fdrs']['num_detected'][:T]), 'num_detected'
    except Exception as exc:                     # noqa: BLE001
        fails.append((done, n, tag, kw, str(exc.args[0] if exc.args else exc)[:200]))
for it in fails[:12]:
    print('FAILED case %d n=%d %s %r\n      %s' % it)
print('%d graphs in %.0f s: %d disagreements' % (done, budget, len(fails)))
print('coverage: ' + '  '.join('%s x%d' % kv for kv in sorted(kinds.items())))
sys.exit(1 if fails else 0)
