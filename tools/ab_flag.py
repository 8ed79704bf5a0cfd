#!/usr/bin/env python3
"""Same-process A/B of a module-level switch of cna_amd.tools._association on repeated association() calls:
    ab_flag.py [MODULE:]FLAG [cells=200000] [samples=50] [calls=300] [rounds=4]
alternates FLAG = True / False in rounds, prints ms per call of every round (box-to-box noise is 10x the effects looked for)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter('ignore')
import numpy as np
import cna_amd as cna
from cna_amd import synth
from cna_amd.engine import get_engine
from cna_amd.tools import _association, _nam, _stats
modname, _, flag = sys.argv[1].rpartition(':')
A = {'': _association, '_association': _association, '_nam': _nam, '_stats': _stats}[modname]
n, N = int(sys.argv[2]) if len(sys.argv) > 2 else 200000, int(sys.argv[3]) if len(sys.argv) > 3 else 50
calls, rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 300, int(sys.argv[5]) if len(sys.argv) > 5 else 4
cna.tune_host_allocator()
data, meta = synth.make_dataset(n, N, k=30, seed=0)
eng = get_engine(); eng.reuse_nam = False
eng.pin_graph(data.obsp['connectivities'])
kw = dict(nsteps=3, Nnull=1000, seed=0)
cna.tl.association(data, meta['y'], 'id', **kw)
if getattr(eng, 'reorder_pending', lambda: False)(): eng.wait_reorder()
for _ in range(60): cna.tl.association(data, meta['y'], 'id', **kw)
assert hasattr(A, flag), flag
for r in range(rounds):
    for v in (True, False):
        setattr(A, flag, v)
        for _ in range(20): cna.tl.association(data, meta['y'], 'id', **kw)
        eng.sync(); t = time.perf_counter()
        for _ in range(calls): cna.tl.association(data, meta['y'], 'id', **kw)
        eng.sync()
        print('%s=%-5s  %.3f ms per call' % (flag, v, (time.perf_counter() - t) / calls * 1e3), flush=True)
