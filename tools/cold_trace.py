#!/usr/bin/env python3
"""Where the FIRST association() of a dataset spends its time (graph preparation, upload, first-use allocations):
every Engine method and the host helpers of _order with enter / exit times.  usage: cold_trace.py [cells] [samples]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, warnings
warnings.simplefilter('ignore')
import cna_amd as cna
cna.tune_host_allocator()
from cna_amd import synth, _order
from cna_amd.engine import get_engine, Engine
n, N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000, int(sys.argv[2]) if len(sys.argv) > 2 else 200
data, meta = synth.make_dataset(n, N, k=30, seed=0)
eng = get_engine(); eng.reuse_nam = False
eng.sync()
ev = []
def wrap(obj, name, label):
    fn = getattr(obj, name)
    def w(*a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); ev.append((t0, time.perf_counter(), label, threading.current_thread().name[:4])); return r
    setattr(obj, name, w)
for name in dir(Engine):
    if name.startswith('__') or name in ('block', 'prof', 'close', 'h'): continue
    import inspect
    raw = inspect.getattr_static(Engine, name)
    fn = getattr(Engine, name)
    if callable(fn) and not isinstance(raw, (staticmethod, classmethod, property)):
        def mk(fn, name):
            def w(self, *a, **k):
                t0 = time.perf_counter(); r = fn(self, *a, **k); ev.append((t0, time.perf_counter(), name, threading.current_thread().name[:4])); return r
            return w
        setattr(Engine, name, mk(fn, name))
for name in ('locality_order', 'permuted_rows', 'cluster_order', 'halo_plan'):
    if hasattr(_order, name): wrap(_order, name, '_order.' + name)
for name in ('cna_graph_upload', 'cna_set_cell_order', 'cna_colsums', 'cna_set_samples'):
    pass
kw = dict(nsteps=3, Nnull=1000, seed=0)
t0 = time.perf_counter(); cna.tl.association(data, meta['y'], 'id', **kw); eng.sync(); t1 = time.perf_counter()
print('cold call %.1f ms' % ((t1 - t0) * 1e3))
ev.sort()
for a, b, name, th in ev:
    if (b - a) * 1e3 >= 0.5:
        print('%9.1f  %-28s %9.1f ms  [%s]' % ((a - t0) * 1e3, name, (b - a) * 1e3, th))
t0 = time.perf_counter(); cna.tl.association(data, meta['y'], 'id', **kw); eng.sync(); t1 = time.perf_counter()
print('second call %.1f ms' % ((t1 - t0) * 1e3))
