#!/usr/bin/env python3
"""Timeline of the FIRST association() on a dataset (graph upload, column sums, first-use allocations, one analysis) and
of the later call that adopts the device order: every Engine call with enter / exit times.
    python tools/cold_trace.py [cells] [samples]"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, warnings
warnings.simplefilter('ignore')
import cna_amd as cna
cna.tune_host_allocator()
from cna_amd import synth
from cna_amd.engine import get_engine, Engine
n, N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000, int(sys.argv[2]) if len(sys.argv) > 2 else 200
data, meta = synth.make_dataset(n, N, k=30, seed=0)
# a context that has seen another (small) dataset: the process is warm (library loaded, streams exist), the dataset is not
warm, wmeta = synth.make_dataset(20000, 24, k=15, seed=1)
eng = get_engine(); eng.reuse_nam = False
kw = dict(nsteps=3, Nnull=1000, seed=0)
if not os.environ.get('COLD_PROCESS'):        # COLD_PROCESS=1: the very first call of the process is the traced one
    cna.tl.association(warm, wmeta['y'], 'id', **kw)
ev = []
for name in dir(Engine):
    if name.startswith('_') or name in ('block', 'prof', 'close'): continue
    fn = getattr(Engine, name)
    if not callable(fn) or isinstance(fn, staticmethod): continue
    def mk(fn, name):
        def w(self, *a, **k):
            t0 = time.perf_counter(); r = fn(self, *a, **k); ev.append((t0, time.perf_counter(), name, threading.current_thread().name[:4])); return r
        return w
    setattr(Engine, name, mk(fn, name))
from cna_amd.tools import _association as _A


def traced(label):
    del ev[:]
    _A._TRACE = []
    t0 = time.perf_counter(); cna.tl.association(data, meta['y'], 'id', **kw); eng.sync(); t1 = time.perf_counter()
    marks, _A._TRACE = _A._TRACE, None
    print('%s: %.3f ms' % (label, (t1 - t0) * 1e3))
    last = t0
    for a, b, name, th in sorted(ev):
        if (b - a) * 1e3 >= 0.3 or (a - last) * 1e3 >= 0.3:
            print('%9.3f  +%8.3f gap  %-26s %9.3f ms  [%s]' % ((a - t0) * 1e3, (a - last) * 1e3, name, (b - a) * 1e3, th))
        if th == 'Main': last = b
    print('%9.3f  +%8.3f gap  end' % ((t1 - t0) * 1e3, (t1 - last) * 1e3))
    print('marks: ' + '  '.join('%s@%.1f' % (k, (v - t0) * 1e3) for k, v in marks))


traced('first call on the dataset')
if getattr(eng, 'reorder_pending', lambda: False)():
    t = time.perf_counter(); eng.wait_reorder(); print('device order ready after another %.1f ms' % ((time.perf_counter() - t) * 1e3))
traced('call that adopts the device order')
traced('next call')
