#!/usr/bin/env python3
"""Timeline of one association() step: every Engine call with enter/exit times and the gaps between them."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, warnings
warnings.simplefilter('ignore')
import cna_amd as cna
cna.tune_host_allocator()
from cna_amd import synth
from cna_amd.engine import get_engine, Engine
n, N = int(sys.argv[1]) if len(sys.argv) > 1 else 200000, int(sys.argv[2]) if len(sys.argv) > 2 else 50
ncov = int(os.environ.get('TRACE_COVS', '0'))
nbat = int(os.environ.get('TRACE_BATCHES', '0'))
data, meta = synth.make_dataset(n, N, k=30, seed=0, n_covs=ncov, n_batches=nbat)
from cna_amd.tools import _stats as _S
_S.DRAW_MEMO = bool(os.environ.get('TRACE_DRAW_MEMO'))
eng = get_engine(); eng.reuse_nam = False; kw = dict(nsteps=3, Nnull=int(os.environ.get('TRACE_NNULL', '1000')), seed=0)
if ncov: kw['covs'] = meta['covs']
if nbat: kw['batches'] = meta['batches']
if os.environ.get('TRACE_NSTEPS') == 'None': kw['nsteps'] = None
if os.environ.get('TRACE_PIN'): eng.pin_graph(data.obsp['connectivities'])
cna.tl.association(data, meta['y'], 'id', **kw)
if getattr(eng, 'reorder_pending', lambda: False)(): eng.wait_reorder()        # the call that adopts the device order is not the one traced
for _ in range(6): cna.tl.association(data, meta['y'], 'id', **kw)
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
_A._TRACE = []
from cna_amd.tools import _fast as _F
_F._TRACE = _A._TRACE
t0 = time.perf_counter(); cna.tl.association(data, meta['y'], 'id', **kw); t1 = time.perf_counter()
marks, _A._TRACE = _A._TRACE, None
_F._TRACE = None
ev.sort()
print('step %.3f ms' % ((t1 - t0) * 1e3))
last = t0
for a, b, name, th in ev:
    print('%8.3f  +%6.3f gap  %-22s %6.3f ms  [%s]' % ((a - t0) * 1e3, (a - last) * 1e3, name, (b - a) * 1e3, th))
    if th == 'Main': last = b
print('%8.3f  +%6.3f gap  end' % ((t1 - t0) * 1e3, (t1 - last) * 1e3))
tm = getattr(eng, 'last_assoc_t_ms', None)
if tm:
    print('cna_assoc_finish stages (ms from its entry): ' + '  '.join('%s=%.3f' % (k, v) for k, v in zip(
        ('posted', 'select_back', 'null_queued', 'verified', 'coef_out', 'null_over+fdr_out', 'null_results', 'eig_joined', 'exit', 'gram_back', 'eig_done', 'ftests_done', 'drawn', 'conditioned'), tm)))
print('marks: ' + '  '.join('%s@%.3f' % (k, (v - t0) * 1e3) for k, v in marks))
