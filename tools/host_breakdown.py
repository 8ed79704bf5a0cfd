#!/usr/bin/env python3
"""Where does one association() step spend wall time?  Wraps every Engine method with a timer
(wall time incl. waiting for the GPU) and reports engine vs pure-Python time.  GPU box only."""
import sys, os, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import warnings; warnings.simplefilter('ignore')
import cna_amd as cna
if os.environ.get('TUNE', '1') == '1': print('tuned', cna.tune_host_allocator())
from cna_amd import synth
from cna_amd.engine import get_engine, Engine

n, N = int(sys.argv[1]) if len(sys.argv) > 1 else 200000, int(sys.argv[2]) if len(sys.argv) > 2 else 50
data, meta = synth.make_dataset(n, N, k=30, seed=0)
eng = get_engine(); eng.reuse_nam = False
kw = dict(nsteps=3, Nnull=1000, seed=0)
for _ in range(3):
    cna.tl.association(data, meta['y'], 'id', **kw)
acc = collections.defaultdict(float); cnt = collections.Counter()
for name in dir(Engine):
    if name.startswith('_') or name in ('block', 'prof', 'close'): continue
    fn = getattr(Engine, name)
    if not callable(fn) or isinstance(fn, staticmethod): continue
    def mk(fn, name):
        def w(self, *a, **k):
            t = time.perf_counter(); r = fn(self, *a, **k); acc[name] += time.perf_counter() - t; cnt[name] += 1; return r
        return w
    setattr(Engine, name, mk(fn, name))
K = 10
t0 = time.perf_counter()
for _ in range(K):
    cna.tl.association(data, meta['y'], 'id', **kw)
tot = (time.perf_counter() - t0) / K * 1e3
eng_ms = sum(acc.values()) / K * 1e3
print('total %.3f ms/step  engine calls %.3f ms  python %.3f ms' % (tot, eng_ms, tot - eng_ms))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print('  %-16s %7.3f ms/step  (%d calls/step)' % (k, v / K * 1e3, cnt[k] // K))

# ---- phase timing of the Python layer (wall, includes engine time inside each phase)
import cna_amd.tools._association as A_
import cna_amd.tools._nam as N_
import cna_amd.tools._stats as S_
pacc = collections.defaultdict(float)
def wrap(mod, name):
    fn = getattr(mod, name)
    def w(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); pacc[name] += time.perf_counter() - t; return r
    setattr(mod, name, w)
for mod, names in ((A_, ['check_inputs', 'compute_nam_and_reindex', '_association', 'sample_codes', '_resid_device',
                         'conditional_permutation', 'minp_stats', '_nam_device', '_qc_device', '_draw_null', '_small_svd']),
                   (N_, ['_prepare_graph'])):
    for nm in names:
        wrap(mod, nm)
acc.clear()
t0 = time.perf_counter()
for _ in range(K):
    cna.tl.association(data, meta['y'], 'id', **kw)
tot = (time.perf_counter() - t0) / K * 1e3
print('phases (ms/step, nested, wall): total %.3f' % tot)
for k, v in sorted(pacc.items(), key=lambda kv: -kv[1]):
    print('  %-26s %7.3f' % (k, v / K * 1e3))
