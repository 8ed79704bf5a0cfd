#!/usr/bin/env python3
"""Line-level wall timing of association() host code on the GPU box (sys.setprofile-free: manual timers)."""
import sys, os, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd, warnings
warnings.simplefilter('ignore')
import cna_amd as cna
if os.environ.get('TUNE', '0') == '1': print('tuned', cna.tune_host_allocator())
from cna_amd import synth
from cna_amd.engine import get_engine
from cna_amd.tools import _association as A_, _nam as N_
n, N = int(sys.argv[1]), int(sys.argv[2])
print("tuned", cna.tune_host_allocator())
data, meta = synth.make_dataset(n, N, k=30, seed=0)
y = meta['y']; eng = get_engine(); eng.reuse_nam = False; kw = dict(nsteps=3, Nnull=1000, seed=0)
for _ in range(2): cna.tl.association(data, y, 'id', **kw)
T = collections.OrderedDict()
PER = collections.defaultdict(list)
def lap(name, t0):
    t = time.perf_counter(); T[name] = T.get(name, 0) + t - t0; PER[name].append((t - t0) * 1e3); return t
K = 5
import gc
if os.environ.get('NOGC'): gc.disable()
for _ in range(K):
    t = time.perf_counter()
    codes, labels = A_.sample_codes(data.obs['id']); t = lap('sample_codes', t)
    used = np.bincount(codes[codes >= 0], minlength=len(labels)) > 0; t = lap('bincount used', t)
    batches, fs = A_.check_inputs(data, y, 'id', None, None, None, False, sids_present=labels[used]); t = lap('check_inputs', t)
    N_._prepare_graph(eng, data, 1); t = lap('prepare_graph', t)
    m_ = codes >= 0; t = lap('C: mask', t)
    cm_ = codes[m_]; t = lap('C: take', t)
    C = np.bincount(cm_, minlength=len(labels)); t = lap('C: bincount', t)
    C = C.astype(np.float64); t = lap('C: astype', t)
    C2 = np.bincount(codes[codes >= 0], minlength=len(labels)).astype(np.float64); t = lap('bincount C again', t)
    eng.set_samples(codes, len(labels), C); t = lap('set_samples', t)
    eng.nam_step(False, True, False); eng.nam_step(False, True, False); eng.nam_step(False, False, True); t = lap('nam_step x3 (launch)', t)
    kept = np.repeat(True, eng.n); t = lap('repeat', t)
    positions = pd.Series(np.arange(len(y)), index=y.index)[fs].values
    sample_index = y.index[positions]; colmap = labels.get_indexer(sample_index); t = lap('positions/colmap', t)
    ys, yn = A_._draw_null(y[fs].values, batches[fs].values, None, 1000, False, 0); t = lap('draw_null', t)
    zv, nz = eng.zero_variance(colmap); t = lap('zero_variance (waits for diffusion)', t)
    ka = kept.all(); t = lap('kept.all', t)
    eng.select(None, colmap); t = lap('select', t)
    eng.standardize(center=True); G = eng.gram(); t = lap('standardize+gram', t)
    _, m = eng.ncorrs(ys); t = lap('ncorrs', t)
    coef, fdr = eng.percell(None, None); t = lap('percell coef', t)
    data.obs['coef'] = coef; t = lap('obs setitem', t)
for k, v in T.items(): print('%-40s %8.3f ms   ' % (k, v / K * 1e3), ' '.join('%.1f' % x for x in PER[k]))
