#!/usr/bin/env python3
"""Distribution of step times of repeated association() calls (graph pinned, NAM cache and draw memo off), with the stage marks of the slow ones."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, warnings
warnings.simplefilter('ignore')
import cna_amd as cna
cna.tune_host_allocator()
from cna_amd import synth
from cna_amd.engine import get_engine
from cna_amd.tools import _stats as _S
_S.DRAW_MEMO = False
n, N, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 40
data, meta = synth.make_dataset(n, N, k=30, seed=0)
eng = get_engine(); eng.reuse_nam = False
eng.pin_graph(data.obsp['connectivities'])
kw = dict(nsteps=3, Nnull=1000, seed=0)
cna.tl.association(data, meta['y'], 'id', **kw)
if eng.reorder_pending(): eng.wait_reorder()
for _ in range(5): cna.tl.association(data, meta['y'], 'id', **kw)

ts, marks = [], []
for i in range(reps):
    t0 = time.perf_counter()
    cna.tl.association(data, meta['y'], 'id', **kw)
    ts.append((time.perf_counter() - t0) * 1e3); marks.append(list(eng.last_assoc_t_ms))
ts = np.array(ts)
print('%d x %d: mean %.3f  median %.3f  min %.3f  max %.3f  p90 %.3f' % (n, N, ts.mean(), np.median(ts), ts.min(), ts.max(), np.percentile(ts, 90)))
print('sorted:', ' '.join('%.2f' % t for t in np.sort(ts)))
names = ('posted', 'select_back', 'null_queued', 'verified', 'coef_out', 'fdr_out', 'null_results', 'eig_joined', 'exit', 'gram_back', 'eig_done', 'ftests_done', 'drawn', 'conditioned')
for i in np.argsort(ts)[[0, len(ts) // 2, -3, -2, -1]]:
    print('%.3f ms: ' % ts[i] + '  '.join('%s=%.3f' % kv for kv in zip(names, marks[i])))
