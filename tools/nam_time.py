import time, sys, numpy as np
sys.path.insert(0, '/root/repo')
import cna_amd as cna
from cna_amd import synth
from cna_amd.engine import Engine
n, N = int(sys.argv[1]), int(sys.argv[2])
data, meta = synth.make_dataset(n, N, k=15, seed=1, n_covs=1, n_batches=4)
e = Engine(device=0)
for it in range(3):
    t0 = time.perf_counter(); fr, keep = cna.tl.nam(data, 'id', batches=meta['batches'], engine=e); t1 = time.perf_counter()
    print('tl.nam ms', round((t1 - t0) * 1e3, 1), fr.shape)
res = cna.tl.association(data, meta['y'], 'id', covs=meta['covs'], batches=meta['batches'], Nnull=1000, return_full=True, engine=e)
for it in range(2):
    res = cna.tl.association(data, meta['y'], 'id', covs=meta['covs'], batches=meta['batches'], Nnull=1000, return_full=True, engine=e)
    for f in ('nam', 'namresid', 'namresid_nbhdXpc'):
        t0 = time.perf_counter(); v = getattr(res, f); t1 = time.perf_counter()
        print(f, 'ms', round((t1 - t0) * 1e3, 1), v.shape)
