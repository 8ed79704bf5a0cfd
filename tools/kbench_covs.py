import sys, os, time, numpy as np, warnings
warnings.simplefilter('ignore')
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import cna_amd as cna
from cna_amd import synth
from cna_amd.engine import get_engine
n, N = int(sys.argv[1]), int(sys.argv[2])
data, meta = synth.make_dataset(n, N, k=30, seed=0, n_covs=3)
eng = get_engine(); eng.reuse_nam = False; eng.pin_graph(data.obsp['connectivities'])
kw = dict(covs=meta['covs'], nsteps=3, Nnull=1000, seed=0)
for _ in range(3): cna.tl.association(data, meta['y'], 'id', **kw)
eng.prof_reset(); eng.prof_enable(True)
t = time.perf_counter()
for _ in range(8): p = cna.tl.association(data, meta['y'], 'id', **kw)
dt = (time.perf_counter() - t) / 8
eng.sync(); pr = eng.prof()
print('ROWPASS16=%s n=%d N=%d: %.3f ms/step p=%.4g  select %.0f us  resid %s' % (os.environ.get('CNA_ROWPASS16', '1'), n, N, dt * 1e3, p, pr['select'][0] / pr['select'][1] * 1e3, pr.get('resid_xb')))
