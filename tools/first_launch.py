import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from cna_amd import synth
from cna_amd.engine import get_engine
from cna_amd.tools._nam import sample_codes
n, N = 200000, 50
data, meta = synth.make_dataset(n, N, k=30, seed=0)
codes, labels = sample_codes(data.obs['id'])
C = np.bincount(codes, minlength=N).astype(float)
eng = get_engine()
eng.ensure_graph(data.obsp['connectivities'].tocsr()); eng.colsums(1)
eng.set_samples(codes, N, C, token='a')
def t(f):
    a = time.perf_counter(); f(); return (time.perf_counter() - a) * 1e3
for rep in range(6):
    eng.sync(); time.sleep(0.002 * (rep % 2))
    r = [t(lambda: eng.set_samples(codes, N, C, token='a'))]
    r += [t(lambda: eng.nam_step(False, True, False)), t(lambda: eng.nam_step(False, True, False)), t(lambda: eng.nam_step(False, False, True))]
    r += [t(eng.sync)]
    print('restart %.3f  step1 %.3f step2 %.3f step3 %.3f sync %.3f' % tuple(r))
