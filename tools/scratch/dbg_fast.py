import sys, os, warnings, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import cna_amd as cna
from cna_amd import synth
from cna_amd.engine import get_engine
from cna_amd.tools import _fast
warnings.simplefilter('ignore')
eng = get_engine(); eng.reuse_nam = False
data, meta = synth.make_dataset(20000, 80, k=15, seed=3, n_covs=2)
kw = dict(nsteps=3, Nnull=200, seed=5, covs=meta['covs'])
def h(a): return hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()[:8]
for on in (False, False, True, True, False, True):
    _fast.ENABLED = on
    r = cna.tl.association(data, meta['y'], 'id', return_full=True, engine=eng, **kw)
    print(on, repr(r.r2), r.p, h(r.nullminps), h(r.ncorrs.values), h(r.namresid_svs.values), h(r.namresid_sampleXpc.values), h(r.M.values), h(r.yresid.values), h(r.namresid.values), h(r.nam.values), _fast.stats)
