import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, warnings
warnings.simplefilter('ignore')
import cna_amd as cna
from cna_amd import synth
from cna_amd.engine import get_engine
from cna_amd.tools import _association as A
A._DEFER_LAST_CELLS = 0
n, N, K = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
data, meta = synth.make_dataset(n, N, k=15, seed=33)
eng = get_engine()
for ov in (K, '0'):
    os.environ['CNA_GRAM_OVERLAP'] = ov
    eng.prof_reset(); eng.prof_enable(True)
    res = cna.tl.association(data, meta['y'], 'id', Nnull=200, seed=5, nsteps=3, return_full=True, engine=eng)
    eng.sync(); eng.prof_enable(False)
    print(ov, res.p, res.k, {k: v[1] for k, v in eng.prof().items()}, 'kept', int(res.kept.sum()), 'G00', eng.gram_fetch()[0, :3])
