import sys, os, time
sys.path.insert(0, os.getcwd())
t00 = time.perf_counter()
import numpy as np, warnings
warnings.simplefilter('ignore')
import cna_amd as cna
from cna_amd import synth
from cna_amd.engine import get_engine
t_imp = time.perf_counter() - t00
data, meta = synth.make_dataset(20000, 24, k=15, seed=1)
t0 = time.perf_counter(); eng = get_engine(); t_eng = time.perf_counter() - t0
import cProfile, pstats, io
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
cna.tl.association(data, meta['y'], 'id', nsteps=3, Nnull=1000, seed=0)
pr.disable()
t1 = time.perf_counter() - t0
t0 = time.perf_counter(); cna.tl.association(data, meta['y'], 'id', nsteps=3, Nnull=1000, seed=0); t2 = time.perf_counter() - t0
print('imports %.0f ms, engine %.0f ms, first call (20k cells) %.1f ms, second %.1f ms' % (t_imp * 1e3, t_eng * 1e3, t1 * 1e3, t2 * 1e3))
buf = io.StringIO(); pstats.Stats(pr, stream=buf).sort_stats('cumulative').print_stats(45); print(buf.getvalue()[:9000])
