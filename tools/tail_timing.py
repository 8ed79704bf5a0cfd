import numpy as np, pandas as pd, time, warnings
warnings.simplefilter('ignore')
n=200000
obs=pd.DataFrame({'id':np.random.randint(0,50,n)})
a=np.random.randn(n); b=np.random.rand(n)
obs['coef']=a; obs['coef_fdr']=b
def t(f,k=200):
    f(); t0=time.perf_counter()
    for _ in range(k): f()
    return (time.perf_counter()-t0)/k*1e3
def setcol():
    obs['coef']=a
print('setitem existing col %.3f ms' % t(setcol))
print("'coef' in obs %.4f ms" % t(lambda: 'coef' in obs))
print('warn %.4f ms' % t(lambda: warnings.warn("Key 'coef' already exists in data.obs. Overwriting.")))
thr=np.arange(0.1,0.4,0.001); f=np.random.rand(len(thr))
print('fmin.accumulate %.4f' % t(lambda: np.fmin.accumulate(f)))
print('np.repeat True %.4f' % t(lambda: np.repeat(True, n)))
k=np.repeat(True,n)
print('kept.all %.4f' % t(lambda: k.all()))
print('copy 1.6MB %.4f' % t(lambda: a.copy()))
print('empty 1.6MB x2 %.4f' % t(lambda: (np.empty(n), np.empty(n))))
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cna_amd.tools._nam import host_blas_threads
def ctx():
    with host_blas_threads(1): pass
print('host_blas_threads enter+exit %.4f ms' % t(ctx))
