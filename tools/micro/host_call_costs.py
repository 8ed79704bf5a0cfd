import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, pandas as pd
import cna_amd as cna
cna.tune_host_allocator()
from cna_amd import synth
from cna_amd.tools._nam import sample_codes_cached, get_connectivity
data, meta = synth.make_dataset(20000, 50, k=15, seed=0, builder='cpu')
y=meta['y']
def T(f,n=3000):
    f(); t=time.perf_counter()
    for i in range(n): f()
    return (time.perf_counter()-t)/n*1e6
sample_codes_cached(data.obs['id'])
print('obs[id]', T(lambda: data.obs['id']))
col=data.obs['id']
print('codes defer', T(lambda: sample_codes_cached(col, defer='caller')))
print('get_conn', T(lambda: get_connectivity(data)))
codes, labels, counts, token = sample_codes_cached(col, defer='caller')
print('equals', T(lambda: labels.equals(y.index)))
yv=y.values
def val():
    fv=~np.isnan(yv); N=int(np.count_nonzero(fv)); 
    with np.errstate(all='ignore'):
        ys=(yv-yv.mean())/yv.std()
    return np.isfinite(ys).all()
print('val core', T(val))
from cna_amd.tools._stats import default_ks, native_draw_start
def k():
    ks_=default_ks(50); a=np.asarray(ks_); return a.ndim!=1 or len(a)<1 or a.dtype.kind not in 'iu' or a.min()<1 or a.max()+0>=50
print('ks', T(k))
ys=(yv-yv.mean())/yv.std()
def d():
    n=native_draw_start(None,ys,1000,0,single_level=True); 
    return n
t=[]; 
for i in range(50):
    t0=time.perf_counter(); n=d(); t.append(time.perf_counter()-t0); n.abandon()
print('draw start us', np.median(t)*1e6)
t=[]
for i in range(50):
    n=d(); time.sleep(0.002); t0=time.perf_counter(); n.wait(); t.append(time.perf_counter()-t0)
print('draw wait (done) us', np.median(t)*1e6)
t=[]
for i in range(50):
    t0=time.perf_counter(); n=d(); n.wait(); t.append(time.perf_counter()-t0)
print('draw total us', np.median(t)*1e6)
print("pd.Index(y.index,name)", T(lambda: pd.Index(y.index, name='id')))
from cna_amd.tools._nam import host_blas_threads
def b():
    with host_blas_threads(1): pass
print('blas', T(b))
n=200000
df=pd.DataFrame({'id':np.random.randint(0,50,n)})
def f():
    df['coef']=np.empty(n); v=df['coef'].values
    df['coef_fdr']=np.empty(n); w=df['coef_fdr'].values
    return v,w
print('2 cols', T(f,200))
a=np.random.randn(n)
def g(): df['coef']=a
print('assign', T(g,200))
import ctypes as C
from cna_amd import _ffi
def mk():
    a=_ffi.AssocArgs(); a.n_sel=5; a.r=0; a.K=3; return a
print('AssocArgs', T(mk))
print('os.cpu_count', os.cpu_count(), len(os.sched_getaffinity(0)))
