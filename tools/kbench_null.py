"""Local null alone on the GPU: integer path (csrc/null_i8.hip) against the f64 kernel on resident data.
python tools/kbench_null.py  (GPU box)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cna_amd.engine import Engine


def run(n, N, P=1000, reps=5):
    eng = Engine()
    rs = np.random.RandomState(0)
    X = rs.randn(n, N)
    X -= X.mean(axis=1, keepdims=True)
    X /= X.std(axis=1, ddof=1)[:, None]
    eng.upload_x(X)
    nc, maxabs = eng.ncorrs(rs.randn(N), fetch=True)
    # a structured phenotype: correlations as large as real data give (max ~ 0.5)
    maxcorr = max(maxabs, 0.3)
    Yc = rs.randn(N, P)
    Yc /= Yc.std(axis=0, ddof=1)
    thr = np.arange(maxcorr / 4, maxcorr, maxcorr / 400)
    edges = thr ** 2 - 1e-8 - 1e-5 * thr ** 2
    tails = eng.null_local(Yc, edges)
    out = {}
    for mode in ('i8', 'f64'):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            if mode == 'i8':
                sums = eng.null_local_resident(0, P, edges, sums_only=True)
            else:
                tails2 = eng.null_local_resident(0, P, edges)
            ts.append(time.perf_counter() - t0)
        out[mode] = min(ts)
        if mode == 'i8':
            used, rechecked, fb = eng.null_local_i8_stats()
    ok = np.array_equal(sums, tails.sum(axis=0))
    fl = 2.0 * n * N * P
    print(f'n={n} N={N} P={P}: i8 {out["i8"]*1e3:.3f} ms ({fl/out["i8"]/1e12:.1f} f64-equivalent TFLOP/s), f64 {out["f64"]*1e3:.3f} ms '
          f'({fl/out["f64"]/1e12:.1f} TFLOP/s); rechecked {rechecked} ({rechecked/(n*P):.2e} of outputs), used={used} fallback={fb} equal={ok}',
          flush=True)
    del eng


if __name__ == '__main__':
    sizes = ((200000, 50), (1000000, 100), (2000000, 200), (500000, 200), (300000, 256), (400000, 130))
    if len(sys.argv) > 1:
        sizes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]]
    for n, N in sizes:
        run(n, N)
