#!/usr/bin/env python3
"""Walk kernels under different device cell orders and XCD chunk sizes:
    kbench_order.py n_cells n_samples   (prints us per launch of nam_first / nam_step)"""
import os, sys, time
os.environ['CNA_REORDER_ASYNC'] = '0'      # read when cna_amd.engine is imported: the order under test must be the one the kernels run in
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cna_amd import synth
from cna_amd.engine import Engine
from cna_amd.tools._nam import sample_codes
n, N = int(sys.argv[1]), int(sys.argv[2])
data, meta = synth.make_dataset(n, N, k=30, seed=0)
A = data.obsp['connectivities'].tocsr()
codes, labels = sample_codes(data.obs['id'])
C = np.bincount(codes, minlength=N).astype(float)
orders = sys.argv[3].split(',') if len(sys.argv) > 3 else ['rcm', 'cluster:16', 'cluster:64', 'cluster:256']
chunks = sys.argv[4].split(',') if len(sys.argv) > 4 else ['', '64', '512']
for order in orders:
    for chunk in chunks:
        os.environ['CNA_ORDER'] = order
        if chunk and chunk != 'n/8': os.environ['CNA_XCD_CHUNK'] = chunk
        else: os.environ.pop('CNA_XCD_CHUNK', None)
        for sparse in (('1', '0') if N >= 96 else ('1',)):
            os.environ['CNA_SPARSE_MIN_N'] = '96' if sparse == '1' else '0'
            eng = Engine(device=0)
            t = time.time(); eng.ensure_graph(A); t_up = time.time() - t
            eng.colsums(1)
            for rep in range(3):
                eng.set_samples(codes, N, C)
                if rep == 1: eng.prof_reset(); eng.prof_enable(True)
                eng.nam_step(False, True, False); eng.nam_step(False, True, False); eng.nam_step(False, False, True)
            eng.sync(); eng.prof_enable(False)
            print('%-12s xcd_chunk=%-5s sparse2=%s upload %.2fs ' % (order, chunk or 'n/8', sparse, t_up),
                  {k: round(v[0] / v[1] * 1e3, 1) for k, v in eng.prof().items() if k.startswith('nam')}, flush=True)
            eng.close()
