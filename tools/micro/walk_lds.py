"""Inputs of tools/micro/walk_lds.hip: a synthetic kNN graph in a cluster order of csrc/host_graph.c
with the blocks of micro_walk_blocks (tools/micro/host_walk.c) (variable-size blocks, sorted source lists).  Run on the GPU box:
    python tools/micro/walk_lds.py /tmp/wl 500000 64 1152 [cluster size] && ./walk_lds /tmp/wl 200"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cna_amd import synth, _order  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import micro_host  # noqa: E402


def main():
    d, n = sys.argv[1], int(sys.argv[2])
    bmax, cap = int(sys.argv[3]), int(sys.argv[4])
    cluster = int(sys.argv[5]) if len(sys.argv) > 5 else 64
    os.makedirs(d, exist_ok=True)
    t = time.time()
    X, _ = synth.mixture_points(n)
    A = synth.fuzzy_knn_graph(X, k=30)
    print('graph %.1fs nnz/row %.1f' % (time.time() - t, A.nnz / n), flush=True)
    t = time.time()
    order = _order.cluster_order(A, cluster)
    t_order = time.time() - t
    indptr, indices, data = _order.permuted_rows(A, order, 0, n)
    t = time.time()
    blk_row, src_ptr, src, slot = micro_host.walk_blocks(indptr, indices, n, bmax, cap, 512)
    nb = len(blk_row) - 1
    print('order(%d) %.2fs, blocks %.2fs: %d blocks, %.1f rows and %.0f sources per block, edges/sources %.2f, '
          'overflow edges %d' % (cluster, t_order, time.time() - t, nb, n / nb, len(src) / nb, len(indices) / len(src),
                                 int((slot == 0xFFFF).sum())), flush=True)
    indptr.astype(np.int64).tofile(d + '/indptr.bin')
    indices.astype(np.int32).tofile(d + '/idx.bin')
    data.astype(np.float32).tofile(d + '/val.bin')
    blk_row.astype(np.int64).tofile(d + '/blkrow.bin')
    src_ptr.astype(np.int64).tofile(d + '/srcptr.bin')
    src.astype(np.int32).tofile(d + '/src.bin')
    slot.astype(np.uint16).tofile(d + '/slot.bin')


if __name__ == '__main__':
    main()
