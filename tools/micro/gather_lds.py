"""Inputs of tools/micro/gather_lds.hip: a synthetic kNN graph in the cluster order of
csrc/host_graph.c (B = 32, 64, 128) with the per-block source lists.  Run on the GPU box:
    python tools/micro/gather_lds.py /tmp/gl 500000 && ./gather_lds /tmp/gl 200"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cna_amd import synth, _order  # noqa: E402


def main():
    d, n = sys.argv[1], int(sys.argv[2])
    os.makedirs(d, exist_ok=True)
    lib = C.CDLL(os.path.join(ROOT, 'cna_amd', 'libcna_hip.so'))
    lib.cna_host_cluster_order.restype = C.c_int64
    lib.cna_host_cluster_order.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import micro_host
    mlib = micro_host.load()
    mlib.micro_block_sources.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_void_p]
    t = time.time()
    X, _ = synth.mixture_points(n)
    A = synth.fuzzy_knn_graph(X, k=30)
    print('graph %.1fs nnz/row %.1f' % (time.time() - t, A.nnz / n), flush=True)
    indptr0 = A.indptr.astype(np.int64)
    indices0 = A.indices.astype(np.int32)
    for B, caps in ((32, (544,)), (64, (960, 1920))):
        order = np.zeros(n, np.int64)
        t = time.time()
        nf = lib.cna_host_cluster_order(n, indptr0.ctypes.data, indices0.ctypes.data, B, order.ctypes.data)
        t_order = time.time() - t
        indptr, indices, data = _order.permuted_rows(A, order, 0, n)
        b = '%s/b%d_' % (d, B)
        indptr.astype(np.int64).tofile(b + 'indptr.bin')
        indices.astype(np.int32).tofile(b + 'idx.bin')
        data.astype(np.float32).tofile(b + 'val.bin')
        nb = (n + B - 1) // B
        for cap in caps:
            sp_ = np.zeros(nb + 1, np.int64)
            src = np.zeros(len(indices), np.int32)
            slot = np.zeros(len(indices), np.uint16)
            t = time.time()
            tot = mlib.micro_block_sources(n, n, indptr.ctypes.data, indices.ctypes.data, B, cap, sp_.ctypes.data,
                                             src.ctypes.data, slot.ctypes.data)
            print('B=%d: order %.2fs (%.1f%% in full clusters), sources cap %d: %.2fs, edges/sources %.2f' % (
                B, t_order, 100.0 * nf / n, cap, time.time() - t, len(indices) / tot), flush=True)
            s = b + 'cap%d_' % cap
            sp_.tofile(s + 'srcptr.bin')
            src[:tot].tofile(s + 'src.bin')
            slot.tofile(s + 'slot.bin')


if __name__ == '__main__':
    main()
