"""Inputs of tools/micro/gather_group.hip: a synthetic kNN graph in several device orders and the
merged neighbour lists of groups of R consecutive rows.  Run on the GPU box:
    python tools/micro/gather_group.py /tmp/gg 500000 && ./gather_group /tmp/gg 200
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cna_amd import synth, _order  # noqa: E402


def greedy_groups(A, perm, R):
    """Seeds in RCM order; a group = seed + its R-1 strongest not yet grouped neighbours; cells whose
    neighbourhood is used up go to the end (grouped consecutively)."""
    n = A.shape[0]
    indptr, indices, data = A.indptr, A.indices, A.data
    grouped = np.zeros(n, bool)
    out, left = [], []
    for i in perm:
        if grouped[i]:
            continue
        lo, hi = indptr[i], indptr[i + 1]
        nb = indices[lo:hi]
        ok = ~grouped[nb]
        nb = nb[ok]
        if len(nb) < R - 1:
            left.append(i)
            grouped[i] = True
            continue
        w = data[lo:hi][ok]
        pick = nb[np.argsort(-w, kind='stable')[:R - 1]]
        grouped[i] = True
        grouped[pick] = True
        out.append(i)
        out.extend(pick.tolist())
    return np.array(out + left, dtype=np.int64), len(left)


def write_case(d, tag, A, order, R):
    n = A.shape[0]
    indptr, indices, data = _order.permuted_rows(A, order, 0, n)
    indptr.astype(np.int64).tofile('%s/%s_indptr.bin' % (d, tag))
    indices.astype(np.int32).tofile('%s/%s_idx.bin' % (d, tag))
    data.astype(np.float32).tofile('%s/%s_val.bin' % (d, tag))
    if not R:
        return
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(indptr))
    grp, slot = rows // R, rows % R
    key = grp * n + indices
    o = np.argsort(key, kind='stable')
    ks = key[o]
    first = np.concatenate([[True], ks[1:] != ks[:-1]])
    eid = np.cumsum(first) - 1
    ne = int(eid[-1]) + 1
    ej = (ks[first] % n).astype(np.int32)
    eg = ks[first] // n
    em = np.zeros(ne, dtype=np.uint32)
    np.add.at(em, eid, (1 << slot[o]).astype(np.uint32))
    ew = np.zeros((ne, R), dtype=np.float32)
    ew[eid, slot[o]] = data[o]
    ng = (n + R - 1) // R
    gptr = np.zeros(ng + 1, dtype=np.int64)
    np.cumsum(np.bincount(eg, minlength=ng), out=gptr[1:])
    gptr.tofile('%s/%s_gptr.bin' % (d, tag))
    ej.tofile('%s/%s_ej.bin' % (d, tag))
    em.tofile('%s/%s_em.bin' % (d, tag))
    ew.tofile('%s/%s_ew.bin' % (d, tag))
    print('%s: R=%d edges/entries = %.2f' % (tag, R, len(indices) / ne), flush=True)


def main():
    d, n = sys.argv[1], int(sys.argv[2])
    os.makedirs(d, exist_ok=True)
    t = time.time()
    X, _ = synth.mixture_points(n)
    A = synth.fuzzy_knn_graph(X, k=30)
    print('graph %.1fs nnz/row %.1f' % (time.time() - t, A.nnz / n), flush=True)
    t = time.time()
    perm = _order.locality_order(A)
    print('rcm %.1fs' % (time.time() - t), flush=True)
    write_case(d, 'rcm', A, perm, 0)
    for R in (8, 16):
        write_case(d, 'c%d' % R, A, perm, R)          # groups = R consecutive rows of the RCM order
    for R in (4, 8, 16):
        t = time.time()
        order, nleft = greedy_groups(A, perm, R)
        print('greedy R=%d %.1fs leftover %.1f%%' % (R, time.time() - t, 100.0 * nleft / n), flush=True)
        write_case(d, 'g%d' % R, A, order, R)


if __name__ == '__main__':
    main()
