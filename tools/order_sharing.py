#!/usr/bin/env python3
"""Host-only probe (no GPU): how many neighbour rows do g CONSECUTIVE destination rows of the device order share?

    order_sharing.py [n_cells=300000] [clusters_sampled=150]

The dense walk step fetches, per destination row, the state rows of its ~40 graph neighbours; a kernel in which one wave
owns g consecutive rows and walks the union of their neighbour lists loads each distinct neighbour row once, so both the
gathered bytes and the bytes from behind the L2 fall by  edges / distinct neighbour rows  of the group.  Round 4 measured
that ratio only on the library's CURRENT order (1.12 / 1.27 / 1.45 / 1.68 at 2 / 4 / 8 / 16 rows).  This probe keeps the
512-cell clusters of that order (host_graph.c:cna_host_cluster_order) and varies the order INSIDE each cluster:

  current      as the library emits it (rows in the order the cluster was grown)
  chain        greedy chain: next row = the unvisited row of the cluster sharing most neighbours with the last g rows
  groups-g     greedy groups of exactly g rows: seed = unvisited row sharing most with the previous group, then g-1 times
               the row sharing most with the group's union (this maximises the very ratio that is printed, greedily)
  bisect       recursive spectral bisection of the cluster's sub-graph (Fiedler vector of the shared-neighbour graph)
  pair-bound   for g = 2 only: every row paired with its best partner anywhere in the cluster (not a valid order -- rows
               are reused -- an upper bound on what any order can give pairs)

Kill criterion of the round-4 review: < 1.9 at 8 rows => the union kernel is not worth building."""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('CNA_SYNTH_CPU', '1')

from cna_amd import synth, _order  # noqa: E402

GS = (2, 4, 8, 16, 64)


def ratio_of_order(B, order, g):
    """edges / distinct columns over aligned groups of g consecutive rows of `order` (B: rows x columns 0/1, dense)."""
    edges = distinct = 0
    for s in range(0, len(order), g):
        blk = B[order[s:s + g]]
        edges += int(blk.sum())
        distinct += int(blk.any(axis=0).sum())
    return edges, distinct


def chain_order(B, S, g):
    n = B.shape[0]
    left = np.ones(n, dtype=bool)
    cur = int(np.argmax(S.sum(axis=1)))
    order = [cur]
    left[cur] = False
    Bf = B.astype(np.float32)
    for _ in range(n - 1):
        u = Bf[order[-(g - 1):]].max(axis=0) if g > 1 else Bf[order[-1]]
        sc = Bf @ u
        sc[~left] = -1
        cur = int(np.argmax(sc))
        order.append(cur)
        left[cur] = False
    return np.array(order)


def group_order(B, g):
    n = B.shape[0]
    left = np.ones(n, dtype=bool)
    Bf = B.astype(np.float32)
    order = []
    prev = None
    while left.any():
        if prev is None:
            seed = int(np.flatnonzero(left)[0])
        else:
            sc = Bf @ prev
            sc[~left] = -1
            seed = int(np.argmax(sc))
        grp = [seed]
        left[seed] = False
        u = Bf[seed].copy()
        while len(grp) < g and left.any():
            sc = Bf @ u
            sc[~left] = -1
            nxt = int(np.argmax(sc))
            grp.append(nxt)
            left[nxt] = False
            np.maximum(u, Bf[nxt], out=u)
        order += grp
        prev = u
    return np.array(order)


def bisect_order(S, idx=None, leaf=2):
    if idx is None:
        idx = np.arange(S.shape[0])
    if len(idx) <= leaf:
        return list(idx)
    W = S[np.ix_(idx, idx)].astype(np.float64)
    np.fill_diagonal(W, 0)
    d = W.sum(axis=1)
    L = np.diag(d) - W
    try:
        w, v = np.linalg.eigh(L)
        f = v[:, 1]
    except Exception:
        f = np.arange(len(idx), dtype=float)
    o = np.argsort(f, kind='stable')
    h = len(idx) // 2
    return bisect_order(S, idx[o[:h]], leaf) + bisect_order(S, idx[o[h:]], leaf)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
    ncl = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    t = time.time()
    data, _ = synth.make_dataset(n, 50, k=30, seed=0, builder='cpu')
    A = data.obsp['connectivities'].tocsr()
    print('graph: %d cells, %.1f nnz/row (%.0f s)' % (n, A.nnz / n, time.time() - t), flush=True)
    t = time.time()
    perm = _order.cluster_order(A, 512)              # device row -> caller's row
    print('cluster order: %.1f s' % (time.time() - t), flush=True)
    nclusters = n // 512
    rs = np.random.RandomState(0)
    pick = np.sort(rs.choice(nclusters, size=min(ncl, nclusters), replace=False))
    names = ['current', 'chain', 'groups-g', 'bisect']
    tot = {(nm, g): [0, 0] for nm in names for g in GS}
    pair_bound = [0, 0]
    in_cluster = [0, 0]
    t = time.time()
    for ci, c in enumerate(pick):
        rows = perm[c * 512:(c + 1) * 512]
        sub = A[rows]
        cols, inv = np.unique(sub.indices, return_inverse=True)
        B = np.zeros((len(rows), len(cols)), dtype=np.uint8)
        B[np.repeat(np.arange(len(rows)), np.diff(sub.indptr)), inv] = 1
        Bf = B.astype(np.float32)
        S = Bf @ Bf.T                                # shared-neighbour counts of every pair of rows
        in_cluster[0] += int(np.isin(sub.indices, rows).sum())
        in_cluster[1] += sub.nnz
        deg = B.sum(axis=1).astype(np.int64)
        S0 = S.copy()
        np.fill_diagonal(S0, -1)
        best = S0.max(axis=1)
        pair_bound[0] += int(2 * deg.sum())
        pair_bound[1] += int((2 * deg - best).sum())  # ~ |N(i)| + |N(j*)| - shared, with |N(j*)| ~ |N(i)|
        cur = np.arange(len(rows))
        bis = np.array(bisect_order(S))
        for g in GS:
            for nm, order in (('current', cur), ('chain', chain_order(B, S, g) if g <= 16 else cur),
                              ('groups-g', group_order(B, g)), ('bisect', bis)):
                e, d = ratio_of_order(B, order, g)
                tot[(nm, g)][0] += e
                tot[(nm, g)][1] += d
        if ci % 25 == 24:
            print('  %d clusters, %.0f s' % (ci + 1, time.time() - t), flush=True)
    print('\n%d cells, k=30 (8-d mixture of 20 Gaussians, seed 0), %d of %d clusters of 512 rows sampled' % (n, len(pick), nclusters))
    print('edges whose neighbour lies in the same 512-row cluster: %.1f %%' % (100.0 * in_cluster[0] / in_cluster[1]))
    print('edges / distinct neighbour rows of g consecutive rows')
    print('%-10s' % 'order' + ''.join('%8s' % ('g=%d' % g) for g in GS))
    for nm in names:
        print('%-10s' % nm + ''.join('%8.2f' % (tot[(nm, g)][0] / tot[(nm, g)][1]) for g in GS))
    print('%-10s%8.2f   (every row with its best partner in the cluster; not an order)' % ('pair-bound', pair_bound[0] / pair_bound[1]))
    g8 = max(tot[(nm, 8)][0] / tot[(nm, 8)][1] for nm in names)
    print('\nbest at 8 rows: %.2f -> %s' % (g8, 'BUILD the union kernel' if g8 >= 1.9 else
                                           'below the 1.9 kill criterion: the union kernel is closed for this formulation'))


if __name__ == '__main__':
    main()
