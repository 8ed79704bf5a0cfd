"""CPU checks of the host-side graph helpers of the C ABI (csrc/host_graph.c): content hash, cluster
order, per-block source lists.  No device involved."""
import pytest
import numpy as np
import scipy.sparse as sp

from cna_amd import _ffi, _order, synth
from cna_amd._ffi import ptr


def _hash(a, threads):
    a = np.ascontiguousarray(a)
    return int(_ffi.load().cna_host_hash64(ptr(a), a.nbytes, threads))


def test_hash_is_thread_count_independent_and_sees_single_bytes():
    rs = np.random.RandomState(0)
    for nbytes in (0, 1, 7, 31, 32, 33, 4096, (1 << 20) - 1, (1 << 20), (1 << 20) + 5, 5 * (1 << 20) + 123):
        buf = rs.randint(0, 256, size=nbytes).astype(np.uint8)
        h1 = _hash(buf, 1)
        assert h1 == _hash(buf, 3) == _hash(buf, 8) == _hash(buf, 200)
        if nbytes:
            for pos in {0, nbytes // 2, nbytes - 1}:
                b2 = buf.copy()
                b2[pos] ^= 1
                assert _hash(b2, 4) != h1
    # length matters, not only content
    z = np.zeros(64, dtype=np.uint8)
    assert _hash(z[:32], 1) != _hash(z, 1)


def _graph(n=6000, k=12, seed=3):
    X, _ = synth.mixture_points(n, seed=seed)
    return synth.fuzzy_knn_graph(X, k=k)


def test_cluster_order_is_a_permutation_and_shares_neighbours():
    A = _graph()
    n = A.shape[0]
    for B in (1, 8, 64, 100):
        order = _order.cluster_order(A, B)
        assert np.array_equal(np.sort(order), np.arange(n))
    # rows of a block share neighbours: fewer distinct columns per block than in the caller's order
    B = 64
    order = _order.cluster_order(A, B)
    def distinct(o):
        ip, ix, _ = _order.permuted_rows(A, o, 0, n)
        return sum(len(np.unique(ix[ip[b]:ip[min(b + B, n)]])) for b in range(0, n, B))
    rs = np.random.RandomState(0)
    assert distinct(order) < 0.6 * distinct(rs.permutation(n))
    # deterministic
    assert np.array_equal(order, _order.cluster_order(A, B))


def test_cluster_order_handles_isolated_cells_and_asymmetric_graphs():
    rs = np.random.RandomState(1)
    n = 500
    A = sp.random(n, n, density=0.01, random_state=rs, format='csr', dtype=np.float32)
    A[10:30] = 0                      # rows without edges
    A = sp.csr_matrix(A)
    A.eliminate_zeros()
    order = _order.cluster_order(A, 16)
    assert np.array_equal(np.sort(order), np.arange(n))
    assert len(_order.cluster_order(sp.csr_matrix((0, 0), dtype=np.float32), 8)) == 0


def test_block_sources_lists_every_column_once_per_block():
    """(a planner of the walk-step probes: tools/micro/host_walk.c since round 6, no longer in the product library)"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'micro'))
    import micro_host
    A = _graph(3000, 10)
    n = A.shape[0]
    B = 32
    order = _order.cluster_order(A, B)
    ip, ix, _ = _order.permuted_rows(A, order, 0, n)
    for cap in (4000, 100):
        src_ptr, src, slot = micro_host.block_sources(ip, ix, n, B, cap)
        assert src_ptr[0] == 0 and src_ptr[-1] == len(src) and len(src_ptr) == (n + B - 1) // B + 1
        for b in range(0, len(src_ptr) - 1, 7):
            lo, hi = ip[b * B], ip[min((b + 1) * B, n)]
            cols, mine = ix[lo:hi], src[src_ptr[b]:src_ptr[b + 1]]
            assert len(np.unique(mine)) == len(mine) <= cap
            listed = slot[lo:hi] != 0xFFFF
            assert np.array_equal(mine[slot[lo:hi][listed]], cols[listed])
            if cap >= 4000:
                assert listed.all() and set(mine.tolist()) == set(cols.tolist())
            else:
                assert not np.isin(cols[~listed], mine).any()       # unlisted columns really are not in the list


def test_permute_rows_equals_numpy_formulation():
    """cna_host_permute_rows (threaded) = rows perm[r0:r1] of A with columns relabelled, values bit for bit,
    neighbours in their original order."""
    rs = np.random.RandomState(0)
    n = 5000
    A = sp.random(n, n, density=0.004, random_state=rs, format='csr', dtype=np.float32)
    A.indices = A.indices.astype(np.int32)
    for dt in (np.float32, np.float64):
        B = A.astype(dt)
        perm = rs.permutation(n).astype(np.int64)
        for r0, r1 in ((0, n), (100, 4000), (7, 7)):
            ip, ix, da = _order.permuted_rows(B, perm, r0, r1)
            ref = sp.csr_matrix(B)[perm[r0:r1]]
            assert np.array_equal(ip, ref.indptr.astype(np.int64))
            assert np.array_equal(ix, _order.inverse(perm)[ref.indices]) and ix.dtype == np.int32
            assert np.array_equal(da, ref.data) and da.dtype == dt


@pytest.mark.parametrize('n,threads', [(0, 4), (1, 4), (131071, 3), (3_000_001, 8), (2_000_000, 1)])
def test_host_copy_is_a_memcpy_whatever_the_thread_count(n, threads):
    """cna_host_copy (the per-cell result column into the caller's frame) on odd sizes and thread counts."""
    from cna_amd import _ffi
    lib = _ffi.load()
    rs = np.random.RandomState(n % 97)
    src = rs.randn(n)
    dst = np.full(n + 2, -7.0)
    assert lib.cna_host_copy(dst[1:1 + n].ctypes.data if n else None, src.ctypes.data if n else None, 8 * n, threads) == 0
    assert np.array_equal(dst[1:1 + n], src) and dst[0] == -7.0 and dst[-1] == -7.0


def test_cluster_graph_counts_edges_between_clusters():
    """cna_host_cluster_graph (what _order.partition_order packs the blocks of a sharded run from): entry (c, d) = number
    of edges between the cells of cluster c and of cluster d of a cell order, against scipy's sparse product P^T A P."""
    import scipy.sparse as sp
    from cna_amd import _order
    rs = np.random.RandomState(4)
    n, B = 1000, 64
    A = sp.random(n, n, density=0.01, random_state=rs, format='csr')
    A = (A + A.T).tocsr()
    A.setdiag(0)
    A.eliminate_zeros()
    order = rs.permutation(n)
    W = _order.cluster_graph(A, order, B)
    nc = -(-n // B)
    cl = np.empty(n, dtype=np.int64)
    cl[order] = np.arange(n) // B
    P = sp.csr_matrix((np.ones(n), (np.arange(n), cl)), shape=(n, nc))
    pattern = A.copy()
    pattern.data[:] = 1.0
    want = (P.T @ pattern @ P).toarray()
    np.fill_diagonal(want, 0.0)
    np.testing.assert_array_equal(W.toarray(), want)
    assert W.shape == (nc, nc) and (W != W.T).nnz == 0
