"""Synthetic inputs for parity tests and benchmarks.

Nothing here is on the accelerated path: it only manufactures the *inputs* of
``cna.tl.association`` -- an AnnData-like object whose ``obsp['connectivities']``
looks like what ``scanpy.pp.neighbors`` emits (UMAP fuzzy-union kNN graph,
CSR float32/int32, empty diagonal, symmetric; /root/reference/demo/demo.ipynb:590)
and whose ``obs[sid]`` assigns every cell to a sample.

scanpy / umap-learn are not installed in this image, so the graph builder below is
our own stand-in: exact kNN with ``scipy.spatial.cKDTree`` plus UMAP's published
smooth-kNN weighting.  It is an input generator, not a parity target.
"""
import numpy as np
import pandas as pd
import scipy.sparse as sp
from scipy.spatial import cKDTree


class CellData:
    """Minimal AnnData duck type: ``.obs`` (DataFrame), ``.obsp``, ``.uns``.

    The reference only touches ``data.obs[...]``, ``data.obsp['connectivities']``
    (or ``data.uns['neighbors']['connectivities']``) -- /root/reference/src/cna/tools/_nam.py:12-19.
    """

    def __init__(self, obs, connectivities):
        self.obs = obs
        self.obsp = {'connectivities': connectivities}
        self.uns = {'neighbors': {'connectivities': connectivities}}

    @property
    def n_obs(self):
        return len(self.obs)

    def __len__(self):
        return len(self.obs)


def mixture_points(n, dim=8, n_clusters=20, seed=0, spread=4.0, cluster_sorted=True):
    """n points in R^dim from a mixture of Gaussians; returns (X float32, cluster int32)."""
    rs = np.random.RandomState(seed)
    centers = rs.randn(n_clusters, dim) * spread
    weights = rs.dirichlet(np.full(n_clusters, 5.0))
    cl = rs.choice(n_clusters, size=n, p=weights).astype(np.int32)
    if cluster_sorted:
        cl.sort()
    X = centers[cl] + rs.randn(n, dim)
    return X.astype(np.float32), cl


def _usable_cpus():
    """CPUs this process may really use: the cgroup v2 quota if there is one, else the affinity mask
    (a 256-thread kd-tree query under a 16-CPU quota only earns throttling)."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def _gpu_builder_available(X, k, dtype):
    import os
    if os.environ.get('CNA_SYNTH_CPU', '0') not in ('0', '', 'off', 'no'):
        return False
    if np.dtype(dtype) != np.float32 or X.shape[1] > 64 or not (2 <= k <= 65) or X.shape[0] <= k:
        return False
    try:
        import ctypes as C
        from . import _ffi
        cnt = C.c_int(0)
        return _ffi.load().cna_device_count(C.byref(cnt)) == 0 and cnt.value > 0
    except Exception:
        return False


_knn_engine = None


def _builder_engine():
    """A context of its own, WITHOUT a communicator, for the input builder: in a multi-rank job only rank 0
    generates the dataset, and creating the process-wide engine there would enter RCCL's collective
    communicator set-up while the other ranks wait elsewhere."""
    global _knn_engine
    if _knn_engine is None:
        from . import dist
        from .engine import Engine
        _knn_engine = Engine(device=dist.current().get('device'), rank=0, nranks=1)
    return _knn_engine


def fuzzy_knn_graph(X, k=30, dtype=np.float32, workers=None, n_iter=40, builder='auto'):
    """UMAP-style connectivities for points X with ``k`` neighbours (self included,
    as scanpy counts them), symmetrised by fuzzy union A + A^T - A*A^T.

    builder: 'auto' uses the device builder (csrc/knn.hip: brute-force kNN, weights and fuzzy union on
    the GPU -- seconds at 2M points) when a GPU is visible, else this function's cKDTree + scipy.sparse
    path ('cpu'); CNA_SYNTH_CPU=1 forces the latter.  Both are exact kNN; they may differ in how ties and
    float32 / float64 distance roundings fall."""
    if builder == 'gpu' or (builder == 'auto' and _gpu_builder_available(X, k, dtype)):
        return _builder_engine().knn_graph(X, k)
    n = X.shape[0]
    kk = min(k, n)
    tree = cKDTree(X)
    dist, idx = tree.query(X, k=kk, workers=workers or _usable_cpus())
    dist = dist[:, 1:].astype(np.float64)      # drop self
    idx = idx[:, 1:]
    m = dist.shape[1]
    rho = dist[:, 0].copy()
    target = np.log2(kk)
    lo = np.zeros(n)
    hi = np.full(n, np.inf)
    sigma = np.ones(n)
    d0 = np.maximum(dist - rho[:, None], 0.0)
    for _ in range(n_iter):
        val = np.exp(-d0 / sigma[:, None]).sum(axis=1)
        too_big = val > target
        hi = np.where(too_big, sigma, hi)
        lo = np.where(too_big, lo, sigma)
        sigma = np.where(np.isinf(hi), sigma * 2.0, 0.5 * (lo + hi))
    w = np.exp(-d0 / sigma[:, None])
    rows = np.repeat(np.arange(n, dtype=np.int64), m)
    A = sp.csr_matrix((w.ravel(), (rows, idx.ravel().astype(np.int64))), shape=(n, n))
    A = A + A.T - A.multiply(A.T)
    A = sp.csr_matrix(A)
    A.setdiag(0)
    A.eliminate_zeros()
    A.sort_indices()
    A = A.astype(dtype)
    A.indices = A.indices.astype(np.int32)
    A.indptr = A.indptr.astype(np.int32)
    return A


def graph_digest(A):
    """sha256 over shape, indptr (as int64), indices (int32) and the value bytes of a CSR matrix: the identity of a
    regenerated input (tests/golden/d02_config2.npz stores the digest instead of 64 MB of graph)."""
    import hashlib
    h = hashlib.sha256()
    h.update(np.asarray(A.shape, dtype=np.int64).tobytes())
    h.update(np.ascontiguousarray(A.indptr, dtype=np.int64).tobytes())
    h.update(np.ascontiguousarray(A.indices, dtype=np.int32).tobytes())
    h.update(str(A.data.dtype).encode())
    h.update(np.ascontiguousarray(A.data).tobytes())
    return h.hexdigest()


def assign_samples(cluster, n_samples, seed=0, skew=1.0):
    """Sample id per cell with a per-sample preference over clusters so the NAM
    carries signal.  Returns (sid int64[n], sample_cluster_props float64[N, K])."""
    rs = np.random.RandomState(seed + 1)
    n = len(cluster)
    K = int(cluster.max()) + 1
    logits = rs.randn(n_samples, K) * skew
    p = np.exp(logits)
    p /= p.sum(axis=0, keepdims=True)          # P(sample | cluster)
    cdf = np.cumsum(p, axis=0)                 # N x K
    u = rs.rand(n)
    sid = (u[None, :] > cdf[:, cluster]).sum(axis=0).astype(np.int64)
    sid = np.minimum(sid, n_samples - 1)
    props = np.zeros((n_samples, K))
    np.add.at(props, (sid, cluster), 1.0)
    props /= np.maximum(props.sum(axis=1, keepdims=True), 1)
    return sid, props


def make_dataset(n_cells, n_samples, k=30, seed=0, dim=8, n_clusters=20,
                 graph_dtype=np.float32, cluster_sorted=True, sid_name='id',
                 sid_kind='int', signal=True, n_covs=0, n_batches=0, builder='auto'):
    """Build a CellData plus sample-level phenotype/covariates.  ``builder``: see fuzzy_knn_graph ('cpu' gives the same
    graph on every machine of this image, GPU or not -- what fixtures captured from the reference are regenerated with).

    Returns (data, meta) where meta has y (Series), covs (DataFrame|None),
    batches (Series|None), props, cluster.
    """
    X, cl = mixture_points(n_cells, dim=dim, n_clusters=n_clusters, seed=seed,
                           cluster_sorted=cluster_sorted)
    A = fuzzy_knn_graph(X, k=k, dtype=graph_dtype, builder=builder)
    sid, props = assign_samples(cl, n_samples, seed=seed)
    labels = np.arange(n_samples)
    if sid_kind == 'str':
        names = np.array(['s%03d' % i for i in range(n_samples)])
        sid_col = names[sid]
        index = pd.Index(names)
    elif sid_kind == 'cat':
        names = np.array(['s%03d' % i for i in range(n_samples)])
        sid_col = pd.Categorical(names[sid], categories=list(names))
        index = pd.Index(names)
    else:
        sid_col = sid
        index = pd.Index(labels)
    obs = pd.DataFrame({sid_name: sid_col},
                       index=pd.Index(['cell_%d' % i for i in range(n_cells)], name='cell'))
    data = CellData(obs, A)
    rs = np.random.RandomState(seed + 2)
    if signal:
        yv = props[:, 0] * 10 + 0.3 * rs.randn(n_samples)
    else:
        yv = rs.randn(n_samples)
    meta = {
        'y': pd.Series(yv, index=index),
        'covs': (pd.DataFrame(rs.randn(n_samples, n_covs), index=index,
                              columns=['cov%d' % j for j in range(n_covs)])
                 if n_covs else None),
        'batches': (pd.Series(np.arange(n_samples) % n_batches, index=index)
                    if n_batches else None),
        'props': props, 'cluster': cl, 'sid': sid,
    }
    return data, meta


def make_demo_like(n_samples=50, n_genes=50, cells_per_sample=200, noise=1.0, k=15, seed=0,
                   graph_dtype=np.float32):
    """The reference's demo dataset, regenerated (recipe: /root/reference/demo/makedata.ipynb cells 2-4):
    `n_samples` samples of `cells_per_sample` cells over `n_genes` genes, three cell populations whose
    per-sample proportions depend on the sample-level covariates `case` and `male`, five batches
    tiled over the samples; expression = population profile + unit Gaussian noise drawn from
    numpy's legacy generator seeded with `seed`, in the notebook's order.  The notebook then calls
    scanpy.pp.neighbors (not installed here): the graph below is this module's own fuzzy kNN stand-in
    on the expression matrix (k = scanpy's default 15), so the dataset is demo-LIKE, not the demo.

    Returns (data, samplem) with samplem a DataFrame indexed by sample id with columns case, male, batch."""
    N, G, C = int(n_samples), int(n_genes), int(cells_per_sample)
    rs = np.random.RandomState(seed)
    samplem = pd.DataFrame(index=pd.Index(np.arange(N), name='id'))
    samplem['case'] = [0] * (N // 2) + [1] * (N - N // 2)
    q = int(2 * N / 8)
    samplem['male'] = [0] * q + [1] * q + [0] * q + [1] * (N - 3 * q)
    H = np.zeros((3, G))
    H[0, :G // 2] = 1
    H[1, G // 2:] = 1
    H[2, :G // 2] = 1
    H[2, :G // 4] = 2
    props = np.array([[0.2, -0.2], [-0.2, 0.0], [0.5, 0.5]])        # rows: case, male, baseline
    blocks = []
    for _, row in samplem.iterrows():
        pr = np.array([row['case'], row['male'], 1.0]).dot(props)     # proportions of populations 0 and 1
        ids = np.concatenate([np.full(int(p * C), i) for i, p in enumerate(pr)])
        ids = np.concatenate([ids, np.full(C - len(ids), len(pr))]).astype(int)
        W = np.zeros((C, len(pr) + 1))
        W[np.arange(C), ids] = 1
        blocks.append(W.dot(H) + noise * rs.randn(C, G))
    X = np.concatenate(blocks).astype(np.float32)
    samplem['batch'] = np.tile(np.arange(5), -(-N // 5))[:N]
    A = fuzzy_knn_graph(X, k=k, dtype=graph_dtype)
    obs = pd.DataFrame({'id': np.repeat(samplem.index.values, C)},
                       index=pd.Index(['cell_%d' % i for i in range(N * C)], name='cell'))
    return CellData(obs, A), samplem
