// Synthetic-input builder on the GPU: exact kNN graph of a point cloud with UMAP's fuzzy-union
// connectivities -- the stand-in for `scanpy.pp.neighbors` (/root/reference/demo/demo.ipynb:590,
// makedata.ipynb:117) that bench.py and the tests use to manufacture `obsp['connectivities']`.
// Not on the analysed path and not a parity target (scanpy's approximate kNN is not reproducible
// here); it exists so that a 2M-cell benchmark input takes seconds instead of half a minute of
// cKDTree + scipy.sparse on the host.
//
//   k_knn        brute force: one thread per query point, candidates streamed through LDS in tiles
//                (every lane reads the same candidate: LDS broadcasts), squared distances in f32,
//                the k-1 nearest kept sorted in registers (insertion is rare once the list is warm)
//   k_smooth     per point: rho = distance to the nearest neighbour, sigma by bisection so that
//                sum_j exp(-(d_j - rho)/sigma) = log2(k)  (UMAP smooth_knn_dist), weights w_ij
//   k_mutual     per directed edge i->j: is i in j's list?  combined weight w_ij + w_ji - w_ij w_ji
//                (fuzzy union A + A^T - A o A^T); edges whose reverse is missing are counted at j
//   k_fill       rows of the result: own edges + the reverse-only edges pointing at the row, sorted by
//                column inside the row (rank by counting, as in the column-sum kernel)
#include "common.h"
#include <algorithm>
#include <cmath>
#include <cstring>

namespace {

constexpr int KNN_MAXK = 64;       // neighbours kept per point (k - 1 <= 64)

template <int D, int KK>
__global__ __launch_bounds__(256) void k_knn(const float* __restrict__ X, int64_t n, int32_t* __restrict__ nbr,
                                             float* __restrict__ dist, int tile) {
  extern __shared__ float cand[];                         // tile x D
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float x[D];
#pragma unroll
  for (int c = 0; c < D; ++c) x[c] = q < n ? X[q * D + c] : 0.0f;
  float best[KK];
  int32_t who[KK];
#pragma unroll
  for (int i = 0; i < KK; ++i) { best[i] = 3.0e38f; who[i] = -1; }
  for (int64_t base = 0; base < n; base += tile) {
    const int m = (int)(n - base < tile ? n - base : tile);
    __syncthreads();
    for (int i = threadIdx.x; i < m * D; i += 256) cand[i] = X[base * D + i];
    __syncthreads();
    for (int j = 0; j < m; ++j) {
      const float* __restrict__ y = cand + j * D;
      float d2 = 0.0f;
#pragma unroll
      for (int c = 0; c < D; ++c) { const float t = x[c] - y[c]; d2 = fmaf(t, t, d2); }
      if (d2 < best[KK - 1] && base + j != q) {
        const int32_t id = (int32_t)(base + j);
#pragma unroll
        for (int i = KK - 1; i > 0; --i) {
          if (best[i - 1] > d2) { best[i] = best[i - 1]; who[i] = who[i - 1]; }
          else if (best[i] > d2) { best[i] = d2; who[i] = id; }
        }
        if (best[0] > d2) { best[0] = d2; who[0] = id; }
      }
    }
  }
  if (q < n) {
#pragma unroll
    for (int i = 0; i < KK; ++i) { nbr[q * KK + i] = who[i]; dist[q * KK + i] = sqrtf(best[i]); }
  }
}

// UMAP smooth_knn_dist + membership strengths, as cna_amd/synth.py:fuzzy_knn_graph evaluates them on the host
__global__ void k_smooth(const float* __restrict__ dist, const int32_t* __restrict__ nbr, int64_t n, int kk, int k,
                         float* __restrict__ w) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* d = dist + i * kk;
  const double rho = d[0];
  const double target = log2((double)k);
  double lo = 0.0, hi = 1.0 / 0.0, sigma = 1.0;
  for (int it = 0; it < 40; ++it) {
    double val = 0.0;
    for (int j = 0; j < kk; ++j)
      if (nbr[i * kk + j] >= 0) val += exp(-fmax((double)d[j] - rho, 0.0) / sigma);
    if (val > target) hi = sigma; else lo = sigma;
    sigma = hi == 1.0 / 0.0 ? sigma * 2.0 : 0.5 * (lo + hi);
  }
  for (int j = 0; j < kk; ++j) w[i * kk + j] = nbr[i * kk + j] >= 0 ? (float)exp(-fmax((double)d[j] - rho, 0.0) / sigma) : 0.0f;
}

// per directed edge: combined weight; reverse-only edges are counted at their target
__global__ void k_mutual(const int32_t* __restrict__ nbr, const float* __restrict__ w, int64_t n, int kk,
                         float* __restrict__ u, unsigned int* __restrict__ extra) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * kk) return;
  const int64_t i = e / kk;
  const int32_t j = nbr[e];
  if (j < 0) { u[e] = 0.0f; return; }
  float back = 0.0f;
  bool found = false;
  for (int t = 0; t < kk; ++t)
    if (nbr[(int64_t)j * kk + t] == (int32_t)i) { back = w[(int64_t)j * kk + t]; found = true; break; }
  const float a = w[e];
  u[e] = a + back - a * back;
  if (!found) atomicAdd(&extra[j], 1u);
}

__global__ void k_row_sizes(const int32_t* __restrict__ nbr, const float* __restrict__ u, const unsigned int* __restrict__ extra,
                            int64_t n, int kk, unsigned int* __restrict__ size) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned int c = extra[i];
  for (int t = 0; t < kk; ++t) c += (nbr[i * kk + t] >= 0 && u[i * kk + t] != 0.0f) ? 1u : 0u;   // eliminate_zeros
  size[i] = c;
}

// own edges first, then (by atomic cursor) the reverse-only edges pointing here
__global__ void k_fill(const int32_t* __restrict__ nbr, const float* __restrict__ w, const float* __restrict__ u, int64_t n,
                       int kk, const unsigned long long* __restrict__ first, unsigned int* __restrict__ cursor,
                       int32_t* __restrict__ col, float* __restrict__ val) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * kk) return;
  const int64_t i = e / kk;
  const int32_t j = nbr[e];
  if (j < 0) return;
  if (u[e] != 0.0f) {
    const unsigned long long p = first[i] + atomicAdd(&cursor[i], 1u);
    col[p] = j;
    val[p] = u[e];
  }
  bool found = false;
  for (int t = 0; t < kk; ++t)
    if (nbr[(int64_t)j * kk + t] == (int32_t)i) { found = true; break; }
  if (!found && w[e] != 0.0f) {                           // j does not list i: (j, i) = w_ij
    const unsigned long long p = first[j] + atomicAdd(&cursor[j], 1u);
    col[p] = (int32_t)i;
    val[p] = w[e];
  }
}

// sort every row by column (scanpy emits sorted indices): rank by counting inside the row
__global__ __launch_bounds__(256) void k_sort_rows(const unsigned long long* __restrict__ first, int64_t n,
                                                   const int32_t* __restrict__ col, const float* __restrict__ val,
                                                   int32_t* __restrict__ col_out, float* __restrict__ val_out) {
  const int lane = threadIdx.x & 63;
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += nw) {
    const unsigned long long lo = first[i];
    const int64_t cnt = (int64_t)(first[i + 1] - lo);
    for (int64_t a = lane; a < cnt; a += 64) {
      const int32_t c = col[lo + a];
      int64_t rank = 0;
      for (int64_t b = 0; b < cnt; ++b) {
        const int32_t cb = col[lo + b];
        rank += (cb < c || (cb == c && b < a)) ? 1 : 0;
      }
      col_out[lo + rank] = c;
      val_out[lo + rank] = val[lo + a];
    }
  }
}

// exclusive scan of u32 sizes into u64 offsets (single workgroup; the builder is not performance critical here)
__global__ __launch_bounds__(1024) void k_scan_sizes(const unsigned int* __restrict__ size, int64_t n,
                                                     unsigned long long* __restrict__ first) {
  __shared__ unsigned long long sm[1024];
  const int64_t per = (n + 1023) / 1024;
  const int64_t a = per * threadIdx.x, b = a + per < n ? a + per : n;
  unsigned long long s = 0;
  for (int64_t i = a; i < b; ++i) s += size[i];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const unsigned long long t = threadIdx.x >= o ? sm[threadIdx.x - o] : 0ull;
    __syncthreads();
    sm[threadIdx.x] += t;
    __syncthreads();
  }
  unsigned long long off = sm[threadIdx.x] - s;
  for (int64_t i = a; i < b; ++i) { first[i] = off; off += size[i]; }
  if (threadIdx.x == 1023) first[n] = sm[1023];
}

template <int D>
int launch_knn_d(hipStream_t st, const float* X, int64_t n, int kk, int32_t* nbr, float* dist) {
  const int tile = D <= 16 ? 2048 : (D <= 32 ? 1024 : 512);
  const size_t smem = sizeof(float) * (size_t)tile * D;
  dim3 grid((unsigned)((n + 255) / 256));
#define KNN_CASE(KK) \
  { static bool once = false; if (!once) { HIP_TRY(hipFuncSetAttribute((const void*)k_knn<D, KK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); once = true; } \
    hipLaunchKernelGGL((k_knn<D, KK>), grid, dim3(256), smem, st, X, n, nbr, dist, tile); }
  if (kk <= 14) KNN_CASE(14)
  else if (kk <= 29) KNN_CASE(29)
  else KNN_CASE(64)
#undef KNN_CASE
  HIP_TRY(hipGetLastError());
  return 0;
}

}  // namespace

// Builds the graph for points X (n x d, float32, row-major) with `k` neighbours counting the point
// itself (scanpy's convention): CSR of the fuzzy union, float32 values, int32 sorted column indices,
// empty diagonal.  indptr_out int64[n+1]; indices_out / data_out must hold 2 n (k-1) entries; *nnz_out
// receives the actual count.
extern "C" int cna_knn_graph(cna_ctx* c, const float* X, int64_t n, int d, int k, int64_t* indptr_out,
                             int32_t* indices_out, float* data_out, int64_t* nnz_out) {
  if (!c) CNA_FAIL(CNA_EINVAL, "null context");
  if (n < 2 || d < 1 || d > 64 || k < 2 || k - 1 > KNN_MAXK || k > n) CNA_FAIL(CNA_EINVAL, "cna_knn_graph: need 2 <= k <= min(n, 65), 1 <= d <= 64");
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  const int D = d <= 8 ? 8 : (d <= 16 ? 16 : (d <= 32 ? 32 : 64));
  const int kreq = k - 1;
  const int kk = kreq <= 14 ? 14 : (kreq <= 29 ? 29 : 64);      // slots kept by the kernel (unused ones stay -1 only if n is tiny)
  // padded copy of the points
  std::vector<float> Xp((size_t)n * D, 0.0f);
  for (int64_t i = 0; i < n; ++i) std::memcpy(&Xp[(size_t)i * D], X + (size_t)i * d, sizeof(float) * d);
  const int64_t ne = n * kk;
  const int64_t b_x = round_up64(4 * n * D, 256), b_e = round_up64(4 * ne, 256), b_n = round_up64(4 * n, 256),
                b_f = round_up64(8 * (n + 1), 256), b_out = round_up64(4 * 2 * ne, 256);
  const int64_t need = b_x + 4 * b_e + 3 * b_n + b_f + 4 * b_out;
  void* buf = nullptr;
  CNA_TRY(dev_alloc(c, &buf, (size_t)need));
  char* p = (char*)buf;
  float* Xd = (float*)p; p += b_x;
  int32_t* nbr = (int32_t*)p; p += b_e;
  float* dist = (float*)p; p += b_e;
  float* w = (float*)p; p += b_e;
  float* u = (float*)p; p += b_e;
  unsigned int* extra = (unsigned int*)p; p += b_n;
  unsigned int* size = (unsigned int*)p; p += b_n;
  unsigned int* cursor = (unsigned int*)p; p += b_n;
  unsigned long long* first = (unsigned long long*)p; p += b_f;
  int32_t* col = (int32_t*)p; p += b_out;
  float* val = (float*)p; p += b_out;
  int32_t* col2 = (int32_t*)p; p += b_out;
  float* val2 = (float*)p; p += b_out;
  int rc = 0;
  int64_t nnz = 0;
  std::vector<unsigned long long> hfirst((size_t)n + 1);
  do {
#define STEP(expr) if ((expr) != hipSuccess) { rc = 1; break; }
    STEP(hipMemcpyAsync(Xd, Xp.data(), sizeof(float) * (size_t)n * D, hipMemcpyHostToDevice, st));
    STEP(hipMemsetAsync(extra, 0, (size_t)(3 * b_n), st));
    int r = 0;
    // the kernel keeps kk slots; only the first kreq are used below (kreq == kk for k = 15 / 30)
    if (D == 8) r = launch_knn_d<8>(st, Xd, n, kk, nbr, dist);
    else if (D == 16) r = launch_knn_d<16>(st, Xd, n, kk, nbr, dist);
    else if (D == 32) r = launch_knn_d<32>(st, Xd, n, kk, nbr, dist);
    else r = launch_knn_d<64>(st, Xd, n, kk, nbr, dist);
    if (r) { rc = 2; break; }
    if (kreq < kk) {                                       // drop the surplus slots: mark them unused
      std::vector<int32_t> h((size_t)ne);
      STEP(hipMemcpyAsync(h.data(), nbr, 4 * (size_t)ne, hipMemcpyDeviceToHost, st));
      STEP(hipStreamSynchronize(st));
      for (int64_t i = 0; i < n; ++i)
        for (int t = kreq; t < kk; ++t) h[(size_t)i * kk + t] = -1;
      STEP(hipMemcpyAsync(nbr, h.data(), 4 * (size_t)ne, hipMemcpyHostToDevice, st));
      STEP(hipStreamSynchronize(st));
    }
    const unsigned gn = (unsigned)((n + 255) / 256), ge = (unsigned)((ne + 255) / 256);
    hipLaunchKernelGGL(k_smooth, dim3(gn), dim3(256), 0, st, dist, nbr, n, kk, k, w);
    hipLaunchKernelGGL(k_mutual, dim3(ge), dim3(256), 0, st, nbr, w, n, kk, u, extra);
    hipLaunchKernelGGL(k_row_sizes, dim3(gn), dim3(256), 0, st, nbr, u, extra, n, kk, size);
    hipLaunchKernelGGL(k_scan_sizes, dim3(1), dim3(1024), 0, st, size, n, first);
    hipLaunchKernelGGL(k_fill, dim3(ge), dim3(256), 0, st, nbr, w, u, n, kk, first, cursor, col, val);
    const unsigned gs = (unsigned)std::min<int64_t>((n + 3) / 4, 65536);
    hipLaunchKernelGGL(k_sort_rows, dim3(gs), dim3(256), 0, st, first, n, col, val, col2, val2);
    STEP(hipGetLastError());
    STEP(hipMemcpyAsync(hfirst.data(), first, 8 * (size_t)(n + 1), hipMemcpyDeviceToHost, st));
    STEP(hipStreamSynchronize(st));
    nnz = (int64_t)hfirst[(size_t)n];
    if (nnz > 2 * ne) { rc = 3; break; }
    STEP(hipMemcpyAsync(indices_out, col2, 4 * (size_t)nnz, hipMemcpyDeviceToHost, st));
    STEP(hipMemcpyAsync(data_out, val2, 4 * (size_t)nnz, hipMemcpyDeviceToHost, st));
    STEP(hipStreamSynchronize(st));
#undef STEP
  } while (0);
  (void)hipStreamSynchronize(st);
  dev_free(c, buf, (size_t)need);
  if (rc) CNA_FAIL(CNA_EINVAL, "cna_knn_graph: device step failed (" + std::to_string(rc) + "): " + hipGetErrorString(hipGetLastError()));
  for (int64_t i = 0; i <= n; ++i) indptr_out[i] = (int64_t)hfirst[(size_t)i];
  if (nnz_out) *nnz_out = nnz;
  return 0;
}
