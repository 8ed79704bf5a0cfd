// Sample-space kernels of the permutation test (reference _association.py:35-61,84,96-97).
// Small (N x Nnull) but on the critical path at Nnull = 10000; as batched GEMMs on the matrix cores:
//   k_cond_gemm + k_cond_scale   Zc = M.Y / std(M.Y, ddof=1) for the observed + permuted phenotypes; the
//                 result stays resident: input of the global F-tests and B operand of the local-null kernel
//   k_gt_project  squared projections on the first kmax sample-PCs (U^T.Zc) and ssered per column
//   k_gt_ftest    one thread per (k in ks, column): F statistic and its survival function (regularised
//                 incomplete beta); k_gt_min: min-p over ks and its r2
// (round 1: one wave per column with serial N-length loops and at most len(ks) busy lanes)
#include "common.h"

namespace {

// regularised incomplete beta I_x(a, b) by the continued fraction (modified Lentz); used as
// the F survival function  sf(f; d1, d2) = I_{d2/(d2+d1 f)}(d2/2, d1/2)  =  scipy.special.fdtrc
__device__ double betacf(double a, double b, double x) {
  const double tiny = 1e-300, eps = 1e-16;
  const double qab = a + b, qap = a + 1.0, qam = a - 1.0;
  double c = 1.0, d = 1.0 - qab * x / qap;
  if (fabs(d) < tiny) d = tiny;
  d = 1.0 / d;
  double h = d;
  for (int m = 1; m <= 500; ++m) {
    const double m2 = 2.0 * m;
    double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
    d = 1.0 + aa * d;
    if (fabs(d) < tiny) d = tiny;
    c = 1.0 + aa / c;
    if (fabs(c) < tiny) c = tiny;
    d = 1.0 / d;
    h *= d * c;
    aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
    d = 1.0 + aa * d;
    if (fabs(d) < tiny) d = tiny;
    c = 1.0 + aa / c;
    if (fabs(c) < tiny) c = tiny;
    d = 1.0 / d;
    const double del = d * c;
    h *= del;
    if (fabs(del - 1.0) < eps) break;
  }
  return h;
}

__device__ double incbet(double a, double b, double x) {
  if (!(x > 0.0)) return 0.0;
  if (x >= 1.0) return 1.0;
  const double lbeta = lgamma(a) + lgamma(b) - lgamma(a + b);
  const double front = exp(a * log(x) + b * log1p(-x) - lbeta);
  if (x < (a + 1.0) / (a + b + 2.0)) return front * betacf(a, b, x) / a;
  return 1.0 - front * betacf(b, a, 1.0 - x) / b;
}

// F survival function with scipy.stats.f.sf's conventions at the edges
__device__ double f_sf(double f, double d1, double d2) {
  if (!(d2 > 0.0) || f != f) return __builtin_nan("");
  if (f <= 0.0) return 1.0;
  if (f == __builtin_inf()) return 0.0;
  return incbet(0.5 * d2, 0.5 * d1, d2 / (d2 + d1 * f));
}

typedef double v4d __attribute__((ext_vector_type(4)));

// ---- Zc = M.Y / std(M.Y, ddof=1)  (_association.py:51-52, 96-97) ---------------------------------------
// M.Y on the matrix cores: one wave per 16 x 16 tile of the product, v_mfma_f64_16x16x4_f64 over the
// sample axis (operand layouts as in mfma.hip: A lane (m = l & 15, k = l >> 4), B lane (k = l >> 4,
// n = l & 15), C register r of lane l = element (m = (l >> 4) + 4 r, n = l & 15)).  M (N x N) and Y
// (N x P, row-major) are L2 resident; rows and columns past the edge enter as zeros.
__global__ __launch_bounds__(256) void k_cond_gemm(const double* __restrict__ M, const double* __restrict__ Y, int N,
                                                   int P, double* __restrict__ Z, int ldy) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i0 = blockIdx.y * 16, p0 = (blockIdx.x * 4 + wv) * 16;
  if (p0 >= P) return;
  const int ai = lane & 15, ak = lane >> 4;
  const bool arow = i0 + ai < N, bcol = p0 + ai < P;
  const double* __restrict__ mp = M + (size_t)(arow ? i0 + ai : 0) * N;
  v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
  for (int j0 = 0; j0 < N; j0 += 4) {
    const int j = j0 + ak;
    const double a = (arow && j < N) ? mp[j] : 0.0;
    const double b = (bcol && j < N) ? Y[(size_t)j * P + p0 + ai] : 0.0;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  if (bcol) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + ak + 4 * r;
      if (i < N) Z[(size_t)i * ldy + p0 + ai] = acc[r];
    }
  }
}

// per column: mean and standard deviation (ddof = 1) over the N rows, then the division -- one thread per
// column, consecutive threads on consecutive columns (coalesced), rows walked in order (two passes, like
// pandas / numpy: mean first, then squared deviations)
// (round 5: the three passes fetch eight rows ahead of the sequential adds -- the loop was one exposed memory latency per
// row: 279 us for 200 x 1001 at C4, 465 us at 10 001 columns; the order of the additions is unchanged)
__global__ __launch_bounds__(256) void k_cond_scale(double* __restrict__ Z, int N, int P, int ldy) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  double* __restrict__ z = Z + p;
  constexpr int U = 8;
  double s = 0.0;
  int i = 0;
  for (; i + U <= N; i += U) {
    double v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = z[(size_t)(i + u) * ldy];
#pragma unroll
    for (int u = 0; u < U; ++u) s += v[u];
  }
  for (; i < N; ++i) s += z[(size_t)i * ldy];
  const double mean = s / (double)N;
  double ss = 0.0;
  for (i = 0; i + U <= N; i += U) {
    double v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = z[(size_t)(i + u) * ldy];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const double d = v[u] - mean;
      ss += d * d;
    }
  }
  for (; i < N; ++i) {
    const double d = z[(size_t)i * ldy] - mean;
    ss += d * d;
  }
  const double sd = sqrt(ss / (double)(N - 1));
  for (i = 0; i + U <= N; i += U) {
    double v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = z[(size_t)(i + u) * ldy];
#pragma unroll
    for (int u = 0; u < U; ++u) z[(size_t)(i + u) * ldy] = v[u] / sd;
  }
  for (; i < N; ++i) z[(size_t)i * ldy] /= sd;
}

// ---- global F-tests of every column (_association.py:35-61,84) ------------------------------------------
// beta = U[:, :kmax]^T . Zc on the matrix cores (one wave per 16 PCs x 16 columns), stored squared; the
// B-operand lanes also sum zc^2 per column (= ssered, _association.py:43)
__global__ __launch_bounds__(256) void k_gt_project(const double* __restrict__ Zc, int ldy, int N, int P,
                                                    const double* __restrict__ U, int kmax,
                                                    double* __restrict__ beta2 /* kmax x P */,
                                                    double* __restrict__ ssered /* P */) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int k0 = blockIdx.y * 16, p0 = (blockIdx.x * 4 + wv) * 16;
  if (p0 >= P) return;
  const int ai = lane & 15, ak = lane >> 4;
  const bool arow = k0 + ai < kmax, bcol = p0 + ai < P;
  v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
  double s2 = 0.0;
  for (int i0 = 0; i0 < N; i0 += 4) {
    const int i = i0 + ak;
    const double a = (arow && i < N) ? U[(size_t)i * kmax + k0 + ai] : 0.0;      // U^T: PC k, sample i
    const double b = (bcol && i < N) ? Zc[(size_t)i * ldy + p0 + ai] : 0.0;
    s2 += b * b;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  if (blockIdx.y == 0) {
    s2 += __shfl_xor(s2, 16);
    s2 += __shfl_xor(s2, 32);
    if (ak == 0 && bcol) ssered[p0 + ai] = s2;
  }
  if (bcol) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = k0 + ak + 4 * r;
      if (k < kmax) beta2[(size_t)k * P + p0 + ai] = acc[r] * acc[r];
    }
  }
}

// one thread per (k in ks, column): fit = sum of the first k squared projections, F statistic, survival
// function (K x P independent incomplete-beta evaluations), r2
__global__ __launch_bounds__(256) void k_gt_ftest(const double* __restrict__ beta2, const double* __restrict__ ssered,
                                                  int N, int P, const int32_t* __restrict__ ks, int K, int r,
                                                  double* __restrict__ pk /* K x P */, double* __restrict__ r2k) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)K * P) return;
  const int a = (int)(t / P), p = (int)(t - (int64_t)a * P);
  const int k = ks[a];
  double fit = 0.0;
  for (int j = 0; j < k; ++j) fit += beta2[(size_t)j * P + p];      // ||Uk Uk^T z||^2
  const double red = ssered[p];
  const double ssefull = red - fit;
  const double n = (double)N;
  const double f = ((red - ssefull) / (double)k) / (ssefull / n);     // _association.py:45
  pk[t] = f_sf(f, (double)k, n - (1.0 + r + k));
  r2k[t] = 1.0 - ssefull / red;
}

// np.nanargmin over ks (_association.py:60)
__global__ __launch_bounds__(256) void k_gt_min(const double* __restrict__ pk, const double* __restrict__ r2k, int P, int K,
                                                double* __restrict__ minp, double* __restrict__ r2out,
                                                int32_t* __restrict__ kidx) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  int best = -1;
  double pb = 0.0;
  for (int a = 0; a < K; ++a) {
    const double v = pk[(size_t)a * P + p];
    if (v == v && (best < 0 || v < pb)) { best = a; pb = v; }
  }
  kidx[p] = best;
  minp[p] = best < 0 ? __builtin_nan("") : pb;
  r2out[p] = best < 0 ? __builtin_nan("") : r2k[(size_t)best * P + p];
}

}  // namespace

int launch_condition(cna_ctx* c, hipStream_t st, const double* M_dev, const double* Y_dev, int N, int P,
                     double* Zc_dev, int ldy) {
  if (P == 0) return 0;
  ProfScope ps(c, CNA_K_CONDITION, st);
  hipLaunchKernelGGL(k_cond_gemm, dim3((P + 63) / 64, (N + 15) / 16), dim3(256), 0, st, M_dev, Y_dev, N, P, Zc_dev, ldy);
  hipLaunchKernelGGL(k_cond_scale, dim3((P + 255) / 256), dim3(256), 0, st, Zc_dev, N, P, ldy);
  HIP_TRY(hipGetLastError());
  return 0;
}

// scratch of the global test: beta2 kmax x P | ssered P | pk K x P | r2k K x P  (doubles)
int64_t global_test_scratch_doubles(int P, int kmax, int K) { return (int64_t)P * (kmax + 1 + 2 * K); }

int launch_global_test(cna_ctx* c, hipStream_t st, const double* Zc_dev, int ldy, int N, int P, const double* U_dev,
                       int kmax, const int32_t* ks_dev, int K, int r, double* work, double* minp_dev, double* r2_dev,
                       int32_t* kidx_dev) {
  if (P == 0) return 0;
  ProfScope ps(c, CNA_K_GLOBAL_TEST, st);
  double* beta2 = work;
  double* ssered = beta2 + (size_t)kmax * P;
  double* pk = ssered + P;
  double* r2k = pk + (size_t)K * P;
  hipLaunchKernelGGL(k_gt_project, dim3((P + 63) / 64, (kmax + 15) / 16), dim3(256), 0, st, Zc_dev, ldy, N, P, U_dev, kmax,
                     beta2, ssered);
  const int64_t nt = (int64_t)K * P;
  hipLaunchKernelGGL(k_gt_ftest, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, beta2, ssered, N, P, ks_dev, K, r, pk, r2k);
  hipLaunchKernelGGL(k_gt_min, dim3((P + 255) / 256), dim3(256), 0, st, pk, r2k, P, K, minp_dev, r2_dev, kidx_dev);
  HIP_TRY(hipGetLastError());
  return 0;
}
