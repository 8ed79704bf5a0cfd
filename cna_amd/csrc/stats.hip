// Sample-space kernels of the permutation test (reference _association.py:35-61,84,96-97).
// They are tiny (N x Nnull) but sit on the host's critical path otherwise:
//   k_condition   Zc = M.Y / std(M.Y, ddof=1) per phenotype column (observed + permutations);
//                 the result stays resident: it is both the input of the global F-tests and the
//                 B operand of the local-null kernel
//   k_global_test per column: projections on the first kmax sample-PCs, F-test for every k in ks
//                 (regularised incomplete beta), min-p and its r2
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

// one workgroup (one wave) per phenotype column
__global__ __launch_bounds__(64) void k_condition(const double* __restrict__ M, const double* __restrict__ Y,
                                                  int N, int P, double* __restrict__ Zc, int ldy) {
  extern __shared__ double sm[];          // z[N]
  const int p = blockIdx.x, lane = threadIdx.x;
  for (int j = lane; j < N; j += 64) sm[j] = Y[(size_t)j * P + p];
  __syncthreads();
  constexpr int Q = 8;                    // up to 512 samples
  double zc[Q];
  double s = 0.0;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int i = lane + 64 * q;
    zc[q] = 0.0;
    if (i < N) {
      double acc = 0.0;
      for (int j = 0; j < N; ++j) acc += M[(size_t)i * N + j] * sm[j];
      zc[q] = acc;
      s += acc;
    }
  }
  const double mean = wave_sum(s) / (double)N;
  double ss = 0.0;
#pragma unroll
  for (int q = 0; q < Q; ++q)
    if (lane + 64 * q < N) {
      const double d = zc[q] - mean;
      ss += d * d;
    }
  const double sd = sqrt(wave_sum(ss) / (double)(N - 1));     // ddof = 1 (_association.py:52)
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int i = lane + 64 * q;
    if (i < N) Zc[(size_t)i * ldy + p] = zc[q] / sd;
  }
}

__global__ __launch_bounds__(64) void k_global_test(const double* __restrict__ Zc, int ldy, int N, int P,
                                                    const double* __restrict__ U, int kmax,
                                                    const int32_t* __restrict__ ks, int K, int r,
                                                    double* __restrict__ minp, double* __restrict__ r2out,
                                                    int32_t* __restrict__ kidx) {
  extern __shared__ double sm[];          // zc[N] | beta2[kmax] | pk[K] | r2k[K]
  double* zc = sm;
  double* beta2 = sm + N;
  double* pk = beta2 + kmax;
  double* r2k = pk + K;
  const int p = blockIdx.x, lane = threadIdx.x;
  double s2 = 0.0;
  for (int i = lane; i < N; i += 64) {
    const double v = Zc[(size_t)i * ldy + p];
    zc[i] = v;
    s2 += v * v;
  }
  const double ssered = wave_sum(s2);
  __syncthreads();
  for (int j = lane; j < kmax; j += 64) {                       // beta_j = U_j . zc  (_association.py:37)
    double b = 0.0;
    for (int i = 0; i < N; ++i) b += U[(size_t)i * kmax + j] * zc[i];
    beta2[j] = b * b;
  }
  __syncthreads();
  for (int a = lane; a < K; a += 64) {
    const int k = ks[a];
    double fit = 0.0;
    for (int j = 0; j < k; ++j) fit += beta2[j];                // ||Uk Uk^T z||^2
    const double ssefull = ssered - fit;
    const double n = (double)N;
    const double f = ((ssered - ssefull) / (double)k) / (ssefull / n);     // _association.py:45
    pk[a] = f_sf(f, (double)k, n - (1.0 + r + k));
    r2k[a] = 1.0 - ssefull / ssered;
  }
  __syncthreads();
  if (lane == 0) {                                              // np.nanargmin (_association.py:60)
    int best = -1;
    for (int a = 0; a < K; ++a)
      if (pk[a] == pk[a] && (best < 0 || pk[a] < pk[best])) best = a;
    kidx[p] = best;
    minp[p] = best < 0 ? __builtin_nan("") : pk[best];
    r2out[p] = best < 0 ? __builtin_nan("") : r2k[best];
  }
}

}  // namespace

int launch_condition(cna_ctx* c, hipStream_t st, const double* M_dev, const double* Y_dev, int N, int P,
                     double* Zc_dev, int ldy) {
  if (P == 0) return 0;
  if (N > 512) CNA_FAIL(CNA_EINVAL, "more than 512 samples are not supported");
  ProfScope ps(c, CNA_K_CONDITION, st);
  hipLaunchKernelGGL(k_condition, dim3(P), dim3(64), sizeof(double) * N, st, M_dev, Y_dev, N, P, Zc_dev, ldy);
  HIP_TRY(hipGetLastError());
  return 0;
}

int launch_global_test(cna_ctx* c, hipStream_t st, const double* Zc_dev, int ldy, int N, int P, const double* U_dev,
                       int kmax, const int32_t* ks_dev, int K, int r, double* minp_dev, double* r2_dev,
                       int32_t* kidx_dev) {
  if (P == 0) return 0;
  ProfScope ps(c, CNA_K_GLOBAL_TEST, st);
  const size_t sm = sizeof(double) * (N + kmax + 2 * K);
  hipLaunchKernelGGL(k_global_test, dim3(P), dim3(64), sm, st, Zc_dev, ldy, N, P, U_dev, kmax, ks_dev, K, r,
                     minp_dev, r2_dev, kidx_dev);
  HIP_TRY(hipGetLastError());
  return 0;
}
