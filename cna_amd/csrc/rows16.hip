// Row-local passes over a cells x samples matrix with SIXTEEN rows per wave in the layout of the f64 matrix
// instruction's result: lane l = 16 g + n, register i holds row g + 4 i, column 16 t + n (t: column tile).  Round 3.
//
//   * the projector of the residualisation, M = I - C.W in factored form (_nam.py:128-148: x <- x - (x.W^T).C^T, r = rank
//     of C <= 16: covariates, or one-hot batches + covariates under a ridge), is two skinny products on
//     v_mfma_f64_16x16x4_f64 -- P = X.W^T (16 rows x 16, K = samples) and X -= P.C^T (16 x 16 tiles, K = 16) -- where the
//     wave-per-row kernel (rows.hip:k_resid_lowrank) spends r dot products with a 64-lane reduction each and r axpys:
//     ~520 vector instructions per row at r = 7, 1.66 ms for the 1.6 GB of a 1M x 100 matrix (here: 0.73 ms).  Up to 128
//     samples only: the A operand of the first product, 16 rows x 4 columns per load, touches every cache line of the
//     tile four times, which the L1 absorbs for 1 KB rows and does not for 1.6 KB rows (10 ms against 2.4 for the
//     selection pass at 2M x 200 with five covariates); the same products in vector instructions on this layout (every
//     factor read serving four rows) need 228-256 registers and spill.  Wider matrices keep the wave-per-row kernels;
//   * every row statistic -- mean (_nam.py:122), batch means and their kurtosis (_nam.py:78-82,150), std with ddof = 1
//     (_nam.py:159), coefficient X.y/N (_association.py:77) -- is four DPP steps inside a row of 16 lanes, for four
//     rows at once (rows.hip:k_select_std16 has the argument).
//
// Same statements as the wave-per-row kernels; sums run in another order, so results agree to rounding (1e-15
// relative).  The projections are taken from the uncentred row and corrected, p_k = x.w_k - mean (1.w_k): the columns
// of C are centred, so 1.w_k is rounding noise and the two forms are the same number to a few ulps of |x|.
#include "common.h"
#include <cstdlib>

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double r16_sum(double v) {
  v = dpp_add(v, 0);
  v = dpp_add(v, 1);
  v = dpp_add(v, 2);
  return dpp_add(v, 3);
}

// operands of k_rowpass16 in one device block: B1 [NP][16] (W^T, zero padded) | Ct [16][NP] | sw[16] = row sums of W |
// inv_cnt[16] = 1 / samples of batch b | bcode[NP] (int32: batch of a column, -1 none)
__global__ __launch_bounds__(256) void k_rowpass16_prep(const double* __restrict__ W, const double* __restrict__ Ctg, int r,
                                                        int N, int NP, const int32_t* __restrict__ order,
                                                        const int32_t* __restrict__ boff, int nb, double* __restrict__ prep,
                                                        int qc_cols) {
  double* B1 = prep;
  double* Ct = prep + (size_t)NP * 16;
  double* sw = Ct + (size_t)16 * NP;
  double* inv = sw + 16;
  int* bcode = (int*)(inv + 16);
  for (int idx = threadIdx.x; idx < NP * 16; idx += 256) {
    const int col = idx >> 4, k = idx & 15;
    B1[idx] = (k < r && col < N) ? W[(size_t)k * N + col] : 0.0;
  }
  for (int idx = threadIdx.x; idx < 16 * NP; idx += 256) {
    const int k = idx / NP, col = idx - k * NP;
    Ct[idx] = (k < r && col < N) ? Ctg[(size_t)k * N + col] : 0.0;
  }
  if (threadIdx.x < 16) {
    const int k = threadIdx.x;
    double s = 0.0;
    if (k < r)
      for (int col = 0; col < N; ++col) s += W[(size_t)k * N + col];
    sw[k] = s;
    inv[k] = (order && k < nb) ? 1.0 / (double)(boff[k + 1] - boff[k]) : 0.0;
  }
  for (int col = threadIdx.x; col < NP; col += 256) bcode[col] = -1;
  __syncthreads();
  if (order)
    for (int b = 0; b < nb; ++b)
      for (int m = boff[b] + threadIdx.x; m < boff[b + 1]; m += 256) bcode[order[m]] = b;
  if (qc_cols && order) {
    // QC mode: columns r .. r + nb - 1 of the first product's B operand are the batch indicators over the batch sizes, so
    // that P = X.[W^T | E] delivers the batch means of the RAW rows with the projections, in the same matrix instructions
    __syncthreads();
    for (int b = 0; b < nb; ++b) {
      const double w = 1.0 / (double)(boff[b + 1] - boff[b]);
      for (int m = boff[b] + threadIdx.x; m < boff[b + 1]; m += 256) B1[(size_t)order[m] * 16 + r + b] = w;
    }
  }
}

__device__ __forceinline__ double r16_max(double v) {
  v = fmax(v, dpp_partner(v, 0));
  v = fmax(v, dpp_partner(v, 1));
  v = fmax(v, dpp_partner(v, 2));
  return fmax(v, dpp_partner(v, 3));
}

// batch kurtosis of row i of the register tile as it stands (_nam.py:78-82: Fisher kurtosis of the batch means, + 3)
template <int NT>
__device__ __forceinline__ double row_batch_kurtosis(const double (&xd)[NT][4], int i, const int (&code)[NT], int nb,
                                                     const double* __restrict__ pinv) {
  double bm[16];
  double bsum = 0.0;
#pragma unroll
  for (int b = 0; b < 16; ++b) {
    bm[b] = 0.0;
    if (b < nb) {
      double part = 0.0;
#pragma unroll
      for (int t = 0; t < NT; ++t) part += code[t] == b ? xd[t][i] : 0.0;
      bm[b] = r16_sum(part) * pinv[b];
      bsum += bm[b];
    }
  }
  const double nbd = (double)nb;
  const double bmean = bsum / nbd;
  double d2s = 0.0, d4s = 0.0;
#pragma unroll
  for (int b = 0; b < 16; ++b) {
    if (b < nb) {
      const double d = bm[b] - bmean;
      const double d2 = d * d;
      d2s += d2;
      d4s += d2 * d2;
    }
  }
  const double m2 = d2s / nbd, m4 = d4s / nbd;
  const double em = 2.220446049250313e-16 * bmean;
  const double kk = (m2 <= em * em) ? __builtin_nan("") : m4 / (m2 * m2);
  return (kk - 3.0) + 3.0;
}

// QC: the pass starts from the NAM itself (src) and also answers what the two passes in front of it would have been asked
// (round 6: selection + QC + first ridge in ONE pass): counters[0] += rows whose batch kurtosis BEFORE anything
// is done to them is NaN (the only way a row can fail _qc_nam's `kurtosis < max(6, 2 median)` with at most seven batches,
// _nam.py:94-96: the kurtosis of so few batch means cannot reach 6), counters[1] += rows that are constant over the samples
// the way pandas' std sees it (_association.py:182; rows.hip:k_select_zv's test).
template <int NT, bool QC = false>
__global__ __launch_bounds__(256) void k_rowpass16(const double* __restrict__ src, int lds_, double* __restrict__ dst, int ldd,
                                                   int64_t nrows, int N, const double* __restrict__ prep, int r, int center,
                                                   int standardize, int write_out, const double* __restrict__ y,
                                                   double* __restrict__ nc, unsigned long long* __restrict__ maxword, int nb,
                                                   double* __restrict__ bk_out, unsigned long long* __restrict__ counters = nullptr) {
  constexpr int NP = 16 * NT;
  extern __shared__ double sm[];
  double* B1 = sm;                               // [NP][16]
  double* Ct = sm + NP * 16;                     // [16][NP]
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double* stage = sm + NP * 32 + wv * (16 * 17); // the 16 x 16 projections of this wave's tile, row stride 17
  const double* psw = prep + (size_t)NP * 32;
  const double* pinv = psw + 16;
  const int* pcode = (const int*)(pinv + 16);
  if (r > 0) {
    for (int i = threadIdx.x; i < NP * 32; i += 256) sm[i] = prep[i];
    __syncthreads();
  }
  const int n = lane & 15, g = lane >> 4;
  const double nn = (double)N;
  const double sw_n = r > 0 ? psw[n] : 0.0;
  const int KS = (N + 3) >> 2;
  const int ns = (r + 3) >> 2;
  double yv[NT];
  int code[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = 16 * t + n;
    yv[t] = (y && col < N) ? y[col] : 0.0;
    code[t] = (bk_out && col < N) ? pcode[col] : -1;
  }
  const int64_t ntiles = (nrows + 15) >> 4;
  double vmax = 0.0;
  bool any_nan = false;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wv; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t row0 = tile << 4;
    double xd[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t row = row0 + g + 4 * i;
        const int col = 16 * t + n;
        xd[t][i] = (row < nrows && col < N) ? src[row * lds_ + col] : 0.0;
      }
    bool flat4[4] = {false, false, false, false};
    if (QC) {                                                // constant rows, tested on the values as they were loaded
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        double hi = -1.7976931348623157e308, lo = 1.7976931348623157e308;
#pragma unroll
        for (int t = 0; t < NT; ++t)
          if (16 * t + n < N) { hi = fmax(hi, xd[t][i]); lo = fmin(lo, xd[t][i]); }
        hi = r16_max(hi);
        lo = -r16_max(-lo);
        bool flat = hi == lo;
        if (flat) {                                          // pandas nanvar: zero iff sum / N == x (the sum taken in order)
          double s = 0.0;
          for (int cidx = 0; cidx < N; ++cidx) s += hi;
          flat = s / nn - hi == 0.0;
        }
        flat4[i] = flat;
      }
    }
    double mean[4] = {0.0, 0.0, 0.0, 0.0};
    if (center) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        double s = 0.0;
#pragma unroll
        for (int t = 0; t < NT; ++t) s += xd[t][i];
        mean[i] = r16_sum(s) / nn;
#pragma unroll
        for (int t = 0; t < NT; ++t)
          if (16 * t + n < N) xd[t][i] -= mean[i];
      }
    }
    if (r > 0) {
      // P = X.W^T: A operand (row m = n of the tile, k = g) straight from memory, B operand (k = g, column n) from LDS
      v4d P = (v4d){0.0, 0.0, 0.0, 0.0};
      const int64_t arow = row0 + n;
      const double* __restrict__ ap = src + (arow < nrows ? arow : 0) * lds_;
      for (int s = 0; s < KS; ++s) {
        const int col = 4 * s + g;
        const double a = (arow < nrows && col < N) ? ap[col] : 0.0;
        P = __builtin_amdgcn_mfma_f64_16x16x4f64(a, B1[col * 16 + n], P, 0, 0, 0);
      }
      if (QC) {
        // lanes r <= n < r + nb of a row's sixteen hold its RAW batch means (P = X.[W^T | E], see k_rowpass16_prep): the
        // row fails the QC iff their kurtosis is NaN -- their variance vanishes against their mean (row_batch_kurtosis'
        // test) -- and it has zero variance iff it is constant (largest == smallest entry, and pandas' own test)
        unsigned bad_qc = 0, flat_rows = 0;
        const bool mine = n >= r && n < r + nb;
        const double nbd = (double)nb;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int64_t row = row0 + g + 4 * i;
          const double bmean = r16_sum(mine ? P[i] : 0.0) / nbd;
          const double d = mine ? P[i] - bmean : 0.0;
          const double m2 = r16_sum(d * d) / nbd;
          const double em = 2.220446049250313e-16 * bmean;
          const bool nan_k = m2 <= em * em;
          if (n == 0 && row < nrows) {
            bad_qc += nan_k ? 1u : 0u;
            flat_rows += flat4[i] ? 1u : 0u;
          }
        }
        if (bad_qc) atomicAdd(counters, (unsigned long long)bad_qc);
        if (flat_rows) atomicAdd(counters + 1, (unsigned long long)flat_rows);
      }
      // lane (g, n) holds p_n of rows g + 4 i: centre, negate, and turn the 16 x 16 block into the A layout through LDS
#pragma unroll
      for (int i = 0; i < 4; ++i) stage[(g + 4 * i) * 17 + n] = -(P[i] - mean[i] * sw_n);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      double pa[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) pa[s] = stage[n * 17 + g + 4 * s];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      // X -= P.C^T, tile by tile of 16 columns: the accumulator starts as the centred rows
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        v4d acc = (v4d){xd[t][0], xd[t][1], xd[t][2], xd[t][3]};
#pragma unroll
        for (int s = 0; s < 4; ++s)
          if (s < ns) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[s], Ct[(4 * s + g) * NP + 16 * t + n], acc, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) xd[t][i] = acc[i];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t row = row0 + g + 4 * i;
      if (bk_out) {
        const double kk3 = row_batch_kurtosis<NT>(xd, i, code, nb, pinv);
        if (n == 0 && row < nrows) bk_out[row] = kk3;
      }
      if (!write_out && !y) continue;
      double sd = 1.0;
      if (standardize) {                          // pandas std: avg = sum/N ; sqrt(sum((avg-x)^2)/(N-1))
        double s2 = 0.0;
#pragma unroll
        for (int t = 0; t < NT; ++t) s2 += xd[t][i];
        const double avg = r16_sum(s2) / nn;
        double ss = 0.0;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (16 * t + n < N) {
            const double dd = avg - xd[t][i];
            ss += dd * dd;
          }
        }
        sd = sqrt(r16_sum(ss) / (nn - 1.0));
      }
      double dot = 0.0;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int col = 16 * t + n;
        const double xs = col < N ? (standardize ? __ddiv_rn(xd[t][i], sd) : xd[t][i]) : 0.0;
        if (write_out && row < nrows && col < ldd) dst[row * ldd + col] = xs;
        dot += yv[t] * xs;
      }
      if (y) {
        const double v = r16_sum(dot) / nn;
        if (row < nrows) {
          if (n == 0) nc[row] = v;
          const double av = fabs(v);
          if (av > vmax) vmax = av;
          any_nan = any_nan || (v != v);
        }
      }
    }
  }
  if (y) {
    const bool wn = __any(any_nan);
    const double wm = wave_max_d(vmax);
    if (lane == 0) {
      const unsigned long long bits = wn ? 0x7ff8000000000000ull : (unsigned long long)__double_as_longlong(wm);
      if (bits > __builtin_nontemporal_load(maxword)) atomicMax(maxword, bits);
    }
  }
}

}  // namespace

// 1: the pass was queued here; 0: not eligible (the caller takes the wave-per-row kernels); < 0: error.
// src / dst: nrows x N with leading dimensions lds / ldd (dst may be src).  W_dev, Ct_dev: r x N (r = 0: no projector);
// bk_order / bk_boff: samples grouped by batch (nb batches; bk_out = null: no batch kurtosis); y_dev: coefficients
// into c->ncorrs and the bits of max |coefficient| into maxword (zeroed here).
int launch_rowpass16(cna_ctx* c, const double* src, int lds, double* dst, int ldd, int64_t nrows, int N, const double* W_dev,
                     const double* Ct_dev, int r, int center, int standardize, int write_out, const double* y_dev,
                     unsigned long long* maxword, const int32_t* bk_order, const int32_t* bk_boff, int nb, double* bk_out,
                     unsigned long long* qc_counters) {
  const char* sw = getenv("CNA_ROWPASS16");
  if (sw && atoi(sw) == 0) return 0;
  if (N < 2 || N > 256 || r < 0 || r > 16 || (r > 0 && N > 128) || (bk_out && (nb < 1 || nb > 16))) return 0;
  // (the QC shortcut holds for at most seven batches -- see k_rowpass16 -- and its batch means ride in the spare columns of
  // the first product's sixteen)
  if (qc_counters && (!bk_out || nb > 7 || r < 1 || r + nb > 16)) return 0;
  const int NT0 = (N + 15) / 16;
  const int NT = NT0 <= 8 ? NT0 : (NT0 <= 10 ? 10 : (NT0 <= 13 ? 13 : 16));      // the instantiation that takes it
  const int NP = 16 * NT;
  const int64_t prep_bytes = 8 * ((int64_t)NP * 32 + 32) + 4 * (int64_t)NP;
  CNA_TRY(dev_reserve(c, &c->rp16_buf, &c->rp16_cap, prep_bytes));
  if (y_dev) HIP_TRY(hipMemsetAsync(maxword, 0, sizeof(unsigned long long), c->stream));
  if (nrows == 0) return 1;
  hipLaunchKernelGGL(k_rowpass16_prep, dim3(1), dim3(256), 0, c->stream, W_dev, Ct_dev, r, N, NP, bk_out ? bk_order : nullptr,
                     bk_boff, bk_out ? nb : 0, (double*)c->rp16_buf, qc_counters ? 1 : 0);
  const size_t smem = sizeof(double) * ((size_t)NP * 32 + 4 * 16 * 17);
  const int64_t want = (nrows + 63) / 64;
  const unsigned grid = (unsigned)(want < 2048 ? want : 2048);
#define RP(T) { static bool once = false, once_qc = false; \
    if (qc_counters) { \
      if (smem > 48 * 1024 && !once_qc) { HIP_TRY(hipFuncSetAttribute((const void*)k_rowpass16<T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)); once_qc = true; } \
      hipLaunchKernelGGL((k_rowpass16<T, true>), dim3(grid), dim3(256), smem, c->stream, src, lds, dst, ldd, nrows, N, (const double*)c->rp16_buf, r, center, standardize, write_out, y_dev, c->ncorrs, maxword, nb, bk_out, qc_counters); \
    } else { \
      if (smem > 48 * 1024 && !once) { HIP_TRY(hipFuncSetAttribute((const void*)k_rowpass16<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)); once = true; } \
      hipLaunchKernelGGL((k_rowpass16<T, false>), dim3(grid), dim3(256), smem, c->stream, src, lds, dst, ldd, nrows, N, (const double*)c->rp16_buf, r, center, standardize, write_out, y_dev, c->ncorrs, maxword, nb, bk_out, (unsigned long long*)nullptr); } }
  switch (NT) {
    case 1: RP(1) break;
    case 2: RP(2) break;
    case 3: RP(3) break;
    case 4: RP(4) break;
    case 5: RP(5) break;
    case 6: RP(6) break;
    case 7: RP(7) break;
    case 8: RP(8) break;
    case 10: RP(10) break;
    case 13: RP(13) break;
    default: RP(16) break;
  }
#undef RP
#undef RP
  HIP_TRY(hipGetLastError());
  return 1;
}
