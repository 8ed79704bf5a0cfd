// Row-local (per-cell) kernels: QC statistics, selection, standardisation, neighbourhood
// coefficients, observed threshold counts, per-cell FDR lookup, transpose for D2H.
// All matrices are cell-major (one row per cell), so every statistic "over samples" is a
// reduction inside one row: one wave per row, lanes across the sample axis.
#include "common.h"
#include <cstddef>

namespace {

__device__ __forceinline__ int64_t uniform64(int64_t v) {
  int lo = __builtin_amdgcn_readfirstlane((int)(v & 0xffffffffll));
  int hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
  return ((int64_t)hi << 32) | (uint32_t)lo;
}

constexpr int MAXQ = 16;  // up to 1024 columns per row

// ---- batch kurtosis of a row held in registers (lane owns columns lane + 64 q), up to BK_FAST batches:
// per batch one masked partial per lane and one wave reduction (six DPP steps) -- every lane takes part, where the
// sum "in sample order" of k_batch_kurtosis keeps nb lanes busy with a chain of dependent LDS reads (0.84 ms for the
// 0.16 ms of traffic of a 1M x 100 matrix).  The batch sums add in another order than there (lane partials, then
// the reduction tree): equal to rounding (1e-16), and likewise deterministic.
constexpr int BK_FAST = 16;
template <int NQ>
__device__ __forceinline__ double batch_kurt_regs(const double (&x)[NQ], const int (&code)[NQ], int nb,
                                                  const double* __restrict__ inv_cnt) {
  double bm[BK_FAST];
  double sum = 0.0;
#pragma unroll
  for (int b = 0; b < BK_FAST; ++b) {
    bm[b] = 0.0;
    if (b < nb) {
      double part = 0.0;
#pragma unroll
      for (int q = 0; q < NQ; ++q) part += code[q] == b ? x[q] : 0.0;
      bm[b] = wave_sum(part) * inv_cnt[b];
      sum += bm[b];
    }
  }
  const double n = (double)nb;
  const double mean = sum / n;
  double d2s = 0.0, d4s = 0.0;
#pragma unroll
  for (int b = 0; b < BK_FAST; ++b) {
    if (b < nb) {
      const double d = bm[b] - mean;
      const double d2 = d * d;
      d2s += d2;
      d4s += d2 * d2;
    }
  }
  const double m2 = d2s / n, m4 = d4s / n;
  const double em = 2.220446049250313e-16 * mean;
  const double k = (m2 <= em * em) ? __builtin_nan("") : m4 / (m2 * m2);
  return (k - 3.0) + 3.0;                                  // Fisher, then "+ 3" as the reference writes it
}
// codes of the columns a lane owns and 1 / (samples of batch b), from the grouped lists k_batch_kurtosis takes
template <int NQ>
__device__ __forceinline__ void batch_codes_of(const int32_t* __restrict__ order, const int32_t* __restrict__ boff, int nb,
                                               int ncols, int lane, int (&code)[NQ], double* inv_cnt_lds) {
#pragma unroll
  for (int q = 0; q < NQ; ++q) code[q] = -1;
  for (int b = 0; b < nb; ++b) {
    const int s0 = boff[b], s1 = boff[b + 1];
    for (int m = s0; m < s1; ++m) {
      const int col = order[m];
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (col == lane + 64 * q) code[q] = b;
    }
    if (threadIdx.x == 0) inv_cnt_lds[b] = 1.0 / (double)(s1 - s0);
  }
  (void)ncols;
}
template <int NQ>
__global__ __launch_bounds__(256) void k_batch_kurtosis_fast(const double* __restrict__ mat, int64_t rows, int ncols,
                                                             int ld, const int32_t* __restrict__ order,
                                                             const int32_t* __restrict__ boff, int nb,
                                                             double* __restrict__ out) {
  __shared__ double inv_cnt[BK_FAST];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int code[NQ];
  batch_codes_of<NQ>(order, boff, nb, ncols, lane, code, inv_cnt);
  __syncthreads();
  // four rows in flight per wave: one row at a time leaves the pass bound by the latency of its 800-byte loads
  constexpr int RPW = NQ <= 4 ? 4 : 2;
  const int64_t stride = (int64_t)gridDim.x * 4 * RPW;
  for (int64_t base = ((int64_t)blockIdx.x * 4 + wv) * RPW; base < rows; base += stride) {
    double x[RPW][NQ];
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        x[r][q] = (base + r < rows && lane + 64 * q < ncols) ? mat[(base + r) * ld + lane + 64 * q] : 0.0;
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      if (base + r >= rows) break;
      const double k = batch_kurt_regs<NQ>(x[r], code, nb, inv_cnt);
      if (lane == 0) out[base + r] = k;
    }
  }
}

// ---- _batch_kurtosis (_nam.py:78-82) -------------------------------------------------
// order[] lists the sample columns grouped by batch (stable), boff[b]..boff[b+1] is batch b.
// Lane b sums its batch's entries in sample order (the order numpy's mean walks them).
__global__ __launch_bounds__(256) void k_batch_kurtosis(const double* __restrict__ mat, int64_t rows,
                                                        int ncols, int ld, const int32_t* __restrict__ order,
                                                        const int32_t* __restrict__ boff, int nb,
                                                        double* __restrict__ out) {
  extern __shared__ double sm[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double* xr = sm + (size_t)wv * ld;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t base = (int64_t)blockIdx.x * 4; base < rows; base += nwaves) {
    const int64_t row = base + wv;
    const bool live = row < rows;
    if (live)
      for (int col = lane; col < ncols; col += 64) xr[col] = mat[row * ld + col];
    __syncthreads();
    double bm[4];
    double sum = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int b = lane + 64 * q;
      bm[q] = 0.0;
      if (live && b < nb) {
        const int s0 = boff[b], s1 = boff[b + 1];
        double s = 0.0;
        for (int m = s0; m < s1; ++m) s += xr[order[m]];
        bm[q] = s / (double)(s1 - s0);
        sum += bm[q];
      }
    }
    const double n = (double)nb;
    const double mean = wave_sum(sum) / n;
    double d2s = 0.0, d4s = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (lane + 64 * q < nb) {
        const double d = bm[q] - mean;
        const double d2 = d * d;
        d2s += d2;
        d4s += d2 * d2;
      }
    }
    const double m2 = wave_sum(d2s) / n, m4 = wave_sum(d4s) / n;
    const double em = 2.220446049250313e-16 * mean;
    const double k = (m2 <= em * em) ? __builtin_nan("") : m4 / (m2 * m2);
    if (live && lane == 0) out[row] = (k - 3.0) + 3.0;   // Fisher, then "+ 3" as the reference writes it
    __syncthreads();
  }
}

// ---- NAM.std(axis=0) == 0 over the selected samples (_association.py:182) -------------
__global__ __launch_bounds__(256) void k_zero_variance(const double* __restrict__ nam, int64_t rows, int ld,
                                                       const int32_t* __restrict__ colmap, int nsel,
                                                       uint8_t* __restrict__ flags, unsigned long long* count) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wv; row < rows; row += nwaves) {
    const double v0 = nam[row * ld + (colmap ? colmap[0] : 0)];
    bool eq = true;
    for (int c = lane; c < nsel; c += 64) {
      const double v = nam[row * ld + (colmap ? colmap[c] : c)];
      eq = eq && (v == v0);
    }
    const bool all_eq = __all(eq);
    if (lane == 0) {
      uint8_t f = 0;
      if (all_eq) {
        // pandas nanvar: avg = sum/N ; var = sum((avg - x)^2)/(N-1) -- zero iff avg == x
        double s = 0.0;
        for (int c = 0; c < nsel; ++c) s += v0;
        const double avg = s / (double)nsel;
        f = (avg - v0 == 0.0) ? 1 : 0;
      }
      flags[row] = f;
      if (f) atomicAdd(count, 1ull);
    }
  }
}

// ---- X[i', c'] = NAM[keep[i'], colmap[c']] ---------------------------------------------
__global__ void k_select(const double* __restrict__ nam, int ld, const int64_t* __restrict__ keep,
                         const int32_t* __restrict__ colmap, double* __restrict__ X, int64_t nx, int Nx,
                         int ldx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nx * ldx) return;
  const int64_t r = i / ldx;
  const int c = (int)(i - r * ldx);
  double v = 0.0;
  if (c < Nx) {
    const int64_t sr = keep ? keep[r] : r;
    const int sc = colmap ? colmap[c] : c;
    v = nam[sr * ld + sc];
  }
  X[i] = v;
}

// ---- select + the zero-variance test of the selected rows in one pass (M != I path): X row = NAM[keep[i]]
// over colmap; count of rows that are constant over the selected samples (the test of k_zero_variance)
__global__ __launch_bounds__(256) void k_select_zv(const double* __restrict__ nam, int ld, const int64_t* __restrict__ keep,
                                                   const int32_t* __restrict__ colmap, double* __restrict__ X, int64_t nx,
                                                   int Nx, int ldx, unsigned long long* count) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t r = (int64_t)blockIdx.x * 4 + wv; r < nx; r += nwaves) {
    const double* __restrict__ src = nam + (keep ? keep[r] : r) * ld;
    const double v0 = src[colmap ? colmap[0] : 0];
    bool eq = true;
    for (int c = lane; c < ldx; c += 64) {
      double v = 0.0;
      if (c < Nx) {
        v = src[colmap ? colmap[c] : c];
        eq = eq && (v == v0);
      }
      X[r * ldx + c] = v;
    }
    if (__all(eq) && lane == 0) {
      double s = 0.0;
      for (int c = 0; c < Nx; ++c) s += v0;               // pandas nanvar: zero iff sum/N == x
      if (s / (double)Nx - v0 == 0.0) atomicAdd(count, 1ull);
    }
  }
}

// ---- X <- (X [- mean]) / std(ddof=1) per row (_nam.py:103-104,159) -----------------------
// A wave walks RPW rows at a time: all their loads are issued before the first reduction so
// the dependent shuffle chains of one row hide under the memory latency of the others.
template <int NQ>
__global__ __launch_bounds__(256) void k_standardize(double* __restrict__ X, int64_t nx, int Nx, int ldx,
                                                     int center) {
  constexpr int RPW = 4;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t stride = (int64_t)gridDim.x * 4 * RPW;
  const double n = (double)Nx;
  for (int64_t base = ((int64_t)blockIdx.x * 4 + wv) * RPW; base < nx; base += stride) {
    double x[RPW][NQ];
    double sum[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      sum[r] = 0.0;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int col = lane + 64 * q;
        x[r][q] = (col < Nx && base + r < nx) ? X[(base + r) * ldx + col] : 0.0;
        sum[r] += x[r][q];
      }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      if (center) {
        const double mean = wave_sum(sum[r]) / n;
        sum[r] = 0.0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          if (lane + 64 * q < Nx) x[r][q] -= mean;
          sum[r] += x[r][q];
        }
      }
      // pandas std: avg = sum/N ; sqrt(sum((avg-x)^2)/(N-1))
      const double avg = wave_sum(sum[r]) / n;
      double ss = 0.0;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        if (lane + 64 * q < Nx) {
          const double d = avg - x[r][q];
          ss += d * d;
        }
      }
      const double sd = sqrt(wave_sum(ss) / (n - 1.0));
      if (base + r < nx) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int col = lane + 64 * q;
          if (col < Nx) X[(base + r) * ldx + col] = __ddiv_rn(x[r][q], sd);
        }
      }
    }
  }
}

// ---- fused fast path when no covariate / batch has to be regressed out (M = I):
//   X[i', c'] = standardise_row( NAM[keep[i'], colmap[c']] )   with centring,
// i.e. NAM.reindex(y.index)[filter] -> drop -> centre -> /std in one read of the NAM and one
// write of X (_association.py:178-185, _nam.py:122,159).  Rows whose selected entries are all
// equal (NAM.std(axis=0) == 0, _association.py:182) are counted; if there are any the caller
// redoes the selection without them (rare).
template <int NQ>
__global__ __launch_bounds__(256) void k_select_std(const double* __restrict__ nam, int ld,
                                                    const int64_t* __restrict__ keep,
                                                    const int32_t* __restrict__ colmap, double* __restrict__ X,
                                                    int64_t nx, int Nx, int ldx, unsigned long long* nzero,
                                                    const double* __restrict__ y, double* __restrict__ nc,
                                                    unsigned long long* __restrict__ blockmax,
                                                    const double* __restrict__ Wg, const double* __restrict__ Ctg, int rk,
                                                    unsigned char* __restrict__ xq, double2* __restrict__ xscale, int Kp) {
  // xq != null: the rows also leave as the 24-bit fixed-point digit planes of the integer local null
  // (null_i8.hip: [row][digit][Kp] bytes, one scale per row, {max |x|, sum |q|} in xscale), so that pass
  // does not have to read X again.
  // y != null: the rows leave this kernel final, so the neighbourhood coefficients
  // ncorrs = X.y/N (_association.py:77) and their max |.| are taken on the way out, as k_ncorrs would.
  // rk > 0: a projector M = I - C.W in factored form (W: rk x Nx, C^T: rk x Nx, see k_resid_lowrank) is
  // applied to the centred row before the std: selection + residualisation + standardisation in one pass
  constexpr int RPW = 4;
  extern __shared__ double lw[];
  __shared__ unsigned long long wmax[4];
  __shared__ unsigned qstage[4][192];
  for (int i = threadIdx.x; i < 2 * rk * Nx; i += 256) lw[i] = i < rk * Nx ? Wg[i] : Ctg[i - rk * Nx];
  if (rk > 0) __syncthreads();
  const double* Wl = lw;
  const double* Ctl = lw + (size_t)rk * Nx;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t stride = (int64_t)gridDim.x * 4 * RPW;
  const double n = (double)Nx;
  int sc[NQ];
  double yv[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int col = lane + 64 * q;
    sc[q] = col < Nx ? (colmap ? colmap[col] : col) : 0;
    yv[q] = (y && col < Nx) ? y[col] : 0.0;
  }
  double vmax = 0.0;
  bool any_nan = false;
  for (int64_t base = ((int64_t)blockIdx.x * 4 + wv) * RPW; base < nx; base += stride) {
    double x[RPW][NQ];
    double sum[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      sum[r] = 0.0;
      const bool live = base + r < nx;
      const int64_t sr = live ? (keep ? keep[base + r] : base + r) : 0;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        x[r][q] = (live && lane + 64 * q < Nx) ? nam[sr * ld + sc[q]] : 0.0;
        sum[r] += x[r][q];
      }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      if (base + r >= nx) break;                         // wave-uniform
      // zero variance the way pandas sees it: avg = sum/N, every (avg - x) == 0
      const double avg0 = wave_sum(sum[r]) / n;
      bool flat = true;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (lane + 64 * q < Nx) flat = flat && (avg0 - x[r][q] == 0.0);
      if (__all(flat) && lane == 0) atomicAdd(nzero, 1ull);
      // centre (_nam.py:122), then std with ddof=1 of the centred values (_nam.py:159)
      double s2 = 0.0;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (lane + 64 * q < Nx) x[r][q] -= avg0;
      if (rk > 0) {                                          // x <- x - (x.W^T).C^T
        double corr[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) corr[q] = 0.0;
        for (int k = 0; k < rk; ++k) {
          double d = 0.0;
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            if (lane + 64 * q < Nx) d += x[r][q] * Wl[k * Nx + lane + 64 * q];
          const double pk = wave_sum(d);
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            if (lane + 64 * q < Nx) corr[q] += pk * Ctl[k * Nx + lane + 64 * q];
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) x[r][q] -= corr[q];
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q) s2 += x[r][q];
      const double avg = wave_sum(s2) / n;
      double ss = 0.0;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        if (lane + 64 * q < Nx) {
          const double d = avg - x[r][q];
          ss += d * d;
        }
      }
      const double sd = sqrt(wave_sum(ss) / (n - 1.0));
      double dot = 0.0, amax = 0.0;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int col = lane + 64 * q;
        const double xs = col < Nx ? __ddiv_rn(x[r][q], sd) : 0.0;
        if (col < ldx) X[(base + r) * ldx + col] = xs;
        dot += yv[q] * xs;
        x[r][q] = xs;
        amax = fmax(amax, fabs(xs));                      // NaN rows (zero variance): fmax drops them, q = 0 below
      }
      if (xq) {
        const double rmax = wave_max_d(amax);
        const double inv = rmax > 0.0 ? I8_QMAX / rmax : 0.0;
        // bytes of a row meet in LDS (one plane = 256 bytes per wave) and leave as dwords: three 4 Kp / 4-lane stores
        // per row instead of twelve byte stores (partial-line writes: +0.6 ms at 2M x 200)
        unsigned char* qb = (unsigned char*)qstage[wv];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int col = lane + 64 * q;
          if (col < Kp) {
            const double v = x[r][q] * inv;
            const int qi = v == v ? (int)rint(v) : 0;
            // balanced base-256 digits: the low byte of qi, of (qi + 128) >> 8 and of ((qi + 128) >> 8) + 128 >> 8
            const int q1 = (qi + 128) >> 8;
            qb[col] = (unsigned char)qi;
            qb[256 + col] = (unsigned char)q1;
            qb[512 + col] = (unsigned char)((q1 + 128) >> 8);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (4 * lane < Kp) {
          unsigned* rq = (unsigned*)(xq + (size_t)(base + r) * 3 * Kp);
          rq[lane] = qstage[wv][lane];
          rq[Kp / 4 + lane] = qstage[wv][64 + lane];
          rq[2 * (Kp / 4) + lane] = qstage[wv][128 + lane];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // sum |q| from above instead of a reduction: the row has sum x^2 = N - 1, so sum |x| < N and
        // sum |q| <= N Q / max|x| + N / 2
        const double l1 = rmax > 0.0 ? n * (I8_QMAX / rmax) + n : 0.0;
        if (lane == 0) xscale[base + r] = make_double2(rmax, l1);
      }
      if (y) {
        const double v = wave_sum(dot) / n;
        if (lane == 0) nc[base + r] = v;
        const double av = fabs(v);
        if (av > vmax) vmax = av;
        any_nan = any_nan || (v != v);
      }
    }
  }
  if (y) {                                   // one slot per workgroup, folded by k_max_fold (see k_ncorrs)
    if (lane == 0)
      wmax[wv] = any_nan ? 0x7ff8000000000000ull : (unsigned long long)__double_as_longlong(vmax);
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long m = wmax[0];
      for (int i = 1; i < 4; ++i) m = wmax[i] > m ? wmax[i] : m;
      blockmax[blockIdx.x] = m;
    }
  }
}

// The same pass for up to 256 samples with SIXTEEN lanes per cell, four cells per wave: lane (r = lane >> 4,
// l = lane & 15) owns the columns 64 g + 4 l + j (g < G, j < 4) of row base + r.  The row statistics (mean,
// mean of the centred values, sum of squares, coefficient, max |x|) are then four DPP steps inside a row of 16
// lanes -- no cross-row combination through scalar reads, which made the wave-per-row kernel VALU-bound
// (400 instructions per cell, 8.0e8 per launch at 2M x 200: 1.5 ms of issue for a 1.2 ms stream) -- and four
// consecutive columns per lane are one 32-byte load, one 32-byte store and, for the digit planes, one dword
// per plane without a detour through LDS.
__device__ __forceinline__ double row16_sum(double v) {
  v = dpp_add(v, 0);
  v = dpp_add(v, 1);
  v = dpp_add(v, 2);
  return dpp_add(v, 3);
}
__device__ __forceinline__ double row16_max(double v) {
  v = fmax(v, dpp_partner(v, 0));
  v = fmax(v, dpp_partner(v, 1));
  v = fmax(v, dpp_partner(v, 2));
  return fmax(v, dpp_partner(v, 3));
}

template <int G>
__global__ __launch_bounds__(256) void k_select_std16(const double* __restrict__ nam, int ld,
                                                      const int64_t* __restrict__ keep,
                                                      const int32_t* __restrict__ colmap, double* __restrict__ X,
                                                      int64_t nx, int Nx, int ldx, unsigned long long* nzero,
                                                      const double* __restrict__ y, double* __restrict__ nc,
                                                      unsigned long long* __restrict__ blockmax,
                                                      const double* __restrict__ Wg, const double* __restrict__ Ctg, int rk,
                                                      unsigned char* __restrict__ xq, double2* __restrict__ xscale, int Kp) {
  extern __shared__ double lw[];
  __shared__ unsigned long long wmax[4];
  for (int i = threadIdx.x; i < 2 * rk * Nx; i += 256) lw[i] = i < rk * Nx ? Wg[i] : Ctg[i - rk * Nx];
  if (rk > 0) __syncthreads();
  const double* Wl = lw;
  const double* Ctl = lw + (size_t)rk * Nx;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r4 = lane >> 4, l16 = lane & 15;
  const int64_t stride = (int64_t)gridDim.x * 16;
  const double n = (double)Nx;
  constexpr int NE = 4 * G;
  int sc[NE];
  double yv[NE];
  bool contig[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    contig[g] = colmap == nullptr && 64 * g + 4 * l16 + 3 < Nx && (ld & 1) == 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = 64 * g + 4 * l16 + j;
      sc[4 * g + j] = col < Nx ? (colmap ? colmap[col] : col) : 0;
      yv[4 * g + j] = (y && col < Nx) ? y[col] : 0.0;
    }
  }
  double vmax = 0.0;
  bool any_nan = false;
  for (int64_t base = ((int64_t)blockIdx.x * 4 + wv) * 4; base < nx; base += stride) {
    const int64_t row = base + r4;
    const bool live = row < nx;
    const int64_t sr = live ? (keep ? keep[row] : row) : 0;
    const double* __restrict__ src = nam + sr * ld;
    double x[NE];
    double sum = 0.0;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (contig[g]) {
        const double2 a = live ? *(const double2*)(src + 64 * g + 4 * l16) : make_double2(0.0, 0.0);
        const double2 b = live ? *(const double2*)(src + 64 * g + 4 * l16 + 2) : make_double2(0.0, 0.0);
        x[4 * g] = a.x; x[4 * g + 1] = a.y; x[4 * g + 2] = b.x; x[4 * g + 3] = b.y;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          x[4 * g + j] = (live && 64 * g + 4 * l16 + j < Nx) ? src[sc[4 * g + j]] : 0.0;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) sum += x[4 * g + j];
    }
    // zero variance the way pandas sees it: avg = sum/N, every (avg - x) == 0
    const double avg0 = row16_sum(sum) / n;
    bool flat = true;
#pragma unroll
    for (int e = 0; e < NE; ++e)
      if (64 * (e >> 2) + 4 * l16 + (e & 3) < Nx) flat = flat && (avg0 - x[e] == 0.0);
    {
      const unsigned long long bal = __ballot(flat);
      const unsigned long long rowmask = 0xffffull << (16 * r4);
      if (live && l16 == 0 && (bal & rowmask) == rowmask) atomicAdd(nzero, 1ull);
    }
    // centre (_nam.py:122), then std with ddof=1 of the centred values (_nam.py:159)
#pragma unroll
    for (int e = 0; e < NE; ++e)
      if (64 * (e >> 2) + 4 * l16 + (e & 3) < Nx) x[e] -= avg0;
    if (rk > 0) {                                            // x <- x - (x.W^T).C^T
      double corr[NE];
#pragma unroll
      for (int e = 0; e < NE; ++e) corr[e] = 0.0;
      for (int k = 0; k < rk; ++k) {
        double d = 0.0;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          const int col = 64 * (e >> 2) + 4 * l16 + (e & 3);
          if (col < Nx) d += x[e] * Wl[k * Nx + col];
        }
        const double pk = row16_sum(d);
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          const int col = 64 * (e >> 2) + 4 * l16 + (e & 3);
          if (col < Nx) corr[e] += pk * Ctl[k * Nx + col];
        }
      }
#pragma unroll
      for (int e = 0; e < NE; ++e) x[e] -= corr[e];
    }
    double s2 = 0.0;
#pragma unroll
    for (int e = 0; e < NE; ++e) s2 += x[e];
    const double avg = row16_sum(s2) / n;
    double ss = 0.0;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      if (64 * (e >> 2) + 4 * l16 + (e & 3) < Nx) {
        const double d = avg - x[e];
        ss += d * d;
      }
    }
    const double sd = sqrt(row16_sum(ss) / (n - 1.0));
    double dot = 0.0, amax = 0.0;
    double* __restrict__ dst = X + row * ldx;
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = 4 * g + j, col = 64 * g + 4 * l16 + j;
        const double xs = col < Nx ? __ddiv_rn(x[e], sd) : 0.0;
        x[e] = xs;
        dot += yv[e] * xs;
        amax = fmax(amax, fabs(xs));                      // NaN rows (zero variance): fmax drops them, q = 0 below
      }
      if (live) {
        const int c0 = 64 * g + 4 * l16;
        if (c0 + 3 < ldx && (ldx & 1) == 0) {
          *(double2*)(dst + c0) = make_double2(x[4 * g], x[4 * g + 1]);
          *(double2*)(dst + c0 + 2) = make_double2(x[4 * g + 2], x[4 * g + 3]);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (c0 + j < ldx) dst[c0 + j] = x[4 * g + j];
        }
      }
    }
    if (xq) {
      const double rmax = row16_max(amax);
      const double inv = rmax > 0.0 ? I8_QMAX / rmax : 0.0;
      unsigned* rq = (unsigned*)(xq + (size_t)row * 3 * Kp);
#pragma unroll
      for (int g = 0; g < G; ++g) {
        unsigned w0 = 0, w1 = 0, w2 = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const double v = x[4 * g + j] * inv;
          const int qi = v == v ? (int)rint(v) : 0;
          // balanced base-256 digits: the low byte of qi, of (qi + 128) >> 8 and of ((qi + 128) >> 8) + 128 >> 8
          const int q1 = (qi + 128) >> 8;
          w0 |= ((unsigned)qi & 255u) << (8 * j);
          w1 |= ((unsigned)q1 & 255u) << (8 * j);
          w2 |= ((unsigned)((q1 + 128) >> 8) & 255u) << (8 * j);
        }
        const int c0 = 64 * g + 4 * l16;
        if (live && c0 < Kp) {
          rq[c0 >> 2] = w0;
          rq[(Kp + c0) >> 2] = w1;
          rq[(2 * Kp + c0) >> 2] = w2;
        }
      }
      // sum |q| from above instead of a reduction: the row has sum x^2 = N - 1, so sum |x| < N and
      // sum |q| <= N Q / max|x| + N / 2
      const double l1 = rmax > 0.0 ? n * (I8_QMAX / rmax) + n : 0.0;
      if (live && l16 == 0) xscale[row] = make_double2(rmax, l1);
    }
    if (y) {
      const double v = row16_sum(dot) / n;
      if (live) {
        if (l16 == 0) nc[row] = v;
        const double av = fabs(v);
        if (av > vmax) vmax = av;
        any_nan = any_nan || (v != v);
      }
    }
  }
  if (y) {                                   // one slot per workgroup, folded by k_max_fold (see k_ncorrs)
    const bool wn = __any(any_nan);
    const double wm = wave_max_d(vmax);
    if (lane == 0) wmax[wv] = wn ? 0x7ff8000000000000ull : (unsigned long long)__double_as_longlong(wm);
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long m = wmax[0];
      for (int i = 1; i < 4; ++i) m = wmax[i] > m ? wmax[i] : m;
      blockmax[blockIdx.x] = m;
    }
  }
}

// ---- X <- (X - mean).M^T for a projector of low rank defect, M = I - C.W (C: N x r standardised batches /
// covariates, W = (C^T C + ridge N L)^-1 C^T: r x N; _nam.py:128-148), row by row:
//   x.M^T = x - (x.W^T).C^T
// i.e. r dot products and r axpys per cell instead of an N x N product: 2 n N r flops twice instead of
// 2 n N^2, and -- being row-local -- fused with the centring before it and, when the caller asks, the
// division by the std (ddof = 1, _nam.py:159) and the neighbourhood coefficients X.y/N (_association.py:77)
// after it: ONE pass over X where the GEMM route takes three (M-apply, standardise, coefficients).
// W and C^T sit in LDS (2 r N doubles).  Agrees with forming M and multiplying to rounding (1e-15).
template <int NQ>
__global__ __launch_bounds__(256) void k_resid_lowrank(double* __restrict__ X, int64_t nx, int Nx, int ldx,
                                                       const double* __restrict__ Wg, const double* __restrict__ Ctg,
                                                       int r, int center, int standardize,
                                                       const double* __restrict__ y, double* __restrict__ nc,
                                                       unsigned long long* __restrict__ blockmax,
                                                       const int32_t* __restrict__ bk_order,
                                                       const int32_t* __restrict__ bk_boff, int nb,
                                                       double* __restrict__ bk_out) {
  // bk_out != null: the batch kurtosis of the residualised row (_nam.py:150, before the division by the std) leaves
  // with it -- k_batch_kurtosis's arithmetic on the row staged in LDS (lane b sums batch b in sample order) -- so
  // the ridge schedule's check needs no pass of its own
  extern __shared__ double lw[];               // W (r x Nx) | C^T (r x Nx) | [4 x ldx: one row per wave]
  __shared__ unsigned long long wmax[4];
  double* W = lw;
  double* Ct = lw + (size_t)r * Nx;
  double* xr = lw + 2 * (size_t)r * Nx + (size_t)(threadIdx.x >> 6) * ldx;
  __shared__ double inv_cnt[BK_FAST];
  int bcode[NQ];
  if (bk_out && nb <= BK_FAST) batch_codes_of<NQ>(bk_order, bk_boff, nb, Nx, threadIdx.x & 63, bcode, inv_cnt);
  for (int i = threadIdx.x; i < 2 * r * Nx; i += 256) lw[i] = i < r * Nx ? Wg[i] : Ctg[i - r * Nx];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // two rows in flight per wave (the loads of the second are issued before the first is worked on): one at a time the
  // pass is bound by the latency of its row loads (0.71 ms for 1.6 GB at 1M x 100)
  constexpr int RPW = NQ <= 4 ? 2 : 1;
  const int64_t stride = (int64_t)gridDim.x * 4 * RPW;
  const double n = (double)Nx;
  double yv[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) yv[q] = (y && lane + 64 * q < Nx) ? y[lane + 64 * q] : 0.0;
  double vmax = 0.0;
  bool any_nan = false;
  for (int64_t base = ((int64_t)blockIdx.x * 4 + wv) * RPW; base < nx; base += stride) {
   double xin[RPW][NQ];
#pragma unroll
   for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
     for (int q = 0; q < NQ; ++q)
       xin[rr][q] = (base + rr < nx && lane + 64 * q < Nx) ? X[(base + rr) * ldx + lane + 64 * q] : 0.0;
#pragma unroll
   for (int rr = 0; rr < RPW; ++rr) {
    const int64_t row = base + rr;
    if (row >= nx) break;
    double x[NQ];
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      x[q] = xin[rr][q];
      s += x[q];
    }
    if (center) {
      const double mean = wave_sum(s) / n;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (lane + 64 * q < Nx) x[q] -= mean;
    }
    // p = x.W^T (all r from the same x), then x -= p.C^T
    double corr[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) corr[q] = 0.0;
    for (int k = 0; k < r; ++k) {
      double d = 0.0;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int col = lane + 64 * q;
        if (col < Nx) d += x[q] * W[k * Nx + col];
      }
      const double p = wave_sum(d);
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int col = lane + 64 * q;
        if (col < Nx) corr[q] += p * Ct[k * Nx + col];
      }
    }
    double s2 = 0.0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      x[q] -= corr[q];
      s2 += x[q];
    }
    if (bk_out && nb <= BK_FAST) {
      const double kk = batch_kurt_regs<NQ>(x, bcode, nb, inv_cnt);
      if (lane == 0) bk_out[row] = kk;
    } else if (bk_out) {
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (lane + 64 * q < Nx) xr[lane + 64 * q] = x[q];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      double bm[4];
      double bsum = 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int b = lane + 64 * q;
        bm[q] = 0.0;
        if (b < nb) {
          const int s0 = bk_boff[b], s1 = bk_boff[b + 1];
          double sb = 0.0;
          for (int m = s0; m < s1; ++m) sb += xr[bk_order[m]];
          bm[q] = sb / (double)(s1 - s0);
          bsum += bm[q];
        }
      }
      const double nbd = (double)nb;
      const double bmean = wave_sum(bsum) / nbd;
      double d2s = 0.0, d4s = 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (lane + 64 * q < nb) {
          const double d = bm[q] - bmean;
          const double d2 = d * d;
          d2s += d2;
          d4s += d2 * d2;
        }
      }
      const double m2 = wave_sum(d2s) / nbd, m4 = wave_sum(d4s) / nbd;
      const double em = 2.220446049250313e-16 * bmean;
      const double kk = (m2 <= em * em) ? __builtin_nan("") : m4 / (m2 * m2);
      if (lane == 0) bk_out[row] = (kk - 3.0) + 3.0;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    double sd = 1.0;
    if (standardize) {                          // pandas std: avg = sum/N ; sqrt(sum((avg-x)^2)/(N-1))
      const double avg = wave_sum(s2) / n;
      double ss = 0.0;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        if (lane + 64 * q < Nx) {
          const double dd = avg - x[q];
          ss += dd * dd;
        }
      }
      sd = sqrt(wave_sum(ss) / (n - 1.0));
    }
    double dot = 0.0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int col = lane + 64 * q;
      const double xs = col < Nx ? (standardize ? __ddiv_rn(x[q], sd) : x[q]) : 0.0;
      if (col < ldx) X[row * ldx + col] = xs;
      dot += yv[q] * xs;
    }
    if (y) {
      const double v = wave_sum(dot) / n;
      if (lane == 0) nc[row] = v;
      const double av = fabs(v);
      if (av > vmax) vmax = av;
      any_nan = any_nan || (v != v);
    }
   }
  }
  if (y) {
    if (lane == 0)
      wmax[wv] = any_nan ? 0x7ff8000000000000ull : (unsigned long long)__double_as_longlong(vmax);
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long m = wmax[0];
      for (int i = 1; i < 4; ++i) m = wmax[i] > m ? wmax[i] : m;
      blockmax[blockIdx.x] = m;
    }
  }
}

// ---- ncorrs = (y[:,None]*NAMresid).mean(axis=0) (_association.py:77) -----------------------
template <int NQ>
__global__ __launch_bounds__(256) void k_ncorrs(const double* __restrict__ X, int64_t nx, int Nx, int ldx,
                                                const double* __restrict__ y, double* __restrict__ out,
                                                unsigned long long* __restrict__ blockmax) {
  constexpr int RPW = 4;
  __shared__ unsigned long long wmax[4];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t stride = (int64_t)gridDim.x * 4 * RPW;
  double yv[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) yv[q] = (lane + 64 * q < Nx) ? y[lane + 64 * q] : 0.0;
  double vmax = 0.0;
  bool any_nan = false;
  for (int64_t base = ((int64_t)blockIdx.x * 4 + wv) * RPW; base < nx; base += stride) {
    double s[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      s[r] = 0.0;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int col = lane + 64 * q;
        if (col < Nx && base + r < nx) s[r] += yv[q] * X[(base + r) * ldx + col];
      }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const double v = wave_sum(s[r]) / (double)Nx;
      if (base + r < nx) {
        if (lane == 0) out[base + r] = v;
        const double av = fabs(v);
        if (av > vmax) vmax = av;
        any_nan = any_nan || (v != v);
      }
    }
  }
  // non-negative doubles order like their bit patterns; NaN (0x7ff8...) sorts above +inf.
  // One slot per workgroup, folded by k_max_fold: 8192 same-address atomics cost ~80 us.
  if (lane == 0)
    wmax[wv] = any_nan ? 0x7ff8000000000000ull : (unsigned long long)__double_as_longlong(vmax);
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long m = wmax[0];
    for (int i = 1; i < 4; ++i) m = wmax[i] > m ? wmax[i] : m;
    blockmax[blockIdx.x] = m;
  }
}

// One pass of an exact radix select over doubles: histogram of the 8-bit digit at `shift` among the
// values whose order-preserving 64-bit key matches `prefix` above that digit; hist[256] counts NaNs.
__device__ __forceinline__ unsigned long long order_key(double x) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(x);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);       // negatives reversed, positives above them
}
__global__ __launch_bounds__(256) void k_digit_hist(const double* __restrict__ v, int64_t n, unsigned long long prefix,
                                                    int shift, unsigned long long* __restrict__ hist) {
  __shared__ unsigned int h[257];
  for (int i = threadIdx.x; i < 257; i += 256) h[i] = 0u;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double x = v[i];
    if (x != x) {
      if (shift == 56) atomicAdd(&h[256], 1u);               // NaNs are counted once, in the first pass
      continue;
    }
    const unsigned long long k = order_key(x);
    if (shift == 56 || (k >> (shift + 8)) == prefix) atomicAdd(&h[(unsigned)(k >> shift) & 255u], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 257; i += 256)
    if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
}

// ---- the walk's stop rule without the host (cna_nam_auto; _nam.py:59,64-68).  After every step the median over the
// cells of the per-cell kurtosis decides whether the walk goes on: medkurt = np.median(kurtosis), stop at the first
// step i + 1 >= 3 with prevmedkurt - medkurt < 3.  Exact radix select as cna_stat_median does it -- eight 8-bit
// digits of the order-preserving key, both middle order statistics at once -- but with the digit choice on the
// device too, so that the next steps can be queued before this one's verdict: a step kernel that finds
// `stopped_at` set returns at once.
struct AutoState {
  unsigned long long prefix[2];       // key prefix of the lower / upper middle element so far
  long long k[2];                     // their rank among the entries that share the prefix
  long long n_tot, n_nan;
  double med[16];                     // medkurt of every step taken
  int stopped_at;                     // 0: still walking; else the number of steps after which the rule was met
  int pad;
  double median;                      // of the last select (launch_device_median)
  double threshold;                   // launch_qc_count: max(6, 2 median) (_nam.py:94)
  unsigned long long n_not_below;     // launch_qc_count: entries that are not < threshold (NaN included)
};
__global__ __launch_bounds__(256) void k_digit_hist2(const double* __restrict__ v, int64_t n,
                                                     const AutoState* __restrict__ st, int shift,
                                                     unsigned long long* __restrict__ hist) {
  if (st->stopped_at) return;
  __shared__ unsigned int h[2][257];
  for (int i = threadIdx.x; i < 2 * 257; i += 256) (&h[0][0])[i] = 0u;
  __syncthreads();
  const unsigned long long p0 = st->prefix[0], p1 = st->prefix[1];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double x = v[i];
    if (x != x) {
      if (shift == 56) atomicAdd(&h[0][256], 1u);            // NaNs are counted once, in the first pass
      continue;
    }
    const unsigned long long k = order_key(x);
    const unsigned d = (unsigned)(k >> shift) & 255u;
    if (shift == 56) { atomicAdd(&h[0][d], 1u); continue; }  // (the second histogram of the first pass is the first)
    const unsigned long long hi = k >> (shift + 8);
    if (hi == p0) atomicAdd(&h[0][d], 1u);
    if (hi == p1) atomicAdd(&h[1][d], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * 257; i += 256) {
    const unsigned c = (&h[0][0])[i];
    if (c) atomicAdd(&hist[i], (unsigned long long)c);
  }
}
// one thread per order statistic: pick the digit, descend; zero the histograms for the next pass; after the last
// pass (decide >= 0) the median and the stop rule
__global__ __launch_bounds__(256) void k_auto_pick(unsigned long long* __restrict__ hist, AutoState* __restrict__ st, int pass, int step,
                                                   int min_steps) {
  if (st->stopped_at) return;
  __shared__ unsigned long long val[2];
  __shared__ long long incl[2][256];
  __shared__ int digit[2];
  const int t = threadIdx.x;                          // 256 threads: one per bin, both order statistics side by side
  // inclusive prefix sums of the two histograms (the first pass has one, shared by both): the serial scan of 256 bins
  // by one thread took 17 us per pass, 0.4 ms per walk with the default stop rule at 1M cells
  for (int w = 0; w < 2; ++w) incl[w][t] = (long long)hist[(pass == 0 ? 0 : 257 * w) + t];
  if (t < 2) digit[t] = 255;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    long long a0 = 0, a1 = 0;
    if (t >= off) { a0 = incl[0][t - off]; a1 = incl[1][t - off]; }
    __syncthreads();
    incl[0][t] += a0;
    incl[1][t] += a1;
    __syncthreads();
  }
  if (t < 2) {
    const int w = t;
    long long k;
    if (pass == 0) {
      const long long tot = incl[w][255];
      const long long nan = (long long)hist[256];
      if (w == 0) { st->n_tot = tot + nan; st->n_nan = nan; }
      k = w == 0 ? (tot - 1) / 2 : tot / 2;
      if (k >= tot) k = tot - 1;
      if (k < 0) k = 0;
    } else {
      k = st->k[w];
    }
    val[w] = (unsigned long long)k;                   // (handed to the search below)
  }
  __syncthreads();
  // the digit: the first bin whose inclusive sum exceeds k (the last one if none does)
  for (int w = 0; w < 2; ++w) {
    const long long k = (long long)val[w];
    const long long before = t ? incl[w][t - 1] : 0;
    if (t < 255 && before <= k && incl[w][t] > k) digit[w] = t;
  }
  __syncthreads();
  if (t < 2) {
    const int w = t;
    const int d = digit[w];
    const long long before = d ? incl[w][d - 1] : 0;
    st->k[w] = (long long)val[w] - before;
    const unsigned long long prefix = ((pass == 0 ? 0ull : st->prefix[w]) << 8) | (unsigned long long)d;
    st->prefix[w] = prefix;
    val[w] = (prefix >> 63) ? (prefix & 0x7fffffffffffffffull) : ~prefix;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * 257; i += blockDim.x) hist[i] = 0ull;
  if (pass == 7 && threadIdx.x == 0) {
    const double lo = __longlong_as_double((long long)val[0]), hi = __longlong_as_double((long long)val[1]);
    const long long tot = st->n_tot;
    const double med = (tot == 0 || st->n_nan > 0) ? __builtin_nan("") : ((tot & 1) ? lo : (lo + hi) / 2.0);
    st->median = med;
    if (step >= 0) {                                        // the walk's rule (_nam.py:64-68)
      st->med[step] = med;
      if (step + 1 >= min_steps && step >= 1 && (st->med[step - 1] - med < 3.0)) st->stopped_at = step + 1;
    }
  }
}
// _qc_nam (_nam.py:94-96) without the vector leaving the device: threshold = max(6, 2 median), and how many entries
// fail `kurtosis < threshold` (NaN fails).  Zero -- the usual outcome: with up to seven batches the kurtosis of the batch
// means cannot reach 6 -- means "keep every cell" and nothing cells-sized has to reach the host.
__global__ __launch_bounds__(256) void k_qc_count(const double* __restrict__ v, int64_t n, AutoState* __restrict__ st) {
  const double med = st->median;
  const double thr = fmax(6.0, 2.0 * med);                  // (NaN median: fmax gives 6, as Python's max(6, nan) does)
  if (blockIdx.x == 0 && threadIdx.x == 0) st->threshold = thr;
  unsigned long long cnt = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) cnt += !(v[i] < thr);
  cnt = (unsigned long long)wave_sum((double)cnt);
  if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&st->n_not_below, cnt);
}

__global__ __launch_bounds__(256) void k_max_fold(const unsigned long long* __restrict__ blockmax, int nblocks,
                                                  unsigned long long* __restrict__ out) {
  __shared__ unsigned long long sm[256];
  unsigned long long m = 0;
  for (int i = threadIdx.x; i < nblocks; i += 256) m = blockmax[i] > m ? blockmax[i] : m;
  sm[threadIdx.x] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o && sm[threadIdx.x + o] > sm[threadIdx.x]) sm[threadIdx.x] = sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = sm[0];
}

// number of t in [0,T) with arr[t] <= x, arr ascending, starting from a guess
__device__ __forceinline__ int count_le(const double* arr, int T, double x, int guess) {
  int h = guess < 0 ? 0 : (guess > T ? T : guess);
  while (h < T && arr[h] <= x) ++h;
  while (h > 0 && !(arr[h - 1] <= x)) --h;
  return h;
}
__device__ __forceinline__ int count_lt(const double* arr, int T, double x, int guess) {
  int h = guess < 0 ? 0 : (guess > T ? T : guess);
  while (h < T && arr[h] < x) ++h;
  while (h > 0 && !(arr[h - 1] < x)) --h;
  return h;
}
__device__ __forceinline__ int linear_guess(double z, double thr0, double inv_step, int T) {
  if (!(z >= thr0)) return 0;
  const double f = (z - thr0) * inv_step;
  return f >= (double)T ? T : (int)f + 1;
}

// ---- ranks / num_detected of the observed coefficients (_stats.py:74, _association.py:108) --
// hist[0..T)   : cells whose ncorr^2 falls in [edges[t], edges[t+1])   (suffix sum -> ranks)
// hist[T..2T)  : cells with thr[t] < |ncorr| <= thr[t+1]               (suffix sum -> num_detected)
__global__ __launch_bounds__(256) void k_obs_counts(const double* __restrict__ ncorrs, int64_t nx,
                                                    const double* __restrict__ edges,
                                                    const double* __restrict__ thr, int T, double thr0,
                                                    double inv_step, unsigned long long* hist) {
  extern __shared__ double smd[];
  double* e_s = smd;
  double* t_s = smd + T;
  unsigned int* h_s = (unsigned int*)(smd + 2 * T);
  for (int i = threadIdx.x; i < T; i += blockDim.x) {
    e_s[i] = edges[i];
    t_s[i] = thr[i];
  }
  for (int i = threadIdx.x; i < 2 * T; i += blockDim.x) h_s[i] = 0;
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nx; i += stride) {
    const double v = ncorrs[i];
    const double z = fabs(v), z2 = v * v;
    const int g = linear_guess(z, thr0, inv_step, T);
    const int hr = count_le(e_s, T, z2, g);
    const int hd = count_lt(t_s, T, z, g);
    if (hr > 0) atomicAdd(&h_s[hr - 1], 1u);
    if (hd > 0) atomicAdd(&h_s[T + hd - 1], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * T; i += blockDim.x)
    if (h_s[i]) atomicAdd(&hist[i], (unsigned long long)h_s[i]);
}

// tails[p][t] = sum_{t' >= t} hist[p][t']: one workgroup per permutation, Hillis-Steele suffix
// scan in LDS (integers -> exact in any order)
__global__ __launch_bounds__(512) void k_suffix_sum(const unsigned long long* __restrict__ hist, int P, int T,
                                                    int64_t* __restrict__ tails) {
  __shared__ long long buf[2][512];
  const int p = blockIdx.x, t = threadIdx.x;
  int cur = 0;
  buf[0][t] = t < T ? (long long)hist[(size_t)p * T + t] : 0;
  __syncthreads();
  for (int o = 1; o < 512; o <<= 1) {
    buf[cur ^ 1][t] = buf[cur][t] + (t + o < 512 ? buf[cur][t + o] : 0);
    cur ^= 1;
    __syncthreads();
  }
  if (t < T) tails[(size_t)p * T + t] = buf[cur][t];
}

// sums[t] = sum_p tails[p][t]  (the FDR is mean_p(tails[p][t]/ranks[t]), _stats.py:79-80).
// block = 64 thresholds x 16 permutation groups: coalesced 512-byte reads, 16-way LDS fold
__global__ __launch_bounds__(1024) void k_tail_sums(const int64_t* __restrict__ tails, int P, int T,
                                                    int64_t* __restrict__ sums) {
  __shared__ long long part[16][64];
  const int t = blockIdx.x * 64 + threadIdx.x, g = threadIdx.y;
  long long s = 0;
  if (t < T)
    for (int p = g; p < P; p += 16) s += tails[(size_t)p * T + t];
  part[g][threadIdx.x] = s;
  __syncthreads();
  if (g == 0 && t < T) {
    long long tot = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) tot += part[k][threadIdx.x];
    sums[t] = tot;
  }
}

__global__ void k_fill(double* v, int64_t n, double x) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = x;
}
__global__ void k_copy_f64(const double* __restrict__ src, double* __restrict__ dst, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}
__global__ void k_scatter(const double* __restrict__ src, const int64_t* __restrict__ keep, int64_t nx,
                          double* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nx) dst[keep ? keep[i] : i] = src[i];
}
// out[idx[i]] = in[i] for up to two vectors at once (per-cell outputs back in the caller's numbering)
__global__ void k_unpermute2(const double* __restrict__ a, const double* __restrict__ b,
                             const int64_t* __restrict__ idx, int64_t n, double* __restrict__ oa,
                             double* __restrict__ ob) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t j = idx[i];
  oa[j] = a[i];
  if (b) ob[j] = b[i];
}
// dst[k] = src[idx[k]] / dst[idx[k]] = src[k] for whole rows of ld doubles (16-byte lanes)
__global__ void k_pack_rows(const double2* __restrict__ src, const int64_t* __restrict__ idx, int64_t nrows,
                            int ld2, double2* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nrows * ld2) return;
  const int64_t k = i / ld2;
  const int col = (int)(i - k * ld2);
  dst[i] = src[idx[k] * ld2 + col];
}
__global__ void k_unpack_rows(const double2* __restrict__ src, const int64_t* __restrict__ idx, int64_t nrows,
                              int ld2, double2* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nrows * ld2) return;
  const int64_t k = i / ld2;
  const int col = (int)(i - k * ld2);
  dst[idx[k] * ld2 + col] = src[i];
}
// fdr_i = min{fdr_t : thr_t <= |coef_i|} else 1 (_association.py:234-237)
__global__ void k_percell_fdr(const double* __restrict__ coef, int64_t n, const double* __restrict__ thr,
                              const double* __restrict__ runmin, int T, double thr0, double inv_step,
                              double* __restrict__ fdr) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double z = fabs(coef[i]);
  const int h = count_le(thr, T, z, linear_guess(z, thr0, inv_step, T));
  fdr[i] = h > 0 ? runmin[h - 1] : 1.0;
}

// h_i = #{t : thr_t <= |coef_i|}, the count k_percell_fdr looks its FDR up with, as 16 bits per cell in the caller's
// cell order (orig null: identity): it depends on the observed coefficients and the thresholds only, so it can leave
// the device while the local null runs; the host finishes fdr_i = h_i > 0 ? runmin[h_i - 1] : 1 when the table arrives
__global__ void k_percell_bins(const double* __restrict__ coef, int64_t n, const double* __restrict__ thr, int T,
                               double thr0, double inv_step, const int64_t* __restrict__ orig,
                               unsigned short* __restrict__ bins) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double z = fabs(coef[i]);
  const int h = count_le(thr, T, z, linear_guess(z, thr0, inv_step, T));
  bins[orig ? orig[i] : i] = (unsigned short)h;
}

// dst[i][j] = src[rows[i]][cols[j]] (rows / cols null = identity), written row-major (n_out x n_cols)
// or transposed (n_cols x n_out): the matrices handed back to the caller leave the device already in
// the caller's cell order and orientation.
__global__ __launch_bounds__(256) void k_gather_rows(const double* __restrict__ src, int ld,
                                                     const int64_t* __restrict__ rows, int64_t n_out,
                                                     const int32_t* __restrict__ cols, int n_cols,
                                                     double* __restrict__ dst, int transposed) {
  __shared__ double tile[32][33];
  const int64_t i0 = (int64_t)blockIdx.x * 32;
  const int j0 = blockIdx.y * 32;
  for (int k = threadIdx.y; k < 32; k += blockDim.y) {
    const int64_t i = i0 + k;
    const int j = j0 + threadIdx.x;
    double v = 0.0;
    if (i < n_out && j < n_cols) v = src[(rows ? rows[i] : i) * ld + (cols ? cols[j] : j)];
    tile[k][threadIdx.x] = v;
  }
  __syncthreads();
  if (!transposed) {
    for (int k = threadIdx.y; k < 32; k += blockDim.y) {
      const int64_t i = i0 + k;
      const int j = j0 + threadIdx.x;
      if (i < n_out && j < n_cols) dst[i * n_cols + j] = tile[k][threadIdx.x];
    }
  } else {
    for (int k = threadIdx.y; k < 32; k += blockDim.y) {
      const int j = j0 + k;
      const int64_t i = i0 + threadIdx.x;
      if (j < n_cols && i < n_out) dst[(int64_t)j * n_out + i] = tile[threadIdx.x][k];
    }
  }
}

__global__ void k_transpose(const double* __restrict__ in, int64_t rows, int cols, int ld,
                            double* __restrict__ out) {
  __shared__ double tile[32][33];
  const int64_t r0 = (int64_t)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  for (int k = threadIdx.y; k < 32; k += blockDim.y) {
    const int64_t r = r0 + k;
    const int c = c0 + threadIdx.x;
    tile[k][threadIdx.x] = (r < rows && c < cols) ? in[r * ld + c] : 0.0;
  }
  __syncthreads();
  for (int k = threadIdx.y; k < 32; k += blockDim.y) {
    const int c = c0 + k;
    const int64_t r = r0 + threadIdx.x;
    if (c < cols && r < rows) out[(int64_t)c * rows + r] = tile[threadIdx.x][k];
  }
}

inline unsigned wave_grid(int64_t rows) {
  const int64_t want = (rows + 3) / 4;
  return (unsigned)(want < 4096 ? (want > 0 ? want : 1) : 4096);
}

}  // namespace

int launch_batch_kurtosis(cna_ctx* c, const double* mat, int64_t rows, int ncols, int ld,
                          const int32_t* order_dev, const int32_t* boff_dev, int n_batches, double* out) {
  if (rows == 0) return 0;
  if (n_batches > 256) CNA_FAIL(CNA_EINVAL, "more than 256 batches are not supported");
  ProfScope ps(c, CNA_K_BATCH_KURT);
  {
    const int rc = launch_rowpass16(c, mat, ld, nullptr, 0, rows, ncols, nullptr, nullptr, 0, 0, 0, 0, nullptr, nullptr,
                                    order_dev, boff_dev, n_batches, out);
    if (rc < 0) return rc;
    if (rc == 1) return 0;
  }
  if (n_batches <= BK_FAST && ncols <= 64 * MAXQ) {
#define BKF(Q) hipLaunchKernelGGL(k_batch_kurtosis_fast<Q>, dim3(wave_grid(rows)), dim3(256), 0, c->stream, mat, rows, ncols, ld, order_dev, boff_dev, n_batches, out)
    switch ((ncols + 63) / 64) {
      case 1: BKF(1); break;
      case 2: BKF(2); break;
      case 3: BKF(3); break;
      case 4: BKF(4); break;
      case 5: case 6: case 7: case 8: BKF(8); break;
      default: BKF(MAXQ); break;
    }
#undef BKF
    HIP_TRY(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(k_batch_kurtosis, dim3(wave_grid(rows)), dim3(256), sizeof(double) * 4 * ld, c->stream,
                     mat, rows, ncols, ld, order_dev, boff_dev, n_batches, out);
  HIP_TRY(hipGetLastError());
  return 0;
}

int launch_zero_variance(cna_ctx* c, const int32_t* colmap_dev, int n_sel, uint8_t* flags_dev,
                         unsigned long long* count_dev) {
  if (c->n_local == 0) return 0;
  ProfScope ps(c, CNA_K_ZEROVAR);
  hipLaunchKernelGGL(k_zero_variance, dim3(wave_grid(c->n_local)), dim3(256), 0, c->stream, c->nam,
                     c->n_local, c->ld, colmap_dev, n_sel, flags_dev, count_dev);
  HIP_TRY(hipGetLastError());
  return 0;
}

int launch_select(cna_ctx* c, const int32_t* colmap_dev) {
  const int64_t tot = c->nx * c->ldx;
  if (tot == 0) return 0;
  ProfScope ps(c, CNA_K_SELECT);
  hipLaunchKernelGGL(k_select, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream, c->nam, c->ld,
                     c->keep_idx, colmap_dev, c->X, c->nx, c->Nx, c->ldx);
  HIP_TRY(hipGetLastError());
  return 0;
}

int launch_select_zv(cna_ctx* c, const int32_t* colmap_dev, unsigned long long* count_dev) {
  HIP_TRY(hipMemsetAsync(count_dev, 0, sizeof(unsigned long long), c->stream));
  if (c->nx == 0) return 0;
  ProfScope ps(c, CNA_K_SELECT);
  hipLaunchKernelGGL(k_select_zv, dim3(wave_grid((c->nx + 3) / 4)), dim3(256), 0, c->stream, c->nam, c->ld, c->keep_idx,
                     colmap_dev, c->X, c->nx, c->Nx, c->ldx, count_dev);
  HIP_TRY(hipGetLastError());
  return 0;
}

int launch_select_std(cna_ctx* c, const int32_t* colmap_dev, unsigned long long* nzero_dev, const double* y_dev,
                      unsigned long long* maxbits_dev, const double* W_dev, const double* Ct_dev, int rk,
                      unsigned char* xq, void* xscale, int Kp) {
  // maxbits_dev (with y_dev): [0] = max |ncorrs| bits, [1 ..] per-workgroup partials (4097 words)
  HIP_TRY(hipMemsetAsync(nzero_dev, 0, sizeof(unsigned long long), c->stream));
  if (maxbits_dev) HIP_TRY(hipMemsetAsync(maxbits_dev, 0, sizeof(unsigned long long), c->stream));
  if (c->nx == 0) return 0;
  if (c->Nx > 64 * MAXQ) CNA_FAIL(CNA_EINVAL, "more than 1024 samples are not supported");
  ProfScope ps(c, CNA_K_SELECT);
  const unsigned grid = wave_grid((c->nx + 3) / 4);
  const size_t smem = sizeof(double) * 2 * (size_t)rk * c->Nx;
  if (smem > 128 * 1024) CNA_FAIL(CNA_EINVAL, "projector factors too large for LDS");
  // (rows16.hip's sixteen-rows-per-wave pass with the projector on the matrix cores, extended by this pass's extras --
  // row selection, zero-variance count, digit planes -- measured slower here: 1217 against 785 us at 1M x 100 with three
  // covariates, 731 against 405 at 500k x 128; it serves the in-place ridge pass and the batch kurtosis only)
  // sixteen lanes per cell up to 256 samples; with a projector the wave-per-cell kernel keeps the lead (its four
  // cells share every LDS read of the factors: 2.42 vs 2.71 ms at 2M x 200 with 5 covariates)
  if (c->Nx <= 256 && rk == 0) {
    const int cols = c->ldx > Kp ? c->ldx : Kp;
    const int G = (cols + 63) / 64;
    const unsigned grid16 = wave_grid((c->nx + 3) / 4);      // 16 rows per workgroup and turn
#define SS16(GG) { static bool once = false; if (smem > 48 * 1024 && !once) { HIP_TRY(hipFuncSetAttribute((const void*)k_select_std16<GG>, hipFuncAttributeMaxDynamicSharedMemorySize, 132 * 1024)); once = true; } \
    hipLaunchKernelGGL(k_select_std16<GG>, dim3(grid16), dim3(256), smem, c->stream, c->nam, c->ld, c->keep_idx, colmap_dev, c->X, c->nx, c->Nx, c->ldx, nzero_dev, y_dev, c->ncorrs, maxbits_dev ? maxbits_dev + 1 : nullptr, W_dev, Ct_dev, rk, xq, (double2*)xscale, Kp); }
    switch (G) {
      case 1: SS16(1) break;
      case 2: SS16(2) break;
      case 3: SS16(3) break;
      case 4: SS16(4) break;
      default: SS16(5) break;
    }
#undef SS16
    if (y_dev) hipLaunchKernelGGL(k_max_fold, dim3(1), dim3(256), 0, c->stream, maxbits_dev + 1, (int)grid16, maxbits_dev);
    HIP_TRY(hipGetLastError());
    return 0;
  }
#define SS_LAUNCH(Q) { static bool once = false; if (smem > 48 * 1024 && !once) { HIP_TRY(hipFuncSetAttribute((const void*)k_select_std<Q>, hipFuncAttributeMaxDynamicSharedMemorySize, 132 * 1024)); once = true; } \
    hipLaunchKernelGGL(k_select_std<Q>, dim3(grid), dim3(256), smem, c->stream, c->nam, c->ld, c->keep_idx, colmap_dev, c->X, c->nx, c->Nx, c->ldx, nzero_dev, y_dev, c->ncorrs, maxbits_dev ? maxbits_dev + 1 : nullptr, W_dev, Ct_dev, rk, xq, (double2*)xscale, Kp); }
  switch ((c->ldx + 63) / 64) {
    case 1: SS_LAUNCH(1) break;
    case 2: SS_LAUNCH(2) break;
    case 3: SS_LAUNCH(3) break;
    case 4: SS_LAUNCH(4) break;
    case 5: case 6: case 7: case 8: SS_LAUNCH(8) break;
    default: SS_LAUNCH(MAXQ) break;
  }
#undef SS_LAUNCH
  if (y_dev) hipLaunchKernelGGL(k_max_fold, dim3(1), dim3(256), 0, c->stream, maxbits_dev + 1, (int)grid, maxbits_dev);
  HIP_TRY(hipGetLastError());
  return 0;
}

size_t auto_state_bytes() { return sizeof(AutoState); }
int auto_state_stopped_offset() { return (int)offsetof(AutoState, stopped_at); }
int auto_state_result_offset() { return (int)offsetof(AutoState, median); }
// step >= 0: the walk's bookkeeping (med[step], the stop rule); step < 0: the median alone (AutoState::median).
// sum_over_ranks: the values are this rank's share of a vector spread over the ranks (X-space statistics): the digit
// histograms are summed over the ranks between counting and picking -- still no host round trip.
int launch_auto_median(cna_ctx* c, const double* v, int64_t n, void* state, unsigned long long* hist, int step, int min_steps,
                       bool sum_over_ranks) {
  const int64_t want = (n + 1023) / 1024;
  const unsigned grid = (unsigned)(want < 1 ? 1 : (want < 1024 ? want : 1024));
  HIP_TRY(hipMemsetAsync(hist, 0, sizeof(unsigned long long) * 2 * 257, c->stream));
  for (int pass = 0; pass < 8; ++pass) {
    hipLaunchKernelGGL(k_digit_hist2, dim3(grid), dim3(256), 0, c->stream, v, n, (const AutoState*)state, 56 - 8 * pass, hist);
    if (sum_over_ranks) CNA_TRY(comm_allreduce_i64_sum(c, (int64_t*)hist, 2 * 257));
    hipLaunchKernelGGL(k_auto_pick, dim3(1), dim3(256), 0, c->stream, hist, (AutoState*)state, pass, step, min_steps);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}
int launch_qc_count(cna_ctx* c, const double* v, int64_t n, void* state) {
  const int64_t want = (n + 1023) / 1024;
  const unsigned grid = (unsigned)(want < 1 ? 1 : (want < 1024 ? want : 1024));
  hipLaunchKernelGGL(k_qc_count, dim3(grid), dim3(256), 0, c->stream, v, n, (AutoState*)state);
  HIP_TRY(hipGetLastError());
  return 0;
}

namespace {
__global__ void k_pair_pack(const unsigned long long* cnt, const unsigned long long* maxbits, unsigned long long* slot) {
  slot[0] = cnt[0];
  slot[1] = maxbits[0];
}
__global__ void k_pair_fold(const unsigned long long* slots, int nranks, unsigned long long* cnt, unsigned long long* maxbits) {
  unsigned long long s = 0, m = 0;
  for (int r = 0; r < nranks; ++r) {              // fixed rank order: identical on every rank
    s += slots[2 * r];
    m = slots[2 * r + 1] > m ? slots[2 * r + 1] : m;
  }
  cnt[0] = s;
  maxbits[0] = m;
}
}  // namespace
int launch_pair_pack(cna_ctx* c, const unsigned long long* cnt, const unsigned long long* maxbits, unsigned long long* slot) {
  hipLaunchKernelGGL(k_pair_pack, dim3(1), dim3(1), 0, c->stream, cnt, maxbits, slot);
  HIP_TRY(hipGetLastError());
  return 0;
}
int launch_pair_fold(cna_ctx* c, const unsigned long long* slots, int nranks, unsigned long long* cnt, unsigned long long* maxbits) {
  hipLaunchKernelGGL(k_pair_fold, dim3(1), dim3(1), 0, c->stream, slots, nranks, cnt, maxbits);
  HIP_TRY(hipGetLastError());
  return 0;
}
int launch_max_fold(cna_ctx* c, const unsigned long long* blockmax, int nblocks, unsigned long long* out) {
  hipLaunchKernelGGL(k_max_fold, dim3(1), dim3(256), 0, c->stream, blockmax, nblocks, out);
  HIP_TRY(hipGetLastError());
  return 0;
}

int launch_digit_hist(cna_ctx* c, const double* v, int64_t n, unsigned long long prefix, int shift,
                      unsigned long long* hist_dev) {
  HIP_TRY(hipMemsetAsync(hist_dev, 0, sizeof(unsigned long long) * 257, c->stream));
  if (n > 0) {
    const int64_t want = (n + 1023) / 1024;
    const unsigned grid = (unsigned)(want < 1024 ? want : 1024);
    hipLaunchKernelGGL(k_digit_hist, dim3(grid), dim3(256), 0, c->stream, v, n, prefix, shift, hist_dev);
    HIP_TRY(hipGetLastError());
  }
  return 0;
}

int launch_standardize(cna_ctx* c, int center) {
  if (c->nx == 0) return 0;
  if (c->Nx > 64 * MAXQ) CNA_FAIL(CNA_EINVAL, "more than 1024 samples are not supported");
  ProfScope ps(c, CNA_K_STANDARDIZE);
  const unsigned grid = wave_grid((c->nx + 3) / 4);
  switch ((c->Nx + 63) / 64) {
#define STD_CASE(Q) case Q: hipLaunchKernelGGL(k_standardize<Q>, dim3(grid), dim3(256), 0, c->stream, c->X, c->nx, c->Nx, c->ldx, center); break
    STD_CASE(1); STD_CASE(2); STD_CASE(3); STD_CASE(4);
    case 5: case 6: case 7: STD_CASE(8);
    default: hipLaunchKernelGGL(k_standardize<MAXQ>, dim3(grid), dim3(256), 0, c->stream, c->X, c->nx, c->Nx, c->ldx, center);
#undef STD_CASE
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// W_dev: r x Nx, Ct_dev: r x Nx (C transposed); maxbits_dev as in launch_ncorrs (only with y_dev)
int launch_resid_lowrank(cna_ctx* c, const double* W_dev, const double* Ct_dev, int r, int center, int standardize,
                         const double* y_dev, unsigned long long* maxbits_dev, const int32_t* bk_order,
                         const int32_t* bk_boff, int nb, double* bk_out) {
  if (y_dev) HIP_TRY(hipMemsetAsync(maxbits_dev, 0, sizeof(unsigned long long), c->stream));
  if (c->nx == 0) return 0;
  if (c->Nx > 64 * MAXQ) CNA_FAIL(CNA_EINVAL, "more than 1024 samples are not supported");
  if (bk_out && nb > 256) CNA_FAIL(CNA_EINVAL, "more than 256 batches are not supported");
  const size_t smem = sizeof(double) * (2 * (size_t)r * c->Nx + (bk_out ? 4 * (size_t)c->ldx : 0));
  if (smem > 128 * 1024) CNA_FAIL(CNA_EINVAL, "cna_resid_lowrank: r x N too large for LDS");
  ProfScope ps(c, CNA_K_RESID);
  {
    const int rc = launch_rowpass16(c, c->X, c->ldx, c->X, c->ldx, c->nx, c->Nx, W_dev, Ct_dev, r, center, standardize, 1,
                                    y_dev, maxbits_dev, bk_order, bk_boff, nb, bk_out);
    if (rc < 0) return rc;
    if (rc == 1) return 0;
  }
  const int64_t want = (c->nx + 15) / 16;
  const unsigned grid = (unsigned)(want < 2048 ? want : 2048);
#define LR_CASE(Q) { static bool once = false; if (!once) { HIP_TRY(hipFuncSetAttribute((const void*)k_resid_lowrank<Q>, hipFuncAttributeMaxDynamicSharedMemorySize, 132 * 1024)); once = true; } \
    hipLaunchKernelGGL(k_resid_lowrank<Q>, dim3(grid), dim3(256), smem, c->stream, c->X, c->nx, c->Nx, c->ldx, W_dev, Ct_dev, r, center, standardize, y_dev, c->ncorrs, y_dev ? maxbits_dev + 1 : nullptr, bk_order, bk_boff, nb, bk_out); }
  switch ((c->Nx + 63) / 64) {
    case 1: LR_CASE(1) break;
    case 2: LR_CASE(2) break;
    case 3: LR_CASE(3) break;
    case 4: LR_CASE(4) break;
    case 5: case 6: case 7: case 8: LR_CASE(8) break;
    default: LR_CASE(MAXQ) break;
  }
#undef LR_CASE
  if (y_dev) hipLaunchKernelGGL(k_max_fold, dim3(1), dim3(256), 0, c->stream, maxbits_dev + 1, (int)grid, maxbits_dev);
  HIP_TRY(hipGetLastError());
  return 0;
}

int launch_ncorrs(cna_ctx* c, const double* y_dev, unsigned long long* maxbits_dev) {
  // maxbits_dev[0] = result, maxbits_dev[1 ..] = per-workgroup partials (caller reserves 2049 words)
  HIP_TRY(hipMemsetAsync(maxbits_dev, 0, sizeof(unsigned long long), c->stream));
  if (c->nx == 0) return 0;
  if (c->Nx > 64 * MAXQ) CNA_FAIL(CNA_EINVAL, "more than 1024 samples are not supported");
  ProfScope ps(c, CNA_K_NCORRS);
  const int64_t want = (c->nx + 15) / 16;
  const unsigned grid = (unsigned)(want < 2048 ? want : 2048);
  switch ((c->Nx + 63) / 64) {
#define NC_CASE(Q) case Q: hipLaunchKernelGGL(k_ncorrs<Q>, dim3(grid), dim3(256), 0, c->stream, c->X, c->nx, c->Nx, c->ldx, y_dev, c->ncorrs, maxbits_dev + 1); break
    NC_CASE(1); NC_CASE(2); NC_CASE(3); NC_CASE(4);
    case 5: case 6: case 7: NC_CASE(8);
    default: hipLaunchKernelGGL(k_ncorrs<MAXQ>, dim3(grid), dim3(256), 0, c->stream, c->X, c->nx, c->Nx, c->ldx, y_dev, c->ncorrs, maxbits_dev + 1);
#undef NC_CASE
  }
  hipLaunchKernelGGL(k_max_fold, dim3(1), dim3(256), 0, c->stream, maxbits_dev + 1, (int)grid, maxbits_dev);
  HIP_TRY(hipGetLastError());
  return 0;
}

int launch_obs_counts(cna_ctx* c, const double* edges_dev, const double* thr_dev, int T, double thr0,
                      double inv_step, unsigned long long* hist_dev) {
  HIP_TRY(hipMemsetAsync(hist_dev, 0, sizeof(unsigned long long) * 2 * T, c->stream));
  if (c->nx == 0 || T == 0) return 0;
  ProfScope ps(c, CNA_K_OBS_COUNTS);
  const int64_t want = (c->nx + 255) / 256;
  const unsigned grid = (unsigned)(want < 1024 ? want : 1024);
  const size_t sm = sizeof(double) * 2 * T + sizeof(unsigned int) * 2 * T;
  hipLaunchKernelGGL(k_obs_counts, dim3(grid), dim3(256), sm, c->stream, c->ncorrs, c->nx, edges_dev, thr_dev,
                     T, thr0, inv_step, hist_dev);
  HIP_TRY(hipGetLastError());
  return 0;
}

int launch_suffix_sum(cna_ctx* c, const unsigned long long* hist, int P, int T, int64_t* tails) {
  if (P == 0 || T == 0) return 0;
  if (T > 512) CNA_FAIL(CNA_EINVAL, "more than 512 FDR thresholds are not supported");
  hipLaunchKernelGGL(k_suffix_sum, dim3((unsigned)P), dim3(512), 0, c->stream, hist, P, T, tails);
  HIP_TRY(hipGetLastError());
  return 0;
}

int launch_tail_sums(cna_ctx* c, const int64_t* tails, int P, int T, int64_t* sums) {
  if (T == 0) return 0;
  hipLaunchKernelGGL(k_tail_sums, dim3((unsigned)((T + 63) / 64)), dim3(64, 16), 0, c->stream, tails, P, T, sums);
  HIP_TRY(hipGetLastError());
  return 0;
}

int launch_copy_f64(cna_ctx* c, const double* src, double* dst, int64_t n) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_copy_f64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, src, dst, n);
  HIP_TRY(hipGetLastError());
  return 0;
}

int launch_percell_fdr(cna_ctx* c, const double* thr_dev, const double* runmin_dev, int T, double thr0,
                       double inv_step, double* coef_local, double* fdr_local) {
  const int64_t n = c->n_local;
  if (n == 0) return 0;
  ProfScope ps(c, CNA_K_PERCELL_FDR);
  const unsigned g = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(k_fill, dim3(g), dim3(256), 0, c->stream, coef_local, n, __builtin_nan(""));
  if (c->nx > 0)
    hipLaunchKernelGGL(k_scatter, dim3((unsigned)((c->nx + 255) / 256)), dim3(256), 0, c->stream, c->ncorrs,
                       c->keep_idx, c->nx, coef_local);
  if (fdr_local)
    hipLaunchKernelGGL(k_percell_fdr, dim3(g), dim3(256), 0, c->stream, coef_local, n, thr_dev, runmin_dev, T,
                       thr0, inv_step, fdr_local);
  HIP_TRY(hipGetLastError());
  return 0;
}

// fdr[t] = sums[t] / ranks[t] / P and its running fmin (reference _stats.py:79-80, _association.py:234):
// the per-threshold FDR table, on the device so that the per-cell lookup can follow the local null
// without a host round trip.  Same operations in the same order as the numpy expressions.
__global__ __launch_bounds__(512) void k_fdr_table(const int64_t* __restrict__ sums, const int64_t* __restrict__ ranks,
                                                   int T, int P, double* __restrict__ fdr, double* __restrict__ runmin) {
  // one thread per threshold (T <= 512); the running fmin is an inclusive scan -- fmin with NaN as
  // its neutral element is associative, so the scan equals numpy's left-to-right accumulate
  __shared__ double sh[2][512];
  const int t = threadIdx.x;
  double f = __builtin_nan("");
  if (t < T) {
    f = __ddiv_rn(__ddiv_rn((double)sums[t], (double)ranks[t]), (double)P);
    fdr[t] = f;
  }
  int cur = 0;
  sh[0][t] = f;
  __syncthreads();
  for (int d = 1; d < 512; d <<= 1) {
    const double v = t >= d ? fmin(sh[cur][t - d], sh[cur][t]) : sh[cur][t];
    sh[cur ^ 1][t] = v;
    cur ^= 1;
    __syncthreads();
  }
  if (t < T) runmin[t] = sh[cur][t];
}

int launch_fdr_table(cna_ctx* c, const int64_t* sums, const int64_t* ranks, int T, int P, double* fdr, double* runmin) {
  hipLaunchKernelGGL(k_fdr_table, dim3(1), dim3(512), 0, c->stream, sums, ranks, T, P, fdr, runmin);
  HIP_TRY(hipGetLastError());
  return 0;
}

int launch_percell_bins(cna_ctx* c, hipStream_t st, const double* coef_local, const double* thr_dev, int T, double thr0,
                        double inv_step, unsigned short* bins) {
  const int64_t n = c->n_local;
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_percell_bins, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, coef_local, n, thr_dev, T, thr0,
                     inv_step, c->orig_idx, bins);
  HIP_TRY(hipGetLastError());
  return 0;
}

int launch_unpermute2(cna_ctx* c, const double* a, const double* b, const int64_t* idx, int64_t n, double* oa,
                      double* ob) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_unpermute2, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, a, b, idx, n, oa, ob);
  HIP_TRY(hipGetLastError());
  return 0;
}

int launch_pack_rows(cna_ctx* c, const double* src, const int64_t* idx, int64_t nrows, int ld, double* dst, hipStream_t st) {
  if (nrows == 0) return 0;
  const int64_t work = nrows * (ld / 2);
  hipLaunchKernelGGL(k_pack_rows, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st ? st : c->stream, (const double2*)src,
                     idx, nrows, ld / 2, (double2*)dst);
  HIP_TRY(hipGetLastError());
  return 0;
}
int launch_unpack_rows(cna_ctx* c, const double* src, const int64_t* idx, int64_t nrows, int ld, double* dst, hipStream_t st) {
  if (nrows == 0) return 0;
  const int64_t work = nrows * (ld / 2);
  hipLaunchKernelGGL(k_unpack_rows, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st ? st : c->stream,
                     (const double2*)src, idx, nrows, ld / 2, (double2*)dst);
  HIP_TRY(hipGetLastError());
  return 0;
}

int launch_gather_rows(cna_ctx* c, const double* src, int ld, const int64_t* rows_dev, int64_t n_out,
                       const int32_t* cols_dev, int n_cols, double* dst, int transposed) {
  if (n_out == 0 || n_cols == 0) return 0;
  ProfScope ps(c, CNA_K_TRANSPOSE);
  dim3 grid((unsigned)((n_out + 31) / 32), (unsigned)((n_cols + 31) / 32)), block(32, 8);
  hipLaunchKernelGGL(k_gather_rows, grid, block, 0, c->stream, src, ld, rows_dev, n_out, cols_dev, n_cols, dst, transposed);
  HIP_TRY(hipGetLastError());
  return 0;
}

int launch_transpose(cna_ctx* c, const double* in, int64_t rows, int cols, int ld, double* out) {
  if (rows == 0 || cols == 0) return 0;
  ProfScope ps(c, CNA_K_TRANSPOSE);
  dim3 grid((unsigned)((rows + 31) / 32), (unsigned)((cols + 31) / 32)), block(32, 8);
  hipLaunchKernelGGL(k_transpose, grid, block, 0, c->stream, in, rows, cols, ld, out);
  HIP_TRY(hipGetLastError());
  return 0;
}
