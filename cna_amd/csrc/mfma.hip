// Dense float64 contractions over the cell-major matrix X (cells x samples) on the gfx950
// matrix cores, v_mfma_f64_16x16x4_f64:
//   k_xb    OUT = (X [- rowmean]) . B     residualisation M.NAM (_nam.py:135,148), V = NAM^T U/sqrt(svs) (:106)
//   k_gram  G   = X^T X                   NAM.dot(NAM.T) (_nam.py:105), contraction over cells
//   k_null  fused |X.Yc|/N -> square -> threshold bin -> per-permutation histogram
//           (_association.py:96-99 + _stats.py:34-62), never materialising cells x Nnull.
//
// Operand maps of __builtin_amdgcn_mfma_f64_16x16x4f64 (one f64 per lane for A and B):
//   A[i][k]: i = lane & 15, k = lane >> 4        B[k][j]: k = lane >> 4, j = lane & 15
//   D[r][j]: r = (lane >> 4) + 4*reg, j = lane & 15,  reg in 0..3
#include "common.h"
#include <algorithm>
#include <array>
#include <cstdlib>
#include <utility>

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

// ======================================================================== OUT = X . B
// One wave owns a 16-cell tile and keeps it in registers in MFMA A layout (lane (i,k): X[r0+i][4q+k],
// one instantiation per exact k-depth), optionally centred by its row means (two cross-lane adds);
// the workgroup stages B (K x ldb, zero padded, L2 resident) through LDS in strips of 16*NS columns
// and every wave runs NS independent accumulator chains per strip.  A wave reads its 16 rows
// completely before it writes any of them, so OUT may be X itself (in-place residualisation).
// (The first version fed one dependent chain per wave with B fragments read from global for every
// MFMA: 28.9 ms at 2M x 200, 0.07 of the f64 MFMA peak.)
// N = 132 ... 208: capped at 128 VGPRs (a handful of spills) two workgroups share a CU, 4928 -> 4176 us
// at 2M x 200; deeper tiles spill too much (N = 256: 3853 -> 4996 us) and keep the full register file
#ifndef CNA_NULL_PD
#define CNA_NULL_PD 0   // measured on MI355X: no gain (N = 200: 53.7 vs 54.6 TFLOP/s without), worse at N = 128 (43.5 vs 50.8)
#endif
#ifndef CNA_XB_CAP
#define CNA_XB_CAP 52
#endif
#define XB_MIN_WAVES(KQ) (((KQ) > 32 && (KQ) <= CNA_XB_CAP) ? 4 : 1)
// More than 256 samples (k-depth beyond 64 quads): the product is split along k into parts of up to 256
// columns; part p reads columns [k_off, k_off + 4 KQ) of X (row stride ldx) and the matching rows of B, and
// adds into OUT (accumulate), which then cannot be X itself -- launch_xb works into a second buffer and
// swaps.  Centring uses row means taken beforehand over the whole row (k_row_means).
template <int KQ, int NS>
__global__ __launch_bounds__(512, XB_MIN_WAVES(KQ)) void k_xb(const double* __restrict__ X, int64_t nx, int Nx,
                                            const double* __restrict__ B, int ldb, int center,
                                            double* out, int ld_out, int ldx, int k_off,
                                            const double* __restrict__ means, int accumulate) {
  constexpr int LDX = 4 * KQ, PT = 16 * NS, LDB = PT + 16;   // LDB = 16 mod 32: conflict-free B fragments
  extern __shared__ double sm[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ai = lane & 15, ak = lane >> 4;
  const int64_t r0 = ((int64_t)blockIdx.x * 8 + wv) * 16;
  double a[KQ];
  {
    const int64_t row = r0 + ai;
    if (row < nx) {
      const double* __restrict__ xp = X + row * ldx + k_off + ak;
#pragma unroll
      for (int q = 0; q < KQ; ++q) a[q] = xp[4 * q];
    } else {
#pragma unroll
      for (int q = 0; q < KQ; ++q) a[q] = 0.0;
    }
  }
  if (means) {                               // split product: the row mean comes from k_row_means
    const int64_t row = r0 + ai;
    const double mean = row < nx ? means[row] : 0.0;
#pragma unroll
    for (int q = 0; q < KQ; ++q)
      if (k_off + 4 * q + ak < Nx) a[q] -= mean;
  } else if (center) {                       // pad columns of X are zero, so they do not disturb the sum
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < KQ; ++q) s += a[q];
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    const double mean = s / (double)Nx;
#pragma unroll
    for (int q = 0; q < KQ; ++q)
      if (4 * q + ak < Nx) a[q] -= mean;
  }
  const double* bp = sm + ak * LDB + ai;
  for (int c0 = 0; c0 < ldb; c0 += PT) {
    __syncthreads();                          // the previous strip has been consumed
    for (int i = tid; i < LDX * PT; i += 512) {
      const int k = i / PT, j = i - k * PT;
      sm[k * LDB + j] = (c0 + j < ldb) ? B[(size_t)k * ldb + c0 + j] : 0.0;
    }
    __syncthreads();
    v4d acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
#pragma unroll
      for (int s = 0; s < NS; ++s)
        acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], bp[4 * q * LDB + 16 * s], acc[s], 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int col = c0 + 16 * s + ai;
      if (col < ld_out) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t gr = r0 + ak + 4 * r;
          if (gr < nx) out[gr * ld_out + col] = accumulate ? out[gr * ld_out + col] + acc[s][r] : acc[s][r];
        }
      }
    }
  }
}

// mean over the Nx columns of every row of X (pad columns are zero); one wave per row
__global__ __launch_bounds__(256) void k_row_means(const double* __restrict__ X, int64_t nx, int Nx, int ldx,
                                                   double* __restrict__ means) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nx) return;
  double s = 0.0;
  for (int col = lane; col < Nx; col += 64) s += X[row * ldx + col];
  s = wave_sum(s);
  if (lane == 0) means[row] = s / (double)Nx;
}

// Up to 128 samples the whole of B fits in LDS: one load per workgroup, then 16 waves walk the row
// tiles with no further barrier (k_xb re-stages B and synchronises twice per 16 x 64 output strip,
// which at N = 100 leaves only 200 MFMAs per wave between barriers: 0.35 of the MFMA peak).
template <int KQ>
__global__ __launch_bounds__(1024) void k_xb_res(const double* __restrict__ X, int64_t nx, int Nx,
                                                 const double* __restrict__ B, int ldb, int center,
                                                 double* out, int ld_out, int64_t ntile) {
  constexpr int LDX = 4 * KQ, NS = 4;
  extern __shared__ double sm[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ai = lane & 15, ak = lane >> 4;
  const int nct = ((ldb + 63) / 64) * 4;                // 16-column tiles of B, padded with zero tiles to whole strips of 4
  const int LDB = 16 * nct + 16;                        // = 16 mod 32: conflict-free B fragments
  for (int i = tid; i < LDX * LDB; i += 1024) {
    const int k = i / LDB, j = i - k * LDB;
    sm[i] = j < ldb ? B[(size_t)k * ldb + j] : 0.0;
  }
  __syncthreads();
  const double* bp = sm + ak * LDB + ai;
  for (int64_t tile = (int64_t)blockIdx.x * 16 + wv; tile < ntile; tile += (int64_t)gridDim.x * 16) {
    const int64_t r0 = tile * 16;
    double a[KQ];
    {
      const int64_t row = r0 + ai;
      if (row < nx) {
        const double* __restrict__ xp = X + row * LDX + ak;
#pragma unroll
        for (int q = 0; q < KQ; ++q) a[q] = xp[4 * q];
      } else {
#pragma unroll
        for (int q = 0; q < KQ; ++q) a[q] = 0.0;
      }
    }
    if (center) {
      double s = 0.0;
#pragma unroll
      for (int q = 0; q < KQ; ++q) s += a[q];
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      const double mean = s / (double)Nx;
#pragma unroll
      for (int q = 0; q < KQ; ++q)
        if (4 * q + ak < Nx) a[q] -= mean;
    }
    for (int ct = 0; ct < nct; ct += NS) {
      v4d acc[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) acc[s] = (v4d){0.0, 0.0, 0.0, 0.0};
      const double* bs = bp + 16 * ct;
#pragma unroll
      for (int q = 0; q < KQ; ++q) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
          acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], bs[4 * q * LDB + 16 * s], acc[s], 0, 0, 0);
      }
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int col = 16 * (ct + s) + ai;
        if (col < ld_out) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int64_t gr = r0 + ak + 4 * r;
            if (gr < nx) out[gr * ld_out + col] = acc[s][r];
          }
        }
      }
    }
  }
}

// ======================================================================== G = X^T X
// 16 waves per workgroup; the nt*(nt+1)/2 upper-triangular 16x16 tiles of G are dealt to the
// waves (TPW accumulator tiles each, kept in registers for the whole kernel); the workgroup
// streams 32-cell slabs of X through LDS and every wave feeds its tiles with 8 k-steps of 4
// cells.  Per-workgroup partial tiles are written out and summed in a fixed order by
// k_gram_reduce (deterministic; no float atomics).
template <int TPW, int NW = 8, int SLAB = 32>
__global__ __launch_bounds__(64 * NW) void k_gram(const double* __restrict__ X, int64_t nx, int ldx, int nt,
                                              int ldp, int ntri, const int32_t* __restrict__ tiles,
                                              double* __restrict__ partial, int64_t slab0, int64_t slab1, int accumulate) {
  extern __shared__ double sm[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ak = lane >> 4, ai = lane & 15;
  v4d acc[TPW];
  int off_i[TPW], off_j[TPW];
  bool live[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
    const int tix = (blockIdx.y * NW + wv) * TPW + t;
    live[t] = tix < ntri;
    const int packed = live[t] ? __builtin_amdgcn_readfirstlane(tiles[tix]) : 0;
    off_i[t] = (packed >> 16) * 16;
    off_j[t] = (packed & 0xffff) * 16;
    if (accumulate && live[t]) {                          // a later row range of the same product (launch_gram_range)
      const double* p = partial + ((size_t)blockIdx.x * ntri + tix) * 256;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][r] = p[r * 64 + lane];
    }
  }
  for (int i = tid; i < SLAB * ldp; i += 64 * NW) sm[i] = 0.0;
  const int64_t nslab = slab1;
  for (int64_t slab = slab0 + blockIdx.x; slab < nslab; slab += gridDim.x) {
    __syncthreads();
    const int64_t r0 = slab * SLAB;
    for (int r = wv; r < SLAB; r += NW) {
      const int64_t gr = r0 + r;
      for (int col = lane; col < ldx; col += 64) sm[r * ldp + col] = (gr < nx) ? X[gr * ldx + col] : 0.0;
    }
    __syncthreads();
#pragma unroll 1
    for (int kq = 0; kq < SLAB / 4; ++kq) {
      const double* rowp = sm + (4 * kq + ak) * ldp + ai;
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        if (live[t]) {
          const double a = rowp[off_i[t]];
          const double b = rowp[off_j[t]];
          acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    if (live[t]) {
      const int tix = (blockIdx.y * NW + wv) * TPW + t;
      double* p = partial + ((size_t)blockIdx.x * ntri + tix) * 256;
#pragma unroll
      for (int r = 0; r < 4; ++r) p[r * 64 + lane] = acc[t][r];
    }
  }
}
// X^T X with the tiles of a wave arranged as a 3 x 3 block (three tile rows x three tile columns of the upper
// triangle): per k step three A and three B fragments from LDS feed up to nine MFMAs -- 0.67-1 LDS reads per
// MFMA against 1.25 (k_gram_db with the A reuse) or 2 (without).  On this chip an f64 MFMA does not hide the
// LDS -> VGPR return of its operands (tools/micro/mfma_f64_rate.hip), so reads per MFMA set the pipe's duty.
// One block per wave, sixteen waves: fits when the triangle has at most 16 blocks, i.e. 11 <= tiles per side
// <= 15 (161 ... 240 samples); `blocks` holds per wave {i0, i1, i2, j0, j1, j2} (tile indices, -1: unused) in
// an order that spreads the work over the four SIMDs; tixmap[i * nt + j] = index of tile (i, j) in the
// upper-triangular table k_gram_reduce walks.
template <int NW, int SLAB, bool ACC = false, int PF = 5>
__global__ __launch_bounds__(64 * NW) void k_gram_blk(const double* __restrict__ X, int64_t nx, int ldx, int nt,
                                                  int ldp, int ntri, const int32_t* __restrict__ blocks,
                                                  const int32_t* __restrict__ tixmap, double* __restrict__ partial,
                                                  int64_t slab0, int64_t slab1, int accumulate) {
  extern __shared__ double sm[];
  constexpr int NT = 64 * NW;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ak = lane >> 4, ai = lane & 15;
  int ri[3], cj[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    ri[r] = __builtin_amdgcn_readfirstlane(blocks[wv * 8 + r]);
    cj[r] = __builtin_amdgcn_readfirstlane(blocks[wv * 8 + 3 + r]);
  }
  bool live[9];
  v4d acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
    live[t] = ri[t / 3] >= 0 && cj[t % 3] >= ri[t / 3];
    if (ACC && live[t]) {                                 // a later row range of the same product (launch_gram_range)
      const double* p = partial + ((size_t)blockIdx.x * ntri + tixmap[ri[t / 3] * nt + cj[t % 3]]) * 256;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][r] = p[r * 64 + lane];
    }
  }
  for (int i = tid; i < 2 * SLAB * ldp; i += NT) sm[i] = 0.0;
  const int64_t nslab = slab1;
  const int slab2 = SLAB * ldx / 2;
  int lrow[PF], lcol[PF];
#pragma unroll
  for (int u = 0; u < PF; ++u) {
    const int i = 2 * (tid + u * NT);
    lrow[u] = i / ldx;
    lcol[u] = i - lrow[u] * ldx;
  }
  double2 v[PF];
  auto gload = [&](int64_t slab) {
    const int64_t r0 = slab * SLAB;
    const int64_t rows = nx - r0 < SLAB ? nx - r0 : SLAB;
    const int lim = (int)(rows * ldx / 2);
    const double2* __restrict__ src = (const double2*)(X + r0 * ldx);
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int i2 = tid + u * NT;
      v[u] = i2 < lim ? src[i2] : make_double2(0.0, 0.0);
    }
  };
  auto lstore = [&](double* buf) {
#pragma unroll
    for (int u = 0; u < PF; ++u)
      if (tid + u * NT < slab2) *(double2*)(buf + lrow[u] * ldp + lcol[u]) = v[u];
  };
  int64_t slab = slab0 + blockIdx.x;
  __syncthreads();
  if (slab < nslab) { gload(slab); lstore(sm); }
  __syncthreads();
  int cur = 0;
  const int oa0 = 16 * (ri[0] > 0 ? ri[0] : 0), oa1 = 16 * (ri[1] > 0 ? ri[1] : 0), oa2 = 16 * (ri[2] > 0 ? ri[2] : 0);
  const int ob0 = 16 * (cj[0] > 0 ? cj[0] : 0), ob1 = 16 * (cj[1] > 0 ? cj[1] : 0), ob2 = 16 * (cj[2] > 0 ? cj[2] : 0);
  for (; slab < nslab; slab += gridDim.x) {
    const int64_t next = slab + gridDim.x;
    if (next < nslab) gload(next);
    const double* buf = sm + cur * SLAB * ldp;
#pragma unroll        // all eight k steps of a slab: the next fragments are read under the MFMAs (1965 -> 1908 us at 2M x 200)
    for (int kq = 0; kq < SLAB / 4; ++kq) {
      const double* rowp = buf + (4 * kq + ak) * ldp + ai;
      const double a0 = rowp[oa0], a1 = rowp[oa1], a2 = rowp[oa2];
      const double b0 = rowp[ob0], b1 = rowp[ob1], b2 = rowp[ob2];
      if (live[0]) acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0], 0, 0, 0);
      if (live[1]) acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[1], 0, 0, 0);
      if (live[2]) acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b2, acc[2], 0, 0, 0);
      if (live[3]) acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[3], 0, 0, 0);
      if (live[4]) acc[4] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[4], 0, 0, 0);
      if (live[5]) acc[5] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b2, acc[5], 0, 0, 0);
      if (live[6]) acc[6] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b0, acc[6], 0, 0, 0);
      if (live[7]) acc[7] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b1, acc[7], 0, 0, 0);
      if (live[8]) acc[8] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, acc[8], 0, 0, 0);
    }
    if (next < nslab) lstore(sm + (cur ^ 1) * SLAB * ldp);
    __syncthreads();
    cur ^= 1;
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    if (live[t]) {
      const int tix = tixmap[ri[t / 3] * nt + cj[t % 3]];
      double* p = partial + ((size_t)blockIdx.x * ntri + tix) * 256;
#pragma unroll
      for (int r = 0; r < 4; ++r) p[r * 64 + lane] = acc[t][r];
    }
  }
}


// ---- selection pass and Gram matrix in ONE kernel (round 3; 161 ... 240 samples, all cells, samples in place).
// The standardisation of the NAM rows is row-local (_nam.py:122,159), so a Gram workgroup can standardise its own
// 32-cell slab: the raw NAM rows of the NEXT slab arrive by LDS-DMA (inline assembly: hipcc neither counts nor
// serialises these copies) while the 3 x 3 tile blocks of k_gram_blk run on the current one; then the eight waves --
// sixteen lanes per cell, four cells per wave, the arithmetic of rows.hip:k_select_std16 statement by statement --
// standardise the slab from LDS into the MFMA buffer and write X, the digit planes of the integer local null, the
// neighbourhood coefficients (X.y/N, _association.py:77) and the zero-variance count on the way.  X is read by
// nobody before the local null any more: 3.2 GB less traffic and one launch less at 2M x 200.  Same slabs per
// workgroup, same order of the MFMAs and of k_gram_reduce: G is bit-identical to k_select_std16 + k_gram_blk.
__device__ __forceinline__ double sg_row16_sum(double v) {
  v = dpp_add(v, 0);
  v = dpp_add(v, 1);
  v = dpp_add(v, 2);
  return dpp_add(v, 3);
}
__device__ __forceinline__ double sg_row16_max(double v) {
  v = fmax(v, dpp_partner(v, 0));
  v = fmax(v, dpp_partner(v, 1));
  v = fmax(v, dpp_partner(v, 2));
  return fmax(v, dpp_partner(v, 3));
}
__device__ __forceinline__ void sg_dma16(const void* gsrc, unsigned lds_base) {      // 16 bytes per lane, LDS = base + 16 lane
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}

#ifndef SG_UNROLL
#define SG_UNROLL 1     // deeper unrolling spills out of the 256 registers (18 accumulator tiles are live)
#endif
template <int G>
__global__ __launch_bounds__(512) void k_selgram_blk(const double* __restrict__ nam, int ld, double* __restrict__ X,
                                                     int64_t nx, int Nx, int ldx, unsigned long long* nzero,
                                                     const double* __restrict__ y, double* __restrict__ nc,
                                                     unsigned long long* __restrict__ blockmax,
                                                     unsigned char* __restrict__ xq, double2* __restrict__ xscale, int Kp,
                                                     int nt, int ldp, int ntri, const int32_t* __restrict__ blocks,
                                                     const int32_t* __restrict__ tixmap, double* __restrict__ partial) {
  // EIGHT waves (two per SIMD, 256 registers each): a wave carries the two 3 x 3 tile blocks that waves w and w + 8 of
  // k_gram_blk carry (the same four blocks per SIMD) -- 18 accumulator tiles -- and still has registers for the
  // standardisation of four cells, in which all eight waves take part (32 cells per slab).
  extern __shared__ double sm[];
  constexpr int NW = 8, SLAB = 32, NT = 64 * NW;
  double* buf = sm;                                   // SLAB x ldp: the standardised slab (MFMA operands)
  double* raw = sm + SLAB * ldp;                      // SLAB x ld : raw NAM rows of the next slab
  double* ysh = raw + SLAB * ld;                      // 64 G: the phenotype, zero beyond Nx
  unsigned long long* wmax = (unsigned long long*)(ysh + 64 * G);   // 8: per-wave max |coefficient|
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ak = lane >> 4, ai = lane & 15;
  int ri[2][3], cj[2][3];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      ri[h][r] = __builtin_amdgcn_readfirstlane(blocks[(wv + 8 * h) * 8 + r]);
      cj[h][r] = __builtin_amdgcn_readfirstlane(blocks[(wv + 8 * h) * 8 + 3 + r]);
    }
  bool live_t[2][9];
  v4d acc[2][9];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      acc[h][t] = (v4d){0.0, 0.0, 0.0, 0.0};
      live_t[h][t] = ri[h][t / 3] >= 0 && cj[h][t % 3] >= ri[h][t / 3];
    }
  for (int i = tid; i < SLAB * ldp; i += NT) buf[i] = 0.0;
  for (int i = tid; i < 64 * G; i += NT) ysh[i] = (y && i < Nx) ? y[i] : 0.0;
  const int64_t nslab = (nx + SLAB - 1) / SLAB;
  const unsigned raw_lds = (unsigned)(size_t)raw;
  const int slab_bytes = SLAB * ld * 8;
  const int64_t nam_bytes = nx * (int64_t)ld * 8;
  auto dma_raw = [&](int64_t slab) {                  // rows [32 slab, 32 slab + 32) x ld doubles are contiguous in the NAM
    const int64_t b0 = slab * (int64_t)slab_bytes;
    for (int pc = wv; pc * 1024 < slab_bytes; pc += NW) {
      int64_t off = b0 + pc * 1024 + lane * 16;
      if (off > nam_bytes - 16) off = nam_bytes - 16;  // past the last row: any valid address (those rows are not live)
      sg_dma16((const char*)nam + off, raw_lds + (unsigned)(pc * 1024));
    }
  };
  const int r4 = lane >> 4, l16 = lane & 15;
  const double n = (double)Nx;
  double vmax = 0.0;
  bool any_nan = false;
  // waves 0-7: rows 4 wv + r4 of the slab from `raw` into `buf` (and out to memory)
  // Registers are scarce here (nine accumulator tiles per lane stay live): every pass over the row re-reads its four
  // columns per group from LDS instead of keeping the row in registers -- the same values, the same operations in
  // the same order as rows.hip:k_select_std16.
  auto standardize = [&](int64_t slab) {
    const int srow = 4 * wv + r4;
    const int64_t row = slab * SLAB + srow;
    const bool live = row < nx;
    const double* rrow = raw + srow * ld;
    auto rawval = [&](int col) -> double { return (live && col < Nx) ? rrow[col] : 0.0; };
    double sum = 0.0;
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) sum += rawval(64 * g + 4 * l16 + j);
    const double avg0 = sg_row16_sum(sum) / n;
    bool flat = true;
    double s2 = 0.0;
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = 64 * g + 4 * l16 + j;
        double xv = rawval(col);
        if (col < Nx) {
          flat = flat && (avg0 - xv == 0.0);
          xv -= avg0;
        }
        s2 += xv;
      }
    {
      const unsigned long long bal = __ballot(flat);
      const unsigned long long rowmask = 0xffffull << (16 * r4);
      if (live && l16 == 0 && (bal & rowmask) == rowmask) atomicAdd(nzero, 1ull);
    }
    const double avg = sg_row16_sum(s2) / n;
    double ss = 0.0;
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = 64 * g + 4 * l16 + j;
        if (col < Nx) {
          double xv = rawval(col);
          xv -= avg0;
          const double d = avg - xv;
          ss += d * d;
        }
      }
    const double sd = sqrt(sg_row16_sum(ss) / (n - 1.0));
    double dot = 0.0, amax = 0.0;
    double* __restrict__ dst = X + row * ldx;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int c0 = 64 * g + 4 * l16;
      double xs4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = c0 + j;
        double xv = rawval(col);
        if (col < Nx) xv -= avg0;
        const double xs = col < Nx ? __ddiv_rn(xv, sd) : 0.0;
        xs4[j] = xs;
        dot += ysh[col] * xs;
        amax = fmax(amax, fabs(xs));
      }
      if (live) {
        if (c0 + 3 < ldx && (ldx & 1) == 0) {
          *(double2*)(dst + c0) = make_double2(xs4[0], xs4[1]);
          *(double2*)(dst + c0 + 2) = make_double2(xs4[2], xs4[3]);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (c0 + j < ldx) dst[c0 + j] = xs4[j];
        }
      }
      if (c0 + 3 < ldp) {                              // the slab for the MFMAs (rows past the end: zero)
        *(double2*)(buf + srow * ldp + c0) = live ? make_double2(xs4[0], xs4[1]) : make_double2(0.0, 0.0);
        *(double2*)(buf + srow * ldp + c0 + 2) = live ? make_double2(xs4[2], xs4[3]) : make_double2(0.0, 0.0);
      }
    }
    if (xq) {
      const double rmax = sg_row16_max(amax);
      const double inv = rmax > 0.0 ? I8_QMAX / rmax : 0.0;
      unsigned* rq = (unsigned*)(xq + (size_t)row * 3 * Kp);
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int c0 = 64 * g + 4 * l16;
        unsigned w0 = 0, w1 = 0, w2 = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          double xv = rawval(c0 + j);                  // the standardised value again (the slab may be narrower than Kp)
          if (c0 + j < Nx) xv -= avg0;
          const double xs = c0 + j < Nx ? __ddiv_rn(xv, sd) : 0.0;
          const double v = xs * inv;
          const int qi = v == v ? (int)rint(v) : 0;
          const int q1 = (qi + 128) >> 8;
          w0 |= ((unsigned)qi & 255u) << (8 * j);
          w1 |= ((unsigned)q1 & 255u) << (8 * j);
          w2 |= ((unsigned)((q1 + 128) >> 8) & 255u) << (8 * j);
        }
        if (live && c0 < Kp) {
          rq[c0 >> 2] = w0;
          rq[(Kp + c0) >> 2] = w1;
          rq[(2 * Kp + c0) >> 2] = w2;
        }
      }
      const double l1 = rmax > 0.0 ? n * (I8_QMAX / rmax) + n : 0.0;
      if (live && l16 == 0) xscale[row] = make_double2(rmax, l1);
    }
    if (y) {
      const double v = sg_row16_sum(dot) / n;
      if (live) {
        if (l16 == 0) nc[row] = v;
        const double av = fabs(v);
        if (av > vmax) vmax = av;
        any_nan = any_nan || (v != v);
      }
    }
  };
  auto lds_barrier = [&]() {                          // LDS traffic only: the global stores of X need not have landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  int64_t slab = blockIdx.x;
  __syncthreads();
  if (slab < nslab) {
    dma_raw(slab);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    standardize(slab);
    lds_barrier();
  }
  int oa[2][3], ob[2][3];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      oa[h][r] = 16 * (ri[h][r] > 0 ? ri[h][r] : 0);
      ob[h][r] = 16 * (cj[h][r] > 0 ? cj[h][r] : 0);
    }
  for (; slab < nslab; slab += gridDim.x) {
    const int64_t next = slab + gridDim.x;
    if (next < nslab) dma_raw(next);
#pragma unroll SG_UNROLL
    for (int kq = 0; kq < SLAB / 4; ++kq) {
      const double* rowp = buf + (4 * kq + ak) * ldp + ai;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const double a0 = rowp[oa[h][0]], a1 = rowp[oa[h][1]], a2 = rowp[oa[h][2]];
        const double b0 = rowp[ob[h][0]], b1 = rowp[ob[h][1]], b2 = rowp[ob[h][2]];
        if (live_t[h][0]) acc[h][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[h][0], 0, 0, 0);
        if (live_t[h][1]) acc[h][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[h][1], 0, 0, 0);
        if (live_t[h][2]) acc[h][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b2, acc[h][2], 0, 0, 0);
        if (live_t[h][3]) acc[h][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[h][3], 0, 0, 0);
        if (live_t[h][4]) acc[h][4] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[h][4], 0, 0, 0);
        if (live_t[h][5]) acc[h][5] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b2, acc[h][5], 0, 0, 0);
        if (live_t[h][6]) acc[h][6] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b0, acc[h][6], 0, 0, 0);
        if (live_t[h][7]) acc[h][7] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b1, acc[h][7], 0, 0, 0);
        if (live_t[h][8]) acc[h][8] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, acc[h][8], 0, 0, 0);
      }
    }
    if (next < nslab) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of the next slab's raw rows
      lds_barrier();                                        // everybody's; and all MFMAs have read `buf`
      standardize(next);
      lds_barrier();
    }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if (live_t[h][t]) {
        const int tix = tixmap[ri[h][t / 3] * nt + cj[h][t % 3]];
        double* p = partial + ((size_t)blockIdx.x * ntri + tix) * 256;
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r * 64 + lane] = acc[h][t][r];
      }
    }
  if (y) {
    {
      const bool wn = __any(any_nan);
      const double wm = wave_max_d(vmax);
      if (lane == 0) wmax[wv] = wn ? 0x7ff8000000000000ull : (unsigned long long)__double_as_longlong(wm);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long m = wmax[0];
      for (int i = 1; i < 8; ++i) m = wmax[i] > m ? wmax[i] : m;
      blockmax[blockIdx.x] = m;
    }
  }
}

// The same with two LDS slabs: the next slab's global loads are issued before the MFMAs of the current one
// and written to the other buffer after them -- one barrier per slab instead of two, and the load latency
// (~2 us per 53 KB slab at N = 200) disappears behind 48 MFMAs per wave.
template <int TPW, int NW, int SLAB>
__global__ __launch_bounds__(64 * NW) void k_gram_db(const double* __restrict__ X, int64_t nx, int ldx, int nt,
                                                 int ldp, int ntri, const int32_t* __restrict__ tiles,
                                                 double* __restrict__ partial, int64_t slab0, int64_t slab1, int accumulate) {
  extern __shared__ double sm[];
  constexpr int NT = 64 * NW, PF = 5;                  // PF double2 per thread cover SLAB x ldx doubles up to ldx = 320
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ak = lane >> 4, ai = lane & 15;
  v4d acc[TPW];
  int off_i[TPW], off_j[TPW];
  bool live[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
    const int tix = (blockIdx.y * NW + wv) * TPW + t;
    live[t] = tix < ntri;
    const int packed = live[t] ? __builtin_amdgcn_readfirstlane(tiles[tix]) : 0;
    off_i[t] = (packed >> 16) * 16;
    off_j[t] = (packed & 0xffff) * 16;
    if (accumulate && live[t]) {
      const double* p = partial + ((size_t)blockIdx.x * ntri + tix) * 256;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][r] = p[r * 64 + lane];
    }
  }
  for (int i = tid; i < 2 * SLAB * ldp; i += NT) sm[i] = 0.0;
  const int64_t nslab = slab1;
  const int slab2 = SLAB * ldx / 2;                     // double2 elements of a slab (contiguous in X)
  int lrow[PF], lcol[PF];
#pragma unroll
  for (int u = 0; u < PF; ++u) {
    const int i = 2 * (tid + u * NT);
    lrow[u] = i / ldx;
    lcol[u] = i - lrow[u] * ldx;
  }
  double2 v[PF];
  auto gload = [&](int64_t slab) {
    const int64_t r0 = slab * SLAB;
    const int64_t rows = nx - r0 < SLAB ? nx - r0 : SLAB;
    const int lim = (int)(rows * ldx / 2);
    const double2* __restrict__ src = (const double2*)(X + r0 * ldx);
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int i2 = tid + u * NT;
      v[u] = i2 < lim ? src[i2] : make_double2(0.0, 0.0);
    }
  };
  auto lstore = [&](double* buf) {
#pragma unroll
    for (int u = 0; u < PF; ++u)
      if (tid + u * NT < slab2) *(double2*)(buf + lrow[u] * ldp + lcol[u]) = v[u];
  };
  int64_t slab = slab0 + blockIdx.x;
  __syncthreads();                                      // zero fill done
  if (slab < nslab) { gload(slab); lstore(sm); }
  __syncthreads();
  int cur = 0;
  for (; slab < nslab; slab += gridDim.x) {
    const int64_t next = slab + gridDim.x;
    if (next < nslab) gload(next);
    const double* buf = sm + cur * SLAB * ldp;
#pragma unroll 1
    for (int kq = 0; kq < SLAB / 4; ++kq) {
      const double* rowp = buf + (4 * kq + ak) * ldp + ai;
      // consecutive tiles of a wave mostly share their row of the tile table (row-major upper triangle): the A
      // fragment is read again only when it changes (uniform compare) -- 12 -> ~7.5 LDS reads per 6 MFMAs
      double a = 0.0;
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        if (live[t]) {
          if (TPW < 5 || t == 0 || off_i[t] != off_i[t - 1]) a = rowp[off_i[t]];   // (short lists: the compare costs more than it saves -- 318 -> 565 us at N = 130)
          const double b = rowp[off_j[t]];
          acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
        }
      }
    }
    if (next < nslab) lstore(sm + (cur ^ 1) * SLAB * ldp);
    __syncthreads();
    cur ^= 1;
  }
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    if (live[t]) {
      const int tix = (blockIdx.y * NW + wv) * TPW + t;
      double* p = partial + ((size_t)blockIdx.x * ntri + tix) * 256;
#pragma unroll
      for (int r = 0; r < 4; ++r) p[r * 64 + lane] = acc[t][r];
    }
  }
}

__global__ __launch_bounds__(1024) void k_gram_reduce(const double* __restrict__ partial, int nblocks, int ntri,
                                                      const int32_t* __restrict__ tiles, int Nx,
                                                      double* __restrict__ G, double* __restrict__ G_host) {
  // block = 64 accumulator lanes x 16 strided groups of workgroup partials; fixed summation
  // order (group-strided, then groups 0..15) -> bit-reproducible
  __shared__ double red[16][64];
  const int tix = blockIdx.x >> 2, r = blockIdx.x & 3;
  const int lane = threadIdx.x, g = threadIdx.y;
  double s = 0.0;
  for (int b = g; b < nblocks; b += 16) s += partial[((size_t)b * ntri + tix) * 256 + r * 64 + lane];
  red[g][lane] = s;
  __syncthreads();
  if (g == 0) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][lane];
    const int packed = tiles[tix];
    const int i = (packed >> 16) * 16 + (lane >> 4) + 4 * r;
    const int j = (packed & 0xffff) * 16 + (lane & 15);
    if (i < Nx && j < Nx) {
      G[(size_t)i * Nx + j] = t;
      G[(size_t)j * Nx + i] = t;
      if (G_host) {                                         // pinned copy for the eigen-solver on the host: no copy engine in between
        G_host[(size_t)i * Nx + j] = t;
        G_host[(size_t)j * Nx + i] = t;
      }
    }
  }
}

// ======================================================================== local null
// Fused |X.Yc| -> threshold bin -> per-permutation histogram; the cells x permutations matrix of
// _association.py:99 never exists.  grid = (row chunks, ceil(P/PT)), PT = 16*NS permutations.
//
// The block's strip of Yc (4*KQ x PT doubles) sits in LDS for the whole kernel; every wave walks
// its own 16-cell tiles of the row chunk (stride = waves per block), loads the A operand straight
// from global memory in MFMA layout (lane (i,k): X[r0+i][4q+k]; the tile of the next iteration is
// prefetched into a second register set), runs NS independent accumulator chains over it
// (v_mfma_f64_16x16x4_f64, 64 cycles each) and bins its own outputs.  There is no barrier in the
// main loop: waves drift apart freely.  (A first version staged 64-cell slabs of X through LDS
// between two barriers with Yc in registers; all eight waves then alternated between MFMA and
// epilogue in lockstep.  Measured at 200k x 50 x 1000: 590 us -> 5xx us, see DESIGN.md.)
//
// Epilogue.  The reference bins z^2 = (|x.yc|/N)^2 against edges[t] (_stats.py:47-54).  Both
// roundings are monotone in |x.yc|, so the host converts every edge into the smallest double
// cut[t] with fl(fl(cut/N)^2) >= edges[t]; counting cut[t] <= |acc| is then *exactly* the
// reference's count.  With c_s[k] = cuts[k-1] the count of an output is h = #{k in 1..T : c_s[k] <= x};
// the cuts are an arithmetic progression to within `eps` steps (the host measures the worst
// deviation), so g = (x - cut0)/step + OFF has h = floor(g) - OFF + 1 unless g lies within eps of an
// integer.  u = trunc(g * 65536) (one fma, one saturating convert) gives both: h from u >> 16 and
// the fraction u & 0xffff, whose distance from 0 / 65536 is compared with an integer margin
// E >= eps * 65536 + 2; only outputs inside the margin (all of them when the progression test
// failed: lim = 0) walk the exact table.  On this chip VALU work does not hide under f64 MFMAs
// (ablation: arithmetic alone +130 us on a 366 us MFMA pipeline), so the epilogue is priced per
// instruction: 32-bit LDS counters (address = one shift-add, increment = 1), nine VALU
// instructions per output.  Each block stores its counters as one plain slab; k_hist_reduce sums
// the slabs (integers: bit-reproducible) -- global atomics cost 60 us here.
#ifndef CNA_PAIR_A
#define CNA_PAIR_A 1
#endif
// LDS row of sample (k index) `k` of a B operand whose A operand is loaded in 16-byte pairs: inside every
// block of 8 samples the even ones come first (rows 0-3: MFMA step 2q', lanes ak = 0..3 hold samples
// 8q' + 2ak) and the odd ones second (rows 4-7: step 2q' + 1); an unpaired last step keeps its rows
template <int KQ>
__device__ __forceinline__ int pair_row(int k) {
#if CNA_PAIR_A
  if (k < 8 * (KQ / 2)) return (k & ~7) + 4 * (k & 1) + ((k >> 1) & 3);
#endif
  return k;
}

template <int KQ, int NS, int LOOPM, int NW, int MODE = 0>      // LOOPM: 0 one tile at a time, 1 with A prefetch, 2 two tiles share every B fragment
__global__ __launch_bounds__(64 * NW) void k_null(const double* __restrict__ X, int64_t nx, int64_t chunk_rows,
                                              const double* __restrict__ Yc, int ldy, int P,
                                              const double* __restrict__ cuts, int T, double cut0,
                                              double inv_step, double eps, unsigned int* __restrict__ partial,
                                              int pt0, const int* __restrict__ guard) {
  extern __shared__ double sm[];
  if (guard && *guard == 0) return;                             // stand-by launch behind the integer path (null_i8.hip)
  constexpr int PT = 16 * NS, LDB = PT + 16, LDX = 4 * KQ;   // LDB = 16 mod 32: the two k-rows of a 32-lane group hit disjoint banks
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ak = lane >> 4, aj = lane & 15;
  const int TP = (T + 4) & ~1;                                // 0, cuts[0..T), +inf, +inf (even count)
  const int TW = T + 1;                                       // counter row: [0] unused, [h] = bin h-1; odd/even either way fine
  double* c_s = sm;
  double* bs = sm + TP;                                       // LDX x LDB
  unsigned int* hist = (unsigned int*)(bs + LDX * LDB);       // PT x TW words
  const int pt = pt0 + blockIdx.y;
  for (int i = tid; i < TP; i += 64 * NW) c_s[i] = i == 0 ? 0.0 : (i <= T ? cuts[i - 1] : __builtin_inf());
  for (int i = tid; i < PT * TW; i += 64 * NW) hist[i] = 0u;
  for (int i = tid; i < LDX * PT; i += 64 * NW) {
    const int k = i / PT, j = i - k * PT;
    bs[pair_row<KQ>(k) * LDB + j] = Yc[(size_t)k * ldy + pt * PT + j];
  }
  __syncthreads();

  const int64_t row_begin = (int64_t)blockIdx.x * chunk_rows;
  int64_t row_end = row_begin + chunk_rows;
  if (row_end > nx) row_end = nx;
  const int64_t ntile = row_begin < row_end ? (row_end - row_begin + 15) / 16 : 0;
  const double* bp = bs + ak * LDB + aj;
  unsigned int* hrow = hist + aj * TW;                        // this lane's permutation within strip 0

  auto load_a = [&](double (&a)[KQ], int64_t tile) {
    const int64_t row = row_begin + 16 * tile + aj;
    if (tile < ntile && row < row_end) {
#if CNA_PAIR_A
      // 16-byte loads: lane (row, ak) takes columns 8q' + 2ak, 8q' + 2ak + 1 -- the k index of MFMA steps 2q'
      // and 2q' + 1 (the Yc strip sits in LDS with its rows permuted to match, pair_row) -- so that one
      // load instruction covers 64 contiguous bytes of each of the 16 rows instead of 32
      const double* __restrict__ xp = X + row * LDX + 2 * ak;
#pragma unroll
      for (int q2 = 0; q2 < KQ / 2; ++q2) {
        const double2 v = *(const double2*)(xp + 8 * q2);
        a[2 * q2] = v.x;
        a[2 * q2 + 1] = v.y;
      }
      if (KQ & 1) a[KQ - 1] = X[row * LDX + 4 * (KQ - 1) + ak];
#else
      const double* __restrict__ xp = X + row * LDX + ak;
#pragma unroll
      for (int q = 0; q < KQ; ++q) a[q] = xp[4 * q];
#endif
    } else {
#pragma unroll
      for (int q = 0; q < KQ; ++q) a[q] = 0.0;               // zero rows never reach cut0 > 0
    }
  };
  unsigned sink = 0;
  // g is offset by OFF >= cut0/step + 2 so that outputs below the first cut keep a meaningful
  // fraction instead of saturating at 0: they are rejected on the fast path too.
  const double ratio = cut0 * inv_step;
  const bool linear = eps < 0.25 && ratio < 60000.0;
  const int OFF = linear ? (int)ratio + 2 : 1;
  const double K1 = inv_step * 65536.0, K0 = ((double)OFF - ratio) * 65536.0;
  const unsigned E = linear ? (unsigned)(eps * 65536.0) + 3u : 65536u;
  const unsigned lim = linear ? 65536u - 2u * E : 0u;
  // one strip (4 outputs per lane) at a time: independent instructions for the scheduler without
  // holding all NS*4 intermediate values in registers; the exact walk is one rarely taken region
  auto count_strip = [&](const v4d& acc, unsigned int* row) {
    unsigned u[4];
    int h[4];
    bool near = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) u[i] = (unsigned)__builtin_fma(fabs(acc[i]), K1, K0);   // v_cvt_u32_f64 (saturating)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int hi = (int)(u[i] >> 16) - (OFF - 1);                  // floor((x - cut0)/step) + 1
      h[i] = hi < T ? hi : T;
      near |= !(((u[i] & 0xffffu) - E) < lim);
    }
    if (__builtin_expect(near, 0)) {                                  // some output within the margin of a cut
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (!(((u[i] & 0xffffu) - E) < lim)) {
          const double x = fabs(acc[i]);
          int hh = h[i] > 0 ? h[i] : 0;
          while (c_s[hh + 1] <= x) ++hh;                              // c_s[T+1] = +inf
          while (c_s[hh] > x) --hh;                                   // c_s[0] = 0
          h[i] = hh;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (h[i] > 0) {
        if (MODE == 2) sink += (unsigned)h[i];                        // experiment: arithmetic only
        else atomicAdd(&row[h[i]], 1u);
      }
    }
  };
  auto count_all = [&](const v4d (&acc)[NS]) {
#pragma unroll
    for (int s = 0; s < NS; ++s) count_strip(acc[s], hrow + 16 * s * TW);
  };
  auto tile_body = [&](const double (&a)[KQ]) {
    v4d acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = (v4d){0.0, 0.0, 0.0, 0.0};
#if CNA_NULL_PD > 0
    // B fragments are read from LDS CNA_NULL_PD k-steps ahead of the MFMA that consumes them (left to
    // itself the compiler issues each ds_read one MFMA before its use and waits on it: 64 cycles of cover for
    // an LDS round trip of ~100+, i.e. a stall on every second MFMA at N = 200)
    constexpr int PD = CNA_NULL_PD < KQ ? CNA_NULL_PD : KQ;
    double bq[PD][NS];
#pragma unroll
    for (int q = 0; q < PD; ++q)
#pragma unroll
      for (int s = 0; s < NS; ++s) bq[q][s] = bp[4 * q * LDB + 16 * s];
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      double bc[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) bc[s] = bq[q % PD][s];
      if (q + PD < KQ) {
#pragma unroll
        for (int s = 0; s < NS; ++s) bq[q % PD][s] = bp[4 * (q + PD) * LDB + 16 * s];
      }
#pragma unroll
      for (int s = 0; s < NS; ++s) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], bc[s], acc[s], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#else
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
#pragma unroll
      for (int s = 0; s < NS; ++s)
        acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], bp[4 * q * LDB + 16 * s], acc[s], 0, 0, 0);
    }
#endif
    if (MODE == 1) {                                                     // experiment: no counting
      double z = 0.0;
#pragma unroll
      for (int s = 0; s < NS; ++s) z += acc[s][0] + acc[s][1] + acc[s][2] + acc[s][3];
      if (z == 123.456) atomicAdd(&hist[0], 1u);
      return;
    }
    count_all(acc);
  };
  if (LOOPM == 2) {
    // Two 16-cell tiles per wave and iteration: every B fragment read from LDS feeds 2 x NS MFMAs.  On this
    // chip the LDS -> VGPR return of an operand is not hidden under f64 MFMAs (tools/micro/mfma_f64_rate.hip:
    // 78 TFLOP/s from registers, 56 with one ds_read_b64 per MFMA at 4 waves x 4 chains per SIMD), so bytes
    // of B per MFMA are what count: 512 -> 256.  Deep tiles take their A operands in two k halves (the
    // accumulators persist) to stay within the register budget of two waves per SIMD.
    constexpr int NPH = KQ > 28 ? 2 : 1;
    constexpr int KH = (((KQ + NPH - 1) / NPH) + 1) & ~1;
    auto load_part = [&](double (&a)[KH], int64_t tile, int ph) {
      const int64_t row = row_begin + 16 * tile + aj;
      const bool ok = tile < ntile && row < row_end;
      const double* __restrict__ xr = X + (ok ? row : row_begin) * LDX;
#pragma unroll
      for (int j = 0; j < KH; j += 2) {
        const int q = ph * KH + j;
        if (q + 1 < 2 * (KQ / 2) + 0 && q < 2 * (KQ / 2)) {
          const double2 v = *(const double2*)(xr + 8 * (q >> 1) + 2 * ak);
          a[j] = ok ? v.x : 0.0;
          a[j + 1] = ok ? v.y : 0.0;
        } else if (q == KQ - 1) {
          a[j] = ok ? xr[4 * q + ak] : 0.0;
          a[j + 1] = 0.0;
        } else {
          a[j] = 0.0;
          a[j + 1] = 0.0;
        }
      }
    };
    for (int64_t t = 2 * (int64_t)wv; t < ntile; t += 2 * NW) {
      v4d acc0[NS], acc1[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) { acc0[s] = (v4d){0.0, 0.0, 0.0, 0.0}; acc1[s] = (v4d){0.0, 0.0, 0.0, 0.0}; }
#pragma unroll
      for (int ph = 0; ph < NPH; ++ph) {
        double a0[KH], a1[KH];
        load_part(a0, t, ph);
        load_part(a1, t + 1, ph);
#pragma unroll
        for (int j = 0; j < KH; ++j) {
          const int q = ph * KH + j;
          if (q < KQ) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
              const double b = bp[4 * q * LDB + 16 * s];
              acc0[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[j], b, acc0[s], 0, 0, 0);
              acc1[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[j], b, acc1[s], 0, 0, 0);
            }
          }
        }
      }
      count_all(acc0);
      count_all(acc1);
    }
  } else if (LOOPM == 1) {
    double a0[KQ], a1[KQ];
    load_a(a0, wv);
    for (int64_t t = wv; t < ntile; t += 2 * NW) {
      load_a(a1, t + NW);
      tile_body(a0);
      if (t + NW >= ntile) break;
      load_a(a0, t + 2 * NW);
      tile_body(a1);
    }
  } else {
    double a0[KQ];
    for (int64_t t = wv; t < ntile; t += NW) {
      load_a(a0, t);
      tile_body(a0);
    }
  }
  if (MODE == 2 && sink == 0x12345u) hist[0] = 1u;
  __syncthreads();
  // slab of this block: partial[chunk][p][t], t contiguous
  unsigned int* out = partial + ((size_t)blockIdx.x * P + (size_t)pt * PT) * T;
  for (int i = tid; i < PT * T; i += 64 * NW) {
    const int pl = i / T, t = i - pl * T;
    if (pt * PT + pl < P) out[(size_t)pl * T + t] = hist[pl * TW + t + 1];
  }
}

// More than 256 samples: the k-depth is a run-time loop, the A operand comes from memory at every step
// (two 16-cell tiles per wave share each B fragment), one strip of 16 permutations per block.  Same
// counting rule as k_null (exact cuts; the linear guess with the table walk inside the margin).  A
// fallback for large cohorts, not a tuned kernel: ~0.1-0.2 of the MFMA peak.
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_null_big(const double* __restrict__ X, int64_t nx, int64_t chunk_rows, int ldx,
                                                      const double* __restrict__ Yc, int ldy, int P,
                                                      const double* __restrict__ cuts, int T, double cut0,
                                                      double inv_step, double eps, unsigned int* __restrict__ partial,
                                                      int LDB, const int* __restrict__ guard) {
  extern __shared__ double sm[];
  if (guard && *guard == 0) return;
  constexpr int PT = 16;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ak = lane >> 4, aj = lane & 15;
  const int TP = (T + 4) & ~1, TW = T + 1;
  double* c_s = sm;
  double* bs = sm + TP;                                       // ldx x LDB
  unsigned int* hist = (unsigned int*)(bs + (size_t)ldx * LDB);
  const int pt = blockIdx.y;
  for (int i = tid; i < TP; i += 64 * NW) c_s[i] = i == 0 ? 0.0 : (i <= T ? cuts[i - 1] : __builtin_inf());
  for (int i = tid; i < PT * TW; i += 64 * NW) hist[i] = 0u;
  for (int i = tid; i < ldx * PT; i += 64 * NW) {
    const int k = i / PT, j = i - k * PT;
    bs[k * LDB + j] = Yc[(size_t)k * ldy + pt * PT + j];
  }
  __syncthreads();
  const int64_t row_begin = (int64_t)blockIdx.x * chunk_rows;
  int64_t row_end = row_begin + chunk_rows;
  if (row_end > nx) row_end = nx;
  const int64_t ntile = row_begin < row_end ? (row_end - row_begin + 15) / 16 : 0;
  const double* bp = bs + ak * LDB + aj;
  unsigned int* hrow = hist + aj * TW;
  const double ratio = cut0 * inv_step;
  const bool linear = eps < 0.25 && ratio < 60000.0;
  const int OFF = linear ? (int)ratio + 2 : 1;
  const double K1 = inv_step * 65536.0, K0 = ((double)OFF - ratio) * 65536.0;
  const unsigned E = linear ? (unsigned)(eps * 65536.0) + 3u : 65536u;
  const unsigned lim = linear ? 65536u - 2u * E : 0u;
  auto count = [&](const v4d& acc) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const double x = fabs(acc[i]);
      const unsigned u = (unsigned)__builtin_fma(x, K1, K0);
      int h = (int)(u >> 16) - (OFF - 1);
      h = h < T ? h : T;
      if (!(((u & 0xffffu) - E) < lim)) {
        int hh = h > 0 ? h : 0;
        while (c_s[hh + 1] <= x) ++hh;
        while (c_s[hh] > x) --hh;
        h = hh;
      }
      if (h > 0) atomicAdd(&hrow[h], 1u);
    }
  };
  const int kq = ldx / 4;
  for (int64_t t = 2 * (int64_t)wv; t < ntile; t += 2 * NW) {
    const int64_t row0 = row_begin + 16 * t + aj, row1 = row0 + 16;
    const bool ok0 = row0 < row_end, ok1 = row1 < row_end;
    const double* __restrict__ x0 = X + (ok0 ? row0 : row_begin) * ldx + ak;
    const double* __restrict__ x1 = X + (ok1 ? row1 : row_begin) * ldx + ak;
    v4d acc0 = (v4d){0.0, 0.0, 0.0, 0.0}, acc1 = (v4d){0.0, 0.0, 0.0, 0.0};
    int q = 0;
    for (; q + 4 <= kq; q += 4) {
      double a0[4], a1[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { a0[u] = x0[4 * (q + u)]; a1[u] = x1[4 * (q + u)]; b[u] = bp[(size_t)4 * (q + u) * LDB]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ok0 ? a0[u] : 0.0, b[u], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ok1 ? a1[u] : 0.0, b[u], acc1, 0, 0, 0);
      }
    }
    for (; q < kq; ++q) {
      const double b = bp[(size_t)4 * q * LDB];
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ok0 ? x0[4 * q] : 0.0, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ok1 ? x1[4 * q] : 0.0, b, acc1, 0, 0, 0);
    }
    count(acc0);                                              // zero rows never reach cut0 > 0
    count(acc1);
  }
  __syncthreads();
  unsigned int* out = partial + ((size_t)blockIdx.x * P + (size_t)pt * PT) * T;
  for (int i = tid; i < PT * T; i += 64 * NW) {
    const int pl = i / T, tt = i - pl * T;
    if (pt * PT + pl < P) out[(size_t)pl * T + tt] = hist[pl * TW + tt + 1];
  }
}

// hist[p][t] = sum over row chunks of the per-block slabs (fixed order, integers)
__global__ void k_hist_reduce(const unsigned int* __restrict__ partial, int nchunks, int64_t PT_total,
                              unsigned long long* __restrict__ hist, const int* __restrict__ guard) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= PT_total) return;
  if (guard && *guard == 0) return;
  unsigned long long s = 0;
  for (int c = 0; c < nchunks; ++c) s += partial[(size_t)c * PT_total + i];
  hist[i] = s;
}

template <int TPW, int NW = 8, int SLAB = 32>
int launch_gram_t(cna_ctx* c, int nt, int ldp, int ntri, const int32_t* tiles_dev, double* partial, int nblocks,
                  size_t smem, int64_t slab0, int64_t slab1, int accumulate, hipStream_t st) {
  const int npass = (ntri + NW * TPW - 1) / (NW * TPW);
  static bool attr_set = false;
  if (!attr_set) {
    HIP_TRY(hipFuncSetAttribute((const void*)k_gram<TPW, NW, SLAB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  // two slabs in LDS when they fit (up to ~256 samples): loads of the next slab overlap the MFMAs
  if (SLAB == 32 && 2 * smem <= 150 * 1024 && c->ldx <= 320) {
    static bool attr_db = false;
    if (!attr_db) {
      HIP_TRY(hipFuncSetAttribute((const void*)k_gram_db<TPW, NW, SLAB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_db = true;
    }
    hipLaunchKernelGGL((k_gram_db<TPW, NW, SLAB>), dim3(nblocks, npass), dim3(64 * NW), 2 * smem, st, c->X, c->nx, c->ldx,
                       nt, ldp, ntri, tiles_dev, partial, slab0, slab1, accumulate);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL((k_gram<TPW, NW, SLAB>), dim3(nblocks, npass), dim3(64 * NW), smem, st, c->X, c->nx, c->ldx, nt, ldp,
                     ntri, tiles_dev, partial, slab0, slab1, accumulate);
  HIP_TRY(hipGetLastError());
  return 0;
}

typedef int (*null_launch_fn)(cna_ctx*, dim3, size_t, int64_t, const double*, int, int, const double*, int, double,
                              double, double, unsigned int*, const int*);
template <int KQ, int NS>
int launch_null_t(cna_ctx* c, dim3 grid, size_t smem, int64_t chunk_rows, const double* Yc, int ldy, int P,
                  const double* cuts, int T, double cut0, double inv_step, double eps, unsigned int* partial,
                  const int* guard) {
  // 16 waves per block (4 per SIMD) while a wave fits 128 VGPRs, i.e. up to N = 128, 8 waves beyond;
  // no second A register set: four waves per SIMD hide the A loads better than a software prefetch
  // (-6 % at N = 100, -9 % at N = 60, -1 % at N = 128 against 8 waves with it; -2 % at N = 50 against
  // 16 waves with it)
#ifndef CNA_NULL_TWO
#define CNA_NULL_TWO 0   // measured: 54.4 vs 55.3 TFLOP/s at N = 200, 49.1 vs 52.3 at N = 100, 38.2 vs 42.6 at N = 50 (the compiler waits on every shared fragment right before its four MFMAs)
#endif
  // loop shape: two tiles per wave (every B fragment feeds 2 x NS MFMAs) -- measured against one tile per
  // wave in tools/kbench_null.py; A prefetch (shape 1) never paid (53.6 vs 55.3 TFLOP/s at N = 200)
  constexpr int PF = CNA_NULL_TWO ? 2 : 0;
#ifndef CNA_NULL_NW_DEEP
#define CNA_NULL_NW_DEEP 8
#endif
  constexpr int NW = PF == 2 ? 8 : (KQ <= 32 ? 16 : CNA_NULL_NW_DEEP);
  static bool attr_set = false;
  if (!attr_set) {
    HIP_TRY(hipFuncSetAttribute((const void*)k_null<KQ, NS, PF, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  // two launches over the permutation tiles: with ~2 workgroup rounds in total, the boundary lets the
  // short high-priority kernels of the second stream (global F-tests) in half-way instead of after
  // the whole kernel
  const unsigned half = (grid.y + 1) / 2;
  for (unsigned y0 = 0; y0 < grid.y; y0 += half) {
    const unsigned ny = grid.y - y0 < half ? grid.y - y0 : half;
    hipLaunchKernelGGL((k_null<KQ, NS, PF, NW>), dim3(grid.x, ny), dim3(64 * NW), smem, c->stream, c->X, c->nx, chunk_rows,
                       Yc, ldy, P, cuts, T, cut0, inv_step, eps, partial, (int)y0, guard);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}
// one instantiation per exact k-depth (ceil(N/4), N <= 256) so the MFMA loop has no guards
template <int NS, int... KQ>
constexpr std::array<null_launch_fn, sizeof...(KQ)> null_table(std::integer_sequence<int, KQ...>) {
  return {{&launch_null_t<KQ + 1, NS>...}};
}
const auto kNullNS4 = null_table<4>(std::make_integer_sequence<int, 40>{});   // KQ 1..40
const auto kNullNS2 = null_table<2>(std::make_integer_sequence<int, 64>{});   // KQ 1..64
const auto kNullNS1 = null_table<1>(std::make_integer_sequence<int, 64>{});   // KQ 1..64 (many thresholds)

}  // namespace

namespace {
typedef int (*xb_launch_fn)(cna_ctx*, unsigned, size_t, const double*, int, int, double*, int, int, const double*, int);
template <int KQ, int NS>
int launch_xb_t(cna_ctx* c, unsigned grid, size_t smem, const double* B_dev, int ldb, int center, double* out, int ld_out,
                int k_off, const double* means, int accumulate) {
  static bool attr_set = false;
  if (!attr_set) {
    HIP_TRY(hipFuncSetAttribute((const void*)k_xb<KQ, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((k_xb<KQ, NS>), dim3(grid), dim3(512), smem, c->stream, c->X, c->nx, c->Nx, B_dev, ldb, center, out, ld_out,
                     c->ldx, k_off, means, accumulate);
  HIP_TRY(hipGetLastError());
  return 0;
}
template <int NS, int... KQ>
constexpr std::array<xb_launch_fn, sizeof...(KQ)> xb_table(std::integer_sequence<int, KQ...>) {
  return {{&launch_xb_t<KQ + 1, NS>...}};
}
const auto kXbNS4 = xb_table<4>(std::make_integer_sequence<int, 32>{});   // KQ 1..32
typedef int (*xbres_launch_fn)(cna_ctx*, unsigned, size_t, const double*, int, int, double*, int, int64_t);
template <int KQ>
int launch_xbres_t(cna_ctx* c, unsigned grid, size_t smem, const double* B_dev, int ldb, int center, double* out,
                   int ld_out, int64_t ntile) {
  static bool attr_set = false;
  if (!attr_set) {
    HIP_TRY(hipFuncSetAttribute((const void*)k_xb_res<KQ>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((k_xb_res<KQ>), dim3(grid), dim3(1024), smem, c->stream, c->X, c->nx, c->Nx, B_dev, ldb, center, out,
                     ld_out, ntile);
  HIP_TRY(hipGetLastError());
  return 0;
}
template <int... KQ>
constexpr std::array<xbres_launch_fn, sizeof...(KQ)> xbres_table(std::integer_sequence<int, KQ...>) {
  return {{&launch_xbres_t<KQ + 1>...}};
}
const auto kXbRes = xbres_table(std::make_integer_sequence<int, 32>{});     // KQ 1..32
const auto kXbNS2 = xb_table<2>(std::make_integer_sequence<int, 64>{});   // KQ 1..64
}  // namespace

int launch_xb(cna_ctx* c, const double* B_dev, int ldb, int n_out, bool center, double* out, int ld_out) {
  (void)n_out;
  if (c->nx == 0) return 0;
  const int kq = c->ldx / 4;
  if (kq < 1 || kq > 256) CNA_FAIL(CNA_EINVAL, "more than 1024 samples are not supported");
  ProfScope ps(c, out == c->X ? CNA_K_RESID : CNA_K_PROJECT);
  const int64_t ntile = (c->nx + 15) / 16;
  if (kq > 64) {
    // k-split: parts of up to 64 quads (256 samples) accumulated into a buffer other than X
    double* dst = out;
    if (out == c->X) {
      void* p2 = c->X2;
      CNA_TRY(dev_reserve(c, &p2, &c->x2_cap, (int64_t)sizeof(double) * std::max<int64_t>(c->nx, 1) * ld_out));
      c->X2 = (double*)p2;
      dst = c->X2;
    }
    const double* means = nullptr;
    if (center) {
      CNA_TRY(dev_reserve(c, &c->scratch2, &c->scratch2_cap, (int64_t)sizeof(double) * c->nx));
      hipLaunchKernelGGL(k_row_means, dim3((unsigned)((c->nx + 3) / 4)), dim3(256), 0, c->stream, c->X, c->nx, c->Nx, c->ldx,
                         (double*)c->scratch2);
      means = (const double*)c->scratch2;
    }
    const unsigned grid = (unsigned)((ntile + 7) / 8);
    for (int q0 = 0; q0 < kq; q0 += 64) {
      const int kqp = kq - q0 < 64 ? kq - q0 : 64;
      const size_t smem = sizeof(double) * (size_t)(4 * kqp) * (16 * 2 + 16);
      CNA_TRY(kXbNS2[kqp - 1](c, grid, smem, B_dev + (size_t)(4 * q0) * ldb, ldb, 0, dst, ld_out, 4 * q0, means, q0 > 0));
    }
    if (out == c->X) {                        // in-place request: the result replaces X
      std::swap(c->X, c->X2);
      std::swap(c->x_cap, c->x2_cap);
    }
    return 0;
  }
  const int NS = kq <= 32 ? 4 : 2;
  {
    const int nct = ((ldb + 63) / 64) * 4;
    const size_t whole = sizeof(double) * (size_t)c->ldx * (16 * nct + 16);
    if (kq <= 32 && whole <= 160 * 1024 && ntile >= 64) {   // all of B resident in LDS
      const int64_t want = (ntile + 15) / 16;
      return kXbRes[kq - 1](c, (unsigned)(want < 512 ? want : 512), whole, B_dev, ldb, center ? 1 : 0, out, ld_out, ntile);
    }
  }
  const size_t smem = sizeof(double) * (size_t)c->ldx * (16 * NS + 16);
  const unsigned grid = (unsigned)((ntile + 7) / 8);
  xb_launch_fn fn = NS == 4 ? kXbNS4[kq - 1] : kXbNS2[kq - 1];
  return fn(c, grid, smem, B_dev, ldb, center ? 1 : 0, out, ld_out, 0, nullptr, 0);
}

// selgram != nullptr: the fused selection + Gram kernel (k_selgram_blk) instead of the Gram kernel on X -- the caller
// has made sure the selection is "all cells, samples in place, no projector" and that gram_fused_ok() holds.
struct SelGramArgs {
  unsigned long long* nzero; const double* y; unsigned long long* maxbits; unsigned char* xq; void* xscale; int Kp;
};
bool gram_fused_ok(const cna_ctx* c, int Nx, int ldx, int Kp) {
  const int nt = (Nx + 15) / 16, ng = (nt + 2) / 3;
  const int ldp = 16 * nt + ((nt & 1) ? 0 : 16);
  // OFF by default: measured at 2M x 200 the fused kernel takes 5.05 ms against 1.63 + 1.75 ms for the two kernels.
  // The standardisation is ~700 vector instructions per wave and slab (sixteen f64 divisions per lane expand to ~35
  // instructions each) and on this chip nothing overlaps an f64 MFMA on the same SIMD: the vector work of the selection
  // pass (about a millisecond of whole-chip issue) ADDS to the 1.75 ms of matrix work instead of hiding under it, and
  // with 18 accumulator tiles per wave the standardisation runs out of spilled registers.  Kept for the parity test and
  // as the measured answer to "fuse selection and Gram" (DESIGN 8); CNA_SELGRAM=1 selects it (read per call).
  const bool on = getenv("CNA_SELGRAM") != nullptr;
  const int cols = ldx > Kp ? ldx : Kp;
  const size_t lds = sizeof(double) * ((size_t)32 * ldp + (size_t)32 * c->ld + 64 * ((cols + 63) / 64) + 8);
  return on && nt >= 11 && ng * (ng + 1) / 2 <= 16 && ldx <= 320 && (c->ld & 1) == 0 && cols <= 256 &&
         lds <= 158 * 1024 && c->nx >= 32;
}
// The product in three parts, so that it can also be taken range by range behind the pass that writes X
// (c_api.hip:cna_nam_step, the walk's last step in row ranges): gram_plan fixes shapes, tile tables and the buffer of
// per-workgroup partial tiles; launch_gram_range queues the kernel for the slabs [slab0, slab1) of X -- workgroup b
// takes slabs slab0 + b, slab0 + b + nblocks, ...; a range that starts at a multiple of nblocks and continues from the
// partial tiles of the ranges before it (accumulate) therefore adds the same products in the same order as ONE launch
// over all slabs, and the result is bit-identical to it; launch_gram_finish sums the partial tiles in a fixed order.
static int launch_gram_impl(cna_ctx* c, double* G_dev, const SelGramArgs* selgram);
int launch_gram(cna_ctx* c, double* G_dev) { return launch_gram_impl(c, G_dev, nullptr); }
int launch_selgram(cna_ctx* c, double* G_dev, unsigned long long* nzero, const double* y, unsigned long long* maxbits,
                   unsigned char* xq, void* xscale, int Kp) {
  const SelGramArgs a{nzero, y, maxbits, xq, xscale, Kp};
  return launch_gram_impl(c, G_dev, &a);
}

// own_partial: the partial tiles live in c->gram_part (kept apart from c->scratch2, which other entry points reuse
// while a ranged product is still under way)
static int gram_plan(cna_ctx* c, GramPlan& g, bool own_partial) {
  const int Nx = c->Nx;
  g.Nx = Nx;
  const int nt = g.nt = (Nx + 15) / 16;
  const int ntri = g.ntri = nt * (nt + 1) / 2;
  const int ldp = g.ldp = 16 * nt + ((nt & 1) ? 0 : 16);
  // tile table (ti<<16 | tj), upper triangle, row-major
  std::vector<int32_t> tiles;
  for (int i = 0; i < nt; ++i)
    for (int j = i; j < nt; ++j) tiles.push_back((i << 16) | j);
  // 32-cell slabs of X in LDS (32 x ldp doubles) up to 512 samples, 16-cell slabs up to 1024; beyond 256
  // samples the upper-triangular tiles exceed one pass of 16 waves x 9 tiles and grid.y walks the passes
  const int slab_rows = g.slab_rows = (size_t)32 * ldp * sizeof(double) <= 150 * 1024 ? 32 : 16;
  if ((size_t)slab_rows * ldp * sizeof(double) > 160 * 1024) CNA_FAIL(CNA_EINVAL, "more than 1024 samples are not supported");
  const int64_t nslab = g.nslab = (c->nx + slab_rows - 1) / slab_rows;
  int nblocks = (int)(nslab < 512 ? nslab : 512);
  const int64_t blocks_1g = ((int64_t)1 << 30) / ((int64_t)ntri * 2048);      // per-block partial tiles: keep the slab under 1 GiB
  if (nblocks > blocks_1g) nblocks = (int)(blocks_1g > 1 ? blocks_1g : 1);
  // 3 x 3 blocks of tiles, one per wave (k_gram_blk), when the triangle has at most 16 of them: 11 ... 15 tiles per side
  // (161 ... 240 samples) on sixteen waves; round 6: also 6 ... 10 tiles per side (81 ... 160 samples) on as many waves as
  // there are blocks (4 / 8 / 12: 0.67-1 LDS reads per matrix instruction where the tile-per-wave kernel needs 1.25 --
  // 453 -> 342 us at 1M x 100, 278 -> 226 at 500k x 128, 423 -> 332 at 500k x 160; at 5 tiles per side the three blocks
  // leave a SIMD idle and the old kernel stays, 119 vs 136 us at 500k x 80: profiles/r06_kbench_gram.txt), two workgroups
  // per CU where their slabs fit
  const int ng = (nt + 2) / 3;
  const int nblk = ng * (ng + 1) / 2;
  const char* bs = getenv("CNA_GRAM_BLK_SMALL");             // test hook (read per call): 0 = the tile-per-wave kernel
  const bool blk_small = !(bs && atoi(bs) == 0);
  const bool use_blk = g.use_blk = (nt >= 11 || (nt >= 6 && blk_small)) && nblk <= 16 && slab_rows == 32 &&
                                   2 * (size_t)32 * ldp * sizeof(double) <= 150 * 1024 && c->ldx <= 320;
  g.nw = !use_blk ? 16 : (nt >= 11 ? 16 : (nblk <= 4 ? 4 : (nblk <= 8 ? 8 : 12)));
  const int wg_per_cu = use_blk && 2 * (2 * (size_t)32 * ldp * sizeof(double)) <= 150 * 1024 && g.nw <= 8 ? 2 : 1;
  if (use_blk && nblocks > 256 * wg_per_cu) nblocks = 256 * wg_per_cu;   // a workgroup (or two) per CU, every slab after the first prefetched
  if (nblocks < 1) nblocks = 1;
  g.nblocks = nblocks;
  const int64_t part_bytes = (int64_t)sizeof(double) * nblocks * ntri * 256;
  if (own_partial) {
    void* pp = c->gram_part;
    CNA_TRY(dev_reserve(c, &pp, &c->gram_part_cap, part_bytes));
    c->gram_part = pp;
    g.partial = (double*)pp;
  } else {
    CNA_TRY(dev_reserve(c, &c->scratch2, &c->scratch2_cap, part_bytes));
    g.partial = (double*)c->scratch2;
  }
  if (c->gram_tiles_nt != 2 * nt + (use_blk ? 1 : 0)) {   // the tables depend on nt (and the kernel family) only: upload once
    std::vector<int32_t> extra(128 + (size_t)nt * nt, -1);   // [16 waves x 8] block table | tixmap
    if (use_blk) {
      struct Blk { int rg, cg, work; };
      std::vector<Blk> bl;
      for (int rg = 0; rg < ng; ++rg)
        for (int cg = rg; cg < ng; ++cg) {
          int w = 0;
          for (int r = 0; r < 3; ++r)
            for (int q = 0; q < 3; ++q) {
              const int i = 3 * rg + r, j = 3 * cg + q;
              w += i < nt && j < nt && j >= i;
            }
          bl.push_back({rg, cg, w});
        }
      std::stable_sort(bl.begin(), bl.end(), [](const Blk& a, const Blk& b) { return a.work > b.work; });
      for (size_t p = 0; p < bl.size(); ++p) {               // snake over the four SIMDs (waves w, w+4, w+8, w+12 share one)
        const int round = (int)p / 4, pos = (int)p % 4;
        const int wave = 4 * round + ((round & 1) ? 3 - pos : pos);
        if (wave >= g.nw) continue;                          // (never: nw is the block count rounded up to whole rounds)
        for (int r = 0; r < 3; ++r) {
          const int i = 3 * bl[p].rg + r, j = 3 * bl[p].cg + r;
          extra[wave * 8 + r] = i < nt ? i : -1;
          extra[wave * 8 + 3 + r] = j < nt ? j : -1;
        }
      }
      for (int t = 0; t < ntri; ++t) extra[128 + (tiles[t] >> 16) * nt + (tiles[t] & 0xffff)] = t;
    }
    void* tp = c->gram_tiles_ptr;
    CNA_TRY(dev_reserve(c, &tp, &c->gram_tiles_cap, (int64_t)sizeof(int32_t) * (ntri + extra.size())));
    c->gram_tiles_ptr = tp;
    HIP_TRY(hipMemcpyAsync(tp, tiles.data(), sizeof(int32_t) * ntri, hipMemcpyHostToDevice, c->copy_stream));
    HIP_TRY(hipMemcpyAsync((int32_t*)tp + ntri, extra.data(), sizeof(int32_t) * extra.size(), hipMemcpyHostToDevice, c->copy_stream));
    HIP_TRY(hipStreamSynchronize(c->copy_stream));       // the vectors go out of scope (the copy stream: the main one may be busy with a walk)
    c->gram_tiles_nt = 2 * nt + (use_blk ? 1 : 0);
  }
  g.tiles_dev = (int32_t*)c->gram_tiles_ptr;
  g.smem = sizeof(double) * slab_rows * ldp;
  return 0;
}

static int launch_gram_range(cna_ctx* c, const GramPlan& g, int64_t slab0, int64_t slab1, int accumulate, hipStream_t st) {
  if (slab1 <= slab0) return 0;
  const int nt = g.nt, ldp = g.ldp, ntri = g.ntri, nblocks = g.nblocks;
  ProfScope ps(c, CNA_K_GRAM, st);
  if (g.use_blk) {
#define GRAM_BLK(NW_, PF_) do { \
      static bool attr = false; \
      if (!attr) { \
        HIP_TRY(hipFuncSetAttribute((const void*)k_gram_blk<NW_, 32, false, PF_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        HIP_TRY(hipFuncSetAttribute((const void*)k_gram_blk<NW_, 32, true, PF_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        attr = true; \
      } \
      if (accumulate) \
        hipLaunchKernelGGL((k_gram_blk<NW_, 32, true, PF_>), dim3(nblocks), dim3(64 * NW_), 2 * g.smem, st, c->X, c->nx, c->ldx, nt, ldp, \
                           ntri, g.tiles_dev + ntri, g.tiles_dev + ntri + 128, g.partial, slab0, slab1, 1); \
      else \
        hipLaunchKernelGGL((k_gram_blk<NW_, 32, false, PF_>), dim3(nblocks), dim3(64 * NW_), 2 * g.smem, st, c->X, c->nx, c->ldx, nt, ldp, \
                           ntri, g.tiles_dev + ntri, g.tiles_dev + ntri + 128, g.partial, slab0, slab1, 0); \
    } while (0)
    // (PF double2 loads per thread must cover a slab: 64 NW x PF x 2 >= 32 ldx -- 4 waves: ldx <= 96, 8: <= 160, 12: <= 240)
    switch (g.nw) {
      case 4: GRAM_BLK(4, 6); break;
      case 8: GRAM_BLK(8, 5); break;
      case 12: GRAM_BLK(12, 5); break;
      default: GRAM_BLK(16, 5); break;
    }
#undef GRAM_BLK
    HIP_TRY(hipGetLastError());
    return 0;
  }
  int r;
  // 16 waves per workgroup, the tiles dealt ceil(ntri/16) to a wave: against 8 waves with up to 12
  // tiles each (154 VGPRs, one workgroup of two waves per SIMD per CU) 2025 -> 1489 us at 1M x 200,
  // 3958 -> 2034 us at 1M x 256, 57 -> 45 us at 200k x 50; faster at every N tried (20 ... 256)
#define GRAM_T(...) launch_gram_t<__VA_ARGS__>(c, nt, ldp, ntri, g.tiles_dev, g.partial, nblocks, g.smem, slab0, slab1, accumulate, st)
  switch ((ntri + 15) / 16) {
    case 1: r = GRAM_T(1, 16); break;
    case 2: r = GRAM_T(2, 16); break;
    case 3: r = GRAM_T(3, 16); break;
    case 4: r = GRAM_T(4, 16); break;
    case 5: r = GRAM_T(5, 16); break;
    case 6: r = GRAM_T(6, 16); break;
    case 7: r = GRAM_T(7, 16); break;
    case 8: r = GRAM_T(8, 16); break;
    default: r = g.slab_rows == 32 ? GRAM_T(9, 16)   // 144 tiles per pass
                                   : GRAM_T(9, 16, 16);
             break;
  }
#undef GRAM_T
  return r;
}

static int launch_gram_finish(cna_ctx* c, const GramPlan& g, double* G_dev, hipStream_t st) {
  ProfScope ps(c, CNA_K_GRAM_REDUCE, st);
  hipLaunchKernelGGL(k_gram_reduce, dim3((unsigned)g.ntri * 4), dim3(64, 16), 0, st, g.partial, g.nblocks, g.ntri,
                     g.tiles_dev, g.Nx, G_dev, c->gram_mirror);
  HIP_TRY(hipGetLastError());
  if (c->gram_mirror) c->gram_mirrored = true;
  return 0;
}

// ---- the product behind a pass that writes X range by range (c_api.hip:cna_nam_step): opaque handle for c_api.hip
int gram_pre_begin(cna_ctx* c, int64_t* unit_rows) {
  CNA_TRY(gram_plan(c, c->gram_pre_plan, true));
  *unit_rows = (int64_t)c->gram_pre_plan.nblocks * c->gram_pre_plan.slab_rows;
  return 0;
}
int gram_pre_range(cna_ctx* c, int64_t row0, int64_t row1, hipStream_t st) {
  const GramPlan& g = c->gram_pre_plan;
  const int64_t s0 = row0 / g.slab_rows, s1 = (row1 + g.slab_rows - 1) / g.slab_rows;
  return launch_gram_range(c, g, s0, s1 < g.nslab ? s1 : g.nslab, row0 > 0 ? 1 : 0, st);
}
int gram_pre_finish(cna_ctx* c, double* G_dev, hipStream_t st) { return launch_gram_finish(c, c->gram_pre_plan, G_dev, st); }

static int launch_gram_impl(cna_ctx* c, double* G_dev, const SelGramArgs* selgram) {
  const int Nx = c->Nx;
  HIP_TRY(hipMemsetAsync(G_dev, 0, sizeof(double) * Nx * Nx, c->stream));
  if (c->nx == 0) return 0;
  GramPlan g;
  CNA_TRY(gram_plan(c, g, false));
  if (selgram) {
    const int nt = g.nt, ldp = g.ldp, ntri = g.ntri, nblocks = g.nblocks;
    int32_t* tiles_dev = g.tiles_dev;
    double* partial = g.partial;
    if (!g.use_blk) CNA_FAIL(CNA_ESTATE, "launch_selgram: shape outside the fused kernel's range");
    HIP_TRY(hipMemsetAsync(selgram->nzero, 0, sizeof(unsigned long long), c->stream));
    if (selgram->maxbits) HIP_TRY(hipMemsetAsync(selgram->maxbits, 0, sizeof(unsigned long long), c->stream));
    const int cols = c->ldx > selgram->Kp ? c->ldx : selgram->Kp;
    const int G = (cols + 63) / 64;
    const size_t lds = sizeof(double) * ((size_t)32 * ldp + (size_t)32 * c->ld + 64 * G + 8);
    {
      ProfScope ps(c, CNA_K_GRAM);
#define SG_LAUNCH(GG) { static bool once = false; if (!once) { HIP_TRY(hipFuncSetAttribute((const void*)k_selgram_blk<GG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); once = true; } \
      hipLaunchKernelGGL(k_selgram_blk<GG>, dim3(nblocks), dim3(512), lds, c->stream, c->nam, c->ld, c->X, c->nx, c->Nx, c->ldx, selgram->nzero, selgram->y, c->ncorrs, selgram->maxbits ? selgram->maxbits + 1 : nullptr, selgram->xq, (double2*)selgram->xscale, selgram->Kp, nt, ldp, ntri, tiles_dev + ntri, tiles_dev + ntri + 128, partial); }
      switch (G) {
        case 3: SG_LAUNCH(3) break;
        default: SG_LAUNCH(4) break;
      }
#undef SG_LAUNCH
      HIP_TRY(hipGetLastError());
    }
    if (selgram->y) launch_max_fold(c, selgram->maxbits + 1, nblocks, selgram->maxbits);
    return launch_gram_finish(c, g, G_dev, c->stream);
  }
  CNA_TRY(launch_gram_range(c, g, 0, g.nslab, 0, c->stream));
  return launch_gram_finish(c, g, G_dev, c->stream);
}

int launch_null_local(cna_ctx* c, const double* Yc_dev, int ldy, int P, const double* cuts_dev, int T,
                      double cut0, double inv_step, double eps, unsigned long long* hist_dev, const int* guard) {
  if (c->nx == 0 || P == 0 || T == 0) {
    HIP_TRY(hipMemsetAsync(hist_dev, 0, sizeof(unsigned long long) * (size_t)P * T, c->stream));
    return 0;
  }
  const int kq = c->ldx / 4;
  if (!(cut0 > 0.0)) CNA_FAIL(CNA_EINVAL, "local-null kernel needs strictly positive thresholds");
  if (kq > 64) {
    // generic depth (more than 256 samples): k_null_big, 16 permutations per block
    const size_t cap = 160 * 1024;
    auto lds_big = [&](int ldb) {
      return sizeof(double) * ((T + 4) & ~1) + sizeof(double) * (size_t)c->ldx * ldb + sizeof(unsigned int) * (size_t)16 * (T + 1);
    };
    const int LDB = lds_big(32) <= cap ? 32 : 16;
    if (lds_big(LDB) > cap) CNA_FAIL(CNA_EINVAL, "local-null kernel: thresholds/samples exceed LDS (more than 1024 samples?)");
    const int nptile = (P + 15) / 16;
    const int64_t ntile = (c->nx + 15) / 16;
    int64_t nchunks = (512 + nptile - 1) / nptile;
    if (nchunks > (ntile + 31) / 32) nchunks = (ntile + 31) / 32;
    if (nchunks < 1) nchunks = 1;
    const int64_t chunk_rows = ((ntile + nchunks - 1) / nchunks) * 16;
    nchunks = (c->nx + chunk_rows - 1) / chunk_rows;
    void* part = c->null_part;
    CNA_TRY(dev_reserve(c, &part, &c->null_part_cap, (int64_t)sizeof(unsigned int) * nchunks * P * T));
    c->null_part = part;
    ProfScope ps(c, guard ? -1 : CNA_K_NULL_LOCAL);     // stand-by behind the integer path: that one is timed
    static bool attr_set = false;
    if (!attr_set) {
      HIP_TRY(hipFuncSetAttribute((const void*)k_null_big<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_set = true;
    }
    hipLaunchKernelGGL(k_null_big<8>, dim3((unsigned)nchunks, (unsigned)nptile), dim3(512), lds_big(LDB), c->stream, c->X, c->nx,
                       chunk_rows, c->ldx, Yc_dev, ldy, P, cuts_dev, T, cut0, inv_step, eps, (unsigned int*)c->null_part, LDB, guard);
    const int64_t tot = (int64_t)P * T;
    hipLaunchKernelGGL(k_hist_reduce, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream,
                       (const unsigned int*)c->null_part, (int)nchunks, tot, hist_dev, guard);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  // LDS: cuts, the Yc strip (4kq x (PT+16) doubles), PT x (T+1) 32-bit counters
  auto lds = [&](int ns) {
    return sizeof(double) * ((T + 4) & ~1) + sizeof(double) * (size_t)c->ldx * (16 * ns + 16) +
           sizeof(unsigned int) * (size_t)16 * ns * (T + 1);
  };
  const size_t cap = 160 * 1024;
  const int NS = (kq <= 40 && lds(4) <= cap) ? 4 : (lds(2) <= cap ? 2 : 1);
  if (lds(NS) > cap) CNA_FAIL(CNA_EINVAL, "local-null kernel: thresholds/samples exceed LDS");
  const int nptile = (P + 16 * NS - 1) / (16 * NS);
  const int64_t ntile = (c->nx + 15) / 16;
  // one block per CU fits (LDS); two rounds of blocks over the 256 CUs even out the tail
  int64_t nchunks = (512 + nptile - 1) / nptile;
  if (nchunks > (ntile + 15) / 16) nchunks = (ntile + 15) / 16;
  if (nchunks < 1) nchunks = 1;
  const int64_t chunk_rows = ((ntile + nchunks - 1) / nchunks) * 16;
  nchunks = (c->nx + chunk_rows - 1) / chunk_rows;
  void* part = c->null_part;
  CNA_TRY(dev_reserve(c, &part, &c->null_part_cap, (int64_t)sizeof(unsigned int) * nchunks * P * T));
  c->null_part = part;
  dim3 grid((unsigned)nchunks, (unsigned)nptile);
  ProfScope ps(c, guard ? -1 : CNA_K_NULL_LOCAL);     // stand-by behind the integer path: that one is timed
  null_launch_fn fn = NS == 4 ? kNullNS4[kq - 1] : (NS == 2 ? kNullNS2[kq - 1] : kNullNS1[kq - 1]);
  CNA_TRY(fn(c, grid, lds(NS), chunk_rows, Yc_dev, ldy, P, cuts_dev, T, cut0, inv_step, eps, (unsigned int*)c->null_part, guard));
  const int64_t tot = (int64_t)P * T;
  hipLaunchKernelGGL(k_hist_reduce, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream,
                     (const unsigned int*)c->null_part, (int)nchunks, tot, hist_dev, guard);
  HIP_TRY(hipGetLastError());
  return 0;
}
