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
#include <array>
#include <cstdlib>
#include <utility>

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

// ======================================================================== OUT = X . B
// One wave owns a 16-cell tile: its rows are staged (and optionally centred) in a wave
// private LDS panel, padded to ldp = ldx + 2 doubles so the A-fragment ds_read_b64 of 16
// different rows is bank-conflict free.  B (K x ldb, ldb % 16 == 0, zero padded) is small
// and L2 resident; its fragments are read straight from global as 128-byte row segments.
__global__ __launch_bounds__(256) void k_xb(const double* __restrict__ X, int64_t nx, int Nx, int ldx,
                                            const double* __restrict__ B, int ldb, int center,
                                            double* __restrict__ out, int ld_out) {
  extern __shared__ double sm[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ldp = ldx + 2;
  double* xp = sm + (size_t)wv * 16 * ldp;
  const int64_t r0 = ((int64_t)blockIdx.x * 4 + wv) * 16;
  for (int r = 0; r < 16; ++r) {
    const int64_t gr = r0 + r;
    for (int col = lane; col < ldx; col += 64) xp[r * ldp + col] = (gr < nx) ? X[gr * ldx + col] : 0.0;
  }
  __syncthreads();
  if (center) {
    for (int r = 0; r < 16; ++r) {
      double s = 0.0;
      for (int col = lane; col < Nx; col += 64) s += xp[r * ldp + col];
      const double mean = wave_sum(s) / (double)Nx;
      for (int col = lane; col < Nx; col += 64) xp[r * ldp + col] -= mean;
    }
  }
  __syncthreads();
  const int kq = ldx >> 2;
  const int ai = lane & 15, ak = lane >> 4;
  const int ntile = ldb >> 4;
  for (int jt = 0; jt < ntile; ++jt) {
    v4d acc = {0.0, 0.0, 0.0, 0.0};
    const double* bp = B + (size_t)ak * ldb + jt * 16 + ai;
    const double* ap = xp + ai * ldp + ak;
#pragma unroll 4
    for (int q = 0; q < kq; ++q) {
      const double a = ap[4 * q];
      const double b = bp[(size_t)4 * q * ldb];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    const int col = jt * 16 + ai;
    if (col < ld_out) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t gr = r0 + ak + 4 * r;
        if (gr < nx) out[gr * ld_out + col] = acc[r];
      }
    }
  }
}

// ======================================================================== G = X^T X
// 8 waves per workgroup; the nt*(nt+1)/2 upper-triangular 16x16 tiles of G are dealt to the
// waves (TPW accumulator tiles each, kept in registers for the whole kernel); the workgroup
// streams 32-cell slabs of X through LDS and every wave feeds its tiles with 8 k-steps of 4
// cells.  Per-workgroup partial tiles are written out and summed in a fixed order by
// k_gram_reduce (deterministic; no float atomics).
template <int TPW>
__global__ __launch_bounds__(512) void k_gram(const double* __restrict__ X, int64_t nx, int ldx, int nt,
                                              int ldp, int ntri, const int32_t* __restrict__ tiles,
                                              double* __restrict__ partial) {
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
    const int tix = (blockIdx.y * 8 + wv) * TPW + t;
    live[t] = tix < ntri;
    const int packed = live[t] ? __builtin_amdgcn_readfirstlane(tiles[tix]) : 0;
    off_i[t] = (packed >> 16) * 16;
    off_j[t] = (packed & 0xffff) * 16;
  }
  for (int i = tid; i < 32 * ldp; i += 512) sm[i] = 0.0;
  const int64_t nslab = (nx + 31) / 32;
  for (int64_t slab = blockIdx.x; slab < nslab; slab += gridDim.x) {
    __syncthreads();
    const int64_t r0 = slab * 32;
    for (int r = wv; r < 32; r += 8) {
      const int64_t gr = r0 + r;
      for (int col = lane; col < ldx; col += 64) sm[r * ldp + col] = (gr < nx) ? X[gr * ldx + col] : 0.0;
    }
    __syncthreads();
#pragma unroll 1
    for (int kq = 0; kq < 8; ++kq) {
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
      const int tix = (blockIdx.y * 8 + wv) * TPW + t;
      double* p = partial + ((size_t)blockIdx.x * ntri + tix) * 256;
#pragma unroll
      for (int r = 0; r < 4; ++r) p[r * 64 + lane] = acc[t][r];
    }
  }
}

__global__ __launch_bounds__(1024) void k_gram_reduce(const double* __restrict__ partial, int nblocks, int ntri,
                                                      const int32_t* __restrict__ tiles, int Nx,
                                                      double* __restrict__ G) {
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
    }
  }
}

// ======================================================================== local null
// grid = (row chunks, ceil(P/64)); 8 waves.  Wave w owns permutations 64*pt + 16*(w&3) .. +15 and
// keeps that 16-column strip of Yc in registers (KQ doubles per lane) for the whole kernel;
// waves 0-3 take the first half of each slab of cells, waves 4-7 the second half, TS 16-cell
// tiles each (independent accumulators, so the 64-cycle f64 MFMAs of one tile hide the
// dependent-issue latency of the other).  Slabs of 32*TS cells stream through LDS with a
// register prefetch of the next slab under the MFMAs.
//
// Epilogue.  The reference bins z^2 = (|x.yc|/N)^2 against edges[t] (_stats.py:47-54).  Both
// roundings are monotone in |x.yc|, so the host converts every edge into the smallest double
// cut[t] with fl(fl(cut/N)^2) >= edges[t]; counting cut[t] <= |acc| is then *exactly* the
// reference's count and costs one compare instead of a division and a square per element.
// A linear guess + two exact compares finds the bin; counters are packed 16-bit pairs in LDS
// (a chunk has < 65536 cells) and are flushed with integer global atomics -> bit-reproducible.
template <int KQ, int TS, int MODE = 0>
__global__ __launch_bounds__(512) void k_null(const double* __restrict__ X, int64_t nx, int64_t chunk_rows,
                                              const double* __restrict__ Yc, int ldy, int P,
                                              const double* __restrict__ cuts, int T, double cut0,
                                              double inv_step, double eps, unsigned long long* __restrict__ ghist) {
  // KQ is the exact number of k-steps: the leading dimension of X is 4*KQ
  extern __shared__ double sm[];
  constexpr int ROWS = 32 * TS;
  constexpr int LDX = 4 * KQ, LDP = LDX + 2;
  constexpr int ND2 = ROWS * LDX / 2;                    // double2 elements per slab
  constexpr int PF = (ND2 + 511) / 512;                  // double2 prefetch registers per thread
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int strip = wv & 3, half = wv >> 2;
  const int ak = lane >> 4, ai = lane & 15;
  const int HW = ((T + 1) >> 1) | 1;                     // odd row stride: the 16 strips' counters spread over all banks
  const int TP = (T + 4) & ~1;                           // 0, cuts[0..T), +inf, +inf (even count)
  double* c_s = sm;
  double* xt = sm + TP;                                  // ROWS * LDP doubles
  unsigned int* hist = (unsigned int*)(xt + ROWS * LDP);  // 64 * HW words
  for (int i = tid; i < TP; i += 512) c_s[i] = i == 0 ? 0.0 : (i <= T ? cuts[i - 1] : __builtin_inf());
  for (int i = tid; i < 64 * HW; i += 512) hist[i] = 0u;

  const int pt = blockIdx.y;
  double b[KQ];
  {
    const double* bp = Yc + (size_t)ak * ldy + pt * 64 + strip * 16 + ai;
#pragma unroll
    for (int q = 0; q < KQ; ++q) b[q] = bp[(size_t)4 * q * ldy];
  }
  // prefetch slots: slab-relative element offset (global) and padded offset (LDS) of each double2
  unsigned goff[PF], loff[PF];
#pragma unroll
  for (int i = 0; i < PF; ++i) {
    const unsigned f = tid + 512u * i;
    const unsigned e = 2u * f;
    const unsigned r = e / LDX;
    goff[i] = (f < (unsigned)ND2) ? e : 0xffffffffu;
    loff[i] = r * LDP + (e - r * LDX);
  }
  const int64_t row_begin = (int64_t)blockIdx.x * chunk_rows;
  int64_t row_end = row_begin + chunk_rows;
  if (row_end > nx) row_end = nx;

  // register prefetch two slabs ahead (pfA: even slabs, pfB: odd): one slab of MFMAs is shorter
  // than an L2/Infinity-Cache round trip under load
  double2 pfA[PF], pfB[PF];
  auto prefetch = [&](double2 (&pf)[PF], int64_t r0) {
    const double* __restrict__ slab = X + r0 * LDX;      // uniform base, 32-bit lane offsets
    const int64_t left = r0 < row_end ? (row_end - r0) * LDX : 0;
    const unsigned lim = left > (int64_t)(ROWS * LDX) ? (unsigned)(ROWS * LDX) : (unsigned)left;
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      pf[i] = make_double2(0.0, 0.0);
      if (goff[i] < lim) pf[i] = *reinterpret_cast<const double2*>(slab + goff[i]);
    }
  };
  const unsigned hp = (unsigned)((strip * 16 + ai) * HW);   // this lane's counter row (word index)
  // Output -> counter.  c_s[k] (k>=1) = cuts[k-1], c_s[0] = 0, c_s[T+1..] = +inf; the count of an
  // output is h = #{k in 1..T : c_s[k] <= x}.  The cuts are an arithmetic progression to within
  // `eps` steps (the host measures the worst deviation), so h = floor((x-cut0)/step) + 1 is exact
  // unless x lies within eps of a cut -- only those outputs (and the last bin) consult the table.
  // Rows past row_end were staged as zeros and cut0 > 0, so they never count.
  auto count = [&](double v) {
    const double x = fabs(v);
    if (x >= cut0) {
      const double f = (x - cut0) * inv_step;
      const double fl = floor(f);
      const double fr = f - fl;
      int h = (fl < (double)(T - 1)) ? (int)fl + 1 : T;
      if (!(fl < (double)(T - 1) && fr > eps && fr < 1.0 - eps)) {   // near a cut or in the top bin: exact walk
        while (c_s[h + 1] <= x) ++h;                                  // c_s[T+1] = +inf
        while (c_s[h] > x) --h;                                       // c_s[1] = cut0 <= x
      }
      const unsigned bin = (unsigned)(h - 1);
      atomicAdd(&hist[hp + (bin >> 1)], (bin & 1u) ? 0x10000u : 1u);
    }
  };
  auto count_all = [&](const v4d (&acc)[TS]) {
#pragma unroll
    for (int t = 0; t < TS; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) count(acc[t][r]);
    }
  };
  const double* ap = xt + (half * 16 * TS + ai) * LDP + ak;
  auto stage = [&](double2 (&pf)[PF], int64_t r0) {
    __syncthreads();                       // previous slab fully consumed
#pragma unroll
    for (int i = 0; i < PF; ++i)
      if (goff[i] != 0xffffffffu) *reinterpret_cast<double2*>(xt + loff[i]) = pf[i];
    __syncthreads();
    prefetch(pf, r0 + 2 * ROWS);           // this register set is free again: fetch two slabs ahead
  };
  auto slab = [&]() {
    v4d acc[TS];
#pragma unroll
    for (int t = 0; t < TS; ++t) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
#pragma unroll
      for (int t = 0; t < TS; ++t) {
        if (MODE == 2) {
          if (q < 2) acc[t][q] += ap[t * 16 * LDP + 4 * q] * b[q];       // experiment: no matrix work
        } else {
          acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[t * 16 * LDP + 4 * q], b[q], acc[t], 0, 0, 0);
        }
      }
    }
    if (MODE == 1) {                                                     // experiment: no counting
      double z = 0.0;
#pragma unroll
      for (int t = 0; t < TS; ++t) z += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
      if (z == 123.456) atomicAdd(&hist[hp], 1u);
    } else {
      count_all(acc);
    }
  };
  if (row_begin < row_end) {
    prefetch(pfA, row_begin);
    prefetch(pfB, row_begin + ROWS);
    for (int64_t r0 = row_begin;;) {
      stage(pfA, r0);
      slab();
      r0 += ROWS;
      if (r0 >= row_end) break;
      stage(pfB, r0);
      slab();
      r0 += ROWS;
      if (r0 >= row_end) break;
    }
  }
  __syncthreads();
  for (int i = tid; i < 64 * HW; i += 512) {
    const unsigned int w = hist[i];
    if (w) {
      const int pl = i / HW, hw = i - pl * HW;
      const int p = pt * 64 + pl;
      if (p < P) {
        const unsigned int lo = w & 0xffffu, hi = w >> 16;
        if (lo) atomicAdd(&ghist[(size_t)p * T + 2 * hw], (unsigned long long)lo);
        if (hi && 2 * hw + 1 < T) atomicAdd(&ghist[(size_t)p * T + 2 * hw + 1], (unsigned long long)hi);
      }
    }
  }
}

template <int TPW>
int launch_gram_t(cna_ctx* c, int nt, int ldp, int ntri, const int32_t* tiles_dev, double* partial, int nblocks,
                  size_t smem) {
  const int npass = (ntri + 8 * TPW - 1) / (8 * TPW);
  hipLaunchKernelGGL((k_gram<TPW>), dim3(nblocks, npass), dim3(512), smem, c->stream, c->X, c->nx, c->ldx, nt, ldp,
                     ntri, tiles_dev, partial);
  HIP_TRY(hipGetLastError());
  return 0;
}

template <int KQ, int TS>
int launch_null_t(cna_ctx* c, dim3 grid, size_t smem, int64_t chunk_rows, const double* Yc, int ldy, int P,
                  const double* cuts, int T, double cut0, double inv_step, double eps, unsigned long long* hist) {
  static bool attr_set = false;
  if (!attr_set) {
    HIP_TRY(hipFuncSetAttribute((const void*)k_null<KQ, TS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((k_null<KQ, TS>), grid, dim3(512), smem, c->stream, c->X, c->nx, chunk_rows, Yc, ldy, P, cuts,
                     T, cut0, inv_step, eps, hist);
  HIP_TRY(hipGetLastError());
  return 0;
}

// one instantiation per exact k-depth (ceil(N/4), N <= 256) so the MFMA loop has no guards
typedef int (*null_launch_fn)(cna_ctx*, dim3, size_t, int64_t, const double*, int, int, const double*, int, double,
                              double, double, unsigned long long*);
template <int TS, int... KQ>
constexpr std::array<null_launch_fn, sizeof...(KQ)> null_table(std::integer_sequence<int, KQ...>) {
  return {{&launch_null_t<KQ + 1, TS>...}};
}
const auto kNullTS2 = null_table<2>(std::make_integer_sequence<int, 54>{});   // KQ 1..54
const auto kNullTS1 = null_table<1>(std::make_integer_sequence<int, 64>{});   // KQ 1..64

}  // namespace

int launch_xb(cna_ctx* c, const double* B_dev, int ldb, int n_out, bool center, double* out, int ld_out) {
  (void)n_out;
  if (c->nx == 0) return 0;
  ProfScope ps(c, out == c->X ? CNA_K_RESID : CNA_K_PROJECT);
  const size_t smem = sizeof(double) * 4 * 16 * (c->ldx + 2);
  if (smem > 160 * 1024) CNA_FAIL(CNA_EINVAL, "too many samples for the residualisation kernel (max ~300)");
  static bool attr_set = false;
  if (!attr_set) {
    HIP_TRY(hipFuncSetAttribute((const void*)k_xb, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const int64_t ntile = (c->nx + 15) / 16;
  const unsigned grid = (unsigned)((ntile + 3) / 4);
  hipLaunchKernelGGL(k_xb, dim3(grid), dim3(256), smem, c->stream, c->X, c->nx, c->Nx, c->ldx, B_dev, ldb,
                     center ? 1 : 0, out, ld_out);
  HIP_TRY(hipGetLastError());
  return 0;
}

int launch_gram(cna_ctx* c, double* G_dev) {
  const int Nx = c->Nx;
  const int nt = (Nx + 15) / 16;
  const int ntri = nt * (nt + 1) / 2;
  const int ldp = 16 * nt + ((nt & 1) ? 0 : 16);
  const int tpw = (ntri + 7) / 8;
  HIP_TRY(hipMemsetAsync(G_dev, 0, sizeof(double) * Nx * Nx, c->stream));
  if (c->nx == 0) return 0;
  // tile table (ti<<16 | tj), upper triangle, row-major
  std::vector<int32_t> tiles;
  for (int i = 0; i < nt; ++i)
    for (int j = i; j < nt; ++j) tiles.push_back((i << 16) | j);
  const int64_t nslab = (c->nx + 31) / 32;
  const int nblocks = (int)(nslab < 512 ? nslab : 512);
  CNA_TRY(dev_reserve(c, &c->scratch2, &c->scratch2_cap, (int64_t)sizeof(double) * nblocks * ntri * 256));
  double* partial = (double*)c->scratch2;
  if (c->gram_tiles_nt != nt) {                          // the tile table depends on nt only: upload once
    void* tp = c->gram_tiles_ptr;
    CNA_TRY(dev_reserve(c, &tp, &c->gram_tiles_cap, (int64_t)sizeof(int32_t) * ntri));
    c->gram_tiles_ptr = tp;
    HIP_TRY(hipMemcpyAsync(tp, tiles.data(), sizeof(int32_t) * ntri, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));            // tiles vector goes out of scope
    c->gram_tiles_nt = nt;
  }
  int32_t* tiles_dev = (int32_t*)c->gram_tiles_ptr;
  const size_t smem = sizeof(double) * 32 * ldp;
  {
    ProfScope ps(c, CNA_K_GRAM);
    int r;
    if (tpw <= 1) r = launch_gram_t<1>(c, nt, ldp, ntri, tiles_dev, partial, nblocks, smem);
    else if (tpw <= 2) r = launch_gram_t<2>(c, nt, ldp, ntri, tiles_dev, partial, nblocks, smem);
    else if (tpw <= 4) r = launch_gram_t<4>(c, nt, ldp, ntri, tiles_dev, partial, nblocks, smem);
    else if (tpw <= 8) r = launch_gram_t<8>(c, nt, ldp, ntri, tiles_dev, partial, nblocks, smem);
    else r = launch_gram_t<12>(c, nt, ldp, ntri, tiles_dev, partial, nblocks, smem);   // extra passes beyond 96 tiles
    CNA_TRY(r);
  }
  {
    ProfScope ps(c, CNA_K_GRAM_REDUCE);
    hipLaunchKernelGGL(k_gram_reduce, dim3((unsigned)ntri * 4), dim3(64, 16), 0, c->stream, partial, nblocks, ntri,
                       tiles_dev, Nx, G_dev);
    HIP_TRY(hipGetLastError());
  }
  return 0;
}

int launch_null_local(cna_ctx* c, const double* Yc_dev, int ldy, int P, const double* cuts_dev, int T,
                      double cut0, double inv_step, double eps, unsigned long long* hist_dev) {
  HIP_TRY(hipMemsetAsync(hist_dev, 0, sizeof(unsigned long long) * (size_t)P * T, c->stream));
  if (c->nx == 0 || P == 0 || T == 0) return 0;
  const int kq = c->ldx / 4;
  if (kq > 64) CNA_FAIL(CNA_EINVAL, "more than 256 samples are not supported by the local-null kernel yet");
  if (!(cut0 > 0.0)) CNA_FAIL(CNA_EINVAL, "local-null kernel needs strictly positive thresholds");
  const int HW = ((T + 1) / 2) | 1;
  const size_t fixed = sizeof(double) * ((T + 4) & ~1) + sizeof(unsigned int) * 64 * HW;
  const size_t slab64 = sizeof(double) * 64 * (c->ldx + 2), slab32 = slab64 / 2;
  const int TS = (kq <= 54 && fixed + slab64 <= 150 * 1024) ? 2 : 1;
  const size_t smem = fixed + (TS == 2 ? slab64 : slab32);
  if (smem > 160 * 1024) CNA_FAIL(CNA_EINVAL, "local-null kernel: thresholds/samples exceed LDS");
  const int ROWS = 32 * TS;
  const int nptile = (P + 63) / 64;
  const int64_t nslab = (c->nx + ROWS - 1) / ROWS;
  int64_t nchunks = (1024 + nptile - 1) / nptile;
  if (nchunks > nslab) nchunks = nslab;
  int64_t chunk_rows = ((nslab + nchunks - 1) / nchunks) * ROWS;
  if (chunk_rows > 65472) chunk_rows = 65472;            // 16-bit packed counters; multiple of 64
  nchunks = (c->nx + chunk_rows - 1) / chunk_rows;
  dim3 grid((unsigned)nchunks, (unsigned)nptile);
  ProfScope ps(c, CNA_K_NULL_LOCAL);
  null_launch_fn fn = TS == 2 ? kNullTS2[kq - 1] : kNullTS1[kq - 1];
  if (const char* dbg = getenv("CNA_NULL_DEBUG")) {                      // experiments, N=50 only
    if (kq == 13 && TS == 2 && atoi(dbg) == 1) {
      hipLaunchKernelGGL((k_null<13, 2, 1>), grid, dim3(512), smem, c->stream, c->X, c->nx, chunk_rows, Yc_dev, ldy, P,
                         cuts_dev, T, cut0, inv_step, eps, hist_dev);
      return 0;
    }
    if (kq == 13 && TS == 2 && atoi(dbg) == 2) {
      hipLaunchKernelGGL((k_null<13, 2, 2>), grid, dim3(512), smem, c->stream, c->X, c->nx, chunk_rows, Yc_dev, ldy, P,
                         cuts_dev, T, cut0, inv_step, eps, hist_dev);
      return 0;
    }
  }
  return fn(c, grid, smem, chunk_rows, Yc_dev, ldy, P, cuts_dev, T, cut0, inv_step, eps, hist_dev);
}
