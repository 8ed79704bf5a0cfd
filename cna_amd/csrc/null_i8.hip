// Local null on the integer matrix cores (gfx950: v_mfma_i32_32x32x32_i8), exact.
//
// What is counted is the same as in mfma.hip:k_null -- for every kept cell x and every conditioned
// permuted phenotype yc, which of the exact cuts |x.yc| reaches (/root/reference/src/cna/tools/
// _association.py:98-103 with _stats.py:47-54, 64-83) -- but only the SUM over permutations of the
// per-threshold tail counts, which is all `empirical_fdrs` uses: fdr[t] = mean_p(tails[p][t] / ranks[t]).
// (Callers that ask for the per-permutation tails get the f64 kernel.)
//
// The f64 matrix instruction runs at 1/64 of the i8 one, and the counts are integers: they do not need
// the 53-bit products, they need every output on the right side of every cut.  So:
//
//   1. rows of X and columns of Yc become 24-bit fixed point, q = rint(x / s), |q| <= Q, one scale per row
//      of X (s_r = max|x_r| / Q) and one for Yc; q = d2*2^16 + d1*2^8 + d0 with balanced digits in
//      [-128, 127] (k_quant_x, k_quant_y);
//   2. qx.qy = 2^32 S2 + 2^24 S3 + 2^16 S4 + (two lower groups, left out), S2 = d2x.d2y, S3 = d2x.d1y +
//      d1x.d2y, S4 = d2x.d0y + d1x.d1y + d0x.d2y: six i8 matrix products, int32 accumulation, EXACT (at most
//      256 terms of at most 2^14 each);
//   3. an error bound that holds for every output of row r, in units of one threshold step:
//        m_r = s_r s_y / step * (sum|qx_r| / 2 + max_p sum|qy_p| / 2 + N (2*128*128*256 + 128*128 + 1/4))
//              + deviation of the cuts from an arithmetic progression + float rounding of the epilogue
//      (first two terms: rounding of the operands; third: the digit products left out and the product of the
//      two rounding errors);
//   4. every output is binned from the integer result; those whose position (|x.yc| - cut0) / step lies within m_r
//      of an integer -- a fraction of 2 m_r of those in range, 1e-4...1e-3 of all -- ALSO go to a queue, with the bin
//      they were counted in: k_null_recheck recomputes them in f64 against the exact table of cuts, takes that count
//      back and adds the exact one (its slabs are signed corrections, k_i8_reduce).
//
// Result: counts identical to the f64 kernel's (tests/test_gpu_parity.py::test_local_null_i8_*), at the
// price of 6 x 2nNP' integer operations on a pipe 64x as fast.  If the queue overflows (pathological
// thresholds) a status word is raised and the f64 kernel, launched behind with a guard on that word,
// does the work instead: no host round trip either way.
#include "common.h"
#include <algorithm>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define I8_Q I8_QMAX                        // largest |q| (common.h)
#define I8_DROP (2.0 * 128 * 128 * 256 + 128.0 * 128 + 0.25)
#define I8_QBUF 192

// q -> three balanced base-256 digits (each as a byte)
__device__ __forceinline__ void digits3(int q, unsigned& d0, unsigned& d1, unsigned& d2) {
  const int e0 = ((q + 128) & 255) - 128;
  const int q1 = (q - e0) >> 8;
  const int e1 = ((q1 + 128) & 255) - 128;
  const int e2 = (q1 - e1) >> 8;
  d0 = (unsigned)e0 & 255u;
  d1 = (unsigned)e1 & 255u;
  d2 = (unsigned)e2 & 255u;
}

// max |Yc| over the N x P block (positive doubles order like their bit patterns)
__global__ __launch_bounds__(1024) void k_y_absmax(const double* __restrict__ Y, int ldy, int N, int P,
                                                   unsigned long long* __restrict__ scal) {
  __shared__ double red[16];
  double m = 0.0;
  const int64_t tot = (int64_t)N * P;
  for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 1024) {
    const int k = (int)(i / P), p = (int)(i - (int64_t)k * P);
    m = fmax(m, fabs(Y[(size_t)k * ldy + p]));
  }
  m = wave_max_d(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w) m = fmax(m, red[w]);
    atomicMax(&scal[0], (unsigned long long)__double_as_longlong(m));
  }
}

// Yc (k rows, permutation columns) -> digit planes in the B-operand layout of the 32x32x32 instruction,
// [strip of 64 permutations][digit][k step of 32][k half][column][16 bytes], a transposed f64 copy for the
// recheck, and sum|q| per column.  One thread per (permutation, 16 samples).
__global__ __launch_bounds__(256) void k_quant_y(const double* __restrict__ Y, int ldy, int N, int P, int KS, int Ppad,
                                                 const unsigned long long* __restrict__ scal, v4i* __restrict__ Yq,
                                                 double* __restrict__ Yt, int ldt,
                                                 unsigned long long* __restrict__ colL1) {
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int p = (int)(id % Ppad), ch = (int)(id / Ppad);
  if (ch >= 2 * KS) return;
  const double ymax = __longlong_as_double((long long)scal[0]);
  const double inv = ymax > 0.0 ? I8_Q / ymax : 0.0;
  unsigned w[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  unsigned long long l1 = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int k = ch * 16 + j;
    const double y = (k < N && p < P) ? Y[(size_t)k * ldy + p] : 0.0;
    const int q = (int)rint(y * inv);
    l1 += (unsigned long long)(q < 0 ? -q : q);
    unsigned d0, d1, d2;
    digits3(q, d0, d1, d2);
    w[0][j >> 2] |= d0 << (8 * (j & 3));
    w[1][j >> 2] |= d1 << (8 * (j & 3));
    w[2][j >> 2] |= d2 << (8 * (j & 3));
    if (k < ldt) Yt[(size_t)p * ldt + k] = y;
  }
  const int strip = p >> 6, col = p & 63, s = ch >> 1, kh = ch & 1;
  v4i* out = Yq + (size_t)strip * (3 * KS * 128);
#pragma unroll
  for (int d = 0; d < 3; ++d)
    out[((d * KS + s) * 2 + kh) * 64 + col] = (v4i){(int)w[d][0], (int)w[d][1], (int)w[d][2], (int)w[d][3]};
  if (p < P && l1) atomicAdd(&colL1[p], l1);
}

__global__ __launch_bounds__(1024) void k_y_finish(const unsigned long long* __restrict__ colL1, int P,
                                                   unsigned long long* __restrict__ scal) {
  __shared__ unsigned long long red[16];
  unsigned long long m = 0;
  for (int p = threadIdx.x; p < P; p += 1024) m = colL1[p] > m ? colL1[p] : m;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long v = __shfl_xor(m, o);
    m = v > m ? v : m;
  }
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w) m = red[w] > m ? red[w] : m;
    scal[1] = m;
  }
}

// X (row per cell) -> digit planes [row][digit][32 KS] bytes and {max |x|, sum |q|} per row, for working matrices
// whose producer did not write them itself (rows.hip:k_select_std does).  One wave per row, 4 samples per lane.
__global__ __launch_bounds__(256) void k_quant_x(const double* __restrict__ X, int ldx, int64_t nx, int N, int Kp,
                                                 unsigned char* __restrict__ Xq, double2* __restrict__ xscale) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int k0 = 4 * lane;
  const bool act = k0 < Kp;
  for (int r = 0; r < 8; ++r) {
    const int64_t row = (int64_t)blockIdx.x * 32 + wv * 8 + r;
    if (row >= nx) return;
    double x[4] = {0.0, 0.0, 0.0, 0.0};
    if (act) {
      const double* xr = X + (size_t)row * ldx;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (k0 + j < N) x[j] = xr[k0 + j];
    }
    const double rmax = wave_max_d(fmax(fmax(fabs(x[0]), fabs(x[1])), fmax(fabs(x[2]), fabs(x[3]))));
    const double inv = rmax > 0.0 ? I8_Q / rmax : 0.0;
    unsigned w0 = 0, w1 = 0, w2 = 0;
    double l1 = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const double v = x[j] * inv;
      const int q = v == v ? (int)rint(v) : 0;
      l1 += (double)(q < 0 ? -q : q);
      unsigned d0, d1, d2;
      digits3(q, d0, d1, d2);
      w0 |= d0 << (8 * j);
      w1 |= d1 << (8 * j);
      w2 |= d2 << (8 * j);
    }
    l1 = wave_sum(l1);
    if (act) {
      unsigned* rq = (unsigned*)(Xq + (size_t)row * 3 * Kp);
      rq[lane] = w0;
      rq[Kp / 4 + lane] = w1;
      rq[2 * (Kp / 4) + lane] = w2;
    }
    if (lane == 0) xscale[row] = make_double2(rmax, l1);
  }
}

// per row {a_r, m_r}: position of an output = |w| * a_r + (1 - cut0/step), w = 256 S2 + S3 + (S4 >> 8);
// margin m_r as in the header
__global__ void k_rowinfo(const double2* __restrict__ xscale, int64_t nx, int64_t nrows, int N,
                          const unsigned long long* __restrict__ scal, double inv_step, double slack,
                          float2* __restrict__ rowinfo) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= nrows) return;
  float a = 0.f, m = 0.f;
  if (row < nx) {
    const double2 sc = xscale[row];
    if (sc.x > 0.0) {
      const double ymax = __longlong_as_double((long long)scal[0]);
      const double l1y = (double)scal[1];
      const double unit = (sc.x / I8_Q) * (ymax / I8_Q) * inv_step;
      // + 2 * 2^24: the low byte of S4, which the epilogue drops (w = 256 S2 + S3 + (S4 >> 8), exact in int32)
      const double U = 0.5 * sc.y + 0.5 * l1y + (double)N * I8_DROP + 2.0 * 16777216.0;
      a = (float)(unit * 16777216.0);
      m = (float)((unit * U + slack) * 1.0001);
    }
  }
  rowinfo[row] = make_float2(a, m);
}

// The products.  Block = 8 waves, each with its own tile of 32 cells whose 3 x KS A operands stay in
// registers while the block sweeps the permutations in stages of G strips of 64 (double-buffered in LDS by
// LDS-DMA, one barrier per stage); then the next work item.  Lane (j = lane & 31, kh = lane >> 5) of a 32x32
// result holds column j and rows (reg & 3) + 8 (reg >> 2) + 4 kh.
//
// Matrix and vector work of DIFFERENT waves of a SIMD overlap on this chip (tools/micro/mfma_i8_overlap.hip:
// an i8 MFMA keeps the vector port for ~16 of its 32 cycles), but two waves that walk the same
// "products, binning, products, binning" sequence between the same barriers do both at the same time.  So
// the second wave of every SIMD (waves 4-7) runs the sequence rotated by half a period: it bins the LAST
// column tile of a stage at the beginning of the next stage (the accumulators simply stay in registers
// across the barrier), while its partner is in its first block of products.
template <int KS, int G, int MODE = 0>     // MODE: experiments (1: products only, 2: no counters, 3: no products)
__global__ __launch_bounds__(512) void k_null_i8(const unsigned char* __restrict__ Xq, const float2* __restrict__ rowinfo,
                                                 int64_t ntiles, const v4i* __restrict__ Yq, int nstages, int nsplit,
                                                 int spp, int T, float bconst, unsigned int* __restrict__ partial,
                                                 uint2* __restrict__ queue, unsigned long long* __restrict__ qcount,
                                                 unsigned long long qcap, int* __restrict__ status, int rotate) {
  extern __shared__ v4i sm[];
  constexpr int SB = 3 * KS * 128;                            // int4 per strip of 64 permutations
  constexpr int SG = G * SB;                                  // per stage
  constexpr int LPT = (SG + 511) / 512;
  constexpr int NCT = 2 * G;                                  // column tiles (32 permutations) per stage
  const int TW = T + 1;
  v4i* bbuf = sm;
  unsigned* hist = (unsigned*)(sm + 2 * SG);                  // [TW][32]; row 0 takes what does not count
  uint2* qb_all = (uint2*)(hist + (size_t)TW * 32 + 64);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, kh = lane >> 5;
  uint2* qb = qb_all + wv * I8_QBUF;                          // (not volatile: a volatile generic pointer into LDS trips the gfx950 backend)
  const float Thi = (float)T + 0.5f;
  const bool rot = rotate != 0 && wv >= 4;
  unsigned* histj = hist + j;
  for (int i = tid; i < TW * 32 + 64; i += 512) hist[i] = 0u;

  // work items: (group of 8 tiles, part of the stages); dealt round-robin so that short inputs still balance
  const int64_t nitems = ((ntiles + 7) / 8) * nsplit;
  int wcount = 0;

  auto flush = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(qcount, (unsigned long long)wcount);
    base = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
           (unsigned)__builtin_amdgcn_readfirstlane((int)(base & 0xffffffffu));
    if (base + (unsigned long long)wcount <= qcap) {
      for (int i = lane; i < wcount; i += 64) queue[base + i] = qb[i];
    } else if (lane == 0) {
      atomicOr(status, 1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    wcount = 0;
  };

  if (blockIdx.x < nitems) {
    // stages travel global -> LDS by LDS-DMA (16 bytes per lane to a wave-uniform base + 16 * lane: the stage is
    // stored in exactly that order), no staging registers; the issuing wave waits for its own copies (vmcnt) in
    // front of the barrier that hands the buffer over
    auto stage = [&](int st, int b) {
      const v4i* src = Yq + (size_t)st * SG;
      __attribute__((address_space(3))) v4i* dst = (__attribute__((address_space(3))) v4i*)sm + b * SG;
#pragma unroll
      for (int u = 0; u < LPT; ++u) {
        const int ch = (u * 8 + wv) * 64;
        if (ch < SG)
          __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + ch + lane),
                                           (__attribute__((address_space(3))) void*)(dst + ch), 16, 0, 0);
      }
    };
    stage((int)(blockIdx.x % nsplit) * spp, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int it = 0;
    v16i S2, S3, S4;
    bool carry = false;
    unsigned carry_perm0 = 0;
    for (int64_t item = blockIdx.x; item < nitems; item += gridDim.x) {
      const int64_t tile = (item / nsplit) * 8 + wv;
      const int st0 = (int)(item % nsplit) * spp;
      const int st1 = st0 + spp < nstages ? st0 + spp : nstages;
      const bool last = item + gridDim.x >= nitems;
      const int nxt0 = (int)((item + gridDim.x) % nsplit) * spp;
      const bool have = tile < ntiles;
      v4i a[3][KS];
      float ai[16], mi[16];
      {
        // row i = lane & 31 of the tile, bytes [32 s + 16 kh, + 16) of digit plane d (the k order inside a step is
        // the same for A and B, whatever the instruction makes of it)
        const unsigned char* ap = Xq + ((size_t)(have ? tile : 0) * 32 + j) * (3 * 32 * KS) + 16 * kh;
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
          for (int s = 0; s < KS; ++s) a[d][s] = have ? *(const v4i*)(ap + d * 32 * KS + 32 * s) : (v4i){0, 0, 0, 0};
        const float2* rp = rowinfo + (size_t)(have ? tile : 0) * 32 + 4 * kh;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float2 ri = rp[(r & 3) + 8 * (r >> 2)];
          ai[r] = have ? ri.x : 0.f;
          mi[r] = have ? 0.5f - ri.y : 0.5f;                  // zero rows (a = m = 0): never near
        }
      }
      // Branch-free binning of the 16 outputs a lane holds of one column tile.  u = position clamped to
      // [1/2, T + 1/2]: out-of-range outputs get the fraction 1/2 (never near a cut) and the bins 0 / T.  Bin 0
      // (below the first cut) lands in row 0 of the counters, which nobody reads.  Outputs within the margin of a cut are
      // counted like the others and go to the recheck queue as well.
      auto epilogue = [&](unsigned perm0) {
        unsigned long long nm[16], anym = 0ull;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int w = S2[r] * 256 + S3[r] + (S4[r] >> 8);          // v / 256, the low byte of S4 is inside the margin
          const float t = fmaf(fabsf((float)w), ai[r], bconst);
          const float u = __builtin_amdgcn_fmed3f(t, 0.5f, Thi);
          const bool near = fabsf(__builtin_amdgcn_fractf(u) - 0.5f) > mi[r];   // mi = 1/2 - margin
          nm[r] = __ballot(near);
          anym |= nm[r];
          // counted where it falls even when near a cut: the queue entry carries the bin and the recheck takes it back (a
          // select per output less: 2.37 -> 2.34 ms at 2M x 200, 0.99 -> 0.92 at 1M x 100)
          if (MODE == 2) { if ((int)u == 0x7fffff) atomicAdd(&hist[j], 1u); }
          else atomicAdd(&histj[(int)u * 32], 1u);
        }
        if (__builtin_expect(anym != 0ull, 0)) {
          const unsigned perm = perm0 + (unsigned)j;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const unsigned long long bal = nm[r];
            if (bal == 0ull) continue;
            if ((bal >> lane) & 1ull) {
              const int pos = wcount + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
              qb[pos].x = (unsigned)(tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh);
              const int w = S2[r] * 256 + S3[r] + (S4[r] >> 8);
              const int h = (int)__builtin_amdgcn_fmed3f(fmaf(fabsf((float)w), ai[r], bconst), 0.5f, Thi);
              qb[pos].y = perm | ((unsigned)h << 16);
            }
            wcount += __popcll(bal);
            if (wcount > I8_QBUF - 64) flush();
          }
        }
      };
      for (int st = st0; st < st1; ++st, ++it) {
        // next stage (of this item or the first one of the next) on its way while this one is used
        const bool more = st + 1 < st1 || !last;
        const int nst = st + 1 < st1 ? st + 1 : nxt0;
        if (more) stage(nst, (it + 1) & 1);
        if (carry) {                                          // rotated waves: the last column tile of the previous stage
          epilogue(carry_perm0);
          carry = false;
        }
        const v4i* buf = bbuf + ((it & 1) * SG + kh * 64 + j);
        // B fragments one k step ahead of the products that use them, across the column-tile boundary as well
        v4i bq[2][3];
#pragma unroll
        for (int d = 0; d < 3; ++d) bq[0][d] = buf[((d * KS) * 2) * 64];
        if (MODE != 3) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
#pragma unroll
          for (int r = 0; r < 16; ++r) { S2[r] = 0; S3[r] = 0; S4[r] = 0; }
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            const int cs = c * KS + s;
            if (cs + 1 < NCT * KS) {
              const int c1 = (cs + 1) / KS, s1 = (cs + 1) % KS;
#pragma unroll
              for (int d = 0; d < 3; ++d)
                bq[(cs + 1) & 1][d] = buf[(c1 >> 1) * SB + ((d * KS + s1) * 2) * 64 + (c1 & 1) * 32];
            }
            const v4i b0 = bq[cs & 1][0], b1 = bq[cs & 1][1], b2 = bq[cs & 1][2];
            if (MODE == 3) { S4[0] += b0[0] + a[0][s][0]; S3[1] += b1[1] + a[1][s][1]; S2[2] += b2[2] + a[2][s][2]; }
            else {
              S4 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[0][s], b2, S4, 0, 0, 0);
              S3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[1][s], b2, S3, 0, 0, 0);
              S2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[2][s], b2, S2, 0, 0, 0);
              S4 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[1][s], b1, S4, 0, 0, 0);
              S3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[2][s], b1, S3, 0, 0, 0);
              S4 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[2][s], b0, S4, 0, 0, 0);
              // pin the order "three fragment reads of the next step, then the six products of this one": left alone the
              // scheduler sinks every read to just before its first use (one register quad, a wait per fragment)
              if (cs + 1 < NCT * KS) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
            }
          }
          if (MODE == 1) {
            int z = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) z |= S2[r] ^ S3[r] ^ S4[r];
            if (z == 0x12345678) atomicAdd(&hist[j], 1u);
            continue;
          }
          const unsigned perm0 = (unsigned)((st * G + (c >> 1)) * 64 + (c & 1) * 32);
          if (c == NCT - 1 && rot && st + 1 < st1) {          // binned at the top of the next stage
            carry = true;
            carry_perm0 = perm0;
          } else {
            epilogue(perm0);
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
    }
    if (wcount > 0) flush();
  }
  __syncthreads();
  for (int t = tid; t < TW; t += 512) {
    unsigned s = 0;
#pragma unroll 8
    for (int q = 0; q < 32; ++q) s += hist[t * 32 + q];
    partial[(size_t)blockIdx.x * TW + t] = s;
  }
}

// The queued outputs again, in f64 against the exact cuts: 16 lanes per entry.
__global__ __launch_bounds__(256) void k_null_recheck(const double* __restrict__ X, int ldx, int N,
                                                      const double* __restrict__ Yt, int ldt,
                                                      const uint2* __restrict__ queue,
                                                      const unsigned long long* __restrict__ qcount,
                                                      unsigned long long qcap, const double* __restrict__ cuts, int T,
                                                      double cut0, double inv_step,
                                                      unsigned int* __restrict__ partial) {
  extern __shared__ unsigned rh[];                            // T + 1
  const int TW = T + 1;
  for (int i = threadIdx.x; i < TW; i += 256) rh[i] = 0u;
  __syncthreads();
  unsigned long long cnt = *qcount;
  if (cnt > qcap) cnt = 0;                                    // overflow: the f64 kernel takes over
  const int sub = threadIdx.x & 15;
  const unsigned long long ngrp = (unsigned long long)gridDim.x * 16;
  for (unsigned long long e = (unsigned long long)blockIdx.x * 16 + (threadIdx.x >> 4); e < cnt; e += ngrp) {
    const uint2 en = queue[e];
    const double* xr = X + (size_t)en.x * ldx;
    const double* yr = Yt + (size_t)(en.y & 0xffffu) * ldt;
    const unsigned counted = en.y >> 16;                       // bin the products kernel counted it in (0: none)
    double s = 0.0;
    for (int k = sub; k < N; k += 16) s = fma(xr[k], yr[k], s);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (sub == 0) {
      const double x = fabs(s);
      int h = 0;
      if (x >= cut0) {
        double g = (x - cut0) * inv_step;
        h = g < (double)T ? (int)g + 1 : T;
        if (h > T) h = T;
      }
      // count = #{k : cuts[k] <= x}
      while (h < T && cuts[h] <= x) ++h;
      while (h > 0 && cuts[h - 1] > x) --h;
      // (unsigned wrap-around: k_i8_reduce reads these slabs as signed)
      if (h != (int)counted) {
        if (h > 0) atomicAdd(&rh[h], 1u);
        if (counted > 0) atomicSub(&rh[counted], 1u);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < TW; i += 256) partial[(size_t)blockIdx.x * TW + i] = rh[i];
}

// hist[t] = outputs with exactly t+1 cuts reached, over all slabs (integers: any order); one wave per t
// (slabs from `nplain` on are the recheck's: corrections, signed)
__global__ __launch_bounds__(256) void k_i8_reduce(const unsigned int* __restrict__ partial, int nslabs, int nplain, int T,
                                                   unsigned long long* __restrict__ hist) {
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (t >= T) return;
  unsigned long long s = 0;
  for (int b = lane; b < nslabs; b += 64) {
    const unsigned v = partial[(size_t)b * (T + 1) + t + 1];
    s += b < nplain ? (unsigned long long)v : (unsigned long long)(long long)(int)v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) hist[t] = s;
}

typedef int (*i8_launch_fn)(cna_ctx*, unsigned, size_t, const unsigned char*, const float2*, int64_t, const v4i*, int, int, int, int, float,
                            unsigned int*, uint2*, unsigned long long*, unsigned long long, int*);
// strips of 64 permutations per stage: as many as keep a stage near 40-50 KB (two stages + the counters in LDS)
constexpr int i8_stage_strips(int KS) { return KS >= 5 ? 1 : (KS >= 3 ? 2 : 4); }
template <int KS>
static int launch_i8_t(cna_ctx* c, unsigned grid, size_t smem, const unsigned char* Xq, const float2* rowinfo, int64_t ntiles,
                       const v4i* Yq, int nstages, int nsplit, int spp, int T, float bconst, unsigned int* partial, uint2* queue,
                       unsigned long long* qcount, unsigned long long qcap, int* status) {
  constexpr int G = i8_stage_strips(KS);
  static bool attr_set = false;
  if (!attr_set) {
    HIP_TRY(hipFuncSetAttribute((const void*)k_null_i8<KS, G>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const int rotate = 0;       // (the second wave of a SIMD half a period behind the first: measured 2.53 -> 2.76 ms at 2M x 200)
  hipLaunchKernelGGL((k_null_i8<KS, G>), dim3(grid), dim3(512), smem, c->stream, Xq, rowinfo, ntiles, Yq, nstages, nsplit, spp, T,
                     bconst, partial, queue, qcount, qcap, status, rotate);
  HIP_TRY(hipGetLastError());
  return 0;
}
static const i8_launch_fn kI8[8] = {launch_i8_t<1>, launch_i8_t<2>, launch_i8_t<3>, launch_i8_t<4>,
                                    launch_i8_t<5>, launch_i8_t<6>, launch_i8_t<7>, launch_i8_t<8>};

static size_t i8_lds(int KS, int T) {
  return (size_t)2 * i8_stage_strips(KS) * 3 * KS * 128 * 16 + (size_t)(T + 1) * 32 * 4 + 256 + (size_t)8 * I8_QBUF * 8;
}

// usable for this pass?  (samples within the register budget of the A operands, cuts an arithmetic
// progression to well within a step and starting more than a step above zero -- zero rows and the
// padding must stay below the first cut by more than any margin -- and the counters fit in LDS)
bool null_i8_enabled() {
  const char* e = getenv("CNA_NULL_F64");                  // (read per call: tests/test_gpu_parity.py flips it)
  return !(e && atoi(e) != 0);
}
bool null_i8_eligible(const cna_ctx* c, int P, int T, double cut0, double inv_step, double eps) {
  if (!null_i8_enabled()) return false;
  if (c->Nx > 256 || c->Nx < 2 || c->nx < 1) return false;
  if (P > 65535 - 64 || T > 65535) return false;          // a queue entry packs the permutation and the bin into 16 bits each
  if (!(inv_step > 0.0) || !(eps < 0.05) || !(cut0 * inv_step > 2.0) || !(cut0 * inv_step + T < 5e4)) return false;
  const int KS = (c->Nx + 31) / 32;
  return i8_lds(KS, T) <= 160 * 1024 && c->nx < (int64_t)1 << 31;
}

// Room for the digit planes of the current X: (tiles of 32 rows) x 3 x 32 KS bytes and one {max |x|, sum |q|} per
// row; the rows past nx of the last tile are zero.  A change of shape invalidates what is there.
int ensure_xq(cna_ctx* c, int KS) {
  const int64_t rows = (c->nx + 31) / 32 * 32;
  const int64_t need = rows * 3 * 32 * KS, need_s = rows * 16;
  if (c->xq_valid && (c->xq_rows != rows || c->xq_KS != KS)) c->xq_valid = false;
  if (c->xq_valid) return 0;
  CNA_TRY(dev_reserve(c, &c->xq, &c->xq_cap, std::max<int64_t>(need, 256)));
  CNA_TRY(dev_reserve(c, &c->xq_scale, &c->xq_scale_cap, std::max<int64_t>(need_s, 256)));
  if (rows > c->nx) {
    HIP_TRY(hipMemsetAsync((char*)c->xq + c->nx * 3 * 32 * KS, 0, (size_t)(rows - c->nx) * 3 * 32 * KS, c->stream));
    HIP_TRY(hipMemsetAsync((char*)c->xq_scale + c->nx * 16, 0, (size_t)(rows - c->nx) * 16, c->stream));
  }
  c->xq_rows = rows;
  c->xq_KS = KS;
  return 0;
}

// sums_dev[t] = sum over permutations of #{cells : |x.yc| >= cuts[t]}; *status_out (device) = 0 when they are valid
int launch_null_local_i8(cna_ctx* c, const double* Yc_dev, int ldy, int P, const double* cuts_dev, int T, double cut0,
                         double inv_step, double eps, int64_t** sums_out, int** status_out) {
  const int N = c->Nx, KS = (N + 31) / 32;
  const int64_t ntiles = (c->nx + 31) / 32;
  const int nstrips = (P + 63) / 64, Ppad = nstrips * 64;
  const int ldt = round_up(N, 16);
  const int nrecheck = 1024;
  // deviation of the cuts from the progression + the float roundings of the epilogue (v: two fmas, a_r, the
  // position fma: each 2^-24 relative to a position of at most cut0/step + T + 1)
  const double slack = eps + (cut0 * inv_step + T + 2.0) * 5e-7 + 1e-6;
  const int64_t ngroups = (ntiles + 7) / 8;
  // short inputs: split the strips of a group over several work items (>= ~12 items per workgroup)
  const int G = i8_stage_strips(KS);
  const int nstages = (nstrips + G - 1) / G;                 // the strips past P are zero: below every cut
  int nsplit = (int)((12 * 256 + ngroups - 1) / ngroups);
  if (nsplit > nstages) nsplit = nstages;
  if (nsplit < 1) nsplit = 1;
  const int spp = (nstages + nsplit - 1) / nsplit;
  nsplit = (nstages + spp - 1) / spp;
  const int64_t nitems = ngroups * nsplit;
  const unsigned grid = (unsigned)(nitems < 256 ? nitems : 256);
  const int nslabs = (int)grid + nrecheck;
  uint64_t qcap = (uint64_t)(c->nx / 16) * (uint64_t)Ppad / 4;          // 1/64 of all outputs
  if (qcap < ((uint64_t)1 << 20)) qcap = (uint64_t)1 << 20;
  if (const char* e = getenv("CNA_I8_QCAP")) qcap = (uint64_t)atoll(e) > 0 ? (uint64_t)atoll(e) : qcap;   // tests: force the overflow path
  const int64_t sizes[] = {
      (int64_t)256,                                  // (X digit planes live in c->xq: written with X where possible)
      (int64_t)8 * 32 * ntiles,                      // rowinfo
      (int64_t)16 * 3 * KS * 128 * nstages * G,      // Yq
      (int64_t)8 * Ppad * ldt,                       // Yt
      (int64_t)8 * Ppad + 64,                        // colL1 | scal[2] | qcount | status
      (int64_t)8 * (int64_t)qcap,                    // queue
      (int64_t)4 * nslabs * (T + 1),                 // slabs
      (int64_t)8 * T, (int64_t)8 * T};               // hist, tails
  int64_t need = 0;
  for (int64_t s : sizes) need += round_up64(s, 256);
  CNA_TRY(dev_reserve(c, &c->i8_buf, &c->i8_cap, need));
  char* base = (char*)c->i8_buf;
  int64_t off = 0;
  auto take = [&](int idx) { char* p = base + off; off += round_up64(sizes[idx], 256); return p; };
  take(0);
  CNA_TRY(ensure_xq(c, KS));
  const unsigned char* Xq = (const unsigned char*)c->xq;
  float2* rowinfo = (float2*)take(1);
  v4i* Yq = (v4i*)take(2);
  double* Yt = (double*)take(3);
  unsigned long long* colL1 = (unsigned long long*)take(4);
  unsigned long long* scal = colL1 + Ppad;
  unsigned long long* qcount = scal + 2;
  int* status = (int*)(scal + 3);
  uint2* queue = (uint2*)take(5);
  unsigned int* slabs = (unsigned int*)take(6);
  unsigned long long* hist = (unsigned long long*)take(7);
  int64_t* sums_dev = (int64_t*)take(8);
  ProfScope ps(c, CNA_K_NULL_LOCAL);
  HIP_TRY(hipMemsetAsync(colL1, 0, (size_t)8 * Ppad + 64, c->stream));
  if (nstages * G > nstrips)
    HIP_TRY(hipMemsetAsync(Yq + (size_t)nstrips * 3 * KS * 128, 0, (size_t)16 * 3 * KS * 128 * (nstages * G - nstrips), c->stream));
  hipLaunchKernelGGL(k_y_absmax, dim3(16), dim3(1024), 0, c->stream, Yc_dev, ldy, N, P, scal);
  {
    const int64_t nthr = (int64_t)Ppad * 2 * KS;
    hipLaunchKernelGGL(k_quant_y, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, c->stream, Yc_dev, ldy, N, P, KS, Ppad,
                       scal, Yq, Yt, ldt, colL1);
  }
  hipLaunchKernelGGL(k_y_finish, dim3(1), dim3(1024), 0, c->stream, colL1, P, scal);
  if (!c->xq_valid) {                                        // X did not come with its digit planes
    hipLaunchKernelGGL(k_quant_x, dim3((unsigned)ntiles), dim3(256), 0, c->stream, c->X, c->ldx, c->nx, N, 32 * KS,
                       (unsigned char*)c->xq, (double2*)c->xq_scale);
    c->xq_valid = true;
  }
  hipLaunchKernelGGL(k_rowinfo, dim3((unsigned)((ntiles * 32 + 255) / 256)), dim3(256), 0, c->stream, (const double2*)c->xq_scale,
                     c->nx, ntiles * 32, N, scal, inv_step, slack, rowinfo);
  const float bconst = (float)(1.0 - cut0 * inv_step);
  CNA_TRY(kI8[KS - 1](c, grid, i8_lds(KS, T), Xq, rowinfo, ntiles, Yq, nstages, nsplit, spp, T, bconst, slabs, queue, qcount, qcap, status));
  hipLaunchKernelGGL(k_null_recheck, dim3(nrecheck), dim3(256), (size_t)4 * (T + 1), c->stream, c->X, c->ldx, N, Yt, ldt, queue,
                     qcount, qcap, cuts_dev, T, cut0, inv_step, slabs + (size_t)grid * (T + 1));
  hipLaunchKernelGGL(k_i8_reduce, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, c->stream, slabs, nslabs, (int)grid, T, hist);
  HIP_TRY(hipGetLastError());
  CNA_TRY(launch_suffix_sum(c, hist, 1, T, sums_dev));
  *sums_out = sums_dev;
  *status_out = status;
  c->i8_qcount = qcount;
  return 0;
}
