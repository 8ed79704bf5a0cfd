// Random-walk diffusion over the kNN connectivities graph (reference: _nam.py:21-76).
//
//   colsums = A.sum(axis=0) + w                       (_nam.py:28)
//   s <- A.(s/colsums) + w*s/colsums                   (_nam.py:33)
//
// Device layout: the *scaled* state T = s/colsums is kept for every global cell
// (n_pad x ld float64, row-major, one row per cell) because a step gathers the rows of
// the neighbours j of cell i:  s'[i,:] = sum_e A[i,j_e]*T[j_e,:] + w*T[i,:].
// One 64-lane wave owns one destination row; lanes span the sample axis so every
// gathered neighbour row is one coalesced 8*N-byte read.  Products and sums are issued
// unfused and in CSR order, i.e. the same rounding sequence as scipy's csr_matvecs.
// Workgroup -> row mapping is XCD-aware: each of the 8 XCDs walks a contiguous range of
// rows so the neighbour rows it gathers stay in that XCD's 4 MiB L2.
#include "common.h"
#include <cstdlib>
#include <algorithm>
#include <type_traits>

// The walk's products and sums are meant to round like scipy's csr_matvecs (a multiply, then an add).
// hipcc contracts a*b+c into an fma by default -- also through __dmul_rn / __dadd_rn, which are plain
// operators carrying the header's contract flag (the gather loop compiled to v_fmac_f64) -- so
// contraction is switched off for this file and the walk uses its own mul_rn / add_rn, defined below
// the pragma.  The division sequences use explicit fma builtins and are not affected.
#pragma clang fp contract(off)
__device__ __forceinline__ double mul_rn(double a, double b) { return a * b; }
__device__ __forceinline__ double add_rn(double a, double b) { return a + b; }

namespace {

struct CellInfo {      // one gather per edge in the first step
  double inv_colsum;
  int32_t sid;
  int32_t pad;
};

struct StepArgs {
  const int64_t* indptr;
  const int32_t* idx;
  const void* val;
  const int32_t* sid;
  const double* colsum;
  const double* counts;
  const double* Tin;
  double* Tout;
  double* nam;
  double* stat;
  double* dense_out;
  int64_t n_local, row0;
  // this launch covers rows[0 .. n_local) of the block (halo overlap: the rows other ranks wait for first, then the
  // rest); null: all rows 0 .. n_local in order
  const int32_t* rows;
  int width, ld;
  double w;
  int want_kurt, write_t, write_nam;
  int xcd_chunk;
  // compressed copy of the state after the first step (see k_nam_step_sparse); null = not kept
  struct SpPair* sp_pair;   // n x SP_CAP {value, sample index} records of the non-zeros of a row
  unsigned char* sp_cnt;    // n: how many (SP_DENSE: more than SP_CAP, use the dense row)
  // walk with the stop rule on the device (cna_nam_auto): steps queued behind the one that met the rule find this
  // word non-zero and return at once; null = unconditional step
  const int* stop;
  int sp_keep_dense;        // sharded: the first step writes the dense row of every cell besides its pairs (halo exchange)
  // the selection pass as a by-product of the last step (select_tail below); sel_X == null: not wanted
  double* sel_X;
  const double* sel_y;
  double* sel_nc;
  unsigned char* sel_xq;    // digit planes of X for the integer local null; null: none
  double2* sel_xscale;
  unsigned long long* sel_nz;   // [0] rows of zero variance, [1] bits of max |coefficient| (NaN pattern if any is NaN)
  int sel_ldx, sel_Kp;
  int skip_nam;             // with the by-product: the raw NAM is not stored (c_api.hip:need_nam materialises it on demand)
  // 4-byte state between steps (common.h:state_f32_mode); host-side dispatch only: the kernels are instantiated per format
  int in32, out32;
};
#define STEP_STOPPED(a) ((a).stop != nullptr && __builtin_nontemporal_load((a).stop) != 0)
// One 16-byte record per non-zero: the second step then fetches an edge's whole neighbour row with ONE
// global_load_dwordx4 (lane l = pair l).  With the indices and the values in two arrays the step issued two
// gathers per edge: 4.59 -> 4.42 ms at 2M x 200.  What the step spends its time on is the scatter, not the
// gather (tools/micro/lds_scatter_rate.hip: a ds_add_f64 of 35 lanes into random columns occupies the CU's LDS
// for ~15 clk, 8 of them bank conflicts of the address pattern alone = 2.1 ms of the launch; with the pair
// gathers replaced by constants the launch still takes 3.9 ms).
struct SpPair {
  double v;
  int32_t col;
  int32_t pad;
};
static_assert(sizeof(SpPair) == 16, "one dwordx4 per pair");
constexpr int SP_CAP = 64;
constexpr int SP_DENSE = 255;

__device__ __forceinline__ double readlane_d(double v, int l) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, l);
  hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ int64_t uniform64(int64_t v) {
  int lo = __builtin_amdgcn_readfirstlane((int)(v & 0xffffffffll));
  int hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
  return ((int64_t)hi << 32) | (uint32_t)lo;
}

// write-out shared by both step kernels: s[] holds the new unscaled state of one row; element k
// of lane `lane` is column col_of(lane, k)
struct ColStride1 {   // one double per lane: col = lane + 64*k
  __device__ static int col(int lane, int k) { return lane + 64 * k; }
};
struct ColPair {      // double2 per lane: cols 2*(lane + 64*(k/2)) + (k&1)
  __device__ static int col(int lane, int k) { return 2 * (lane + 64 * (k >> 1)) + (k & 1); }
};

struct ColQuad {      // four adjacent columns per lane (4-byte state: one float4): cols 4*(lane + 64*(k/4)) + (k&3)
  __device__ static int col(int lane, int k) { return 4 * (lane + 64 * (k >> 2)) + (k & 3); }
};

// What rows.hip:k_select_std16 makes of a NAM row when every cell and every sample stays and nothing is regressed out
// (_association.py:182 zero-variance count, _nam.py:122 centring, :159 division by the std with ddof = 1,
// _association.py:77 coefficient X.y/N, the 24-bit digit planes of null_i8.hip), done by the wave that has just
// formed the row: x[] holds s/C in the ColPair layout (lane l: columns 2l, 2l+1, 128+2l, ...).  Same statements as
// that kernel; its sums run over 16 lanes x 4 columns, these over 64 lanes x 2, so the results agree to rounding.
template <int NV>
__device__ __forceinline__ void select_tail(const StepArgs& a, int64_t row, int lane, double (&x)[NV]) {
  const double n = (double)a.width;
  double sum = 0.0;
#pragma unroll
  for (int k = 0; k < NV; ++k) sum += x[k];
  const double avg0 = wave_sum(sum) / n;
  bool flat = true;
#pragma unroll
  for (int k = 0; k < NV; ++k)
    if (ColPair::col(lane, k) < a.width) flat = flat && (avg0 - x[k] == 0.0);
  if (__all(flat) && lane == 0) atomicAdd(a.sel_nz, 1ull);
#pragma unroll
  for (int k = 0; k < NV; ++k)
    if (ColPair::col(lane, k) < a.width) x[k] -= avg0;
  double s2 = 0.0;
#pragma unroll
  for (int k = 0; k < NV; ++k) s2 += x[k];
  const double avg = wave_sum(s2) / n;
  double ss = 0.0;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    if (ColPair::col(lane, k) < a.width) {
      const double d = avg - x[k];
      ss += d * d;
    }
  }
  const double sd = sqrt(wave_sum(ss) / (n - 1.0));
  double dot = 0.0, amax = 0.0;
  double* __restrict__ dst = a.sel_X + row * a.sel_ldx;
#pragma unroll
  for (int k = 0; k < NV; k += 2) {
    const int c0 = ColPair::col(lane, k);
    const double x0 = c0 < a.width ? __ddiv_rn(x[k], sd) : 0.0;
    const double x1 = c0 + 1 < a.width ? __ddiv_rn(x[k + 1], sd) : 0.0;
    x[k] = x0;
    x[k + 1] = x1;
    if (c0 < a.width) dot += a.sel_y[c0] * x0;
    if (c0 + 1 < a.width) dot += a.sel_y[c0 + 1] * x1;
    amax = fmax(amax, fmax(fabs(x0), fabs(x1)));          // NaN rows (zero variance): fmax drops them, q = 0 below
    if (c0 + 1 < a.sel_ldx) *(double2*)(dst + c0) = make_double2(x0, x1);
  }
  if (a.sel_xq) {
    const double rmax = wave_max_d(amax);
    const double inv = rmax > 0.0 ? I8_QMAX / rmax : 0.0;
    unsigned char* rq = a.sel_xq + (size_t)row * 3 * a.sel_Kp;
#pragma unroll
    for (int k = 0; k < NV; k += 2) {
      unsigned h0 = 0, h1 = 0, h2 = 0;                      // this lane's two bytes of each plane
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const double v = x[k + j] * inv;
        const int qi = v == v ? (int)rint(v) : 0;
        const int q1 = (qi + 128) >> 8;
        h0 |= ((unsigned)qi & 255u) << (8 * j);
        h1 |= ((unsigned)q1 & 255u) << (8 * j);
        h2 |= ((unsigned)((q1 + 128) >> 8) & 255u) << (8 * j);
      }
      // even lanes store dwords: their own half and the odd neighbour's
      const unsigned p01 = h0 | (h1 << 16);
      const unsigned o01 = (unsigned)__builtin_amdgcn_ds_bpermute(4 * (lane + 1), (int)p01);
      const unsigned o2 = (unsigned)__builtin_amdgcn_ds_bpermute(4 * (lane + 1), (int)h2);
      const int c0 = ColPair::col(lane, k);
      if ((lane & 1) == 0 && c0 < a.sel_Kp) {
        *(unsigned*)(rq + c0) = (p01 & 0xffffu) | (o01 << 16);
        *(unsigned*)(rq + a.sel_Kp + c0) = (p01 >> 16) | (o01 & 0xffff0000u);
        *(unsigned*)(rq + 2 * a.sel_Kp + c0) = (h2 & 0xffffu) | (o2 << 16);
      }
    }
    const double l1 = rmax > 0.0 ? n * (I8_QMAX / rmax) + n : 0.0;
    if (lane == 0) a.sel_xscale[row] = make_double2(rmax, l1);
  }
  const double v = wave_sum(dot) / n;
  if (lane == 0) {
    a.sel_nc[row] = v;
    const unsigned long long bits = v != v ? 0x7ff8000000000000ull : (unsigned long long)__double_as_longlong(fabs(v));
    // the maximum over all rows: most waves find the word already above their value and skip the atomic
    if (bits > __builtin_nontemporal_load(a.sel_nz + 1)) atomicMax(a.sel_nz + 1, bits);
  }
}

// The same statements for the four-columns-per-lane layout of k_nam_step32 (ColQuad).  A lane holds four adjacent bytes of
// each digit plane, so the planes are stored as dwords without the exchange between neighbouring lanes.
template <int NV>
__device__ __forceinline__ void select_tail_quad(const StepArgs& a, int64_t row, int lane, double (&x)[NV]) {
  const double n = (double)a.width;
  double sum = 0.0;
#pragma unroll
  for (int k = 0; k < NV; ++k) sum += x[k];
  const double avg0 = wave_sum(sum) / n;
  bool flat = true;
#pragma unroll
  for (int k = 0; k < NV; ++k)
    if (ColQuad::col(lane, k) < a.width) flat = flat && (avg0 - x[k] == 0.0);
  if (__all(flat) && lane == 0) atomicAdd(a.sel_nz, 1ull);
#pragma unroll
  for (int k = 0; k < NV; ++k)
    if (ColQuad::col(lane, k) < a.width) x[k] -= avg0;
  double s2 = 0.0;
#pragma unroll
  for (int k = 0; k < NV; ++k) s2 += x[k];
  const double avg = wave_sum(s2) / n;
  double ss = 0.0;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    if (ColQuad::col(lane, k) < a.width) {
      const double d = avg - x[k];
      ss += d * d;
    }
  }
  const double sd = sqrt(wave_sum(ss) / (n - 1.0));
  double dot = 0.0, amax = 0.0;
  double* __restrict__ dst = a.sel_X + row * a.sel_ldx;
#pragma unroll
  for (int k = 0; k < NV; k += 2) {
    const int c0 = ColQuad::col(lane, k);
    const double x0 = c0 < a.width ? __ddiv_rn(x[k], sd) : 0.0;
    const double x1 = c0 + 1 < a.width ? __ddiv_rn(x[k + 1], sd) : 0.0;
    x[k] = x0;
    x[k + 1] = x1;
    if (c0 < a.width) dot += a.sel_y[c0] * x0;
    if (c0 + 1 < a.width) dot += a.sel_y[c0 + 1] * x1;
    amax = fmax(amax, fmax(fabs(x0), fabs(x1)));
    if (c0 + 1 < a.sel_ldx) *(double2*)(dst + c0) = make_double2(x0, x1);
  }
  if (a.sel_xq) {
    const double rmax = wave_max_d(amax);
    const double inv = rmax > 0.0 ? I8_QMAX / rmax : 0.0;
    unsigned char* rq = a.sel_xq + (size_t)row * 3 * a.sel_Kp;
#pragma unroll
    for (int k = 0; k < NV; k += 4) {
      unsigned h0 = 0, h1 = 0, h2 = 0;                      // this lane's four bytes of each plane
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const double v = x[k + j] * inv;
        const int qi = v == v ? (int)rint(v) : 0;
        const int q1 = (qi + 128) >> 8;
        h0 |= ((unsigned)qi & 255u) << (8 * j);
        h1 |= ((unsigned)q1 & 255u) << (8 * j);
        h2 |= ((unsigned)((q1 + 128) >> 8) & 255u) << (8 * j);
      }
      const int c0 = ColQuad::col(lane, k);
      if (c0 < a.sel_Kp) {
        *(unsigned*)(rq + c0) = h0;
        *(unsigned*)(rq + a.sel_Kp + c0) = h1;
        *(unsigned*)(rq + 2 * a.sel_Kp + c0) = h2;
      }
    }
    const double l1 = rmax > 0.0 ? n * (I8_QMAX / rmax) + n : 0.0;
    if (lane == 0) a.sel_xscale[row] = make_double2(rmax, l1);
  }
  const double v = wave_sum(dot) / n;
  if (lane == 0) {
    a.sel_nc[row] = v;
    const unsigned long long bits = v != v ? 0x7ff8000000000000ull : (unsigned long long)__double_as_longlong(fabs(v));
    if (bits > __builtin_nontemporal_load(a.sel_nz + 1)) atomicMax(a.sel_nz + 1, bits);
  }
}

// FL: what a launch's instantiation compiles in -- bit 0: the selection by-product (select_tail; the last step of a
// walk with a hint), bit 1: a row list (halo overlap), bit 2: the scaled state is written in 4 bytes per entry.  Kept out of the plain instantiations on purpose: the fields
// they need are kernel arguments that stay live in scalar registers across the gather loop, and the compressed second
// step lost 9 % (4.09 -> 4.45 ms at 2M x 200) when it carried them.
template <int NV, typename CM, int FL = 0>
__device__ __forceinline__ void finish_row(const StepArgs& a, int64_t row, int64_t grow, int lane,
                                           const double (&s)[NV]) {
  constexpr bool BYP = (FL & 1) != 0;
  constexpr bool T32 = (FL & 4) != 0;
  const double cs = a.colsum[grow];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int col = CM::col(lane, k);
    if (col < a.ld) {
      const bool in = col < a.width;
      if constexpr (T32) {
        if (a.write_t) ((float*)a.Tout)[grow * a.ld + col] = in ? (float)__ddiv_rn(s[k], cs) : 0.0f;
      } else {
        if (a.write_t) a.Tout[grow * a.ld + col] = in ? __ddiv_rn(s[k], cs) : 0.0;
      }
      if (a.dense_out) a.dense_out[row * a.ld + col] = in ? s[k] : 0.0;
    }
  }
  if (a.write_nam || a.want_kurt) {
    double x[NV];
    double sum = 0.0;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int col = CM::col(lane, k);
      const bool in = col < a.width;
      x[k] = in ? __ddiv_rn(s[k], a.counts[col]) : 0.0;     // s / C   (_nam.py:59,73)
      if (a.write_nam && col < a.ld && !(BYP && a.skip_nam)) a.nam[row * a.ld + col] = x[k];
      sum += x[k];
    }
    if (a.want_kurt) {
      // scipy.stats.kurtosis(s/C, axis=1): Fisher, biased (_nam.py:59)
      const double n = (double)a.width;
      const double mean = wave_sum(sum) / n;
      double d2s = 0.0, d4s = 0.0;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        if (CM::col(lane, k) < a.width) {
          const double d = x[k] - mean;
          const double d2 = d * d;
          d2s += d2;
          d4s += d2 * d2;
        }
      }
      const double m2 = wave_sum(d2s) / n;
      const double m4 = wave_sum(d4s) / n;
      const double em = 2.220446049250313e-16 * mean;
      const double k4 = (m2 <= em * em) ? __builtin_nan("") : m4 / (m2 * m2);
      if (lane == 0) a.stat[grow] = k4 - 3.0;
    }
    if constexpr (BYP && std::is_same<CM, ColPair>::value) {
      if (a.sel_X) select_tail<NV>(a, row, lane, x);
    }
    if constexpr (BYP && std::is_same<CM, ColQuad>::value) {
      if (a.sel_X) select_tail_quad<NV>(a, row, lane, x);
    }
  }
}

template <typename VT>
__device__ __forceinline__ void load_edges(const StepArgs& a, int64_t start, int64_t end, int lane, int& jl,
                                           double& al) {
  const int64_t e = start + lane;
  const bool ok = e < end;
  jl = ok ? a.idx[e] : 0;
  al = ok ? (double)((const VT*)a.val)[e] : 0.0;
}

// Workgroup -> rows: 4 consecutive rows per workgroup (one per wave); XCD x (= blockIdx % 8,
// observed dispatch) walks the contiguous block range [x*cpx, (x+1)*cpx) so the neighbour rows
// it gathers stay in its own 4 MiB L2 (+4 % measured; a pure speed choice).
__device__ __forceinline__ int64_t my_row(int wv, int chunk) {
  // chunk = consecutive workgroups (4 rows each) one XCD takes before the next XCD's chunk begins;
  // chunk = gridDim/8 gives every XCD one contiguous eighth of the rows (best when a cluster of
  // cells fits the 4 MiB L2), a small chunk keeps all XCDs inside one window of rows (best when it
  // only fits the 256 MiB Infinity Cache)
  const int64_t b = blockIdx.x >> 3, x = blockIdx.x & 7;
  const int64_t blk = (b / chunk) * (8 * (int64_t)chunk) + x * chunk + (b % chunk);
  return uniform64(blk * 4 + wv);
}

// what follows the neighbours of a row in the first step: the row's own term, the (sample, value) pairs of the
// new state for the second step, the write-out
template <int NQ>
__device__ __forceinline__ void first_tail(const StepArgs& a, const double* accl, int64_t row, int64_t grow, int lane,
                                           const CellInfo me, const double cs) {
  const double self = __ddiv_rn(a.w, cs);        // (w*1)/colsums[i]
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  double s[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) s[q] = add_rn(accl[lane + 64 * q], (lane + 64 * q == me.sid) ? self : 0.0);
  if (!a.sp_cnt) {
    finish_row<NQ, ColStride1>(a, row, grow, lane, s);
    return;
  }
  // after one step a row is non-zero only at the samples of the cell's neighbours: keep those
  // (sample, value) pairs side by side so the second step gathers ~40 entries instead of N.  The dense row
  // of the state is then dead weight -- the second step (k_nam_step_sparse) reads the pairs, its own row
  // included -- and is written only for rows that overflow the SP_CAP pairs.
  int base = 0;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int col = lane + 64 * q;
    const double t = col < a.width ? __ddiv_rn(s[q], cs) : 0.0;        // what finish_row stores in T
    const unsigned long long m = __ballot(t != 0.0);
    const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
    if (t != 0.0 && pos < SP_CAP) a.sp_pair[grow * SP_CAP + pos] = SpPair{t, col, 0};
    base += __popcll(m);
  }
  if (lane == 0) a.sp_cnt[grow] = (unsigned char)(base <= SP_CAP ? base : SP_DENSE);
  StepArgs b = a;
  b.write_t = a.write_t && (base > SP_CAP || a.sp_keep_dense);
  finish_row<NQ, ColStride1>(b, row, grow, lane, s);
}

// First step: the input is the one-hot sample indicator, so neighbour j contributes
// A[i,j] * (1/colsums[j]) to column sid[j] only.  One 16-byte record {1/colsums, sid} per cell
// makes that a single gather per edge; every lane takes one edge and adds its term into the
// wave's LDS accumulator row with ds_add_f64 (edges of one row rarely share a sample, and
// same-address adds of one instruction retire in lane order, i.e. CSR order).
template <typename VT, int NQ, int FL = 0>
__global__ __launch_bounds__(256) void k_nam_first(StepArgs a, const CellInfo* __restrict__ info) {
  extern __shared__ double sm[];
  if (STEP_STOPPED(a)) return;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double* accl = sm + (size_t)wv * 64 * NQ;
#pragma unroll
  for (int q = 0; q < NQ; ++q) accl[lane + 64 * q] = 0.0;
  int64_t row = my_row(wv, a.xcd_chunk);
  if (row >= a.n_local) return;
  if ((FL & 2) != 0) row = uniform64(a.rows[row]);
  const int64_t grow = a.row0 + row;
  const int64_t start = uniform64(a.indptr[row]), end = uniform64(a.indptr[row + 1]);
  for (int64_t base = start; base < end; base += 64) {
    int jl;
    double al;
    load_edges<VT>(a, base, end, lane, jl, al);
    if (base + lane < end) {
      const CellInfo ci = info[jl];
      if (ci.sid >= 0) unsafeAtomicAdd(&accl[ci.sid], mul_rn(al, ci.inv_colsum));
    }
  }
  first_tail<NQ>(a, accl, row, grow, lane, info[grow], a.colsum[grow]);
}

// The same with TWO rows per wave, one after the other in every phase: the loads of a phase (row pointers,
// edges, the neighbours' records, the row's own record and column sum) are four dependent round trips to memory
// per row, and with one row per wave at full occupancy (8 waves per SIMD) part of the step is the latency of
// that chain.  Both rows' loads of a phase are in flight together: 1002 -> 917-928 us at 2M x 200, 462 -> 399 at
// 1M x 100, 77 -> 61 at 200k x 50; four rows per wave: 1120 / 418 / 62 (profiles/r02_kbench_first_rows.txt).
// (amdgpu_num_sgpr(80): with more than 80 scalar registers a CU admits 7 or 6 of these workgroups instead of 8 --
// MI355X_MICROARCH.md, residency -- and the latency-bound steps feel it: this kernel 0.94 -> 0.84 ms at 2M x 200 with its
// 98 registers capped, the spills are a handful of lane moves outside the loops; the same cap on the compressed step, the selection and the
// batch-kurtosis kernels changed nothing or cost 2-4 %)
template <typename VT, int NQ, int R, int FL = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80))) void k_nam_first2(StepArgs a, const CellInfo* __restrict__ info) {
  extern __shared__ double sm[];
  if (STEP_STOPPED(a)) return;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double* accl[R];
  int64_t row[R], start[R], end[R];
  bool live[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    accl[r] = sm + ((size_t)wv * R + r) * 64 * NQ;
#pragma unroll
    for (int q = 0; q < NQ; ++q) accl[r][lane + 64 * q] = 0.0;
    const int64_t b = R * (int64_t)(blockIdx.x >> 3) + r, x = blockIdx.x & 7;       // my_row() of the two workgroups this one stands for
    const int64_t blk = (b / a.xcd_chunk) * (8 * (int64_t)a.xcd_chunk) + x * a.xcd_chunk + (b % a.xcd_chunk);
    row[r] = uniform64(blk * 4 + wv);
    live[r] = row[r] < a.n_local;
    if ((FL & 2) != 0 && live[r]) row[r] = uniform64(a.rows[row[r]]);
    start[r] = end[r] = 0;
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
    if (live[r]) { start[r] = uniform64(a.indptr[row[r]]); end[r] = uniform64(a.indptr[row[r] + 1]); }
  int jl[R];
  double al[R];
#pragma unroll
  for (int r = 0; r < R; ++r) { jl[r] = 0; al[r] = 0.0; if (live[r]) load_edges<VT>(a, start[r], end[r], lane, jl[r], al[r]); }
  CellInfo ci[R], me[R];
  double cs[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    ci[r] = CellInfo{0.0, -1, 0};
    me[r] = CellInfo{0.0, -1, 0};
    cs[r] = 1.0;
    if (live[r]) {
      if (start[r] + lane < end[r]) ci[r] = info[jl[r]];
      me[r] = info[a.row0 + row[r]];
      cs[r] = a.colsum[a.row0 + row[r]];
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (!live[r]) continue;
    if (start[r] + lane < end[r] && ci[r].sid >= 0) unsafeAtomicAdd(&accl[r][ci[r].sid], mul_rn(al[r], ci[r].inv_colsum));
    for (int64_t base = start[r] + 64; base < end[r]; base += 64) {          // rows of more than 64 neighbours
      int j2;
      double a2;
      load_edges<VT>(a, base, end[r], lane, j2, a2);
      if (base + lane < end[r]) {
        const CellInfo c2 = info[j2];
        if (c2.sid >= 0) unsafeAtomicAdd(&accl[r][c2.sid], mul_rn(a2, c2.inv_colsum));
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
    if (live[r]) first_tail<NQ>(a, accl[r], row[r], a.row0 + row[r], lane, me[r], cs[r]);
}

// Steps >= 2: gather-accumulate over neighbour rows of the scaled state T.  Each lane owns two
// adjacent columns and gathers them with one 16-byte load (measured: the L2 -> L1 gather path
// saturates near 12 TB/s of useful bytes on this chip whatever the instruction mix; 16-byte
// lanes cost half the load instructions of 8-byte ones, -18 % time at N=100).  Neighbour index
// and weight travel lane -> SGPR by v_readlane, the row base T + j*ld is scalar arithmetic, and
// products / sums are unfused and in CSR order (scipy's csr_matvecs rounding sequence).
template <typename VT, int NQ2, int U = 8, int FL = 0>       // U: neighbour rows in flight per wave (8 beats 16 and 32; 4 beyond 512 columns: registers)
__global__ __launch_bounds__(256) void k_nam_step(StepArgs a) {
  if (STEP_STOPPED(a)) return;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int64_t row = my_row(wv, a.xcd_chunk);
  if (row >= a.n_local) return;
  if ((FL & 2) != 0) row = uniform64(a.rows[row]);
  const int64_t grow = a.row0 + row;
  const int64_t start = uniform64(a.indptr[row]), end = uniform64(a.indptr[row + 1]);
  const double2* __restrict__ Tin = (const double2*)a.Tin;
  const int ld2 = a.ld >> 1;
  bool act[NQ2];
  unsigned off[NQ2];
#pragma unroll
  for (int q = 0; q < NQ2; ++q) {
    act[q] = lane + 64 * q < ld2;
    off[q] = act[q] ? (unsigned)(lane + 64 * q) : 0u;
  }
  double2 acc[NQ2];
#pragma unroll
  for (int q = 0; q < NQ2; ++q) acc[q] = make_double2(0.0, 0.0);
  // B neighbour rows requested together, then added one by one in CSR order.  PAD: the last, partial batch of a row --
  // its missing rows are requested from the row's own state (wanted next anyway) and never added: one memory latency for
  // the rest of a row instead of one per edge (the rows of a kNN graph have ~40 edges: with U = 8 a row waited for five
  // batches and 3.5 single rows)
  auto batch = [&](auto bc, auto pc, int jl, double al, int l0, int nvalid) {
    constexpr int B = decltype(bc)::value;
    constexpr bool PAD = decltype(pc)::value;
    double2 t[B][NQ2];
#pragma unroll
    for (int u = 0; u < B; ++u) {
      int64_t j = __builtin_amdgcn_readlane(jl, (l0 + u) & 63);
      if (PAD && u >= nvalid) j = grow;
      const double2* __restrict__ rowp = Tin + j * ld2;
#pragma unroll
      for (int q = 0; q < NQ2; ++q) t[u][q] = act[q] ? rowp[off[q]] : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int u = 0; u < B; ++u) {
      if (!PAD || u < nvalid) {              // (uniform; not a `break`: the batch must unroll completely or t[] goes to scratch)
        const double av = readlane_d(al, (l0 + u) & 63);
#pragma unroll
        for (int q = 0; q < NQ2; ++q) {
          acc[q].x = add_rn(acc[q].x, mul_rn(av, t[u][q].x));
          acc[q].y = add_rn(acc[q].y, mul_rn(av, t[u][q].y));
        }
      }
    }
  };
  for (int64_t base = start; base < end; base += 64) {
    int jl;
    double al;
    load_edges<VT>(a, base, end, lane, jl, al);
    const int cnt = (int)((end - base) < 64 ? (end - base) : 64);
    int l = 0;
    for (; l + U <= cnt; l += U) batch(std::integral_constant<int, U>{}, std::false_type{}, jl, al, l, U);
    const int rem = cnt - l;
    if (rem > (U + 1) / 2) batch(std::integral_constant<int, U>{}, std::true_type{}, jl, al, l, rem);
    else if (rem > 0) batch(std::integral_constant<int, (U + 1) / 2>{}, std::true_type{}, jl, al, l, rem);
  }
  double s[2 * NQ2];
#pragma unroll
  for (int q = 0; q < NQ2; ++q) {
    const double2 own = act[q] ? Tin[grow * ld2 + off[q]] : make_double2(0.0, 0.0);
    s[2 * q] = add_rn(acc[q].x, mul_rn(a.w, own.x));          // + w*s/colsums  (exact for w=1)
    s[2 * q + 1] = add_rn(acc[q].y, mul_rn(a.w, own.y));
  }
  finish_row<2 * NQ2, ColPair, FL>(a, row, grow, lane, s);
}

// The same step on a state stored in 4 bytes per entry (cna_set_state_f32): a lane owns FOUR adjacent columns and one
// 16-byte load fetches them; every entry is widened to f64 and enters a fused multiply-add, in the same CSR order.
// Half the bytes per edge through the vector L1 and from behind the L2 -- the two limits of k_nam_step.
template <typename VT, int NQ4, int U = 8, int FL = 0>
__global__ __launch_bounds__(256) void k_nam_step32(StepArgs a) {
  if (STEP_STOPPED(a)) return;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t row = my_row(wv, a.xcd_chunk);
  if (row >= a.n_local) return;
  const int64_t grow = a.row0 + row;
  const int64_t start = uniform64(a.indptr[row]), end = uniform64(a.indptr[row + 1]);
  const float4* __restrict__ Tin = (const float4*)a.Tin;
  const int ld4 = a.ld >> 2;
  // lanes past the row width fetch the row's first 16 bytes (same cache line as lane 0's) and their sums are never read
  // (finish_row looks at columns below ld only): with the load under a condition the compiler puts the widening to f64
  // next to it and waits for every row before it asks for the next one
  unsigned off[NQ4];
#pragma unroll
  for (int q = 0; q < NQ4; ++q) off[q] = lane + 64 * q < ld4 ? (unsigned)(lane + 64 * q) : 0u;
  double acc[NQ4][4];
#pragma unroll
  for (int q = 0; q < NQ4; ++q)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[q][i] = 0.0;
  // (fused: nothing here is bit-identical to the 8-byte walk anyway.  Measured at 2M x 200, HISTORY.md "Round 5": 5.5 ms;
  // with the arithmetic removed 5.3 ms -- 800-byte rows touch 7.25 lines of 128 bytes, the vector L1 passes ~30 bytes of
  // LINES per clock --; with every gather aimed at one resident row 4.6 ms: both sides are within 15 % of the launch)
  auto add_row = [&](int q, double av, const float4& t) {
    acc[q][0] = __builtin_fma(av, (double)t.x, acc[q][0]);
    acc[q][1] = __builtin_fma(av, (double)t.y, acc[q][1]);
    acc[q][2] = __builtin_fma(av, (double)t.z, acc[q][2]);
    acc[q][3] = __builtin_fma(av, (double)t.w, acc[q][3]);
  };
  // (batches as in k_nam_step: the partial batch at the end of a row asks for the row's own state in its empty slots)
  auto batch = [&](auto bc, auto pc, int jl, double al, int l0, int nvalid) {
    constexpr int B = decltype(bc)::value;
    constexpr bool PAD = decltype(pc)::value;
    float4 t[B][NQ4];
#pragma unroll
    for (int u = 0; u < B; ++u) {
      int64_t j = __builtin_amdgcn_readlane(jl, (l0 + u) & 63);
      if (PAD && u >= nvalid) j = grow;
      const float4* __restrict__ rowp = Tin + j * ld4;
#pragma unroll
      for (int q = 0; q < NQ4; ++q) t[u][q] = rowp[off[q]];
    }
    // (all rows requested before the first is consumed: left alone the scheduler starts widening row 0 after three
    // requests and waits for each of them with vmcnt(0))
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < B; ++u) {
      if (!PAD || u < nvalid) {
        const double av = readlane_d(al, (l0 + u) & 63);
#pragma unroll
        for (int q = 0; q < NQ4; ++q) add_row(q, av, t[u][q]);
      }
    }
  };
  for (int64_t base = start; base < end; base += 64) {
    int jl;
    double al;
    load_edges<VT>(a, base, end, lane, jl, al);
    const int cnt = (int)((end - base) < 64 ? (end - base) : 64);
    int l = 0;
    for (; l + U <= cnt; l += U) batch(std::integral_constant<int, U>{}, std::false_type{}, jl, al, l, U);
    const int rem = cnt - l;
    if (rem > (U + 1) / 2) batch(std::integral_constant<int, U>{}, std::true_type{}, jl, al, l, rem);
    else if (rem > 0) batch(std::integral_constant<int, (U + 1) / 2>{}, std::true_type{}, jl, al, l, rem);
  }
  double s[4 * NQ4];
#pragma unroll
  for (int q = 0; q < NQ4; ++q) {
    const float4 own = Tin[grow * ld4 + off[q]];
    s[4 * q] = __builtin_fma(a.w, (double)own.x, acc[q][0]);
    s[4 * q + 1] = __builtin_fma(a.w, (double)own.y, acc[q][1]);
    s[4 * q + 2] = __builtin_fma(a.w, (double)own.z, acc[q][2]);
    s[4 * q + 3] = __builtin_fma(a.w, (double)own.w, acc[q][3]);
  }
  finish_row<4 * NQ4, ColQuad, FL>(a, row, grow, lane, s);
}

// ... and for rows of at most 128 columns (32 lanes of four): TWO edges of the row per wave instruction, one per
// half-wave -- the instruction stream per edge (one load, four widenings, four fused multiply-adds) is what bounds the
// kernel above, and a 100-column row leaves 39 of its 64 lanes idle.  Lanes 0-31 take the even edges of a batch, lanes
// 32-63 the odd ones (index and weight reach them through the LDS crossbar, ds_bpermute: no VALU time); the two partial
// sums of a column meet at the end.  Byte offsets into the state are 32 bits wide (launch_step_q checks).
__device__ __forceinline__ double bperm_d(int byte_idx, double v) {
  const int lo = __builtin_amdgcn_ds_bpermute(byte_idx, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(byte_idx, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
template <typename VT, int U = 8, int FL = 0>
__global__ __launch_bounds__(256) void k_nam_step32h(StepArgs a) {
  if (STEP_STOPPED(a)) return;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t row = my_row(wv, a.xcd_chunk);
  if (row >= a.n_local) return;
  const int64_t grow = a.row0 + row;
  const int64_t start = uniform64(a.indptr[row]), end = uniform64(a.indptr[row + 1]);
  const char* __restrict__ Tin = (const char*)a.Tin;
  const int half = lane >> 5, hl = lane & 31;
  const unsigned row_bytes = (unsigned)a.ld * 4u;
  const unsigned off = (hl < (a.ld >> 2) ? (unsigned)hl : 0u) * 16u;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  auto add_row = [&](double av, const float4& t) {
    acc[0] = __builtin_fma(av, (double)t.x, acc[0]);
    acc[1] = __builtin_fma(av, (double)t.y, acc[1]);
    acc[2] = __builtin_fma(av, (double)t.z, acc[2]);
    acc[3] = __builtin_fma(av, (double)t.w, acc[3]);
  };
  // B PAIRS of edges requested together (see k_nam_step: the partial batch at the end of a row costs one latency)
  auto batch = [&](auto bc, auto pc, int jl, double al, int p0, int nvalid) {
    constexpr int B = decltype(bc)::value;
    constexpr bool PAD = decltype(pc)::value;
    float4 t[B];
    double av[B];
#pragma unroll
    for (int u = 0; u < B; ++u) {
      const int src = 4 * ((2 * (p0 + u) + half) & 63);
      unsigned j = (unsigned)__builtin_amdgcn_ds_bpermute(src, jl);
      av[u] = bperm_d(src, al);
      if (PAD && u >= nvalid) j = (unsigned)grow;
      t[u] = *(const float4*)(Tin + (size_t)(j * row_bytes + off));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < B; ++u) {
      if (!PAD || u < nvalid) add_row(av[u], t[u]);
    }
  };
  for (int64_t base = start; base < end; base += 64) {
    int jl;
    double al;
    load_edges<VT>(a, base, end, lane, jl, al);            // (lanes past the end: neighbour 0 with weight 0)
    const int cnt = (int)((end - base) < 64 ? (end - base) : 64);
    const int npair = (cnt + 1) >> 1;
    int p = 0;
    for (; p + U <= npair; p += U) batch(std::integral_constant<int, U>{}, std::false_type{}, jl, al, p, U);
    const int rem = npair - p;
    if (rem > (U + 1) / 2) batch(std::integral_constant<int, U>{}, std::true_type{}, jl, al, p, rem);
    else if (rem > 0) batch(std::integral_constant<int, (U + 1) / 2>{}, std::true_type{}, jl, al, p, rem);
  }
  // the odd edges' sums join the even edges' (lane l <- lane l + 32), then the row's own term
  double s[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) s[i] = acc[i] + bperm_d(4 * ((lane + 32) & 63), acc[i]);
  const float4 own = *(const float4*)(Tin + ((size_t)grow * row_bytes + off));
  s[0] = __builtin_fma(a.w, (double)own.x, s[0]);
  s[1] = __builtin_fma(a.w, (double)own.y, s[1]);
  s[2] = __builtin_fma(a.w, (double)own.z, s[2]);
  s[3] = __builtin_fma(a.w, (double)own.w, s[3]);
  // (lanes 32-63 now hold copies of columns 4 hl ...: finish_row reads a lane's columns as 4 lane + k, i.e. >= 128 >= ld
  // for them, and skips them)
  finish_row<4, ColQuad, FL>(a, row, grow, lane, s);
}

// Narrow states (ld <= 64 columns, i.e. at most 32 column pairs): TWO destination rows per wave, one
// per half-wave.  The vector-memory path spends ~16 clk per wave instruction whatever its width or
// exec mask (tools/micro/gather_pair.hip: 8.5 -> 12.6 TB/s on 400-byte rows), so a row that fills
// only 25 lanes wastes most of it; here one load instruction fetches a neighbour row for each of the
// two destination rows.  The neighbour (index, weight) records of both rows sit in LDS, 16 bytes
// each, and a half-wave reads its row's record with one broadcast ds_read_b128.  Every row still
// adds its own products one by one in CSR order: same bits as k_nam_step.
struct alignas(16) EdgeRec {
  double a;
  unsigned byte_off;    // j * row bytes
  unsigned lane_mask;   // ~0 for an edge, 0 for padding past the end of the row
};

__device__ __forceinline__ double half_sum(double v, int h) {
  v = dpp_add(v, 0);
  v = dpp_add(v, 1);
  v = dpp_add(v, 2);
  v = dpp_add(v, 3);
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
  const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
  const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
  const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
  return h ? (r2 + r3) : (r0 + r1);           // what wave_sum gives when the other half holds zeros
}

template <typename VT, int FL = 0>
__global__ __launch_bounds__(256) void k_nam_step_pair(StepArgs a) {
  if (STEP_STOPPED(a)) return;
  constexpr int U = 4;                          // neighbour rows in flight per half-wave (8: +2 % time, 16: +16 %)
  __shared__ EdgeRec recs[4][2][32];
  const int lane = threadIdx.x & 63, hl = lane & 31, h = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t row_a = my_row(wv, a.xcd_chunk) * 2;          // this wave: rows row_a, row_a + 1
  if (row_a >= a.n_local) return;
  const bool have = row_a + h < a.n_local;                    // odd n_local: the last wave's upper half idles
  const int64_t row = (FL & 2) != 0 ? (int64_t)a.rows[have ? row_a + h : row_a] : (have ? row_a + h : row_a);
  const int64_t grow = a.row0 + row;
  const int64_t start = a.indptr[row];
  const int deg = have ? (int)(a.indptr[row + 1] - start) : 0;
  const int d0 = __builtin_amdgcn_readlane(deg, 0), d1 = __builtin_amdgcn_readlane(deg, 32);
  const int degmax = d0 > d1 ? d0 : d1;
  const int ld2 = a.ld >> 1;
  const unsigned rowbytes = (unsigned)a.ld * 8u;
  const char* __restrict__ Tb = (const char*)a.Tin;
  const bool act = hl < ld2;
  const unsigned off = act ? (unsigned)hl * 16u : 0u;
  EdgeRec* mine = recs[wv][h];
  const unsigned own_off = (unsigned)grow * rowbytes;         // < 4 GiB (checked by the launcher)
  double2 acc = make_double2(0.0, 0.0);
  for (int base = 0; base < degmax; base += 32) {
    {
      const bool ok = base + hl < deg;
      const int64_t e = start + base + hl;
      EdgeRec r;
      // past the end of the row: weight 0 on the first 16 bytes of the row's own (finite) state row
      // -- all lanes of the half on one address, a single L1 access -- so the gather below needs no
      // per-edge branch and padding costs next to nothing in the vector-memory pipeline
      r.a = ok ? (double)((const VT*)a.val)[e] : 0.0;
      r.byte_off = ok ? (unsigned)a.idx[e] * rowbytes : own_off;
      r.lane_mask = ok ? ~0u : 0u;
      *(uint4*)&mine[hl] = *(const uint4*)&r;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const int left = degmax - base;
    const int cnt = left < 32 ? left : 32;                    // scalar: edges of the longer row in this batch
    for (int l = 0; l < cnt; l += U) {
      double2 t[U];
      double w[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint4 r = *(const uint4*)&mine[l + u];
        w[u] = __hiloint2double((int)r.y, (int)r.x);
        t[u] = *(const double2*)(Tb + (r.z + (off & r.w)));   // idle column lanes re-read column pair 0
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {                           // past a row's end: acc + 0 * finite = acc
        acc.x = add_rn(acc.x, mul_rn(w[u], t[u].x));
        acc.y = add_rn(acc.y, mul_rn(w[u], t[u].y));
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
  if (!have) return;
  const double2 own = act ? *(const double2*)(Tb + ((uint64_t)grow * rowbytes + off)) : make_double2(0.0, 0.0);
  double s[2];
  s[0] = add_rn(acc.x, mul_rn(a.w, own.x));
  s[1] = add_rn(acc.y, mul_rn(a.w, own.y));
  // write-out: finish_row for a half-wave (columns 2*hl, 2*hl + 1)
  const double cs = a.colsum[grow];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int col = 2 * hl + k;
    if (col < a.ld) {
      const bool in = col < a.width;
      if (a.write_t) a.Tout[grow * a.ld + col] = in ? __ddiv_rn(s[k], cs) : 0.0;
      if (a.dense_out) a.dense_out[row * a.ld + col] = in ? s[k] : 0.0;
    }
  }
  if (a.write_nam || a.want_kurt) {
    double x[2];
    double sum = 0.0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int col = 2 * hl + k;
      const bool in = col < a.width;
      x[k] = in ? __ddiv_rn(s[k], a.counts[in ? col : 0]) : 0.0;
      if (a.write_nam && col < a.ld) a.nam[row * a.ld + col] = x[k];
      sum += x[k];
    }
    if (a.want_kurt) {
      const double n = (double)a.width;
      const double mean = half_sum(sum, h) / n;
      double d2s = 0.0, d4s = 0.0;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        if (2 * hl + k < a.width) {
          const double d = x[k] - mean;
          const double d2 = d * d;
          d2s += d2;
          d4s += d2 * d2;
        }
      }
      const double m2 = half_sum(d2s, h) / n;
      const double m4 = half_sum(d4s, h) / n;
      const double em = 2.220446049250313e-16 * mean;
      const double k4 = (m2 <= em * em) ? __builtin_nan("") : m4 / (m2 * m2);
      if (hl == 0) a.stat[grow] = k4 - 3.0;
    }
  }
}

// Second step on the compressed state: the same sums in the same (CSR) order as k_nam_step -- the
// zeros it skips contribute +0 there -- on a fraction of the bytes (~40 x 10 B instead of 8N B per
// edge; at N = 200 a third of the cache lines).  Lane l of an edge adds weight x value_l into column
// sample_l of the wave's LDS accumulator row with ds_add_f64: distinct columns inside one edge, the
// LDS serves one wave's instructions in program order, and the LDS f64 add rounds like v_add_f64
// (tools/micro/lds_add_rounding.hip: 4M single adds and 260k chains of 40, no difference), so the
// result is bit-identical to k_nam_step.  Rows that overflowed the compressed form (more than SP_CAP
// distinct samples) are taken dense.
// (a plain read - add - write of the wave's own row is legal too and slower in the kernel: 4.61 against 4.42 ms;
// two rows per wave side by side, as in k_nam_first2, is slower as well: 4.48 against 4.10 ms at 2M x 200,
// 2.13 against 1.93 at 1M x 100 -- 65 VGPRs, seven waves per SIMD)
__device__ __forceinline__ void lds_add(double* p, double v) { unsafeAtomicAdd(p, v); }

// U, the edges of a batch whose pairs are in flight together: 5 since round 4; before that 6 (us per launch at 2M x 200 / 1M x 100, from the
// averages of profiles/r02_kbench_sparse_u.txt: U = 3: 4.41 / 2.12, 4: 4.20 / 2.03, 6: 4.10 / 1.95, 8: 4.37 / 2.10,
// 12: 4.9 / 2.2, 16: 5.0 / 2.3; the ragged end of a row as one more, predicated batch instead of edge by edge:
// slower at every U).  Round 4, after the batches of edges without an overflowed neighbour got a loop of their own: U = 3:
// 3.98 / 1.79 ms (2M x 200 / 1M x 100), 4: 3.86 / 1.71, 5: 3.85 / 1.67, 6: 4.02 / 1.67, 8: 4.39 / 1.96 -- five.  Round 3: the bank conflicts of the scatter are NOT what the step waits for -- with every lane
// adding into its own column (conflict-free, wrong sums: a timing experiment) the launch goes from 4.45 to 4.25 ms at
// 2M x 200, 1.86 -> 1.78 at 1M x 100; a pair order that spreads a row's columns over the banks
// (tools/micro/lds_scatter_pattern.hip: 14.3 -> 11.7 clk per ds_add_f64, floor 7.4) is therefore not worth its
// bookkeeping.  Counters (profiles/r03_pmc_summary_C4.txt): VALU 42 % and LDS 49 % of the cycles, 61 % of the wave
// cycles waiting, 12.5 vector instructions per edge: no single unit is the limit.
template <typename VT, int NQ2, int U = 5, int FL = 0>
__global__ __launch_bounds__(256) void k_nam_step_sparse(StepArgs a) {
  if (STEP_STOPPED(a)) return;
  extern __shared__ double sm[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double* acc = sm + (size_t)wv * 128 * NQ2;
#pragma unroll
  for (int q = 0; q < 2 * NQ2; ++q) acc[lane + 64 * q] = 0.0;
  int64_t row = my_row(wv, a.xcd_chunk);
  if (row >= a.n_local) return;
  if ((FL & 2) != 0) row = uniform64(a.rows[row]);
  const int64_t grow = a.row0 + row;
  const int64_t start = uniform64(a.indptr[row]), end = uniform64(a.indptr[row + 1]);
  const double2* __restrict__ Tin = (const double2*)a.Tin;
  const SpPair* __restrict__ pairs = a.sp_pair;
  const int ld2 = a.ld >> 1;
  auto dense_edge = [&](int j, double av) {
    const double2* __restrict__ rowp = Tin + (int64_t)j * ld2;
#pragma unroll
    for (int q = 0; q < NQ2; ++q) {
      const int c2 = lane + 64 * q;
      if (c2 < ld2) {
        const double2 t = rowp[c2];
        lds_add(&acc[2 * c2], mul_rn(av, t.x));
        lds_add(&acc[2 * c2 + 1], mul_rn(av, t.y));
      }
    }
  };
  for (int64_t base = start; base < end; base += 64) {
    int jl;
    double al;
    load_edges<VT>(a, base, end, lane, jl, al);
    const int cl = (base + lane < end) ? (int)a.sp_cnt[jl] : 0;
    const int cnt = (int)((end - base) < 64 ? (end - base) : 64);
    int l = 0;
    // (no neighbour of these 64 edges overflowed its pairs -- the rule on one GPU: the batches then carry no test for it)
    if (__ballot(cl == SP_DENSE) == 0ull) {
      for (; l + U <= cnt; l += U) {
        int cc[U];
        SpPair sp[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int j = __builtin_amdgcn_readlane(jl, l + u);
          cc[u] = __builtin_amdgcn_readlane(cl, l + u);
          if (lane < cc[u]) sp[u] = pairs[(int64_t)j * SP_CAP + lane];       // (used under the same condition only)
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const double av = readlane_d(al, l + u);
          if (lane < cc[u]) lds_add(&acc[sp[u].col], mul_rn(av, sp[u].v));
        }
      }
    }
    for (; l + U <= cnt; l += U) {
      int cc[U];
      SpPair sp[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int j = __builtin_amdgcn_readlane(jl, l + u);
        cc[u] = __builtin_amdgcn_readlane(cl, l + u);
        sp[u] = SpPair{0.0, 0, 0};
        if (cc[u] != SP_DENSE && lane < cc[u]) sp[u] = pairs[(int64_t)j * SP_CAP + lane];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const double av = readlane_d(al, l + u);
        if (cc[u] == SP_DENSE) dense_edge(__builtin_amdgcn_readlane(jl, l + u), av);
        else if (lane < cc[u]) lds_add(&acc[sp[u].col], mul_rn(av, sp[u].v));
      }
    }
    for (; l < cnt; ++l) {                       // ragged tail
      const int j = __builtin_amdgcn_readlane(jl, l);
      const int c = __builtin_amdgcn_readlane(cl, l);
      const double av = readlane_d(al, l);
      if (c == SP_DENSE) dense_edge(j, av);
      else if (lane < c) {
        const SpPair p = pairs[(int64_t)j * SP_CAP + lane];
        lds_add(&acc[p.col], mul_rn(av, p.v));
      }
    }
  }
  // + w*s/colsums of the row itself, after the neighbours (the order of scipy's A.dot(s) + s): from its own
  // pairs (adding w * 0 elsewhere would change nothing: the sums are non-negative), or from its dense row
  // when it overflowed the pairs
  const int own_cnt = (int)a.sp_cnt[grow];
  if (own_cnt != SP_DENSE) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (lane < own_cnt) {
      const SpPair p = pairs[grow * SP_CAP + lane];
      lds_add(&acc[p.col], mul_rn(a.w, p.v));
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  double s[2 * NQ2];
#pragma unroll
  for (int q = 0; q < NQ2; ++q) {
    const int c2 = lane + 64 * q;
    const bool act = c2 < ld2;
    if (own_cnt == SP_DENSE) {
      const double2 own = act ? Tin[grow * ld2 + c2] : make_double2(0.0, 0.0);
      s[2 * q] = add_rn(act ? acc[2 * c2] : 0.0, mul_rn(a.w, own.x));
      s[2 * q + 1] = add_rn(act ? acc[2 * c2 + 1] : 0.0, mul_rn(a.w, own.y));
    } else {
      s[2 * q] = act ? acc[2 * c2] : 0.0;
      s[2 * q + 1] = act ? acc[2 * c2 + 1] : 0.0;
    }
  }
  finish_row<2 * NQ2, ColPair, FL>(a, row, grow, lane, s);
}

__global__ void k_cellinfo(const double* __restrict__ colsum, const int32_t* __restrict__ sid, int64_t n,
                           CellInfo* __restrict__ info) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    CellInfo ci;
    ci.inv_colsum = __ddiv_rn(1.0, colsum[i]);      // s/colsums of the one-hot state (_nam.py:33)
    ci.sid = sid[i];
    ci.pad = 0;
    info[i] = ci;
  }
}

// The same records for the rows of the compact state (cna_ctx::t_compact): row t is this rank's row row0 + t, or the
// (t - n_local)-th row of the receive list.
__global__ void k_cellinfo_compact(const double* __restrict__ colsum, const int32_t* __restrict__ sid, int64_t row0, int64_t n_local,
                                   const int64_t* __restrict__ recv_rows, int64_t t_rows, CellInfo* __restrict__ info) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < t_rows) {
    const int64_t i = t < n_local ? row0 + t : recv_rows[t - n_local];
    CellInfo ci;
    ci.inv_colsum = __ddiv_rn(1.0, colsum[i]);
    ci.sid = sid[i];
    ci.pad = 0;
    info[t] = ci;
  }
}

// flags[row] = the row has a neighbour outside this rank's block (its step needs rows another rank sends)
__global__ void k_rows_need_halo(const int64_t* __restrict__ indptr, const int32_t* __restrict__ idx, int64_t n_local, int64_t row0,
                                 unsigned char* __restrict__ flags) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_local) return;
  unsigned char f = 0;
  for (int64_t e = indptr[r]; e < indptr[r + 1] && !f; ++e) {
    const int64_t j = idx[e];
    f = (j < row0 || j >= row0 + n_local) ? 1 : 0;
  }
  flags[r] = f;
}

// Column indices (global rows) -> rows of the compact state: a local column is its row in the block, any other column its
// position in the ascending receive list, behind the block.  *bad: some index is neither (the halo plan does not cover
// the graph block -- cna_set_halo refuses it).
__global__ void k_remap_idx(const int32_t* __restrict__ idx, int64_t nnz, int64_t row0, int64_t n_local,
                            const int64_t* __restrict__ recv_rows, int64_t nr, int32_t* __restrict__ out, int* __restrict__ bad) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = idx[e];
    if (j >= row0 && j < row0 + n_local) { out[e] = (int32_t)(j - row0); continue; }
    int64_t lo = 0, hi = nr;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (recv_rows[mid] < j) lo = mid + 1; else hi = mid;
    }
    if (lo < nr && recv_rows[lo] == j) out[e] = (int32_t)(n_local + lo);
    else { out[e] = 0; *bad = 1; }
  }
}

// ---- column sums, deterministic: colsums = A.sum(axis=0) (_nam.py:28) --------------------------------
// scipy sums a CSR over axis 0 by walking the rows in ascending order and adding every entry into its
// column's accumulator: column j receives its entries in ascending (row, position in row) order.  In
// float64 that order matters as soon as the weights of a column span more than 2^29 in magnitude (at 2M
// cells some do), so the sums here are formed in exactly that order, without float atomics:
//   k_col_count    in-degree of every column (integer atomics: the result is order-free)
//   k_scan_*       exclusive prefix sum -> first slot of every column
//   k_col_scatter  every edge drops {key = caller's row << 32 | position in row, value} into a slot of its
//                  column (integer atomic cursor: WHICH slot is arbitrary, the set per column is not)
//   k_col_sum      one wave per column ranks the column's keys and adds the values by rank
// Rows keep the caller's row index (c->orig_idx) and their entries the caller's order, whatever the
// device numbering.  Sharded runs sum their own rows this way and all-reduce the partial sums.
__global__ void k_col_count(const int32_t* __restrict__ idx, int64_t nnz, unsigned int* __restrict__ cnt) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += stride) atomicAdd(&cnt[idx[e]], 1u);
}

constexpr int SCAN_TILE = 1024;           // elements per workgroup of 256 threads
__global__ __launch_bounds__(256) void k_scan_tiles(const unsigned int* __restrict__ cnt, int64_t n,
                                                    unsigned long long* __restrict__ tile_sum) {
  __shared__ unsigned long long part[4];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + threadIdx.x * 4;
  unsigned long long s = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) s += base + k < n ? cnt[base + k] : 0u;
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) tile_sum[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
__global__ __launch_bounds__(1024) void k_scan_tile_sums(unsigned long long* __restrict__ tile_sum, int64_t ntiles) {
  // one workgroup, exclusive scan in place, 1024 tiles per round
  __shared__ unsigned long long sm[1024];
  __shared__ unsigned long long carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < ntiles; base += 1024) {
    const int64_t i = base + threadIdx.x;
    const unsigned long long v = i < ntiles ? tile_sum[i] : 0ull;
    sm[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const unsigned long long a = threadIdx.x >= o ? sm[threadIdx.x - o] : 0ull;
      __syncthreads();
      sm[threadIdx.x] += a;
      __syncthreads();
    }
    const unsigned long long c0 = carry;
    if (i < ntiles) tile_sum[i] = c0 + sm[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry = c0 + sm[1023];
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void k_scan_finish(const unsigned int* __restrict__ cnt, int64_t n,
                                                     const unsigned long long* __restrict__ tile_sum,
                                                     unsigned long long* __restrict__ first /* n + 1 */) {
  __shared__ unsigned long long part[4];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + threadIdx.x * 4;
  unsigned long long v[4], s = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) { v[k] = base + k < n ? cnt[base + k] : 0u; s += v[k]; }
  unsigned long long incl = s;                 // inclusive scan of the per-thread sums inside the wave
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long a = __shfl_up(incl, o);
    if ((int)(threadIdx.x & 63) >= o) incl += a;
  }
  if ((threadIdx.x & 63) == 63) part[threadIdx.x >> 6] = incl;
  __syncthreads();
  unsigned long long off = tile_sum[blockIdx.x] + incl - s;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) off += part[w];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (base + k < n) first[base + k] = off;
    off += v[k];
    if (base + k == n - 1) first[n] = off;
  }
}

struct ColEntry { unsigned long long key; double val; };

template <typename VT>
__global__ __launch_bounds__(256) void k_col_scatter(const int64_t* __restrict__ indptr, const int32_t* __restrict__ idx,
                                                     const VT* __restrict__ val, const int64_t* __restrict__ orig,
                                                     int64_t n_local, int64_t row0_key,
                                                     const unsigned long long* __restrict__ first,
                                                     unsigned int* __restrict__ cursor, ColEntry* __restrict__ ent) {
  const int lane = threadIdx.x & 63;
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < n_local; row += nw) {
    const int64_t start = indptr[row], end = indptr[row + 1];
    const unsigned long long rk = (unsigned long long)(orig ? orig[row] : row0_key + row) << 32;
    for (int64_t e = start + lane; e < end; e += 64) {
      const int32_t j = idx[e];
      const unsigned long long slot = first[j] + atomicAdd(&cursor[j], 1u);
      ColEntry ce;
      ce.key = rk | (unsigned long long)(e - start);
      ce.val = (double)val[e];
      ent[slot] = ce;
    }
  }
}

__global__ __launch_bounds__(256) void k_col_sum(const unsigned long long* __restrict__ first,
                                                 const ColEntry* __restrict__ ent, int64_t n, double* __restrict__ colsum) {
  __shared__ double sorted[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t j = (int64_t)blockIdx.x * 4 + wv; j < n; j += nw) {
    const unsigned long long lo = first[j];
    const int64_t cnt = (int64_t)(first[j + 1] - lo);
    double s = 0.0;
    if (cnt <= 64) {
      // rank = number of smaller keys (keys of a column are distinct); values to LDS by rank, summed in order
      const bool ok = lane < cnt;
      const ColEntry me = ok ? ent[lo + lane] : ColEntry{~0ull, 0.0};
      const unsigned klo = (unsigned)me.key, khi = (unsigned)(me.key >> 32);
      int rank = 0;
      for (int m = 0; m < (int)cnt; ++m) {
        const unsigned long long km = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)khi, m) << 32) |
                                      (unsigned)__builtin_amdgcn_readlane((int)klo, m);
        rank += km < me.key ? 1 : 0;
      }
      if (ok) sorted[wv][rank] = me.val;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      for (int r = 0; r < (int)cnt; ++r) s = add_rn(s, sorted[wv][r]);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    } else {
      // a hub column: repeatedly take the smallest key above the last one taken (cnt^2 / 64 loads)
      unsigned long long last = 0;
      bool any = false;
      for (int64_t r = 0; r < cnt; ++r) {
        unsigned long long best = ~0ull;
        double bv = 0.0;
        for (int64_t e = lane; e < cnt; e += 64) {
          const ColEntry ce = ent[lo + e];
          if ((!any || ce.key > last) && ce.key < best) { best = ce.key; bv = ce.val; }
        }
        for (int o = 32; o > 0; o >>= 1) {
          const unsigned long long ob = __shfl_xor(best, o);
          const double ov = __shfl_xor(bv, o);
          if (ob < best) { best = ob; bv = ov; }
        }
        s = add_rn(s, bv);
        last = best;
        any = true;
      }
    }
    if (lane == 0) colsum[j] = s;
  }
}

__global__ void k_add_scalar(double* v, int64_t n, double s) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] += s;
}

__global__ void k_scale_rows(const double* __restrict__ s, const double* __restrict__ colsum,
                             double* T, double* dense, int64_t n_local, int64_t row0, int m, int ld) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_local * ld) return;
  const int64_t row = i / ld;
  const int col = (int)(i - row * ld);
  const double v = col < m ? s[row * m + col] : 0.0;
  dense[i] = v;
  T[(row0 + row) * ld + col] = col < m ? __ddiv_rn(v, colsum[row0 + row]) : 0.0;
}

template <typename VT, int NQ>
int launch_first_t(cna_ctx* c, const StepArgs& a, dim3 grid, hipStream_t st) {
  const bool two = (grid.x & 15) == 0;     // an even number of workgroups per XCD: one workgroup takes two of them
  const size_t lds = sizeof(double) * (two ? 8 : 4) * 64 * NQ;
  const CellInfo* info = (const CellInfo*)c->cellinfo;
  if (a.rows) {
    if (two) hipLaunchKernelGGL((k_nam_first2<VT, NQ, 2, 2>), dim3(grid.x / 2), dim3(256), lds, st, a, info);
    else hipLaunchKernelGGL((k_nam_first<VT, NQ, 2>), grid, dim3(256), lds, st, a, info);
  } else {
    if (two) hipLaunchKernelGGL((k_nam_first2<VT, NQ, 2, 0>), dim3(grid.x / 2), dim3(256), lds, st, a, info);
    else hipLaunchKernelGGL((k_nam_first<VT, NQ, 0>), grid, dim3(256), lds, st, a, info);
  }
  return 0;
}
// the instantiation of a step kernel by what the launch needs (finish_row): row list, selection by-product, or neither
template <typename VT, int NQ2, int U = 8>
int launch_step_t(cna_ctx* c, const StepArgs& a, dim3 grid, hipStream_t st) {
  // (resident workgroups per CU, limited by a dynamic LDS allocation: 6 -> 5 changes nothing, 4 costs 4.5 %, 2 costs 47 %:
  // profiles/r04_ab_gram_overlap.txt)
  const size_t pad = 0;
  if (a.rows && a.sel_X) hipLaunchKernelGGL((k_nam_step<VT, NQ2, U, 3>), grid, dim3(256), pad, st, a);   // (row list AND by-product: the last step of a sharded walk, in two launches)
  else if (a.rows) hipLaunchKernelGGL((k_nam_step<VT, NQ2, U, 2>), grid, dim3(256), pad, st, a);
  else if (a.sel_X) hipLaunchKernelGGL((k_nam_step<VT, NQ2, U, 1>), grid, dim3(256), pad, st, a);
  else hipLaunchKernelGGL((k_nam_step<VT, NQ2, U, 0>), grid, dim3(256), pad, st, a);
  return 0;
}

template <typename VT, int NQ4, int U = 8>
int launch_step32_t(cna_ctx* c, const StepArgs& a, dim3 grid, hipStream_t st) {
  if (a.rows) CNA_FAIL(CNA_ESTATE, "4-byte state with a row list");          // (launch_nam_step never produces one)
  if (a.sel_X && a.out32) hipLaunchKernelGGL((k_nam_step32<VT, NQ4, U, 5>), grid, dim3(256), 0, st, a);
  else if (a.sel_X) hipLaunchKernelGGL((k_nam_step32<VT, NQ4, U, 1>), grid, dim3(256), 0, st, a);
  else if (a.out32) hipLaunchKernelGGL((k_nam_step32<VT, NQ4, U, 4>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((k_nam_step32<VT, NQ4, U, 0>), grid, dim3(256), 0, st, a);
  return 0;
}

#ifndef CNA_STEP32_U          // rows (row pairs) in flight per wave of the 4-byte steps; experiments: make EXTRA=-DCNA_STEP32_U=12
#define CNA_STEP32_U 8
#endif
template <typename VT>
int launch_step32h_t(cna_ctx* c, const StepArgs& a, dim3 grid, hipStream_t st) {
  if (a.rows) CNA_FAIL(CNA_ESTATE, "4-byte state with a row list");
  constexpr int U = CNA_STEP32_U;
  if (a.sel_X && a.out32) hipLaunchKernelGGL((k_nam_step32h<VT, U, 5>), grid, dim3(256), 0, st, a);
  else if (a.sel_X) hipLaunchKernelGGL((k_nam_step32h<VT, U, 1>), grid, dim3(256), 0, st, a);
  else if (a.out32) hipLaunchKernelGGL((k_nam_step32h<VT, U, 4>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((k_nam_step32h<VT, U, 0>), grid, dim3(256), 0, st, a);
  return 0;
}

template <typename VT, int NQ2>
int launch_step_sparse_t(cna_ctx* c, const StepArgs& a, dim3 grid, hipStream_t st) {
  const size_t lds = sizeof(double) * 4 * 128 * NQ2;
  if (a.out32) {
    if (a.rows) CNA_FAIL(CNA_ESTATE, "4-byte state with a row list");
    if (a.sel_X) hipLaunchKernelGGL((k_nam_step_sparse<VT, NQ2, 5, 5>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((k_nam_step_sparse<VT, NQ2, 5, 4>), grid, dim3(256), lds, st, a);
    return 0;
  }
  if (a.rows && a.sel_X) hipLaunchKernelGGL((k_nam_step_sparse<VT, NQ2, 5, 3>), grid, dim3(256), lds, st, a);
  else if (a.rows) hipLaunchKernelGGL((k_nam_step_sparse<VT, NQ2, 5, 2>), grid, dim3(256), lds, st, a);
  else if (a.sel_X) hipLaunchKernelGGL((k_nam_step_sparse<VT, NQ2, 5, 1>), grid, dim3(256), lds, st, a);
  else hipLaunchKernelGGL((k_nam_step_sparse<VT, NQ2, 5, 0>), grid, dim3(256), lds, st, a);
  return 0;
}

// two rows per wave when a row fits a half-wave and byte offsets into the state fit 32 bits
static bool step_takes_pairs(const cna_ctx* c, bool first, bool sparse, int ld) {
  return !first && !sparse && ld <= 64 && (c->t_compact ? c->t_rows : c->n_pad) * (int64_t)ld * 8 < (int64_t)4 << 30 &&
         !getenv("CNA_STEP_WIDE");            // (test_two_rows_per_wave_step_matches_wave_per_row compares the two kernels)
}
// consecutive workgroups (4 rows each, 8 for the two-rows-per-wave kernel) one XCD takes per turn
static int64_t step_xcd_chunk(bool pair, int ld, int64_t nblk) {
  const int64_t cpx = (nblk + 7) / 8;
  // measured (tools/kbench.py): one contiguous eighth per XCD is best at 200k x 50 (+4 % over
  // round-robin) and within 2 % of every chunk size at 1M x 100
  int64_t chunk = cpx;
  // wide states (wave-per-row kernels): 128 workgroups = 512 consecutive rows = one cluster of the device
  // order (cna_amd/_order.py) per XCD at a time -- the cluster's neighbour rows then stay in that XCD's L2
  // (2M x 200, dense step: 10.9 ms with one contiguous eighth per XCD under RCM, 7.8 ms this way)
  if (!pair && ld > 64 && cpx > 128) chunk = 128;
  if (const char* e = getenv("CNA_XCD_CHUNK")) { const int64_t v = atoll(e); if (v > 0 && v < cpx) chunk = v; }   // experiments (tools/kbench_order.py)
  return chunk;
}

template <typename VT>
int launch_step_q(cna_ctx* c, bool first, const StepArgs& a_in, hipStream_t st) {
  StepArgs a = a_in;
  const bool pair = step_takes_pairs(c, first, a.sp_cnt != nullptr, a.ld);
  const int64_t nblk = pair ? (a.n_local + 7) / 8 : (a.n_local + 3) / 4;
  int64_t cpx = (nblk + 7) / 8;
  const int64_t chunk = step_xcd_chunk(pair, a.ld, nblk);
  cpx = (cpx + chunk - 1) / chunk * chunk;      // whole chunks per XCD
  a.xcd_chunk = (int)chunk;
  dim3 grid((unsigned)(cpx * 8));
  if (a.ld > 1024) CNA_FAIL(CNA_EINVAL, "more than 1024 samples / state columns are not supported");
  if (first) {
    switch ((a.ld + 63) / 64) {
      case 1: launch_first_t<VT, 1>(c, a, grid, st); break;
      case 2: launch_first_t<VT, 2>(c, a, grid, st); break;
      case 3: launch_first_t<VT, 3>(c, a, grid, st); break;
      case 4: launch_first_t<VT, 4>(c, a, grid, st); break;
      case 5: case 6: launch_first_t<VT, 6>(c, a, grid, st); break;
      case 7: case 8: launch_first_t<VT, 8>(c, a, grid, st); break;
      case 9: case 10: case 11: case 12: launch_first_t<VT, 12>(c, a, grid, st); break;
      default: launch_first_t<VT, 16>(c, a, grid, st); break;
    }
  } else if (a.sp_cnt) {
    switch ((a.ld / 2 + 63) / 64) {
      case 1: CNA_TRY((launch_step_sparse_t<VT, 1>(c, a, grid, st))); break;
      case 2: CNA_TRY((launch_step_sparse_t<VT, 2>(c, a, grid, st))); break;
      case 3: CNA_TRY((launch_step_sparse_t<VT, 3>(c, a, grid, st))); break;
      case 4: CNA_TRY((launch_step_sparse_t<VT, 4>(c, a, grid, st))); break;
      case 5: case 6: CNA_TRY((launch_step_sparse_t<VT, 6>(c, a, grid, st))); break;
      default: CNA_TRY((launch_step_sparse_t<VT, 8>(c, a, grid, st))); break;
    }
  } else if (pair) {
    if (a.in32 || a.out32) CNA_FAIL(CNA_ESTATE, "4-byte state with the two-rows-per-wave step");
    if (a.rows) hipLaunchKernelGGL((k_nam_step_pair<VT, 2>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_nam_step_pair<VT, 0>), grid, dim3(256), 0, st, a);
  } else if (a.in32) {
    const bool halves = a.ld <= 128 && (c->t_compact ? c->t_rows : c->n_pad) * (int64_t)a.ld * 4 < (int64_t)4 << 30 &&
                        !getenv("CNA_STEP32_WIDE");          // (tests compare the two kernels)
    if (halves) CNA_TRY((launch_step32h_t<VT>(c, a, grid, st)));
    else switch ((a.ld / 4 + 63) / 64) {
      case 1: CNA_TRY((launch_step32_t<VT, 1, CNA_STEP32_U>(c, a, grid, st))); break;
      case 2: CNA_TRY((launch_step32_t<VT, 2, 8>(c, a, grid, st))); break;
      case 3: CNA_TRY((launch_step32_t<VT, 3, 4>(c, a, grid, st))); break;
      default: CNA_TRY((launch_step32_t<VT, 4, 4>(c, a, grid, st))); break;
    }
  } else {
    if (a.out32) CNA_FAIL(CNA_ESTATE, "4-byte state out of the 8-byte dense step");
    switch ((a.ld / 2 + 63) / 64) {
      case 1: launch_step_t<VT, 1>(c, a, grid, st); break;
      // 129 ... 256 columns: ten rows in flight (7.82 -> 7.69 ms at 2M x 200; 9: the same, 11 / 12: as 8); with the
      // selection by-product in the write-out, eight (8.45 against 8.55 ms, three runs each on one box; 6: 8.61, 12: 8.95)
      case 2: if (a.sel_X) launch_step_t<VT, 2, 8>(c, a, grid, st); else launch_step_t<VT, 2, 10>(c, a, grid, st); break;
      case 3: launch_step_t<VT, 3>(c, a, grid, st); break;
      case 4: launch_step_t<VT, 4>(c, a, grid, st); break;
      case 5: case 6: launch_step_t<VT, 6, 4>(c, a, grid, st); break;
      default: launch_step_t<VT, 8, 4>(c, a, grid, st); break;
    }
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

}  // namespace

int launch_rows_need_halo(cna_ctx* c, unsigned char* flags_dev) {
  if (c->n_local == 0) return 0;
  hipLaunchKernelGGL(k_rows_need_halo, dim3((unsigned)((c->n_local + 255) / 256)), dim3(256), 0, c->stream, c->indptr, c->indices,
                     c->n_local, c->row0, flags_dev);
  HIP_TRY(hipGetLastError());
  return 0;
}

int launch_remap_indices(cna_ctx* c, const int64_t* recv_rows_dev, int64_t nr, int32_t* out, int* bad) {
  *bad = 0;
  if (c->nnz == 0) return 0;
  int* flag = nullptr;
  HIP_TRY(hipMalloc(&flag, sizeof(int)));
  struct Free { int* p; ~Free() { (void)hipFree(p); } } free_flag{flag};
  HIP_TRY(hipMemsetAsync(flag, 0, sizeof(int), c->stream));
  const unsigned grid = (unsigned)std::min<int64_t>((c->nnz + 255) / 256, 65536);
  hipLaunchKernelGGL(k_remap_idx, dim3(grid), dim3(256), 0, c->stream, c->indices, c->nnz, c->row0, c->n_local, recv_rows_dev, nr,
                     out, flag);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(bad, flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

int launch_colsum(cna_ctx* c) {
  ProfScope ps(c, CNA_K_COLSUM);
  HIP_TRY(hipMemsetAsync(c->colsum, 0, sizeof(double) * c->n_pad, c->stream));
  if (c->nnz == 0) return 0;
  const int64_t n = c->n_global;
  const int64_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  // scratch: cnt u32[n] | cursor u32[n] | tile sums u64[ntiles] | first u64[n+1] | entries 16 B x nnz
  const int64_t b_cnt = round_up64(4 * n, 256), b_tiles = round_up64(8 * ntiles, 256), b_first = round_up64(8 * (n + 1), 256);
  const int64_t need = 2 * b_cnt + b_tiles + b_first + (int64_t)sizeof(ColEntry) * c->nnz;
  void* buf = nullptr;
  CNA_TRY(dev_alloc(c, &buf, (size_t)need));
  char* p = (char*)buf;
  unsigned int* cnt = (unsigned int*)p; p += b_cnt;
  unsigned int* cursor = (unsigned int*)p; p += b_cnt;
  unsigned long long* tiles = (unsigned long long*)p; p += b_tiles;
  unsigned long long* first = (unsigned long long*)p; p += b_first;
  ColEntry* ent = (ColEntry*)p;
  int rc = 0;
  do {
    if (hipMemsetAsync(cnt, 0, (size_t)(2 * b_cnt), c->stream) != hipSuccess) { rc = 1; break; }
    const unsigned g_edges = (unsigned)std::min<int64_t>((c->nnz + 255) / 256, 16384);
    hipLaunchKernelGGL(k_col_count, dim3(g_edges), dim3(256), 0, c->stream, c->indices, c->nnz, cnt);
    hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)ntiles), dim3(256), 0, c->stream, cnt, n, tiles);
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(1), dim3(1024), 0, c->stream, tiles, ntiles);
    hipLaunchKernelGGL(k_scan_finish, dim3((unsigned)ntiles), dim3(256), 0, c->stream, cnt, n, tiles, first);
    const unsigned g_rows = (unsigned)std::min<int64_t>((c->n_local + 3) / 4, 65536);
    if (c->data_f64)
      hipLaunchKernelGGL(k_col_scatter<double>, dim3(g_rows), dim3(256), 0, c->stream, c->indptr, c->indices,
                         (const double*)c->data, c->orig_idx, c->n_local, c->row0, first, cursor, ent);
    else
      hipLaunchKernelGGL(k_col_scatter<float>, dim3(g_rows), dim3(256), 0, c->stream, c->indptr, c->indices,
                         (const float*)c->data, c->orig_idx, c->n_local, c->row0, first, cursor, ent);
    const unsigned g_cols = (unsigned)std::min<int64_t>((n + 3) / 4, 65536);
    hipLaunchKernelGGL(k_col_sum, dim3(g_cols), dim3(256), 0, c->stream, first, ent, n, c->colsum);
  } while (0);
  const hipError_t le = hipGetLastError();
  const hipError_t se = hipStreamSynchronize(c->stream);      // the scratch goes away below
  dev_free(c, buf, (size_t)need);
  if (rc) CNA_FAIL(CNA_EINVAL, "launch_colsum: memset failed");
  HIP_TRY(le);
  HIP_TRY(se);
  return 0;
}

// ---- the resident graph into another cell order, on the device (cna_graph_reorder) --------------------------------
namespace {
__global__ void k_perm_inverse(const int64_t* __restrict__ perm, int64_t n, int32_t* __restrict__ inv, int* __restrict__ bad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t p = perm[i];
  if (p < 0 || p >= n) { atomicExch(bad, 1); return; }
  inv[p] = (int32_t)i;
}
// a permutation iff every position is what the inverse says of its own cell (two positions naming one cell: one of them is not)
__global__ void k_perm_check(const int64_t* __restrict__ perm, int64_t n, const int32_t* __restrict__ inv, int* __restrict__ bad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t p = perm[i];
  if (p < 0 || p >= n || inv[p] != (int32_t)i) atomicExch(bad, 1);
}
__global__ void k_perm_degrees(const int64_t* __restrict__ indptr, const int64_t* __restrict__ perm, int64_t n,
                               unsigned int* __restrict__ deg) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) deg[i] = (unsigned int)(indptr[perm[i] + 1] - indptr[perm[i]]);
}
// one wave per new row: the entries of the old row in their old order, columns renamed (what _order.permuted_rows does
// on the host: the order of the terms of every row sum, hence every result, is unchanged)
template <typename VT>
__global__ __launch_bounds__(256) void k_perm_rows(const int64_t* __restrict__ indptr, const int32_t* __restrict__ idx,
                                                   const VT* __restrict__ val, const int64_t* __restrict__ perm,
                                                   const int32_t* __restrict__ inv, int64_t n,
                                                   const int64_t* __restrict__ new_indptr, int32_t* __restrict__ new_idx,
                                                   VT* __restrict__ new_val) {
  const int lane = threadIdx.x & 63;
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < n; row += nw) {
    const int64_t src = perm[row];
    const int64_t s0 = indptr[src], cnt = indptr[src + 1] - s0, d0 = new_indptr[row];
    for (int64_t e = lane; e < cnt; e += 64) {
      new_idx[d0 + e] = inv[idx[s0 + e]];
      new_val[d0 + e] = val[s0 + e];
    }
  }
}
template <typename T>
__global__ void k_perm_gather(const T* __restrict__ src, const int64_t* __restrict__ perm, int64_t n, T* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[perm[i]];
}
}  // namespace

// perm_dev[i] = the row of the resident graph that becomes row i (one rank, whole graph resident, n_pad == n).  Replaces
// indptr / indices / data, permutes the column sums (keyed by the caller's row ids when they were formed: the same bits
// in any order) and the sample codes.  The caller (c_api.hip) resets what hangs on the order.
int graph_reorder_device(cna_ctx* c, const int64_t* perm_dev) {
  const int64_t n = c->n_global, nnz = c->nnz;
  const size_t vb = c->data_f64 ? 8 : 4;
  const int64_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  const int64_t b_inv = round_up64(4 * n, 256), b_deg = round_up64(4 * n, 256), b_tiles = round_up64(8 * ntiles, 256);
  const size_t tmp_bytes = (size_t)(b_inv + b_deg + b_tiles + 256);
  void* tmp = nullptr;
  int64_t* nip = nullptr;
  int32_t* nix = nullptr;
  void* nval = nullptr;
  double* ncs = nullptr;
  int32_t* nsid = nullptr;
  auto drop = [&]() {
    (void)hipStreamSynchronize(c->stream);
    if (tmp) dev_free(c, tmp, tmp_bytes);
    if (nip) dev_free(c, nip, sizeof(int64_t) * (n + 1));
    if (nix) dev_free(c, nix, sizeof(int32_t) * nnz);
    if (nval) dev_free(c, nval, vb * nnz);
    if (ncs) dev_free(c, ncs, sizeof(double) * c->n_pad);
    if (nsid) dev_free(c, nsid, sizeof(int32_t) * n);
  };
  int rc = dev_alloc(c, &tmp, tmp_bytes);
  if (!rc) rc = dev_alloc(c, (void**)&nip, sizeof(int64_t) * (n + 1));
  if (!rc) rc = dev_alloc(c, (void**)&nix, sizeof(int32_t) * nnz);
  if (!rc) rc = dev_alloc(c, &nval, vb * nnz);
  if (!rc && c->have_colsum) rc = dev_alloc(c, (void**)&ncs, sizeof(double) * c->n_pad);
  if (!rc && c->sid && c->sid_n == n) rc = dev_alloc(c, (void**)&nsid, sizeof(int32_t) * n);
  if (rc) { drop(); return rc; }
  int32_t* inv = (int32_t*)tmp;
  unsigned int* deg = (unsigned int*)((char*)tmp + b_inv);
  unsigned long long* tiles = (unsigned long long*)((char*)tmp + b_inv + b_deg);
  int* bad = (int*)((char*)tmp + b_inv + b_deg + b_tiles);
  hipStream_t st = c->stream;
  const unsigned g_n = (unsigned)((n + 255) / 256);
  hipError_t e = hipMemsetAsync(inv, 0xff, (size_t)b_inv, st);
  if (e == hipSuccess) e = hipMemsetAsync(bad, 0, 4, st);
  if (e != hipSuccess) { drop(); HIP_TRY(e); }
  hipLaunchKernelGGL(k_perm_inverse, dim3(g_n), dim3(256), 0, st, perm_dev, n, inv, bad);
  hipLaunchKernelGGL(k_perm_check, dim3(g_n), dim3(256), 0, st, perm_dev, n, (const int32_t*)inv, bad);
  int bad_h = 0;
  e = hipMemcpyAsync(&bad_h, bad, 4, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) { drop(); HIP_TRY(e); }
  if (bad_h) { drop(); CNA_FAIL(CNA_EINVAL, "cna_graph_reorder: not a permutation of the cells"); }
  hipLaunchKernelGGL(k_perm_degrees, dim3(g_n), dim3(256), 0, st, c->indptr, perm_dev, n, deg);
  hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)ntiles), dim3(256), 0, st, deg, n, tiles);
  hipLaunchKernelGGL(k_scan_tile_sums, dim3(1), dim3(1024), 0, st, tiles, ntiles);
  hipLaunchKernelGGL(k_scan_finish, dim3((unsigned)ntiles), dim3(256), 0, st, deg, n, tiles, (unsigned long long*)nip);
  const unsigned g_rows = (unsigned)std::min<int64_t>((n + 3) / 4, 65536);
  if (c->data_f64)
    hipLaunchKernelGGL(k_perm_rows<double>, dim3(g_rows), dim3(256), 0, st, c->indptr, c->indices, (const double*)c->data, perm_dev,
                       (const int32_t*)inv, n, (const int64_t*)nip, nix, (double*)nval);
  else
    hipLaunchKernelGGL(k_perm_rows<float>, dim3(g_rows), dim3(256), 0, st, c->indptr, c->indices, (const float*)c->data, perm_dev,
                       (const int32_t*)inv, n, (const int64_t*)nip, nix, (float*)nval);
  if (ncs) hipLaunchKernelGGL(k_perm_gather<double>, dim3(g_n), dim3(256), 0, st, (const double*)c->colsum, perm_dev, n, ncs);
  if (nsid) hipLaunchKernelGGL(k_perm_gather<int32_t>, dim3(g_n), dim3(256), 0, st, (const int32_t*)c->sid, perm_dev, n, nsid);
  e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) { drop(); HIP_TRY(e); }
  // swap in
  dev_free(c, c->indptr, sizeof(int64_t) * (n + 1));
  dev_free(c, c->indices, sizeof(int32_t) * nnz);
  dev_free(c, c->data, vb * nnz);
  c->indptr = nip; c->indices = nix; c->data = nval;
  nip = nullptr; nix = nullptr; nval = nullptr;
  if (ncs) { dev_free(c, c->colsum, sizeof(double) * c->n_pad); c->colsum = ncs; ncs = nullptr; }
  if (nsid) {
    (void)hipMemcpyAsync(c->sid, nsid, sizeof(int32_t) * n, hipMemcpyDeviceToDevice, st);      // (sid keeps its buffer: sized elsewhere)
  }
  drop();
  return 0;
}

int launch_add_scalar(cna_ctx* c, double* v, int64_t n, double s) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_add_scalar, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, v, n, s);
  HIP_TRY(hipGetLastError());
  return 0;
}

int64_t nam_step_turn_rows(const cna_ctx* c, int64_t n_rows) {
  if (c->t_ld <= 64) return 0;                               // (the two-rows-per-wave kernel and narrow states are not split)
  return 8 * step_xcd_chunk(false, c->t_ld, (n_rows + 3) / 4) * 4;
}

int launch_nam_step(cna_ctx* c, bool first, bool want_kurt, bool write_t, bool write_nam, bool dense, const int32_t* rows,
                    int64_t n_rows, int64_t base, int64_t count, hipStream_t st_in, bool timed) {
  if (c->n_local == 0 || (rows && n_rows == 0) || count == 0) return 0;
  hipStream_t st = st_in ? st_in : c->stream;
  // an exchange of the input state may still be in flight on the halo stream: every launch waits for it, except the one
  // over the rows that read no foreign row (cna_nam_step)
  if (!first && !c->halo_safe_launch) CNA_TRY(halo_settle(c));
  if (first && !c->cellinfo_valid) {
    void* p = c->cellinfo;
    const int64_t rows = c->t_compact ? c->t_rows : c->n_global;
    CNA_TRY(dev_reserve(c, &p, &c->cellinfo_cap, (int64_t)sizeof(CellInfo) * std::max<int64_t>(rows, 1)));
    c->cellinfo = p;
    if (c->t_compact)
      hipLaunchKernelGGL(k_cellinfo_compact, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, c->colsum, c->sid, c->row0,
                         c->n_local, c->halo_recv_idx, rows, (CellInfo*)c->cellinfo);
    else
      hipLaunchKernelGGL(k_cellinfo, dim3((unsigned)((c->n_global + 255) / 256)), dim3(256), 0, st, c->colsum,
                         c->sid, c->n_global, (CellInfo*)c->cellinfo);
    HIP_TRY(hipGetLastError());
    c->cellinfo_valid = true;
  }
  const bool sparse_step = !first && !dense && c->sp_cnt && c->steps_done == 1;       // launch_step_q takes k_nam_step_sparse then
  ProfScope ps(c, !timed ? -1 : (first ? CNA_K_NAM_FIRST : (sparse_step ? CNA_K_NAM_STEP_SPARSE : CNA_K_NAM_STEP)), st);
  StepArgs a;
  a.indptr = c->indptr;
  a.idx = c->indices;
  a.val = c->data;
  a.sid = c->sid;
  a.colsum = c->colsum;
  a.counts = c->counts;
  a.Tin = c->T[c->t_cur];
  a.Tout = c->T[c->t_cur ^ 1];
  a.nam = c->nam;
  a.stat = c->stat;
  a.dense_out = dense ? c->dense_s : nullptr;
  a.n_local = rows ? n_rows : c->n_local;
  a.rows = rows;
  a.row0 = c->row0;
  if (c->t_compact) {
    // compact row space: the state, the pairs and the cell records are numbered from this rank's first row, and the
    // graph is read through the renumbered column indices; the two arrays that stay global (column sums, per-cell
    // statistic) are handed over shifted by row0 -- the kernels see a block that starts at row 0
    a.idx = c->idx_t;
    a.row0 = 0;
    a.colsum = c->colsum + c->row0;
    a.stat = c->stat ? c->stat + c->row0 : nullptr;
  }
  a.width = c->t_width;
  a.ld = c->t_ld;
  a.w = c->self_weight;
  a.want_kurt = want_kurt;
  a.write_t = write_t;
  a.write_nam = write_nam;
  a.stop = c->auto_stop;
  a.sp_keep_dense = ((c->nranks > 1 || c->halo_on) && !c->sp_dense_interior_off) ? 1 : 0;
  // 4-byte state between steps (opt-in, one rank): written by the compressed second step and by every later step that is
  // not known to be the last, read by k_nam_step32.  The first step, narrow states (two rows per wave), the dense
  // diffusion and sharded walks (the exchange moves 8-byte rows) keep 8 bytes.
  a.in32 = (!first && c->t_f32[c->t_cur]) ? 1 : 0;
  a.out32 = (write_t && !first && !dense && c->state_f32_mode && c->nranks == 1 && !c->halo_on && !c->t_compact && c->t_ld > 64 &&
             !rows && (sparse_step || a.in32)) ? 1 : 0;
  if (write_t) c->t_f32[c->t_cur ^ 1] = a.out32 != 0;
  // compressed state: written by the first step, read by the second (sample indicators only)
  const bool sp = c->sp_cnt && !dense && (first || c->steps_done == 1);
  a.sp_pair = sp ? (SpPair*)c->sp_pair : nullptr;
  a.sp_cnt = sp ? (unsigned char*)c->sp_cnt : nullptr;
  a.sel_X = nullptr;
  a.sel_y = nullptr;
  a.sel_nc = nullptr;
  a.sel_xq = nullptr;
  a.sel_xscale = nullptr;
  a.sel_nz = nullptr;
  a.sel_ldx = a.sel_Kp = 0;
  a.skip_nam = (c->byp_arm && c->byp_skip_nam) ? 1 : 0;
  if (c->byp_arm) {            // c_api.hip:arm_select_byproduct has sized X / planes / coefficients for this launch
    a.sel_X = c->X;
    a.sel_ldx = c->ldx;
    a.sel_nz = (unsigned long long*)c->byp_buf;
    a.sel_y = (const double*)((const char*)c->byp_buf + 16);
    a.sel_nc = c->ncorrs;
    if (c->byp_with_q) {
      a.sel_xq = (unsigned char*)c->xq;
      a.sel_xscale = (double2*)c->xq_scale;
      a.sel_Kp = 32 * ((c->N + 31) / 32);
    }
  }
  if (count > 0) {
    // the rows [base, base + count) only: every per-row array moves up by `base` rows, the kernels number from zero
    // (arrays indexed by the global row -- state, column sums, pairs, cell records -- follow row0)
    if (rows) CNA_FAIL(CNA_EINVAL, "launch_nam_step: a row list and a row range exclude each other");
    a.indptr += base;
    a.row0 += base;
    a.n_local = count;
    if (a.nam) a.nam += base * (int64_t)a.ld;
    if (a.dense_out) a.dense_out += base * (int64_t)a.ld;
    if (a.sel_X) {
      a.sel_X += base * (int64_t)a.sel_ldx;
      a.sel_nc += base;
      if (a.sel_xq) { a.sel_xq += (size_t)base * 3 * a.sel_Kp; a.sel_xscale += base; }
    }
  }
  return c->data_f64 ? launch_step_q<double>(c, first, a, st) : launch_step_q<float>(c, first, a, st);
}

int launch_scale_rows(cna_ctx* c, const double* s_local, double* t_global, int m, int ld) {
  const int64_t tot = c->n_local * ld;
  if (tot == 0) return 0;
  // (compact row space: the block's rows are the first rows of the state)
  hipLaunchKernelGGL(k_scale_rows, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream,
                     s_local, c->t_compact ? c->colsum + c->row0 : c->colsum, t_global, c->dense_s, c->n_local,
                     c->t_compact ? (int64_t)0 : c->row0, m, ld);
  HIP_TRY(hipGetLastError());
  return 0;
}
