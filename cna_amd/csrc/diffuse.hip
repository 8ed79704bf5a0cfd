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

namespace {

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
  int width, ld;
  double w;
  int want_kurt, write_t, write_nam;
};

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

// logical row-block of this workgroup: XCD x (= blockIdx % 8, observed dispatch) walks the
// contiguous range [x*cpx, (x+1)*cpx); a pure speed choice, any placement is correct.
__device__ __forceinline__ int64_t xcd_logical_block() {
  const int64_t cpx = gridDim.x >> 3;
  return (int64_t)(blockIdx.x & 7) * cpx + (blockIdx.x >> 3);
}

// write-out shared by both step kernels: s[] holds the new unscaled state of one row
template <int NQ>
__device__ __forceinline__ void finish_row(const StepArgs& a, int64_t row, int64_t grow, int lane,
                                           const double (&s)[NQ]) {
  const double cs = a.colsum[grow];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int col = lane + 64 * q;
    if (col < a.ld) {
      const bool in = col < a.width;
      if (a.write_t) a.Tout[grow * a.ld + col] = in ? __ddiv_rn(s[q], cs) : 0.0;
      if (a.dense_out) a.dense_out[row * a.ld + col] = in ? s[q] : 0.0;
    }
  }
  if (a.write_nam || a.want_kurt) {
    double x[NQ];
    double sum = 0.0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int col = lane + 64 * q;
      const bool in = col < a.width;
      x[q] = in ? __ddiv_rn(s[q], a.counts[col]) : 0.0;     // s / C   (_nam.py:59,73)
      if (a.write_nam && col < a.ld) a.nam[row * a.ld + col] = x[q];
      sum += x[q];
    }
    if (a.want_kurt) {
      // scipy.stats.kurtosis(s/C, axis=1): Fisher, biased (_nam.py:59)
      const double n = (double)a.width;
      const double mean = wave_sum(sum) / n;
      double d2s = 0.0, d4s = 0.0;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int col = lane + 64 * q;
        if (col < a.width) {
          const double d = x[q] - mean;
          const double d2 = d * d;
          d2s += d2;
          d4s += d2 * d2;
        }
      }
      const double m2 = wave_sum(d2s) / n;
      const double m4 = wave_sum(d4s) / n;
      const double em = 2.220446049250313e-16 * mean;
      const double k = (m2 <= em * em) ? __builtin_nan("") : m4 / (m2 * m2);
      if (lane == 0) a.stat[grow] = k - 3.0;
    }
  }
}

// first step: the input is the one-hot sample indicator, so a neighbour contributes
// A[i,j]/colsums[j] to column sid[j] only -- read 4 B of sid per edge, not an N-wide row.
template <typename VT, int NQ>
__global__ __launch_bounds__(256) void k_nam_first(StepArgs a) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t row = uniform64(xcd_logical_block() * 4 + wv);
  if (row >= a.n_local) return;
  const int64_t grow = a.row0 + row;
  const int64_t start = uniform64(a.indptr[row]), end = uniform64(a.indptr[row + 1]);
  const VT* __restrict__ val = (const VT*)a.val;
  double acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) acc[q] = 0.0;
  for (int64_t base = start; base < end; base += 64) {
    const int64_t e = base + lane;
    const bool ok = e < end;
    const int j = ok ? a.idx[e] : 0;
    const int cj = ok ? a.sid[j] : -1;
    const double v = ok ? __dmul_rn((double)val[e], __ddiv_rn(1.0, a.colsum[j])) : 0.0;
    const int cnt = (int)((end - base) < 64 ? (end - base) : 64);
    for (int l = 0; l < cnt; ++l) {
      const int c = __builtin_amdgcn_readlane(cj, l);
      const double vv = readlane_d(v, l);
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (lane + 64 * q == c) acc[q] = __dadd_rn(acc[q], vv);
    }
  }
  const int sid_i = a.sid[grow];
  const double self = __ddiv_rn(a.w, a.colsum[grow]);        // (w*1)/colsums[i]
  double s[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) s[q] = __dadd_rn(acc[q], (lane + 64 * q == sid_i) ? self : 0.0);
  finish_row<NQ>(a, row, grow, lane, s);
}

template <typename VT, int NQ>
__global__ __launch_bounds__(256) void k_nam_step(StepArgs a) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t row = uniform64(xcd_logical_block() * 4 + wv);
  if (row >= a.n_local) return;
  const int64_t grow = a.row0 + row;
  const int64_t start = uniform64(a.indptr[row]), end = uniform64(a.indptr[row + 1]);
  const VT* __restrict__ val = (const VT*)a.val;
  const double* __restrict__ Tin = a.Tin;
  double acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) acc[q] = 0.0;
  for (int64_t base = start; base < end; base += 64) {
    const int64_t e = base + lane;
    const bool ok = e < end;
    const int jl = ok ? a.idx[e] : 0;
    const double al = ok ? (double)val[e] : 0.0;
    const int cnt = (int)((end - base) < 64 ? (end - base) : 64);
    constexpr int U = (NQ <= 2) ? 16 : 8;      // neighbour rows in flight per wave
    for (int l = 0; l < cnt; l += U) {
      double t[U][NQ];
      double av[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool on = (l + u) < cnt;
        const int lu = on ? (l + u) : l;
        const int64_t j = __builtin_amdgcn_readlane(jl, lu);
        av[u] = readlane_d(al, lu);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int col = lane + 64 * q;
          t[u][q] = (on && col < a.width) ? Tin[j * a.ld + col] : 0.0;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if ((l + u) < cnt) {
#pragma unroll
          for (int q = 0; q < NQ; ++q) acc[q] = __dadd_rn(acc[q], __dmul_rn(av[u], t[u][q]));
        }
      }
    }
  }
  double s[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int col = lane + 64 * q;
    const double own = (col < a.width) ? Tin[grow * a.ld + col] : 0.0;
    s[q] = __dadd_rn(acc[q], __dmul_rn(a.w, own));           // + w*s/colsums  (exact for w=1)
  }
  finish_row<NQ>(a, row, grow, lane, s);
}

template <typename VT>
__global__ void k_colsum(const int32_t* __restrict__ idx, const VT* __restrict__ val, int64_t nnz,
                         double* colsum) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += stride)
    unsafeAtomicAdd(&colsum[idx[e]], (double)val[e]);
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
int launch_step_t(cna_ctx* c, bool first, const StepArgs& a) {
  const int64_t nblk = (c->n_local + 3) / 4;
  const int64_t cpx = (nblk + 7) / 8;
  dim3 grid((unsigned)(cpx * 8)), block(256);
  if (first)
    hipLaunchKernelGGL((k_nam_first<VT, NQ>), grid, block, 0, c->stream, a);
  else
    hipLaunchKernelGGL((k_nam_step<VT, NQ>), grid, block, 0, c->stream, a);
  HIP_TRY(hipGetLastError());
  return 0;
}

template <typename VT>
int launch_step_q(cna_ctx* c, bool first, const StepArgs& a) {
  const int nq = (a.ld + 63) / 64;
  switch (nq) {
    case 1: return launch_step_t<VT, 1>(c, first, a);
    case 2: return launch_step_t<VT, 2>(c, first, a);
    case 3: return launch_step_t<VT, 3>(c, first, a);
    case 4: return launch_step_t<VT, 4>(c, first, a);
    case 5: case 6: return launch_step_t<VT, 6>(c, first, a);
    case 7: case 8: return launch_step_t<VT, 8>(c, first, a);
    default: CNA_FAIL(CNA_EINVAL, "more than 512 samples / state columns are not supported");
  }
}

}  // namespace

int launch_colsum(cna_ctx* c) {
  ProfScope ps(c, CNA_K_COLSUM);
  HIP_TRY(hipMemsetAsync(c->colsum, 0, sizeof(double) * c->n_pad, c->stream));
  if (c->nnz > 0) {
    const int64_t want = (c->nnz + 255) / 256;
    const unsigned grid = (unsigned)(want < 8192 ? want : 8192);
    if (c->data_f64)
      hipLaunchKernelGGL(k_colsum<double>, dim3(grid), dim3(256), 0, c->stream, c->indices,
                         (const double*)c->data, c->nnz, c->colsum);
    else
      hipLaunchKernelGGL(k_colsum<float>, dim3(grid), dim3(256), 0, c->stream, c->indices,
                         (const float*)c->data, c->nnz, c->colsum);
    HIP_TRY(hipGetLastError());
  }
  return 0;
}

int launch_add_scalar(cna_ctx* c, double* v, int64_t n, double s) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_add_scalar, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, v, n, s);
  HIP_TRY(hipGetLastError());
  return 0;
}

int launch_nam_step(cna_ctx* c, bool first, bool want_kurt, bool write_t, bool write_nam, bool dense) {
  if (c->n_local == 0) return 0;
  ProfScope ps(c, first ? CNA_K_NAM_FIRST : CNA_K_NAM_STEP);
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
  a.n_local = c->n_local;
  a.row0 = c->row0;
  a.width = c->t_width;
  a.ld = c->t_ld;
  a.w = c->self_weight;
  a.want_kurt = want_kurt;
  a.write_t = write_t;
  a.write_nam = write_nam;
  return c->data_f64 ? launch_step_q<double>(c, first, a) : launch_step_q<float>(c, first, a);
}

int launch_scale_rows(cna_ctx* c, const double* s_local, double* t_global, int m, int ld) {
  const int64_t tot = c->n_local * ld;
  if (tot == 0) return 0;
  hipLaunchKernelGGL(k_scale_rows, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream,
                     s_local, c->colsum, t_global, c->dense_s, c->n_local, c->row0, m, ld);
  HIP_TRY(hipGetLastError());
  return 0;
}
