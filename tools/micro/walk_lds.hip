// Probe of the LDS-staged dense walk step on a chunk-major state (round 3).
//
//   state   T[chunk][row][W]  (W = 8 columns = 64 B per row and chunk; a chunk plane is n x 64 B)
//   block   <= 64 consecutive device rows with <= SMAX distinct neighbour rows ("sources", sorted)
//   kernel  one workgroup of 512 threads per block; thread = (row, column of the chunk).  Per chunk the
//           sources' 64-byte pieces arrive by LDS-DMA (global_load_lds_dwordx4) into one of two LDS
//           buffers while the previous chunk is consumed: every thread walks its row's edges in CSR order
//           (edge weights and LDS offsets in registers for all chunks) with one ds_read_b64 per edge.
//   checks  bit-for-bit against the wave-per-row gather on the row-major copy of the same state.
//
//   hipcc --offload-arch=gfx950 -O3 walk_lds.hip -o walk_lds && ./walk_lds <dir> <N>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <string>
#include <vector>
#pragma clang fp contract(off)

template <typename T>
static std::vector<T> slurp(const std::string& p) {
  FILE* f = fopen(p.c_str(), "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", p.c_str()); exit(1); }
  fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
  std::vector<T> v(sz / sizeof(T));
  if (fread(v.data(), 1, sz, f) != (size_t)sz) exit(1);
  fclose(f);
  return v;
}
template <typename T>
static T* up(const std::vector<T>& v) {
  T* d; (void)hipMalloc(&d, v.size() * sizeof(T) + 256);
  (void)hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
  return d;
}
__device__ __forceinline__ double readlane_d(double v, int l) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  return __hiloint2double(__builtin_amdgcn_readlane(hi, l), __builtin_amdgcn_readlane(lo, l));
}

// ---- reference: wave per row on the row-major state (the production kernel's shape, U rows in flight)
template <int NQ2, int U>
__global__ __launch_bounds__(256) void k_row(const long* __restrict__ indptr, const int* __restrict__ idx,
                                             const float* __restrict__ val, const double2* __restrict__ T, int ld2,
                                             long n, double2* __restrict__ out, int chunk) {
  const int lane = threadIdx.x & 63;
  const long b = blockIdx.x >> 3, x = blockIdx.x & 7;
  const long blk = (b / chunk) * (8 * (long)chunk) + x * chunk + (b % chunk);
  const long row = blk * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const long start = indptr[row], end = indptr[row + 1];
  double2 acc[NQ2];
#pragma unroll
  for (int q = 0; q < NQ2; ++q) acc[q] = make_double2(0, 0);
  for (long base = start; base < end; base += 64) {
    const bool ok = base + lane < end;
    const int jl = ok ? idx[base + lane] : 0;
    const double al = ok ? (double)val[base + lane] : 0.0;
    const int cnt = (int)((end - base) < 64 ? (end - base) : 64);
    int l = 0;
    for (; l + U <= cnt; l += U) {
      double2 t[U][NQ2];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int j = __builtin_amdgcn_readlane(jl, l + u);
        const double2* rp = T + (long)j * ld2;
#pragma unroll
        for (int q = 0; q < NQ2; ++q) t[u][q] = (lane + 64 * q < ld2) ? rp[lane + 64 * q] : make_double2(0, 0);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const double av = readlane_d(al, l + u);
#pragma unroll
        for (int q = 0; q < NQ2; ++q) { acc[q].x = acc[q].x + av * t[u][q].x; acc[q].y = acc[q].y + av * t[u][q].y; }
      }
    }
    for (; l < cnt; ++l) {
      const int j = __builtin_amdgcn_readlane(jl, l);
      const double av = readlane_d(al, l);
      const double2* rp = T + (long)j * ld2;
#pragma unroll
      for (int q = 0; q < NQ2; ++q) {
        const double2 t = (lane + 64 * q < ld2) ? rp[lane + 64 * q] : make_double2(0, 0);
        acc[q].x = acc[q].x + av * t.x; acc[q].y = acc[q].y + av * t.y;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < NQ2; ++q)
    if (lane + 64 * q < ld2) {
      const double2 own = T[row * ld2 + lane + 64 * q];
      out[row * ld2 + lane + 64 * q] = make_double2(acc[q].x + own.x, acc[q].y + own.y);
    }
}

// ---- the LDS-staged step
struct WalkArgs {
  const long* indptr; const int* idx; const float* val; const unsigned short* slot;
  const long* blk_row; const long* src_ptr; const int* src;
  const double* Tin; double* Tout;        // chunk-major, plane = n_pad * W doubles
  long n, n_pad, nblocks; int nchunks, xcd_chunk;
};

extern __shared__ __align__(16) char sm[];

// MODE 0: full; 1: staging only (no edge walk); 2: edge walk only (no staging: LDS holds whatever)
template <int SMAX, int KMAX, int MODE>
__global__ __launch_bounds__(512) void k_walk(WalkArgs a) {
  constexpr int W = 8, RB = W * 8;                 // bytes of a source's piece per chunk
  constexpr int BUF = SMAX * RB;
  constexpr int NG = SMAX / 16, NS = (NG + 7) / 8;  // staging groups of 16 sources (one wave instruction); per wave
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long bb = blockIdx.x >> 3, x = blockIdx.x & 7;
  const long b = (bb / a.xcd_chunk) * (8 * (long)a.xcd_chunk) + x * a.xcd_chunk + (bb % a.xcd_chunk);
  if (b >= a.nblocks) return;
  const long r0 = a.blk_row[b];
  const int rows = (int)(a.blk_row[b + 1] - r0);
  const long s0 = a.src_ptr[b];
  const int nsrc = (int)(a.src_ptr[b + 1] - s0);
  const int ngroups = (nsrc + 15) >> 4;
  const long plane = a.n_pad * W;                   // doubles per chunk plane
  // this lane's piece of every staging instruction of its wave: byte offset inside a plane
  unsigned goff[NS];
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    int sq = (u * 8 + wv) * 16 + (lane >> 2);
    sq = sq < nsrc ? sq : nsrc - 1;
    goff[u] = (unsigned)a.src[s0 + sq] * (unsigned)RB + (unsigned)(lane & 3) * 16u;
  }
  auto stage = [&](int c, int p) {
    const char* pl = (const char*)(a.Tin + (long)c * plane);
#pragma unroll
    for (int u = 0; u < NS; ++u)
      if ((u * 8 + wv) < ngroups)
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(pl + goff[u]),
                                         (void __attribute__((address_space(3)))*)(sm + p * BUF + (u * 8 + wv) * 1024), 16, 0, 0);
  };
  if (MODE != 2) stage(0, 0);
  // edge records of this thread's row
  const int r = tid >> 3, cc = tid & 7;
  const bool live = r < rows;
  const long es = live ? a.indptr[r0 + r] : 0;
  const int deg = live ? (int)(a.indptr[r0 + r + 1] - es) : 0;
  double w[KMAX];
  unsigned off[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const bool ok = k < deg;
    w[k] = ok ? (double)a.val[es + k] : 0.0;
    off[k] = (ok ? (unsigned)a.slot[es + k] * (unsigned)RB : 0u) + (unsigned)cc * 8u;
  }
  int dmax = deg < KMAX ? deg : KMAX;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(dmax, o); dmax = t > dmax ? t : dmax; }
  dmax = __builtin_amdgcn_readfirstlane(dmax);
  double own = live ? a.Tin[(r0 + r) * W + cc] : 0.0;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int c = 0; c < a.nchunks; ++c) {
    const int p = c & 1;
    double own_next = 0.0;
    if (c + 1 < a.nchunks) {
      if (live) own_next = a.Tin[(long)(c + 1) * plane + (r0 + r) * W + cc];
      if (MODE != 2) stage(c + 1, p ^ 1);
    }
    double acc = 0.0;
    if (MODE != 1) {
      const unsigned pb = (unsigned)(p * BUF);
#pragma unroll
      for (int k0 = 0; k0 < KMAX; k0 += 8) {
        if (k0 < dmax) {
          double v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = *(const double*)(sm + pb + off[k0 + u]);
#pragma unroll
          for (int u = 0; u < 8; ++u) acc = acc + w[k0 + u] * v[u];
        }
      }
      for (int k = KMAX; k < deg; ++k) {            // rows longer than the register copy
        const double v = *(const double*)(sm + pb + (unsigned)a.slot[es + k] * (unsigned)RB + (unsigned)cc * 8u);
        acc = acc + (double)a.val[es + k] * v;
      }
    }
    if (live) a.Tout[(long)c * plane + (r0 + r) * W + cc] = acc + own;
    own = own_next;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
}

// ---- staging-rate variants (no edge walk): what limits the L2 -> LDS transfer?
//   PIECE  bytes of a source per pass (64: W = 8 planes, 128: W = 16 planes, the state reinterpreted)
//   DMA    1: global_load_lds_dwordx4 into LDS, 0: global_load_dwordx4 into registers (discarded)
//   SYNC   0: a workgroup walks all passes of its block (as k_walk); 1: pass-major grid -- consecutive
//          workgroups take consecutive blocks of the SAME pass, so the whole chip reads one plane at a time
template <int PIECE, int DMA, int SYNC, int SMAX>
__global__ __launch_bounds__(512) void k_stage(WalkArgs a, int npass, unsigned long long* sink) {
  constexpr int LPS = PIECE / 16, SPI = 64 / LPS;               // lanes per source, sources per wave instruction
  constexpr int NG = (SMAX + SPI - 1) / SPI, NS = (NG + 7) / 8;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  long b; int c0, c1;
  if (SYNC) { b = blockIdx.x % a.nblocks; c0 = (int)(blockIdx.x / a.nblocks); c1 = c0 + 1; }
  else {
    const long bb = blockIdx.x >> 3, x = blockIdx.x & 7;
    b = (bb / a.xcd_chunk) * (8 * (long)a.xcd_chunk) + x * a.xcd_chunk + (bb % a.xcd_chunk);
    c0 = 0; c1 = npass;
    if (b >= a.nblocks) return;
  }
  const long s0 = a.src_ptr[b];
  const int nsrc = (int)(a.src_ptr[b + 1] - s0);
  const int ngroups = (nsrc + SPI - 1) / SPI;
  const long plane_bytes = a.n_pad * (long)PIECE;
  unsigned goff[NS];
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    int sq = (u * 8 + wv) * SPI + lane / LPS;
    sq = sq < nsrc ? sq : nsrc - 1;
    goff[u] = (unsigned)a.src[s0 + sq] * (unsigned)PIECE + (unsigned)(lane % LPS) * 16u;
  }
  double2 keep = make_double2(0, 0);
  for (int c = c0; c < c1; ++c) {
    const char* pl = (const char*)a.Tin + (long)c * plane_bytes;
    const int p = c & 1;
#pragma unroll
    for (int u = 0; u < NS; ++u)
      if ((u * 8 + wv) < ngroups) {
        if (DMA)
          __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(pl + goff[u]),
                                           (void __attribute__((address_space(3)))*)(sm + ((!SYNC && PIECE == 64) ? p * SMAX * 64 : 0) + (u * 8 + wv) * 1024), 16, 0, 0);
        else {
          const double2 v = *(const double2*)(pl + goff[u]);
          keep.x += v.x; keep.y += v.y;
        }
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (!DMA && keep.x == 1.2345e-300) sink[0] = 1;
}

// row-major copy of a chunk-major matrix (and back) for the comparison
__global__ void k_to_rows(const double* __restrict__ Tc, double* __restrict__ Tr, long n, long n_pad, int ld, int nchunks) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * (long)ld) return;
  const long row = i / ld; const int col = (int)(i % ld);
  Tr[i] = col < nchunks * 8 ? Tc[(long)(col >> 3) * n_pad * 8 + row * 8 + (col & 7)] : 0.0;
}
__global__ void k_compare(const double* __restrict__ Tc, const double* __restrict__ Tr, long n, long n_pad, int ld, int N,
                          unsigned long long* __restrict__ bad) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * (long)N) return;
  const long row = i / N; const int col = (int)(i % N);
  const double a = Tc[(long)(col >> 3) * n_pad * 8 + row * 8 + (col & 7)], b = Tr[row * ld + col];
  if (!(a == b)) atomicAdd(bad, 1ull);
}
__global__ void k_fill(double* p, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned s = (unsigned)(i * 2654435761u) ^ (unsigned)(i >> 32);
  s = s * 1664525u + 1013904223u; s ^= s >> 15; s *= 2246822519u; s ^= s >> 13;
  p[i] = (s >> 8) * (1.0 / 16777216.0);
}

template <typename F> static float time_it(F f, int rep = 3) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); (void)hipEventRecord(e0);
  for (int i = 0; i < rep; ++i) f();
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) printf("      HIP error: %s\n", hipGetErrorString(err));
  return ms / rep;
}

template <int SMAX, int KMAX, int MODE>
static float run_walk(WalkArgs& a, const char* what, double gathered) {
  constexpr size_t smem = (size_t)2 * SMAX * 64;
  (void)hipFuncSetAttribute((const void*)k_walk<SMAX, KMAX, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  const long grid = (a.nblocks + 8 * a.xcd_chunk - 1) / (8 * a.xcd_chunk) * (8 * a.xcd_chunk);
  const float ms = time_it([&] { hipLaunchKernelGGL((k_walk<SMAX, KMAX, MODE>), dim3((unsigned)grid), dim3(512), smem, 0, a); });
  printf("  LDS-staged %-28s SMAX=%d KMAX=%d xcd_chunk=%d  %8.1f us  (%.2f TB/s of edge bytes)\n", what, SMAX, KMAX, a.xcd_chunk,
         ms * 1e3, gathered / (ms * 1e-3) / 1e12);
  return ms;
}

int main(int argc, char** argv) {
  const std::string dir = argv[1];
  const int N = atoi(argv[2]);
  const int smax_file = argc > 3 ? atoi(argv[3]) : 1152;
  const int nchunks = (N + 7) / 8, ld = nchunks * 8 + ((nchunks * 8) % 32 == 0 ? 8 : 0);   // row-major copy: stride off 256 B multiples
  auto indptr = slurp<long>(dir + "/indptr.bin");
  auto idx = slurp<int>(dir + "/idx.bin");
  auto val = slurp<float>(dir + "/val.bin");
  auto slot = slurp<unsigned short>(dir + "/slot.bin");
  auto blkrow = slurp<long>(dir + "/blkrow.bin");
  auto srcptr = slurp<long>(dir + "/srcptr.bin");
  auto src = slurp<int>(dir + "/src.bin");
  const long n = (long)indptr.size() - 1, n_pad = (n + 63) / 64 * 64;
  WalkArgs a{};
  a.indptr = up(indptr); a.idx = up(idx); a.val = up(val); a.slot = up(slot);
  a.blk_row = up(blkrow); a.src_ptr = up(srcptr); a.src = up(src);
  a.n = n; a.n_pad = n_pad; a.nblocks = (long)blkrow.size() - 1; a.nchunks = nchunks; a.xcd_chunk = 8;
  double *Tc, *Tr, *Oc, *Or;
  const size_t cm = (size_t)nchunks * n_pad * 8, rm = (size_t)(n + 64) * ld;
  (void)hipMalloc(&Tc, cm * 8); (void)hipMalloc(&Oc, cm * 8); (void)hipMalloc(&Tr, rm * 8); (void)hipMalloc(&Or, rm * 8);
  hipLaunchKernelGGL(k_fill, dim3((unsigned)((cm + 255) / 256)), dim3(256), 0, 0, Tc, (long)cm);
  hipLaunchKernelGGL(k_to_rows, dim3((unsigned)((n * ld + 255) / 256)), dim3(256), 0, 0, Tc, Tr, n, n_pad, ld, nchunks);
  (void)hipMemset(Oc, 0, cm * 8);
  a.Tin = Tc; a.Tout = Oc;
  const double gathered = (double)idx.size() * N * 8;
  printf("n = %ld, N = %d (%d chunks of 8 columns), nnz/row %.1f, %ld blocks, %.0f sources per block, edges/sources %.2f\n", n, N,
         nchunks, (double)idx.size() / n, a.nblocks, (double)src.size() / a.nblocks, (double)idx.size() / src.size());
  for (int chunk : {128}) {
    const long grid = ((n + 3) / 4 + 8 * chunk - 1) / (8 * chunk) * (8 * chunk);
    float ms;
    if (ld <= 256) ms = time_it([&] { hipLaunchKernelGGL((k_row<2, 10>), dim3((unsigned)grid), dim3(256), 0, 0, a.indptr, a.idx, a.val, (const double2*)Tr, ld / 2, n, (double2*)Or, chunk); });
    else { printf("N too large for the reference kernel\n"); return 1; }
    printf("  wave-per-row (row-major, 10 rows in flight, xcd chunk %d)  %8.1f us  (%.2f TB/s gathered)\n", chunk, ms * 1e3,
           gathered / (ms * 1e-3) / 1e12);
  }
  unsigned long long* bad; (void)hipMalloc(&bad, 8);
  auto check = [&](const char* what) {
    (void)hipMemset(bad, 0, 8);
    hipLaunchKernelGGL(k_compare, dim3((unsigned)((n * (long)N + 255) / 256)), dim3(256), 0, 0, Oc, Or, n, n_pad, ld, N, bad);
    unsigned long long h; (void)hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    printf("      %s: %llu of %ld outputs differ from the wave-per-row result\n", what, h, n * (long)N);
  };
  for (int xc : {8, 1, 32}) {
    a.xcd_chunk = xc;
    if (smax_file <= 960) run_walk<960, 48, 0>(a, "full", gathered);
    else if (smax_file <= 1152) run_walk<1152, 48, 0>(a, "full", gathered);
    else run_walk<1216, 48, 0>(a, "full", gathered);
    if (xc == 8) check("full");
  }
  a.xcd_chunk = 8;
  if (smax_file <= 960) {
    run_walk<960, 48, 1>(a, "staging only", gathered);
    run_walk<960, 48, 2>(a, "edge walk only", gathered);
    run_walk<960, 56, 0>(a, "full, 56 edges in registers", gathered);
  } else if (smax_file <= 1152) {
    run_walk<1152, 48, 1>(a, "staging only", gathered);
    run_walk<1152, 48, 2>(a, "edge walk only", gathered);
    run_walk<1152, 56, 0>(a, "full, 56 edges in registers", gathered);
    run_walk<1152, 40, 0>(a, "full, 40 edges in registers", gathered);
  } else {
    run_walk<1216, 48, 1>(a, "staging only", gathered);
    run_walk<1216, 48, 2>(a, "edge walk only", gathered);
  }

  {   // staging-rate variants
    unsigned long long* sink; (void)hipMalloc(&sink, 8);
    const double staged64 = (double)src.size() * 64.0 * nchunks;
    auto run_stage = [&](auto kern, const char* what, int npass, bool sync, size_t smem, double bytes) {
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      const long grid = sync ? a.nblocks * (long)npass : (a.nblocks + 8 * a.xcd_chunk - 1) / (8 * a.xcd_chunk) * (8 * a.xcd_chunk);
      const float ms = time_it([&] { hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), smem, 0, a, npass, sink); });
      printf("  staging %-44s %8.1f us  %6.2f TB/s staged\n", what, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
    };
    a.xcd_chunk = 32;
    run_stage(k_stage<64, 1, 0, 1152>, "DMA 64 B, block-major (as k_walk), 2 buffers", nchunks, false, 2 * 1152 * 64, staged64);
    run_stage(k_stage<64, 0, 0, 1152>, "loads to registers 64 B, block-major", nchunks, false, 1024, staged64);
    run_stage(k_stage<128, 1, 0, 1152>, "DMA 128 B, block-major, 1 buffer", nchunks / 2, false, 1152 * 128, staged64 / nchunks * (nchunks / 2) * 2);
    run_stage(k_stage<64, 1, 1, 1152>, "DMA 64 B, pass-major grid (1 buffer, 2 WG/CU)", nchunks, true, 1152 * 64, staged64);
    run_stage(k_stage<128, 1, 1, 1152>, "DMA 128 B, pass-major grid", nchunks / 2, true, 1152 * 128, staged64 / nchunks * (nchunks / 2) * 2);
    run_stage(k_stage<64, 0, 1, 1152>, "loads to registers 64 B, pass-major grid", nchunks, true, 1024, staged64);
  }
  return 0;
}
