// Probe (round 4): how fast can the neighbour rows of the dense walk step be STAGED in LDS when the state is walked
// in column slices whose working set fits an XCD's L2?  No edge walk, no arithmetic: only the LDS-DMA traffic of
//
//   block   B consecutive device rows (cluster order), NW waves
//   chunk   PB-byte column slice of the state (PB = 256: 32 columns, one 16-lane group per piece)
//   tile    S of the block's distinct neighbour rows (tile program of cna_host_walk_tiles: ascending caller's index)
//
// in two schedules: (0) every block walks its chunks itself, blocks of an XCD side by side (they share neighbours and
// move through the chunks at the same pace); (1) grid = groups x chunks x blocks with one group of G blocks per XCD
// turn, chunk-major inside the group.  Reported: staged TB/s; run under rocprofv3 --pmc FETCH_SIZE for the bytes that
// come from behind the L2.
//
//   hipcc --offload-arch=gfx950 -O3 stage_rate.hip -o stage_rate && ./stage_rate <dir> <N> <NW> <S> <PB>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <string>
#include <vector>

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
  T* d; (void)hipMalloc(&d, v.size() * sizeof(T) + 4096);
  (void)hipMemset(d, 0, v.size() * sizeof(T) + 4096);
  (void)hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
  return d;
}

extern __shared__ __align__(16) char sm[];

__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}

struct Args {
  const long* blk_tile; const int* tsrc;      // tsrc: S ids per tile (fixed stride)
  const char* T; long nblocks; int ldb, S, nchunk, xcd_chunk, G, buf_bytes, nbuf;
  unsigned long long* sink;
};

// LPP lanes per piece (PB = 16 * LPP bytes), PPI = 64 / LPP pieces per wave instruction
template <int NW, int LPP, int MODE>
__global__ __launch_bounds__(NW * 64) void k_stage(Args a) {
  constexpr int PPI = 64 / LPP, PB = 16 * LPP;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  long b; int c0, c1;
  if (MODE == 0) {
    const long bb = blockIdx.x >> 3, x = blockIdx.x & 7;
    b = (bb / a.xcd_chunk) * (8 * (long)a.xcd_chunk) + x * a.xcd_chunk + (bb % a.xcd_chunk);
    c0 = 0; c1 = a.nchunk;
  } else {
    // XCD x takes groups x, x + 8, ...; inside a group: chunk-major
    const long k = blockIdx.x >> 3, x = blockIdx.x & 7;
    const long per_group = (long)a.G * a.nchunk;
    const long g = (k / per_group) * 8 + x;
    const long r = k % per_group;
    c0 = (int)(r / a.G); c1 = c0 + 1;
    b = g * a.G + (r % a.G);
  }
  if (b >= a.nblocks) return;
  typedef const __attribute__((address_space(4))) long* clong_p;      // constant address space + uniform address = scalar loads
  typedef const __attribute__((address_space(4))) int* cint_p;
  const long T0 = ((clong_p)a.blk_tile)[b];
  const int nt = (int)(((clong_p)a.blk_tile)[b + 1] - T0);
  const unsigned lds0 = (unsigned)(size_t)sm;
  const int ni = (a.S + NW * PPI - 1) / (NW * PPI);     // DMA instructions per wave and tile
  const int g4 = lane / LPP;
  const unsigned within = (unsigned)(lane % LPP) * 16u;
  int it = 0;
  for (int c = c0; c < c1; ++c) {
    const char* Tc = a.T + (long)c * PB + within;
    for (int t = 0; t < nt; ++t, ++it) {
      const int p = it % a.nbuf;
      cint_p ids = (cint_p)a.tsrc + (T0 + t) * a.S;
      for (int u = 0; u < ni; ++u) {
        int p0 = (u * NW + wv) * PPI;                    // first piece of this instruction
        if (p0 + PPI > a.S) p0 = a.S - PPI;              // (ragged end: the last pieces again)
        int id = 0;
#pragma unroll
        for (int q = 0; q < PPI; ++q) {
          const int idq = ids[p0 + q];                     // scalar loads (lgkmcnt): the copies stay in flight
          id = g4 == q ? idq : id;
        }
        dma16(Tc + (long)id * a.ldb, lds0 + (unsigned)(p * a.buf_bytes + p0 * PB));
      }
      // the copies of tile it - (nbuf - 1) have landed once at most (nbuf - 1) tiles' copies are outstanding
      const int keep = ni * (a.nbuf - 1);
      if (keep <= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else if (keep <= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else if (keep <= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (keep <= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (keep <= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (keep <= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if (keep <= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      __syncthreads();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && *(volatile unsigned*)sm == 0x12345u) a.sink[0] = 1;
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

__global__ void k_fill(double* p, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = (double)(i & 1023) * 0.001;
}

template <int NW, int LPP>
static void sweep(Args a, double staged, int rep) {
  const size_t smem = (size_t)a.nbuf * a.buf_bytes;
  (void)hipFuncSetAttribute((const void*)k_stage<NW, LPP, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  (void)hipFuncSetAttribute((const void*)k_stage<NW, LPP, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  for (int xc : {1, 4, 16, 64}) {
    a.xcd_chunk = xc;
    const long grid = (a.nblocks + 8 * xc - 1) / (8 * xc) * (8 * xc);
    const float ms = time_it([&] { hipLaunchKernelGGL((k_stage<NW, LPP, 0>), dim3((unsigned)grid), dim3(NW * 64), smem, 0, a); }, rep);
    printf("  block walks its chunks     xcd_chunk=%-3d          %8.1f us  %.2f TB/s staged\n", xc, ms * 1e3, staged / (ms * 1e-3) / 1e12);
  }
  for (int G : {8, 16, 32, 64, 128}) {
    a.G = G;
    const long ngroups = (a.nblocks + G - 1) / G;
    const long grid = (ngroups + 7) / 8 * 8 * (long)G * a.nchunk;
    const float ms = time_it([&] { hipLaunchKernelGGL((k_stage<NW, LPP, 1>), dim3((unsigned)grid), dim3(NW * 64), smem, 0, a); }, rep);
    printf("  group x chunk x block      G=%-3d                  %8.1f us  %.2f TB/s staged\n", G, ms * 1e3, staged / (ms * 1e-3) / 1e12);
  }
}

int main(int argc, char** argv) {
  const std::string dir = argv[1];
  const int N = atoi(argv[2]), NW = atoi(argv[3]), S = atoi(argv[4]), PB = atoi(argv[5]);
  const int only = argc > 6 ? atoi(argv[6]) : -1;         // >= 0: one configuration (for the counter passes): 0 = mode 0 xcd 4, G otherwise
  auto indptr = slurp<long>(dir + "/indptr.bin");
  auto blktile = slurp<long>(dir + "/blktile.bin");
  auto tilesrc0 = slurp<long>(dir + "/tilesrc0.bin");
  auto tilesrc = slurp<int>(dir + "/tilesrc.bin");
  const long n = (long)indptr.size() - 1;
  const long ntiles = (long)tilesrc0.size() - 1;
  std::vector<int> tsrc((size_t)ntiles * S + 64);
  for (long t = 0; t < ntiles; ++t) {
    const long b0 = tilesrc0[t], c = tilesrc0[t + 1] - b0;
    for (int s = 0; s < S; ++s) tsrc[(size_t)t * S + s] = tilesrc[b0 + (s < c ? s : c - 1)];
  }
  const int nchunk = (N * 8 + PB - 1) / PB;
  const int ldb = nchunk * PB;                           // the state padded to whole chunks
  Args a{};
  a.blk_tile = up(blktile); a.tsrc = up(tsrc);
  a.nblocks = (long)blktile.size() - 1; a.ldb = ldb; a.S = S; a.nchunk = nchunk; a.xcd_chunk = 4; a.G = 32;
  a.buf_bytes = (S * PB + 1023) / 1024 * 1024;
  a.nbuf = getenv("NBUF") ? atoi(getenv("NBUF")) : 2;
  if ((size_t)a.nbuf * a.buf_bytes > 160 * 1024) { printf("ring of %d x %d B exceeds the LDS\n", a.nbuf, a.buf_bytes); return 1; }
  double* T; (void)hipMalloc(&T, (size_t)(n + 64) * ldb);
  hipLaunchKernelGGL(k_fill, dim3((unsigned)(((n + 64) * (long)ldb / 8 + 255) / 256)), dim3(256), 0, 0, T, (n + 64) * (long)ldb / 8);
  a.T = (const char*)T;
  (void)hipMalloc(&a.sink, 8);
  const double staged = (double)tilesrc.size() * PB * nchunk;
  const double edges = (double)indptr[n];
  printf("n = %ld, N = %d: %d chunks of %d B (row stride %d B), %ld blocks, %.1f tiles of %d pieces per block and chunk, "
         "edges/sources %.2f; staged %.2f GB per pass (gather: %.2f GB), LDS %d x %d B\n", n, N, nchunk, PB, ldb, a.nblocks,
         (double)ntiles / a.nblocks, S, edges / tilesrc.size(), staged / 1e9, edges * N * 8 / 1e9, a.nbuf, a.buf_bytes);
  if (only >= 0) {
    const size_t smem = (size_t)a.nbuf * a.buf_bytes;
    if (NW != 8 || PB != 256) { printf("counter mode wants NW=8 PB=256\n"); return 1; }
    if (only == 0) {
      (void)hipFuncSetAttribute((const void*)k_stage<8, 16, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      const long grid = (a.nblocks + 31) / 32 * 32;
      for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_stage<8, 16, 0>), dim3((unsigned)grid), dim3(512), smem, 0, a);
    } else {
      a.G = only;
      (void)hipFuncSetAttribute((const void*)k_stage<8, 16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      const long ngroups = (a.nblocks + a.G - 1) / a.G;
      const long grid = (ngroups + 7) / 8 * 8 * (long)a.G * a.nchunk;
      for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_stage<8, 16, 1>), dim3((unsigned)grid), dim3(512), smem, 0, a);
    }
    (void)hipDeviceSynchronize();
    return 0;
  }
  const int rep = 3;
  if (NW == 8 && PB == 256) sweep<8, 16>(a, staged, rep);
  else if (NW == 16 && PB == 256) sweep<16, 16>(a, staged, rep);
  else if (NW == 8 && PB == 128) sweep<8, 8>(a, staged, rep);
  else if (NW == 8 && PB == 512) sweep<8, 32>(a, staged, rep);
  else if (NW == 4 && PB == 256) sweep<4, 16>(a, staged, rep);
  else printf("no instantiation for NW=%d PB=%d\n", NW, PB);
  return 0;
}
