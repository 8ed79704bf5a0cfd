// Probe of the LDS-tiled dense walk step (round 3), row-major state.
//
//   block   B = NW * R consecutive device rows; wave w keeps the sums of rows w, w + NW, ... in registers
//           (lane = column pair, as the wave-per-row kernel)
//   tiles   the block's distinct neighbour rows in ascending order of the CALLER's column index, S at a time:
//           whole rows (ld * 8 contiguous bytes each) arrive by LDS-DMA into one of two LDS buffers while the
//           previous tile is consumed
//   records the block's CSR entries regrouped by (tile, wave, row): {weight, source inside the tile, row inside
//           the wave}; lane j of a wave holds record j of its segment, v_readlane broadcasts it
//   order   canonical CSR rows list their columns in ascending caller's index, so every row still adds its
//           products in CSR order: bit-identical to the wave-per-row kernel
//
//   hipcc --offload-arch=gfx950 -O3 walk_tiles.hip -o walk_tiles && ./walk_tiles <dir> <N> <NW> <R> <S>
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
  T* d; (void)hipMalloc(&d, v.size() * sizeof(T) + 4096);
  (void)hipMemset(d, 0, v.size() * sizeof(T) + 4096);
  (void)hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
  return d;
}
__device__ __forceinline__ double readlane_d(double v, int l) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  return __hiloint2double(__builtin_amdgcn_readlane(hi, l), __builtin_amdgcn_readlane(lo, l));
}

// ---- reference: wave per row (the production kernel's shape, U rows in flight)
// the same on a column slice: row stride ld2 (double2), columns [c2, c2 + w2) of every row
template <int NQ2, int U>
__global__ __launch_bounds__(256) void k_row_slice(const long* __restrict__ indptr, const int* __restrict__ idx,
                                                   const float* __restrict__ val, const double2* __restrict__ T, int ld2,
                                                   int c2, int w2, long n, double2* __restrict__ out, int chunk) {
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
        const double2* rp = T + (long)j * ld2 + c2;
#pragma unroll
        for (int q = 0; q < NQ2; ++q) t[u][q] = (lane + 64 * q < w2) ? rp[lane + 64 * q] : make_double2(0, 0);
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
      const double2* rp = T + (long)j * ld2 + c2;
#pragma unroll
      for (int q = 0; q < NQ2; ++q) {
        const double2 t = (lane + 64 * q < w2) ? rp[lane + 64 * q] : make_double2(0, 0);
        acc[q].x = acc[q].x + av * t.x; acc[q].y = acc[q].y + av * t.y;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < NQ2; ++q)
    if (lane + 64 * q < w2) {
      const double2 own = T[row * ld2 + c2 + lane + 64 * q];
      out[row * ld2 + c2 + lane + 64 * q] = make_double2(acc[q].x + own.x, acc[q].y + own.y);
    }
}

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

// ---- the LDS-tiled step
struct Rec8 { float w; unsigned short slot; unsigned char row, pad; };
struct TileArgs {
  const long* blk_tile; const int* tile_src;    // tile_src: S ids per tile (fixed stride, last ones repeated)
  const long* seg; const Rec8* rec;
  const double* Tin; double* Tout;
  long n, nblocks; int ld, S, xcd_chunk, buf_bytes, piece_bytes, halves;
};

extern __shared__ __align__(16) char sm[];

// LDS-DMA of 16 bytes per lane: LDS address = lds_base (wave-uniform) + 16 * lane.  Inline assembly: hipcc then
// neither serialises consecutive copies behind s_waitcnt vmcnt(0) nor counts them (waits are placed by hand).
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}

__device__ __forceinline__ void dma4(const void* gsrc, unsigned lds_base) {       // 4 bytes per lane
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}

// MODE 0: full; 1: staging only; 2: edge walk only
//
// Everything the tile loop reads from memory comes in by LDS-DMA issued from inline assembly -- the rows of the next
// tile, this wave's records of the next tile, and (two tiles ahead) the next tile's source ids and segment bounds --
// so that hipcc has no vector-memory operation of its own in the loop: it cannot count the copies, and with loads of
// its own in flight it places s_waitcnt vmcnt(0) in the middle of the edge walk (false register dependencies).
// LDS: 2 row buffers | 2 x NW record slots of 512 B (64 records) | 2 source-id lists of 1 KiB | 2 segment-bound lists of 1 KiB.
template <int NQ2, int R, int NW, int NI, int MODE>
__global__ __launch_bounds__(NW * 64) void k_walk_tiles(TileArgs a) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long bb = blockIdx.x >> 3, x = blockIdx.x & 7;
  const long b = (bb / a.xcd_chunk) * (8 * (long)a.xcd_chunk) + x * a.xcd_chunk + (bb % a.xcd_chunk);
  if (b >= a.nblocks) return;
  const int half = blockIdx.y;
  const int ldb = a.ld * 8, pb = a.piece_bytes;             // row stride of the state; bytes of a source staged per tile
  const int col0b = half * pb;                               // this pass's first byte inside a row
  const long T0 = a.blk_tile[b];
  const int nt = (int)(a.blk_tile[b + 1] - T0);
  const long r0 = b * (long)(NW * R);
  const unsigned lds0 = (unsigned)(size_t)sm;                // LDS byte address of the dynamic segment
  const int REC0 = 2 * a.buf_bytes, ID0 = REC0 + 2 * NW * 512, SEG0 = ID0 + 2048;
  // this lane's 16 bytes of every staging instruction of its wave: (source inside the tile, byte inside its piece)
  int s_of[NI], within[NI];
#pragma unroll
  for (int u = 0; u < NI; ++u) {
    const int o = (u * NW + wv) * 1024 + lane * 16;
    int s = o / pb;
    within[u] = o - s * pb;
    if (s >= a.S) { s = a.S - 1; within[u] = 0; }
    s_of[u] = s;
  }
  const char* Tb = (const char*)a.Tin + col0b;
  auto stage_rows = [&](int t, int p) {                     // source ids of tile t from the id list (LDS), then the copies
    int sid[NI];
#pragma unroll
    for (int u = 0; u < NI; ++u) sid[u] = *(const int*)(sm + ID0 + (t & 1) * 1024 + s_of[u] * 4);
#pragma unroll
    for (int u = 0; u < NI; ++u)
      if ((u * NW + wv) * 1024 < a.buf_bytes)
        dma16(Tb + (long)sid[u] * ldb + within[u], lds0 + (unsigned)(p * a.buf_bytes + (u * NW + wv) * 1024));
  };
  auto stage_recs = [&](int t, long& e0u) -> int {          // this wave's records of tile t; returns their number
    const long e0 = *(const long*)(sm + SEG0 + (t & 1) * 1024 + wv * 8);
    const long e1 = *(const long*)(sm + SEG0 + (t & 1) * 1024 + wv * 8 + 8);
    e0u = __builtin_amdgcn_readfirstlane((int)(e0 & 0xffffffffl)) | ((long)__builtin_amdgcn_readfirstlane((int)(e0 >> 32)) << 32);
    dma4((const char*)(a.rec + e0u) + lane * 4, lds0 + (unsigned)(REC0 + ((t & 1) * NW + wv) * 512));
    dma4((const char*)(a.rec + e0u) + 256 + lane * 4, lds0 + (unsigned)(REC0 + ((t & 1) * NW + wv) * 512 + 256));
    return __builtin_amdgcn_readfirstlane((int)(e1 - e0));
  };
  auto stage_meta = [&](int t) {                            // source ids and segment bounds of tile t (two waves)
    if (wv == 0) dma16((const char*)(a.tile_src + (T0 + t) * a.S) + lane * 16, lds0 + (unsigned)(ID0 + (t & 1) * 1024));
    if (wv == 1 % NW) dma16((const char*)(a.seg + (T0 + t) * NW) + lane * 16, lds0 + (unsigned)(SEG0 + (t & 1) * 1024));
  };
  double2 acc[R][NQ2];
#pragma unroll
  for (int i = 0; i < R; ++i)
#pragma unroll
    for (int q = 0; q < NQ2; ++q) acc[i][q] = make_double2(0.0, 0.0);
  int n_cur = 0, n_next = 0;
  long e_cur = 0, e_next = 0;
  if (nt > 0) {
    stage_meta(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (MODE != 2) stage_rows(0, 0);
    n_cur = stage_recs(0, e_cur);
    if (nt > 1) stage_meta(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  for (int t = 0; t < nt; ++t) {
    const int p = t & 1;
    if (t + 1 < nt) {
      if (MODE != 2) stage_rows(t + 1, p ^ 1);
      n_next = stage_recs(t + 1, e_next);
      // the lists of tile t + 2 replace those of tile t, which every wave read a tile ago; they land before this
      // iteration's barrier and are read at the top of the next one
      if (t + 2 < nt) stage_meta(t + 2);
    }
    if (MODE != 1) {
      const unsigned base = (unsigned)(p * a.buf_bytes) + (unsigned)lane * 16u;
      double w_cur = 0.0;
      unsigned sr_cur = 0xff000000u;
      if (lane < n_cur) {
        const Rec8 r = *(const Rec8*)(sm + REC0 + (p * NW + wv) * 512 + lane * 8);
        w_cur = (double)r.w;
        sr_cur = (unsigned)r.slot * (unsigned)pb + ((unsigned)r.row << 24);   // byte offset of the source in the tile | row
      }
      const unsigned rowj = sr_cur >> 24;
      int j = 0;
      // software pipeline: the source rows of edges j + 1 and j + 2 are on their way from LDS while edge j is added
      const unsigned s0 = __builtin_amdgcn_readlane(sr_cur, 0) & 0xffffffu;
      const unsigned s1 = __builtin_amdgcn_readlane(sr_cur, 1) & 0xffffffu;
      double2 v[NQ2], vn[NQ2], vnn[NQ2];
#pragma unroll
      for (int q = 0; q < NQ2; ++q) v[q] = *(const double2*)(sm + base + s0 + q * 1024);
#pragma unroll
      for (int q = 0; q < NQ2; ++q) vn[q] = *(const double2*)(sm + base + s1 + q * 1024);
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const int c = __popcll(__ballot(rowj == (unsigned)i));
        for (int k = 0; k < c; ++k) {
          const unsigned s2 = __builtin_amdgcn_readlane(sr_cur, (j + 2) & 63) & 0xffffffu;
#pragma unroll
          for (int q = 0; q < NQ2; ++q) vnn[q] = *(const double2*)(sm + base + s2 + q * 1024);
          const double wj = readlane_d(w_cur, j);
#pragma unroll
          for (int q = 0; q < NQ2; ++q) {
            acc[i][q].x = acc[i][q].x + wj * v[q].x;
            acc[i][q].y = acc[i][q].y + wj * v[q].y;
          }
#pragma unroll
          for (int q = 0; q < NQ2; ++q) { v[q] = vn[q]; vn[q] = vnn[q]; }
          ++j;
        }
      }
      if (n_cur > 64) {
        // segments longer than a wave (rare): the rest one record at a time from memory
        for (int jj = 64; jj < n_cur; ++jj) {
          const Rec8 r = a.rec[e_cur + jj];
          const double wj = (double)r.w;
          double2 tq[NQ2];
#pragma unroll
          for (int q = 0; q < NQ2; ++q) tq[q] = *(const double2*)(sm + base + (unsigned)r.slot * (unsigned)pb + q * 1024);
#pragma unroll
          for (int i = 0; i < R; ++i)
            if (r.row == i) {
#pragma unroll
              for (int q = 0; q < NQ2; ++q) { acc[i][q].x = acc[i][q].x + wj * tq[q].x; acc[i][q].y = acc[i][q].y + wj * tq[q].y; }
            }
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    n_cur = n_next; e_cur = e_next;
  }
  const int pairs = pb / 16;
  const double2* Tin2 = (const double2*)((const char*)a.Tin + col0b);
  double2* Tout2 = (double2*)((char*)a.Tout + col0b);
  const int ld2 = a.ld >> 1;
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const long row = r0 + (long)i * NW + wv;
    if (row < a.n) {
#pragma unroll
      for (int q = 0; q < NQ2; ++q)
        if (lane + 64 * q < pairs) {
          const double2 own = Tin2[row * ld2 + lane + 64 * q];
          Tout2[row * ld2 + lane + 64 * q] = make_double2(acc[i][q].x + own.x, acc[i][q].y + own.y);
        }
    }
  }
}

// ---- producer / consumer variant: no workgroup barrier in the tile loop.
//   waves 0 .. NPROD-1   producers: each issues K of the NPROD * K one-KiB LDS-DMA pieces of every tile (the tile's
//                        4 KiB record blob, then its S source rows) into a ring of NB slots, plus the source ids of
//                        the tile NB ahead into its own id ring; a tile is published (LDS counter `landed`) once the
//                        producer's copies of it have landed: s_waitcnt vmcnt(K + 1) after issuing the NEXT tile, so
//                        one to two tiles per producer are always in flight
//   waves NPROD .. 15    consumers: rows cw, cw + NCONS, ... of the block; wait for `landed`, walk their records of the
//                        tile (blob header: uint16 offsets per consumer), count themselves in `done`; a slot is reused
//                        when all NCONS consumers are done with it.  A fast wave runs up to NB - 1 tiles ahead of a
//                        slow one: the per-tile imbalance of the barrier version (x 1.6) averages out.
__global__ void k_compare(const double* __restrict__ A, const double* __restrict__ B, long n, int ld, int N,
                          unsigned long long* __restrict__ bad) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * (long)N) return;
  const long row = i / N; const int col = (int)(i % N);
  if (!(A[row * ld + col] == B[row * ld + col])) atomicAdd(bad, 1ull);
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

template <int NQ2, int R, int NW, int NI, int MODE>
static float run(TileArgs& a, const char* what, double gathered, double staged) {
  const size_t smem = (size_t)2 * a.buf_bytes + 2 * NW * 512 + 4096;
  (void)hipFuncSetAttribute((const void*)k_walk_tiles<NQ2, R, NW, NI, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  const long grid = (a.nblocks + 8 * a.xcd_chunk - 1) / (8 * a.xcd_chunk) * (8 * a.xcd_chunk);
  const float ms = time_it([&] { hipLaunchKernelGGL((k_walk_tiles<NQ2, R, NW, NI, MODE>), dim3((unsigned)grid, a.halves), dim3(NW * 64), smem, 0, a); });
  printf("  LDS-tiled %-16s NW=%d R=%d S=%d halves=%d xcd_chunk=%-3d %8.1f us  (%.2f TB/s of edge bytes, %.2f TB/s staged)\n", what, NW, R,
         a.S, a.halves, a.xcd_chunk, ms * 1e3, gathered / (ms * 1e-3) / 1e12, staged / (ms * 1e-3) / 1e12);
  return ms;
}

int main(int argc, char** argv) {
  const std::string dir = argv[1];
  const int N = atoi(argv[2]), NW = atoi(argv[3]), R = atoi(argv[4]), S = atoi(argv[5]);
  const int halves = argc > 6 ? atoi(argv[6]) : 1;
  int ld = (N + 3) / 4 * 4;
  if ((ld * 8) % 256 == 0) ld += 4;                     // row stride off multiples of 256 bytes
  auto indptr = slurp<long>(dir + "/indptr.bin");
  auto idx = slurp<int>(dir + "/idx.bin");
  auto val = slurp<float>(dir + "/val.bin");
  auto rec = slurp<Rec8>(dir + "/rec.bin");
  auto blktile = slurp<long>(dir + "/blktile.bin");
  auto tilesrc0 = slurp<long>(dir + "/tilesrc0.bin");
  auto tilesrc = slurp<int>(dir + "/tilesrc.bin");
  auto seg = slurp<long>(dir + "/seg.bin");
  const long n = (long)indptr.size() - 1;
  const long ntiles = (long)tilesrc0.size() - 1;
  std::vector<int> tsrc((size_t)ntiles * S + 64);
  for (long t = 0; t < ntiles; ++t) {
    const long b0 = tilesrc0[t], c = tilesrc0[t + 1] - b0;
    for (int s = 0; s < S; ++s) tsrc[(size_t)t * S + s] = tilesrc[b0 + (s < c ? s : c - 1)];
  }
  // (the producer / consumer variant of round 3, k_walk_pc -- two DMA waves feeding fourteen walking waves through a ring of
  // LDS slots -- was removed in round 5: it staged no faster than the barrier version (5.8 TB/s) and its sums were never
  // brought to bit-identity; HISTORY.md 5 keeps the measurement)
  TileArgs a{};
  a.blk_tile = up(blktile); a.tile_src = up(tsrc); a.seg = up(seg); a.rec = up(rec);
  a.n = n; a.nblocks = (long)blktile.size() - 1; a.ld = ld; a.S = S; a.xcd_chunk = 4; a.halves = halves;
  a.piece_bytes = ((N + halves - 1) / halves + 1) / 2 * 16;
  a.buf_bytes = (S * a.piece_bytes + 1023) / 1024 * 1024;
  const long* d_indptr = up(indptr); const int* d_idx = up(idx); const float* d_val = up(val);
  double *T, *O1, *O2;
  const size_t rm = (size_t)(n + 64) * ld;
  (void)hipMalloc(&T, rm * 8); (void)hipMalloc(&O1, rm * 8); (void)hipMalloc(&O2, rm * 8);
  hipLaunchKernelGGL(k_fill, dim3((unsigned)((rm + 255) / 256)), dim3(256), 0, 0, T, (long)rm);
  (void)hipMemset(O1, 0, rm * 8); (void)hipMemset(O2, 0, rm * 8);
  a.Tin = T; a.Tout = O2;
  const double gathered = (double)idx.size() * N * 8, staged = (double)tilesrc.size() * a.piece_bytes * halves;
  printf("n = %ld, N = %d (row stride %d B, %d pass(es) of %d B), nnz/row %.1f, %ld blocks of %d rows, %.1f tiles per block, "
         "edges/sources %.2f, LDS 2 x %d B\n", n, N, ld * 8, halves, a.piece_bytes, (double)idx.size() / n, a.nblocks, NW * R,
         (double)ntiles / a.nblocks, (double)idx.size() / tilesrc.size(), a.buf_bytes);
  {
    const int chunk = 128;
    const long grid = ((n + 3) / 4 + 8 * chunk - 1) / (8 * chunk) * (8 * chunk);
    float ms;
    if (ld <= 128) ms = time_it([&] { hipLaunchKernelGGL((k_row<1, 10>), dim3((unsigned)grid), dim3(256), 0, 0, d_indptr, d_idx, d_val, (const double2*)T, ld / 2, n, (double2*)O1, chunk); });
    else ms = time_it([&] { hipLaunchKernelGGL((k_row<2, 10>), dim3((unsigned)grid), dim3(256), 0, 0, d_indptr, d_idx, d_val, (const double2*)T, ld / 2, n, (double2*)O1, chunk); });
    printf("  wave-per-row (10 rows in flight, xcd chunk %d)  %8.1f us  (%.2f TB/s gathered)\n", chunk, ms * 1e3,
           gathered / (ms * 1e-3) / 1e12);
  }
  unsigned long long* bad; (void)hipMalloc(&bad, 8);
  auto check = [&](const char* what) {
    (void)hipMemset(bad, 0, 8);
    hipLaunchKernelGGL(k_compare, dim3((unsigned)((n * (long)N + 255) / 256)), dim3(256), 0, 0, O2, O1, n, ld, N, bad);
    unsigned long long h; (void)hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    printf("      %s: %llu of %ld outputs differ from the wave-per-row result\n", what, h, n * (long)N);
  };
  if (ld > 128) {
    // the wave-per-row gather in column slices (one launch per slice): does a smaller L2 footprint per cluster pay?
    for (int slices : {2, 4}) {
      for (int chunk : {128, 512, 2048}) {
        const long grid = ((n + 3) / 4 + 8 * chunk - 1) / (8 * chunk) * (8 * chunk);
        const int ld2 = ld / 2, w = (ld2 + slices - 1) / slices;
        (void)hipMemset(O2, 0, rm * 8);
        const float ms = time_it([&] {
          for (int sidx = 0; sidx < slices; ++sidx) {
            const int c2 = sidx * w, w2 = (c2 + w <= ld2) ? w : ld2 - c2;
            hipLaunchKernelGGL((k_row_slice<1, 10>), dim3((unsigned)grid), dim3(256), 0, 0, d_indptr, d_idx, d_val, (const double2*)T, ld2, c2, w2, n, (double2*)O2, chunk);
          }
        });
        printf("  wave-per-row in %d column slices (xcd chunk %4d)  %8.1f us  (%.2f TB/s gathered)\n", slices, chunk, ms * 1e3, gathered / (ms * 1e-3) / 1e12);
        if (chunk == 128) check("sliced");
      }
    }
  }
#define RUNSET(NQ2, R_, NW_, NI_)                                                  \
  for (int xc : {4, 1, 16, 32, 64, 256}) {                                         \
    a.xcd_chunk = xc;                                                              \
    run<NQ2, R_, NW_, NI_, 0>(a, "full", gathered, staged);                        \
    if (xc == 4) check("full");                                                    \
  }                                                                                \
  a.xcd_chunk = 4;                                                                 \
  run<NQ2, R_, NW_, NI_, 1>(a, "staging only", gathered, staged);                  \
  run<NQ2, R_, NW_, NI_, 2>(a, "edge walk only", gathered, staged);
  const int ni = (a.buf_bytes / 1024 + NW - 1) / NW;
  const bool wide = a.piece_bytes > 1024;
  if (wide && NW == 16 && R == 8 && ni <= 5) { RUNSET(2, 8, 16, 5) }
  else if (wide && NW == 8 && R == 16 && ni <= 10) { RUNSET(2, 16, 8, 10) }
  else if (!wide && NW == 16 && R == 16 && ni <= 5) { RUNSET(1, 16, 16, 5) }
  else if (!wide && NW == 8 && R == 32 && ni <= 10) { RUNSET(1, 32, 8, 10) }
  else printf("no instantiation for NW=%d R=%d piece=%d ni=%d\n", NW, R, a.piece_bytes, ni);
  return 0;
}
