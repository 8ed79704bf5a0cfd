// Probe for the LDS-staged diffusion step: a workgroup owns a block of B consecutive rows (a cluster of
// cells that share neighbours, csrc/host_graph.c), stages one WC-column chunk of the state rows of ALL
// distinct neighbours of the block in LDS, then every (row, column pair) thread walks its row's edges in
// CSR order reading LDS.  Compared with the wave-per-row gather on the same order.
//   hipcc --offload-arch=gfx950 -O3 gather_lds.hip -o gather_lds && ./gather_lds <dir> <N>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include <algorithm>
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

template <int NQ2>
__global__ __launch_bounds__(256) void k_row(const long* __restrict__ indptr, const int* __restrict__ idx,
                                             const float* __restrict__ val, const double2* __restrict__ T, int ld2,
                                             long n, double2* __restrict__ out) {
  constexpr int U = 8;
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
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
    for (int l = 0; l < cnt; l += U) {
      double2 t[U][NQ2];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int j = __builtin_amdgcn_readlane(jl, (l + u) & 63);
        const double2* rp = T + (long)j * ld2;
#pragma unroll
        for (int q = 0; q < NQ2; ++q) t[u][q] = (lane + 64 * q < ld2) ? rp[lane + 64 * q] : make_double2(0, 0);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const double av = readlane_d(al, (l + u) & 63);
#pragma unroll
        for (int q = 0; q < NQ2; ++q) { acc[q].x = acc[q].x + av * t[u][q].x; acc[q].y = acc[q].y + av * t[u][q].y; }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < NQ2; ++q)
    if (lane + 64 * q < ld2) out[row * ld2 + lane + 64 * q] = acc[q];
}

struct Rec { float w; unsigned slot; };

template <int B, int WC, int CAP, int ME, int PADW>
__global__ __launch_bounds__(B * WC / 2) void k_lds(const long* __restrict__ indptr, const Rec* __restrict__ emeta,
                                                    const int* __restrict__ eidx, const long* __restrict__ src_ptr,
                                                    const int* __restrict__ src, const double* __restrict__ T, int ld,
                                                    long n, double* __restrict__ out) {
  constexpr int NT = B * WC / 2, TPR = WC / 2, SW = WC + PADW;      // SW: LDS row stride in doubles
  extern __shared__ char sm[];
  double* sdata = (double*)sm;
  Rec* smeta = (Rec*)(sm + (size_t)CAP * SW * 8);
  int* ssrc = (int*)(smeta + ME);
  int* sptr = ssrc + CAP;
  const int tid = threadIdx.x;
  const long b = blockIdx.x, r0 = b * B;
  const int rows = (int)((n - r0) < B ? (n - r0) : B);
  const long e0 = indptr[r0];
  const int nedges = (int)(indptr[r0 + rows] - e0);
  for (int i = tid; i < nedges && i < ME; i += NT) smeta[i] = emeta[e0 + i];
  for (int i = tid; i <= rows; i += NT) sptr[i] = (int)(indptr[r0 + i] - e0);
  const long s0 = src_ptr[b];
  const int nsrc_all = (int)(src_ptr[b + 1] - s0);
  const int nsrc = nsrc_all < CAP ? nsrc_all : CAP;      // sources past CAP: fetched from memory per edge
  for (int i = tid; i < nsrc; i += NT) ssrc[i] = src[s0 + i];
  __syncthreads();
  const int r = tid / TPR, cp = tid % TPR;
  const int es = r < rows ? sptr[r] : 0, ee = r < rows ? sptr[r + 1] : 0;
  for (int c0 = 0; c0 < ld; c0 += WC) {
    for (int s = r; s < nsrc; s += B) {
      const double2 v = *(const double2*)(T + (long)ssrc[s] * ld + c0 + cp * 2);
      *(double2*)(sdata + s * SW + cp * 2) = v;
    }
    __syncthreads();
    double2 acc = make_double2(0.0, 0.0);
#pragma unroll 4
    for (int e = es; e < ee; ++e) {
      const Rec rec = e < ME ? smeta[e] : emeta[e0 + e];
      double2 t;
      if (rec.slot < (unsigned)CAP) t = *(const double2*)(sdata + rec.slot * SW + cp * 2);
      else t = *(const double2*)(T + (long)eidx[e0 + e] * ld + c0 + cp * 2);
      acc.x = acc.x + (double)rec.w * t.x;
      acc.y = acc.y + (double)rec.w * t.y;
    }
    if (r < rows) *(double2*)(out + (r0 + r) * ld + c0 + cp * 2) = acc;
    __syncthreads();
  }
}


// ---- v2: edge records of a thread's row live in registers for all chunks; staging issues its loads in
// batches; chunk width chosen per block so that every listed source of the block fits (16 / 8 / 4 columns);
// sources the host did not list (slot 0xFFFF: more than SMAX distinct neighbours in the block) and edges
// beyond the LDS copy of the records are fetched from memory
extern __shared__ __align__(16) char sm[];
template <int B, int NT, int KMAX, int CAPB, int ME, int SMAX, int W>
__device__ __forceinline__ void lds_body(int rows, int nsrc, const Rec* __restrict__ emeta_blk, const int* __restrict__ eidx_blk,
                                         const double* __restrict__ T, int ld, double* __restrict__ out_blk) {
  constexpr int DUMP = SMAX;                           // slot that absorbs out-of-range staging writes
  const Rec* smeta = (const Rec*)(sm + CAPB);
  const int* ssrc = (const int*)(sm + CAPB + ME * 8);
  const int* sptr = ssrc + SMAX;
  constexpr int TPR = W / 2, SPP = NT / TPR;           // threads per row chunk, sources per staging pass
  static_assert(KMAX % 8 == 0 && KMAX <= 64, "KMAX");
  const int tid = threadIdx.x;
  const int r = tid / TPR, cp = tid % TPR;
  const bool active = r < rows && r < B;
  const int es = active ? sptr[r] : 0;
  const int deg = active ? sptr[r + 1] - es : 0;
  double w[KMAX];
  unsigned off[KMAX];
  unsigned long long miss = 0;                         // edges whose source is not in the block's list
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    Rec rc = Rec{0.0f, 0u};
    if (k < deg) {
      const int ei = es + k;
      rc = smeta[ei < ME ? ei : 0];
      if (ei >= ME) rc = emeta_blk[ei];
    }
    w[k] = (double)rc.w;
    const bool m = rc.slot == 0xFFFFu;
    if (m) miss |= 1ull << k;
    off[k] = (m ? 0u : rc.slot * (unsigned)(W * 8)) + (unsigned)cp * 16u;
  }
  int dmax = deg;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(dmax, o); dmax = t > dmax ? t : dmax; }
  dmax = __builtin_amdgcn_readfirstlane(dmax);
  // staging by LDS-DMA (global_load_lds_dwordx4: 16 bytes per lane straight into LDS at M0 + 16 * lane, no
  // VGPR round trip, asynchronous): one wave instruction brings the chunk of 64 / TPR consecutive sources.
  // The row each lane fetches is the same for every chunk: kept in registers.
  constexpr int NW = NT / 64, SPW = 64 / TPR;          // waves, sources per wave instruction
  constexpr int NS = (SMAX + NW * SPW - 1) / (NW * SPW);
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int part = lane % TPR;
  int sidx[NS];
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    const int sq = (u * NW + wv) * SPW + lane / TPR;
    sidx[u] = ssrc[sq < nsrc ? sq : 0];
  }
  for (int c0 = 0; c0 < ld; c0 += W) {
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      if ((u * NW + wv) * SPW < nsrc)
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(T + (long)sidx[u] * ld + c0 + part * 2),
                                         (void __attribute__((address_space(3)))*)(sm + (u * NW + wv) * SPW * (W * 8)), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    double2 acc = make_double2(0.0, 0.0);
#pragma unroll
    for (int k0 = 0; k0 < KMAX; k0 += 8) {
      if (k0 < dmax) {
        double2 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = *(const double2*)(sm + off[k0 + u]);
        if (__builtin_expect(__ballot(((miss >> k0) & 0xFFull) != 0) != 0, 0)) {
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if ((miss >> (k0 + u)) & 1ull) t[u] = *(const double2*)(T + (long)eidx_blk[es + k0 + u] * ld + c0 + cp * 2);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {                   // padding past a row's end: weight 0 on slot 0 (finite)
          acc.x = acc.x + w[k0 + u] * t[u].x;
          acc.y = acc.y + w[k0 + u] * t[u].y;
        }
      }
    }
    for (int e = es + KMAX; e < es + deg; ++e) {           // rows longer than the register copy
      Rec rc = smeta[e < ME ? e : 0];
      if (e >= ME) rc = emeta_blk[e];
      double2 t = *(const double2*)(sm + (rc.slot == 0xFFFFu ? 0u : rc.slot * (unsigned)(W * 8)) + (unsigned)cp * 16u);
      if (rc.slot == 0xFFFFu) t = *(const double2*)(T + (long)eidx_blk[e] * ld + c0 + cp * 2);
      acc.x = acc.x + (double)rc.w * t.x;
      acc.y = acc.y + (double)rc.w * t.y;
    }
    if (active) *(double2*)(out_blk + (long)r * ld + c0 + cp * 2) = acc;
    __syncthreads();
  }
}

template <int B, int NT, int KMAX, int CAPB, int ME, int SMAX, int W>
__global__ __launch_bounds__(NT) void k_lds2(const long* __restrict__ indptr, const Rec* __restrict__ emeta,
                                             const int* __restrict__ eidx, const long* __restrict__ src_ptr,
                                             const int* __restrict__ src, const double* __restrict__ T, int ld, long n,
                                             double* __restrict__ out) {
  Rec* smeta = (Rec*)(sm + CAPB);
  int* ssrc = (int*)(smeta + ME);
  int* sptr = ssrc + SMAX;
  const int tid = threadIdx.x;
  const long b = blockIdx.x, r0 = b * B;
  const int rows = (int)((n - r0) < B ? (n - r0) : B);
  const long e0 = indptr[r0];
  const int nedges = (int)(indptr[r0 + rows] - e0);
  for (int i = tid; i < nedges && i < ME; i += NT) smeta[i] = emeta[e0 + i];
  for (int i = tid; i <= rows; i += NT) sptr[i] = (int)(indptr[r0 + i] - e0);
  const long s0 = src_ptr[b];
  const int nsrc = (int)(src_ptr[b + 1] - s0);            // <= SMAX = CAPB / (8 W) (host-side cap)
  for (int i = tid; i < nsrc; i += NT) ssrc[i] = src[s0 + i];
  __syncthreads();
  lds_body<B, NT, KMAX, CAPB, ME, SMAX, W>(rows, nsrc, emeta + e0, eidx + e0, T, ld, out + r0 * ld);
}

struct Ctx {
  const long* indptr; const int* idx; const float* val; const Rec* emeta; const long* src_ptr; const int* src;
  const double* T; int ld; long n; double* out; long nblocks;
};
template <int NQ2> static void l_row(Ctx* c) {
  hipLaunchKernelGGL(k_row<NQ2>, dim3((unsigned)((c->n + 3) / 4)), dim3(256), 0, 0, c->indptr, c->idx, c->val,
                     (const double2*)c->T, c->ld / 2, c->n, (double2*)c->out);
}
template <int B, int WC, int CAP, int ME, int PADW> static void l_lds(Ctx* c) {
  constexpr size_t smem = (size_t)CAP * (WC + PADW) * 8 + (size_t)ME * 8 + (size_t)CAP * 4 + (size_t)(B + 1) * 4;
  static bool once = false;
  if (!once) {
    once = true;
    (void)hipFuncSetAttribute((const void*)k_lds<B, WC, CAP, ME, PADW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    printf("      [B=%d WC=%d CAP=%d ME=%d pad=%d: %zu B LDS per workgroup of %d threads]\n", B, WC, CAP, ME, PADW, smem, B * WC / 2);
  }
  hipLaunchKernelGGL((k_lds<B, WC, CAP, ME, PADW>), dim3((unsigned)c->nblocks), dim3(B * WC / 2), smem, 0, c->indptr, c->emeta,
                     c->idx, c->src_ptr, c->src, c->T, c->ld, c->n, c->out);
}

template <int B, int NT, int KMAX, int CAPB, int ME, int SMAX, int W> static void l_lds2(Ctx* c) {
  constexpr size_t smem = (size_t)CAPB + (size_t)ME * 8 + (size_t)SMAX * 4 + (size_t)(B + 1) * 4;
  static bool once = false;
  if (!once) {
    once = true;
    (void)hipFuncSetAttribute((const void*)k_lds2<B, NT, KMAX, CAPB, ME, SMAX, W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    printf("      [v2 B=%d NT=%d KMAX=%d W=%d data %d B, ME=%d SMAX=%d: %zu B LDS]\n", B, NT, KMAX, W, CAPB, ME, SMAX, smem);
  }
  hipLaunchKernelGGL((k_lds2<B, NT, KMAX, CAPB, ME, SMAX, W>), dim3((unsigned)c->nblocks), dim3(NT), smem, 0, c->indptr, c->emeta,
                     c->idx, c->src_ptr, c->src, c->T, c->ld, c->n, c->out);
}
template <typename F> static float time_it(F f, Ctx* c) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(c); (void)hipEventRecord(e0); f(c); f(c); f(c); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) printf("      HIP error: %s\n", hipGetErrorString(err));
  return ms / 3;
}
static double checksum(const double* d, long n, int ld, int N) {
  std::vector<double> h((size_t)n * ld);
  (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (long i = 0; i < n; ++i) for (int j = 0; j < N; ++j) s += h[(size_t)i * ld + j];
  return s;
}

int main(int argc, char** argv) {
  const std::string dir = argv[1];
  const int N = atoi(argv[2]);
  const int ld = (N + 15) / 16 * 16;
  printf("N = %d, ld = %d (row %d bytes)\n", N, ld, ld * 8);
  for (int B : {32, 64, 128}) {
    const std::string b = dir + "/b" + std::to_string(B) + "_";
    FILE* probe = fopen((b + "indptr.bin").c_str(), "rb");
    if (!probe) continue;
    fclose(probe);
    auto indptr = slurp<long>(b + "indptr.bin");
    auto idx = slurp<int>(b + "idx.bin");
    auto val = slurp<float>(b + "val.bin");
    const long n = (long)indptr.size() - 1;
    Ctx c{};
    c.indptr = up(indptr); c.idx = up(idx); c.val = up(val); c.n = n; c.ld = ld;
    double* T; (void)hipMalloc(&T, (size_t)(n + 64) * ld * 8);
    {
      std::vector<double> h((size_t)n * ld);
      unsigned s = 1;
      for (auto& x : h) { s = s * 1664525u + 1013904223u; x = (s >> 8) * (1.0 / 16777216.0); }
      (void)hipMemcpy(T, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    }
    c.T = T;
    (void)hipMalloc(&c.out, (size_t)(n + 256) * ld * 8);
    const double gathered = (double)idx.size() * N * 8;
    float ms = ld <= 128 ? time_it(l_row<1>, &c) : time_it(l_row<2>, &c);
    printf("B=%-3d order: wave-per-row   n=%ld  %8.1f us  %6.2f TB/s gathered  checksum %.9e\n", B, n, ms * 1e3,
           gathered / (ms * 1e-3) / 1e12, checksum(c.out, n, ld, N));
    for (int cap : {544, 960, 1920}) {
      const std::string s = b + "cap" + std::to_string(cap) + "_";
      FILE* pr = fopen((s + "srcptr.bin").c_str(), "rb");
      if (!pr) continue;
      fclose(pr);
      auto srcptr = slurp<long>(s + "srcptr.bin");
      auto src = slurp<int>(s + "src.bin");
      auto slot = slurp<unsigned short>(s + "slot.bin");
      std::vector<Rec> meta(idx.size());
      size_t over = 0;
      for (size_t e = 0; e < idx.size(); ++e) { meta[e].w = val[e]; meta[e].slot = slot[e] == 0xFFFF ? 0xFFFFu : slot[e]; over += slot[e] == 0xFFFF; }
      c.emeta = up(meta); c.src_ptr = up(srcptr); c.src = up(src); c.nblocks = (long)srcptr.size() - 1;
      (void)hipMemset(c.out, 0, (size_t)n * ld * 8);
      float t = -1;
      const char* what = "";
      if (B == 64 && cap == 960) { t = time_it(l_lds2<64, 512, 40, 122880, 3328, 960, 16>, &c); what = "v2 W=16 K40"; }
      if (B == 64 && cap == 1920) { t = time_it(l_lds2<64, 512, 40, 122880, 3328, 1920, 8>, &c); what = "v2 W=8 K40"; }
      if (B == 32 && cap == 544) { t = time_it(l_lds2<32, 256, 40, 69632, 1280, 544, 16>, &c); what = "v2 W=16 B=32 (2WG)"; }
      if (t >= 0)
        printf("B=%-3d LDS-staged %-16s sources=%zu (edges/sources %.2f, %.2f%% edges past cap %d)  %8.1f us   checksum %.9e\n", B, what,
               src.size(), (double)idx.size() / src.size(), 100.0 * over / idx.size(), cap, t * 1e3, checksum(c.out, n, ld, N));
      (void)hipFree((void*)c.emeta); (void)hipFree((void*)c.src_ptr); (void)hipFree((void*)c.src);
    }
    (void)hipFree((void*)c.indptr); (void)hipFree((void*)c.idx); (void)hipFree((void*)c.val); (void)hipFree(T); (void)hipFree(c.out);
  }
  return 0;
}
