// Probe for the grouped diffusion step: how much faster does the row-gather SpMM get when one wave
// owns R destination rows and fetches every distinct neighbour row of the group ONCE (the products
// are formed per destination row from the registers holding that row)?  Inputs are binary files
// written by tools/micro/gather_group.py from a synthetic kNN graph: CSR in device order and, per
// R, the merged neighbour lists of groups of R consecutive rows.
//   hipcc --offload-arch=gfx950 -O3 gather_group.hip -o gather_group && ./gather_group <dir> <N>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
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
  T* d; hipMalloc(&d, v.size() * sizeof(T) + 64);
  hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
  return d;
}
__device__ __forceinline__ double readlane_d(double v, int l) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  return __hiloint2double(__builtin_amdgcn_readlane(hi, l), __builtin_amdgcn_readlane(lo, l));
}

// wave per row (the shape of k_nam_step)
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
        const double av = readlane_d(al, (l + u) & 63);      // 0 past the end
#pragma unroll
        for (int q = 0; q < NQ2; ++q) { acc[q].x = acc[q].x + av * t[u][q].x; acc[q].y = acc[q].y + av * t[u][q].y; }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < NQ2; ++q)
    if (lane + 64 * q < ld2) out[row * ld2 + lane + 64 * q] = acc[q];
}

// wave per group of R rows: entry e = (neighbour row, mask of the group's rows that have the edge,
// R weights); the neighbour row is fetched once
template <int R, int NQ2, int U>
__global__ __launch_bounds__(256) void k_group(const long* __restrict__ gptr, const int* __restrict__ ej,
                                               const unsigned* __restrict__ em, const float* __restrict__ ew,
                                               const double2* __restrict__ T, int ld2, long n_groups,
                                               double2* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long g = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (g >= n_groups) return;
  const long start = gptr[g], end = gptr[g + 1];
  double2 acc[R][NQ2];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int q = 0; q < NQ2; ++q) acc[r][q] = make_double2(0, 0);
  for (long base = start; base < end; base += 64) {
    const bool ok = base + lane < end;
    const int jl = ok ? ej[base + lane] : 0;
    const unsigned ml = ok ? em[base + lane] : 0u;
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
        const unsigned m = __builtin_amdgcn_readlane(ml, (l + u) & 63);
        const float* wp = ew + (base + l + u) * R;               // wave-uniform: scalar loads
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (m & (1u << r)) {
            const double w = (double)wp[r];
#pragma unroll
            for (int q = 0; q < NQ2; ++q) {
              acc[r][q].x = acc[r][q].x + w * t[u][q].x;
              acc[r][q].y = acc[r][q].y + w * t[u][q].y;
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int q = 0; q < NQ2; ++q)
      if (lane + 64 * q < ld2) out[(g * R + r) * ld2 + lane + 64 * q] = acc[r][q];
}

static float time_it(void (*launch)(void*), void* a) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(a); hipEventRecord(e0); launch(a); launch(a); launch(a); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 3;
}

struct Ctx {
  const long* indptr; const int* idx; const float* val; const double2* T; int ld2; long n; double2* out;
  const long* gptr; const int* ej; const unsigned* em; const float* ew; long ng; int R, U;
};
template <int NQ2> static void l_row(void* p) {
  Ctx* c = (Ctx*)p;
  hipLaunchKernelGGL(k_row<NQ2>, dim3((unsigned)((c->n + 3) / 4)), dim3(256), 0, 0, c->indptr, c->idx, c->val, c->T, c->ld2, c->n, c->out);
}
template <int R, int NQ2, int U> static void l_group(void* p) {
  Ctx* c = (Ctx*)p;
  hipLaunchKernelGGL((k_group<R, NQ2, U>), dim3((unsigned)((c->ng + 3) / 4)), dim3(256), 0, 0, c->gptr, c->ej, c->em, c->ew, c->T, c->ld2, c->ng, c->out);
}

static double checksum(const double2* d, long n2) {
  std::vector<double2> h(n2);
  hipMemcpy(h.data(), d, n2 * sizeof(double2), hipMemcpyDeviceToHost);
  double s = 0;
  for (long i = 0; i < n2; ++i) s += h[i].x + h[i].y;
  return s;
}

int main(int argc, char** argv) {
  const std::string dir = argv[1];
  const int N = atoi(argv[2]);
  const int ld = (N + 3) / 4 * 4, ld2 = ld / 2;
  printf("N = %d (row %d bytes)\n", N, ld * 8);
  for (const char* tag : {"rcm", "g4", "g8", "g16", "c8", "c16"}) {
    const std::string b = dir + "/" + tag + "_";
    FILE* probe = fopen((b + "indptr.bin").c_str(), "rb");
    if (!probe) continue;
    fclose(probe);
    auto indptr = slurp<long>(b + "indptr.bin");
    auto idx = slurp<int>(b + "idx.bin");
    auto val = slurp<float>(b + "val.bin");
    const long n = (long)indptr.size() - 1;
    Ctx c{};
    c.indptr = up(indptr); c.idx = up(idx); c.val = up(val); c.n = n; c.ld2 = ld2;
    double2* T; hipMalloc(&T, (size_t)n * ld * 8 + (1 << 20));
    {
      std::vector<double> h((size_t)n * ld);
      unsigned s = 1;
      for (auto& x : h) { s = s * 1664525u + 1013904223u; x = (s >> 8) * (1.0 / 16777216.0); }
      hipMemcpy(T, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    }
    c.T = T;
    hipMalloc(&c.out, (size_t)(n + 64) * ld * 8);
    const double gathered = (double)idx.size() * N * 8;
    float ms = ld2 <= 64 ? time_it(l_row<1>, &c) : time_it(l_row<2>, &c);
    printf("%-4s wave-per-row            n=%ld nnz=%zu  %8.1f us  %6.2f TB/s gathered  checksum %.6e\n", tag, n, idx.size(), ms * 1e3,
           gathered / (ms * 1e-3) / 1e12, checksum(c.out, n * ld2));
    int R = 0;
    if (tag[0] == 'g' || tag[0] == 'c') R = atoi(tag + 1);
    if (R) {
      auto gptr = slurp<long>(b + "gptr.bin");
      auto ej = slurp<int>(b + "ej.bin");
      auto em = slurp<unsigned>(b + "em.bin");
      auto ew = slurp<float>(b + "ew.bin");
      c.gptr = up(gptr); c.ej = up(ej); c.em = up(em); c.ew = up(ew); c.ng = (long)gptr.size() - 1; c.R = R;
      hipMemset(c.out, 0, (size_t)n * ld * 8);
      float t4 = 0, t8 = 0;
      if (ld2 <= 64) {
        if (R == 4) { t4 = time_it(l_group<4, 1, 4>, &c); t8 = time_it(l_group<4, 1, 8>, &c); }
        if (R == 8) { t4 = time_it(l_group<8, 1, 4>, &c); t8 = time_it(l_group<8, 1, 8>, &c); }
        if (R == 16) { t4 = time_it(l_group<16, 1, 4>, &c); t8 = time_it(l_group<16, 1, 8>, &c); }
      } else {
        if (R == 4) { t4 = time_it(l_group<4, 2, 4>, &c); t8 = time_it(l_group<4, 2, 8>, &c); }
        if (R == 8) { t4 = time_it(l_group<8, 2, 4>, &c); t8 = time_it(l_group<8, 2, 8>, &c); }
        if (R == 16) { t4 = time_it(l_group<16, 2, 4>, &c); t8 = time_it(l_group<16, 2, 8>, &c); }
      }
      printf("%-4s wave-per-group R=%-2d      entries=%zu (edges/entries %.2f)  U=4: %8.1f us  U=8: %8.1f us   checksum %.6e\n", tag, R,
             ej.size(), (double)idx.size() / ej.size(), t4 * 1e3, t8 * 1e3, checksum(c.out, n * ld2));
      hipFree((void*)c.gptr); hipFree((void*)c.ej); hipFree((void*)c.em); hipFree((void*)c.ew);
    }
    hipFree((void*)c.indptr); hipFree((void*)c.idx); hipFree((void*)c.val); hipFree(T); hipFree(c.out);
  }
  return 0;
}
