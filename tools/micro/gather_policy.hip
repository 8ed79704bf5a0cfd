// Does the cache policy of the gathering loads move the gather ceiling of the walk?  The probe of
// gather_ceiling.hip (a wave sums 40 rows picked within +-2000 rows of its own; 16-byte lanes, 8 rows in
// flight) with the loads written as inline assembly so that the sc0 / sc1 / nt bits can be set:
//   0 plain   1 sc0   2 sc1   3 sc0 sc1   4 nt   5 nt sc1   6 compiler's own load (reference)
// On gfx942-class caches sc1 makes the vector L1 treat the load as "miss always" while the L2 keeps the line;
// nt additionally marks the line as streaming in the L2 (measured in round 2 inside k_nam_step: worse).
//   hipcc --offload-arch=gfx950 -O3 gather_policy.hip -o gather_policy && ./gather_policy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v2d __attribute__((ext_vector_type(2)));

template <int MODE>
__device__ __forceinline__ v2d ld16(const double* base, unsigned off) {
  v2d r;
  if (MODE == 0) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(off), "s"(base));
  if (MODE == 1) asm volatile("global_load_dwordx4 %0, %1, %2 sc0" : "=v"(r) : "v"(off), "s"(base));
  if (MODE == 2) asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(r) : "v"(off), "s"(base));
  if (MODE == 3) asm volatile("global_load_dwordx4 %0, %1, %2 sc0 sc1" : "=v"(r) : "v"(off), "s"(base));
  if (MODE == 4) asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(r) : "v"(off), "s"(base));
  if (MODE == 5) asm volatile("global_load_dwordx4 %0, %1, %2 sc1 nt" : "=v"(r) : "v"(off), "s"(base));
  return r;
}

template <int NQ2, int MODE>
__global__ __launch_bounds__(256) void k(const double* __restrict__ T, int ld2, const int* __restrict__ idx, int deg,
                                          long n_out, v2d* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_out) return;
  v2d acc[NQ2];
#pragma unroll
  for (int q = 0; q < NQ2; ++q) acc[q] = (v2d){0, 0};
  const int* my = idx + row * deg;
  for (int e = 0; e < deg; e += 8) {
    v2d t[8][NQ2];
    int js[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) js[u] = __builtin_amdgcn_readfirstlane(my[e + u]);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const double* rp = T + (long)js[u] * ld2 * 2;
#pragma unroll
      for (int q = 0; q < NQ2; ++q) {
        const int c2 = lane + 64 * q < ld2 ? lane + 64 * q : ld2 - 1;      // lanes past the row re-read its last pair
        if (MODE == 6) t[u][q] = ((const v2d*)rp)[c2];
        else t[u][q] = ld16<MODE>(rp, (unsigned)c2 * 16u);
      }
    }
    if (MODE != 6) {
      asm volatile("s_waitcnt vmcnt(0)");
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int q = 0; q < NQ2; ++q) asm volatile("" : "+v"(t[u][q]));
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int q = 0; q < NQ2; ++q) acc[q] += t[u][q];
  }
#pragma unroll
  for (int q = 0; q < NQ2; ++q)
    if (lane + 64 * q < ld2) out[row * ld2 + lane + 64 * q] = acc[q];
}

template <int MODE>
static void run(int N, long n_rows, const double* T, const int* idx, v2d* out, double* hout) {
  const int deg = 40, ld = (N + 3) / 4 * 4, ld2 = ld / 2;
  const long n_out = n_rows;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto launch = [&]() {
    dim3 g((unsigned)((n_out + 3) / 4));
    if (ld2 <= 64) hipLaunchKernelGGL((k<1, MODE>), g, dim3(256), 0, 0, T, ld2, idx, deg, n_out, out);
    else hipLaunchKernelGGL((k<2, MODE>), g, dim3(256), 0, 0, T, ld2, idx, deg, n_out, out);
  };
  launch(); (void)hipEventRecord(e0); launch(); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 2;
  (void)hipMemcpy(hout, out, 4096 * 8, hipMemcpyDeviceToHost);
  double cs = 0; for (int i = 0; i < 4096; ++i) cs += hout[i];
  static const char* names[] = {"asm plain", "asm sc0", "asm sc1", "asm sc0 sc1", "asm nt", "asm sc1 nt", "compiler"};
  printf("%6d B rows  %8ld rows  %-12s %9.1f us  %6.2f TB/s  checksum %.6e\n", N * 8, n_rows, names[MODE], ms * 1e3,
         (double)n_out * deg * N * 8 / (ms * 1e-3) / 1e12, cs);
}

int main() {
  const int deg = 40;
  static double hout[4096];
  for (int N : {100, 200}) {
    const int ld = (N + 3) / 4 * 4;
    for (long n_rows : {200000L, 2000000L}) {
      const long window = 2000;
      std::vector<int> h((size_t)n_rows * deg);
      unsigned s = 12345;
      for (long r = 0; r < n_rows; ++r)
        for (int e = 0; e < deg; ++e) {
          s = s * 1664525u + 1013904223u;
          long j = r - window + (long)(s % (2 * window));
          if (j < 0) j = 0;
          if (j >= n_rows) j = n_rows - 1;
          h[(size_t)r * deg + e] = (int)j;
        }
      std::vector<double> ht((size_t)n_rows * ld);
      for (size_t i = 0; i < ht.size(); ++i) ht[i] = 1e-3 * (double)(i % 9973);
      double* T; v2d* out; int* idx;
      (void)hipMalloc(&T, ht.size() * 8); (void)hipMalloc(&out, ht.size() * 8); (void)hipMalloc(&idx, h.size() * 4);
      (void)hipMemcpy(T, ht.data(), ht.size() * 8, hipMemcpyHostToDevice);
      (void)hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
      run<6>(N, n_rows, T, idx, out, hout);
      run<0>(N, n_rows, T, idx, out, hout);
      run<1>(N, n_rows, T, idx, out, hout);
      run<2>(N, n_rows, T, idx, out, hout);
      run<3>(N, n_rows, T, idx, out, hout);
      run<4>(N, n_rows, T, idx, out, hout);
      run<5>(N, n_rows, T, idx, out, hout);
      (void)hipFree(T); (void)hipFree(out); (void)hipFree(idx);
    }
  }
  return 0;
}
