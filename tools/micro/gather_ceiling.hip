// How fast can a wave-per-row kernel gather rows of a table on this chip, independent of the
// diffusion kernel?  Each wave sums `deg` rows of `row_bytes` bytes picked at (pseudo)random from a
// table of `n_rows` rows (16-byte lanes, 8 loads in flight -- the access pattern of k_nam_step
// without the CSR, the weights or the write-out).  Reports useful TB/s for several row widths and
// table sizes (L2-resident ... HBM-resident).
//   hipcc --offload-arch=gfx950 -O3 gather_ceiling.hip -o gather_ceiling && ./gather_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#pragma clang diagnostic ignored "-Wunused-value"

template <int NQ2>
__global__ __launch_bounds__(256) void k(const double2* __restrict__ T, int ld2, const int* __restrict__ idx, int deg,
                                          long n_out, double2* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_out) return;
  double2 acc[NQ2];
#pragma unroll
  for (int q = 0; q < NQ2; ++q) acc[q] = make_double2(0, 0);
  const int* my = idx + row * deg;
  for (int e = 0; e < deg; e += 8) {
    double2 t[8][NQ2];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = __builtin_amdgcn_readfirstlane(my[e + u]);
      const double2* rp = T + (long)j * ld2;
#pragma unroll
      for (int q = 0; q < NQ2; ++q) t[u][q] = (lane + 64 * q < ld2) ? rp[lane + 64 * q] : make_double2(0, 0);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int q = 0; q < NQ2; ++q) { acc[q].x += t[u][q].x; acc[q].y += t[u][q].y; }
  }
#pragma unroll
  for (int q = 0; q < NQ2; ++q)
    if (lane + 64 * q < ld2) out[row * ld2 + lane + 64 * q] = acc[q];
}

int main() {
  const int deg = 40;
  printf("wave-per-row gather of %d rows per output row, 16-byte lanes, 8 loads in flight\n", deg);
  printf("%10s %10s %12s %10s %12s\n", "row bytes", "table MB", "window rows", "us", "useful TB/s");
  for (int N : {50, 100, 200}) {
    const int ld = (N + 3) / 4 * 4, ld2 = ld / 2;
    for (long n_rows : {20000L, 200000L, 2000000L}) {
      if ((long)n_rows * ld * 8 > (6L << 30)) continue;
      for (long window : {2000L, 0L}) {                      // neighbours within +-window rows (banded) or anywhere
        const long n_out = n_rows;
        std::vector<int> h((size_t)n_out * deg);
        unsigned s = 12345;
        for (long r = 0; r < n_out; ++r)
          for (int e = 0; e < deg; ++e) {
            s = s * 1664525u + 1013904223u;
            long j = window ? r - window + (long)(s % (2 * window)) : (long)(s % n_rows);
            if (j < 0) j = 0;
            if (j >= n_rows) j = n_rows - 1;
            h[(size_t)r * deg + e] = (int)j;
          }
        double2 *T, *out; int* idx;
        hipMalloc(&T, (size_t)n_rows * ld * 8); hipMalloc(&out, (size_t)n_out * ld * 8); hipMalloc(&idx, h.size() * 4);
        hipMemset(T, 0, (size_t)n_rows * ld * 8);
        hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto launch = [&]() {
          dim3 g((unsigned)((n_out + 3) / 4));
          if (ld2 <= 64) hipLaunchKernelGGL(k<1>, g, dim3(256), 0, 0, T, ld2, idx, deg, n_out, out);
          else hipLaunchKernelGGL(k<2>, g, dim3(256), 0, 0, T, ld2, idx, deg, n_out, out);
        };
        launch(); hipEventRecord(e0); launch(); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 2;
        const double bytes = (double)n_out * deg * N * 8;
        printf("%10d %10.0f %12s %10.1f %12.2f\n", N * 8, n_rows * ld * 8 / 1e6, window ? "+-2000" : "any", ms * 1e3, bytes / (ms * 1e-3) / 1e12);
        hipFree(T); hipFree(out); hipFree(idx);
      }
    }
  }
  return 0;
}
