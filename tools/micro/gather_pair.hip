// probe: wave-per-row vs two-rows-per-wave (half-wave each) gather, 400-byte rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k_one(const double2* __restrict__ T, int ld2, const int* __restrict__ idx, int deg,
                                          long n_out, double2* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_out) return;
  double2 acc = make_double2(0, 0);
  const int* my = idx + row * deg;
  for (int e = 0; e < deg; e += 8) {
    double2 t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = __builtin_amdgcn_readfirstlane(my[e + u]);
      const double2* rp = T + (long)j * ld2;
      t[u] = (lane < ld2) ? rp[lane] : make_double2(0, 0);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { acc.x += t[u].x; acc.y += t[u].y; }
  }
  if (lane < ld2) out[row * ld2 + lane] = acc;
}
// half-wave per output row; neighbour ids read per lane (same address within a half: broadcast)
__global__ __launch_bounds__(256) void k_pair(const double2* __restrict__ T, int ld2, const int* __restrict__ idx, int deg,
                                          long n_out, double2* __restrict__ out) {
  const int lane = threadIdx.x & 63, hl = lane & 31, h = lane >> 5;
  const long row = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + h;
  if (row >= n_out) return;
  double2 acc = make_double2(0, 0);
  const int* my = idx + row * deg;
  const char* Tb = (const char*)T;
  for (int e = 0; e < deg; e += 8) {
    double2 t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned j = (unsigned)my[e + u];
      const unsigned off = j * (unsigned)(ld2 * 16) + hl * 16;
      t[u] = (hl < ld2) ? *(const double2*)(Tb + off) : make_double2(0, 0);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { acc.x += t[u].x; acc.y += t[u].y; }
  }
  if (hl < ld2) out[row * ld2 + hl] = acc;
}
int main() {
  const int deg = 40, N = 50, ld = 52, ld2 = 26;
  for (long n_rows : {200000L, 1000000L}) {
    std::vector<int> h((size_t)n_rows * deg);
    unsigned s = 12345; const long window = 2000;
    for (long r = 0; r < n_rows; ++r) for (int e = 0; e < deg; ++e) {
      s = s * 1664525u + 1013904223u;
      long j = r - window + (long)(s % (2 * window)); if (j < 0) j = 0; if (j >= n_rows) j = n_rows - 1;
      h[(size_t)r * deg + e] = (int)j;
    }
    double2 *T, *out; int* idx;
    hipMalloc(&T, (size_t)n_rows * ld * 8); hipMalloc(&out, (size_t)n_rows * ld * 8); hipMalloc(&idx, h.size() * 4);
    hipMemset(T, 0, (size_t)n_rows * ld * 8);
    hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
      auto launch = [&]() {
        if (mode == 0) hipLaunchKernelGGL(k_one, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, 0, T, ld2, idx, deg, n_rows, out);
        else hipLaunchKernelGGL(k_pair, dim3((unsigned)((n_rows + 7) / 8)), dim3(256), 0, 0, T, ld2, idx, deg, n_rows, out);
      };
      launch(); hipEventRecord(e0); launch(); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 2;
      printf("n=%ld mode=%s  %.1f us  %.2f TB/s useful\n", n_rows, mode ? "pair" : "one", ms * 1e3, (double)n_rows * deg * N * 8 / (ms * 1e-3) / 1e12);
    }
    hipFree(T); hipFree(out); hipFree(idx);
  }
}
