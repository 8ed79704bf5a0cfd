// What does scattering (column, value) pairs into a wave's own LDS accumulator row cost on gfx950?
// (the inner operation of k_nam_step_sparse, diffuse.hip).  Every wave owns a 256-double row; per "edge"
// ACT lanes add a value into distinct random columns.  Variants:
//   0  ds_add_f64 (no return)                         -- what the kernel does
//   1  ds_read_b64, v_add_f64, ds_write_b64           -- legal: the row belongs to one wave, columns of an edge
//                                                        are distinct, the LDS serves a wave in program order
//   2  ds_add_u64 (integer, same addresses)           -- is it the fp unit or the atomic path?
//   3  ds_add_f32
//   4  ds_write_b64 only                              -- the crossbar / bank cost of the address pattern alone
// Reports clocks per wave instruction and CU at an assumed 2.3 GHz, 16 waves per CU.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_scatter_rate.hip -o lds_scatter_rate && ./lds_scatter_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int VAR>
__global__ __launch_bounds__(256) void k(const int* __restrict__ cols, double* out, int iters, int act) {
  __shared__ double sm[4 * 256];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double* acc = sm + wv * 256;
  for (int q = 0; q < 4; ++q) acc[lane + 64 * q] = 0.0;
  // 8 precomputed column patterns per wave, distinct inside a pattern
  int c[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) c[u] = cols[((blockIdx.x * 4 + wv) & 1023) * 512 + u * 64 + lane];
  const double v = 1.0 + lane;
  const bool on = lane < act;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (on) {
        if (VAR == 0) unsafeAtomicAdd(&acc[c[u]], v);
        if (VAR == 1) acc[c[u]] = acc[c[u]] + v;
        if (VAR == 2) atomicAdd((unsigned long long*)&acc[c[u]], (unsigned long long)lane);
        if (VAR == 3) unsafeAtomicAdd((float*)&acc[c[u]], (float)v);
        if (VAR == 4) acc[c[u]] = v;
      }
      if (VAR == 1 || VAR == 4) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
  }
  __syncthreads();
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc[lane] + acc[lane + 64] + acc[lane + 128] + acc[lane + 192];
}

template <int VAR>
static void run(const int* cols, double* out, int act, const char* what) {
  const int iters = 2000;
  dim3 grid(256 * 4 * 4);          // 4 workgroups of 4 waves per CU at a time, 4 rounds
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<VAR>), grid, dim3(256), 0, 0, cols, out, iters, act);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<VAR>), grid, dim3(256), 0, 0, cols, out, iters, act);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_cu = 8.0 * iters * grid.x * 4 / 256.0;
  printf("%-34s active lanes %2d: %7.3f ms  %6.1f clk per wave instruction and CU (%.2f lanes/clk/CU)\n", what, act, ms,
         ms * 1e-3 * 2.3e9 / instr_per_cu, act * instr_per_cu / (ms * 1e-3 * 2.3e9));
}

int main() {
  int* h = (int*)malloc(1024 * 512 * 4);
  uint64_t s = 88172645463325252ull;
  for (int w = 0; w < 1024 * 8; ++w) {           // a random permutation of 0..199 (+ the rest) per pattern: distinct columns
    int p[256];
    for (int i = 0; i < 256; ++i) p[i] = i;
    for (int i = 0; i < 64; ++i) {
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      const int j = i + (int)(s % (uint64_t)(200 - i));
      const int t = p[i]; p[i] = p[j]; p[j] = t;
    }
    for (int i = 0; i < 64; ++i) h[w * 64 + i] = p[i];
  }
  int* cols; double* out;
  (void)hipMalloc(&cols, 1024 * 512 * 4); (void)hipMemcpy(cols, h, 1024 * 512 * 4, hipMemcpyHostToDevice);
  (void)hipMalloc(&out, (size_t)256 * 16 * 256 * 8);
  for (int act : {16, 35, 64}) {
    run<0>(cols, out, act, "ds_add_f64");
    run<1>(cols, out, act, "ds_read_b64 + v_add_f64 + ds_write_b64");
    run<2>(cols, out, act, "ds_add_u64");
    run<3>(cols, out, act, "ds_add_f32");
    run<4>(cols, out, act, "ds_write_b64 only");
  }
  return 0;
}
