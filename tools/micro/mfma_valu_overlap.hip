// Does VALU work of one wave hide under the f64 MFMAs of another wave on the same SIMD?
// Block = 8 waves (2 per SIMD).  Even waves issue dependent-free v_mfma_f64_16x16x4_f64 chains,
// odd waves issue integer (or f64) VALU chains.  Times: MFMA waves alone, VALU waves alone, both.
//   hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"
typedef double v4d __attribute__((ext_vector_type(4)));

template <int KIND>   // 0: int VALU, 1: f64 VALU, 2: LDS reads (ds_read2_b64), 3: LDS atomics
__global__ __launch_bounds__(512) void k(int do_mfma, int do_valu, int iters, double* out) {
  __shared__ double lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = i;
  __syncthreads();
  const int wv = threadIdx.x >> 6;
  const bool mf = (wv & 4) == 0;            // waves 0-3: one per SIMD; waves 4-7: the other wave of each SIMD
  if (mf) {
    if (!do_mfma) return;
    v4d a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    const double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-6;
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
    }
    out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
  } else {
    if (!do_valu) return;
    if (KIND == 0) {
      unsigned a = threadIdx.x, b = 3, c = 5, d = 7;
      for (int i = 0; i < iters * 16; ++i) {       // 64 int VALU instructions per MFMA-loop iteration (4 x 16 cycles each = 256 clk = 4 MFMAs)
        a = a * 3u + 1u; b = b * 5u + a; c = c ^ (b >> 3); d = d + c;
      }
      out[blockIdx.x * 512 + threadIdx.x] = (double)(a + b + c + d);
    } else if (KIND == 2) {
      double a = 0, b = 0;
      const int lane = threadIdx.x & 63;
      for (int i = 0; i < iters * 4; ++i) {        // 8 x ds_read_b64-class instructions per 4 iterations... (2 doubles each)
        const int o = (i * 64) & 2047;
        a += lds[o + lane]; b += lds[o + 1024 + lane];
        a += lds[o + 64 + lane]; b += lds[o + 1088 + lane];
      }
      out[blockIdx.x * 512 + threadIdx.x] = a + b;
    } else if (KIND == 3) {
      unsigned* h = (unsigned*)lds;
      const int lane = threadIdx.x & 63;
      for (int i = 0; i < iters * 4; ++i) {
        atomicAdd(&h[(lane * 33 + i) & 8191], 1u);
        atomicAdd(&h[(lane * 17 + 3 * i) & 8191], 1u);
      }
      out[blockIdx.x * 512 + threadIdx.x] = 0;
    } else {
      double a = threadIdx.x, b = 3, c = 5, d = 7;
      for (int i = 0; i < iters * 16; ++i) {
        a = __builtin_fma(a, 1.0000001, 1e-9); b = __builtin_fma(b, 0.9999999, a); c = c + b; d = d * 0.999 + c;
      }
      out[blockIdx.x * 512 + threadIdx.x] = a + b + c + d;
    }
  }
}

template <int KIND>
float run(int m, int v, int iters, double* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, m, v, iters, out);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, m, v, iters, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f;
}

int main() {
  double* out; hipMalloc(&out, 256 * 512 * 8);
  const int iters = 20000;
  printf("one block per CU, 2 waves per SIMD; %d iterations (4 MFMAs = 256 clk | 64 VALU instr = 256 clk)\n", iters);
  printf("int VALU : mfma alone %.0f us, valu alone %.0f us, both %.0f us\n", run<0>(1, 0, iters, out), run<0>(0, 1, iters, out), run<0>(1, 1, iters, out));
  printf("LDS reads: mfma alone %.0f us, lds alone %.0f us, both %.0f us\n", run<2>(1, 0, iters, out), run<2>(0, 1, iters, out), run<2>(1, 1, iters, out));
  printf("LDS atom : mfma alone %.0f us, lds alone %.0f us, both %.0f us\n", run<3>(1, 0, iters, out), run<3>(0, 1, iters, out), run<3>(1, 1, iters, out));
  printf("f64 VALU : mfma alone %.0f us, valu alone %.0f us, both %.0f us\n", run<1>(1, 0, iters, out), run<1>(0, 1, iters, out), run<1>(1, 1, iters, out));
  return 0;
}
