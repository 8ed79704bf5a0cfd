// Issue rate of the f32 -> f64 widening on gfx950 against a f64 fused multiply-add and against the same widening done
// with 32-bit integer instructions (what bounds k_nam_step32 / k_nam_step32h).
//   hipcc --offload-arch=gfx950 -O3 -o cvt_rate tools/micro/cvt_rate.hip && ./cvt_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ in, double* __restrict__ out, int iters) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  float x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = in[(t + i * 64) & 1023];
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) {                       // cvt + add
        double d;
        asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(x[i]));
        acc[i] += d;
      } else if (MODE == 1) {                // fma only (same dependent adds)
        acc[i] = __builtin_fma(acc[i], 1.0000001, (double)x[0]);
      } else if (MODE == 2) {                // integer widening of a positive normal float + add
        const unsigned b = __float_as_uint(x[i]);
        unsigned hi, lo;
        asm volatile("v_lshrrev_b32 %0, 3, %1" : "=v"(hi) : "v"(b));
        asm volatile("v_add_u32 %0, 0x38000000, %1" : "=v"(hi) : "v"(hi));
        asm volatile("v_lshlrev_b32 %0, 29, %1" : "=v"(lo) : "v"(b));
        acc[i] += __hiloint2double((int)hi, (int)lo);
      } else {                               // add only
        acc[i] += 1.0000001;
      }
      x[i] += 1.0f * (MODE == 9);
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i];
  out[t] = s;
}

template <int MODE>
double run(const float* in, double* out, int iters) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int blocks = 256 * 8;                // 8 workgroups of 4 waves per CU: 8 waves per SIMD
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, 16);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  // wave-instructions of the inner statement per SIMD: 8 waves x iters x 8
  return ms * 1e-3 / (8.0 * iters * 8) ;      // seconds per wave-statement per SIMD
}

int main() {
  float* in; double* out;
  hipMalloc(&in, 4096); hipMalloc(&out, 8 * 256 * 256 * 8);
  float h[1024];
  for (int i = 0; i < 1024; ++i) h[i] = 1.0f + i * 1e-3f;
  hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
  const int iters = 20000;
  const double ghz = 2.4;
  const double t_add = run<3>(in, out, iters), t_cvt = run<0>(in, out, iters), t_fma = run<1>(in, out, iters), t_int = run<2>(in, out, iters);
  printf("per wave statement and SIMD, in ns (clk at %.1f GHz):\n", ghz);
  printf("  f64 add alone              %.2f ns (%.1f clk)\n", t_add * 1e9, t_add * 1e9 * ghz);
  printf("  f64 fma alone              %.2f ns (%.1f clk)\n", t_fma * 1e9, t_fma * 1e9 * ghz);
  printf("  v_cvt_f64_f32 + add        %.2f ns (%.1f clk) -> cvt %.1f clk\n", t_cvt * 1e9, t_cvt * 1e9 * ghz, (t_cvt - t_add) * 1e9 * ghz);
  printf("  3 int ops + add            %.2f ns (%.1f clk) -> widening %.1f clk\n", t_int * 1e9, t_int * 1e9 * ghz, (t_int - t_add) * 1e9 * ghz);
  return 0;
}
