// Does VALU work hide under v_mfma_i32_32x32x32_i8 on gfx950 -- from another wave of the same SIMD, and from
// the same wave's instruction stream?  And at what clock does the chip run the integer matrix pipe?
// Block = 8 waves (2 per SIMD), one block per CU.  Waves 0-3 (one per SIMD): MFMA chains (4 independent
// accumulators); waves 4-7: f32/int VALU chains.  KIND 1: one wave per SIMD, 4 MFMAs and 24 VALU
// instructions interleaved in ONE stream (8 passes = 32 clk per MFMA leaves 7 VALU slots of 4 clk).
//   hipcc --offload-arch=gfx950 -O3 mfma_i8_overlap.hip -o mfma_i8_overlap && ./mfma_i8_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int KIND, int NV = 0>
__global__ __launch_bounds__(512) void k(int do_mfma, int do_valu, int iters, int* out, unsigned long long* clk) {
  const int wv = threadIdx.x >> 6;
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  v4i a = {(int)threadIdx.x, 3, 5, 7}, b = {1, (int)threadIdx.x, 2, 9};
  int res = 0;
  if (KIND == 1) {
    if (wv >= 4) return;
    v16i c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float x = threadIdx.x, y = 3.f, z = 5.f, w = 7.f;
    for (int i = 0; i < iters; ++i) {
      if (do_mfma) {
        c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c3, 0, 0, 0);
      }
      if (do_valu) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
          x = fmaf(x, 1.0000001f, 1e-9f); y = fmaf(y, 0.9999999f, x); z = z + y; w = fmaf(w, 0.999f, z);
        }
      }
    }
    res = c0[0] + c1[1] + c2[2] + c3[3] + (int)(x + y + z + w);
  } else if (KIND == 2 || KIND == 3 || KIND == 4) {
    // one wave per SIMD: MFMA (KIND 2 only), NV INDEPENDENT vector instructions (8 chains of fma / med3, which do not
    // pack), MFMA, ... fenced so that the order stays
    if (wv >= 4 && KIND != 4) return;                  // KIND 4: two such waves per SIMD
    v16i c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float x[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) x[q] = threadIdx.x + q;
    const float lo = (float)do_valu, hi = 1e30f * do_valu;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        if (KIND == 2 || KIND == 4) {
          if (m == 0) c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
          if (m == 1) c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
          if (m == 2) c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0);
          if (m == 3) c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c3, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < NV; ++u) {
          const int q = (m * NV + u) & 7;
          x[q] = (u & 1) ? __builtin_amdgcn_fmed3f(x[q], lo, hi) : fmaf(x[q], 1.0000001f, lo);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    float sx = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) sx += x[q];
    res = c0[0] + c1[1] + c2[2] + c3[3] + (int)sx;
  } else if (wv < 4) {
    if (!do_mfma) return;
    v16i c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c3, 0, 0, 0);
    }
    res = c0[0] + c1[1] + c2[2] + c3[3];
  } else {
    if (!do_valu) return;
    float x = threadIdx.x, y = 3.f, z = 5.f, w = 7.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {                 // 32 VALU instructions = 128 clk = the 4 MFMAs of the other wave
        x = fmaf(x, 1.0000001f, 1e-9f); y = fmaf(y, 0.9999999f, x); z = z + y; w = fmaf(w, 0.999f, z);
      }
    }
    res = (int)(x + y + z + w);
  }
  out[blockIdx.x * 512 + threadIdx.x] = res;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = __builtin_readcyclecounter() - t0; clk[1] = wall_clock64() - r0; }
}

template <int KIND, int NV = 0>
void run(const char* what, int m, int v, int iters, int* out, unsigned long long* clk) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KIND, NV>), dim3(256), dim3(512), 0, 0, m, v, iters, out, clk);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND, NV>), dim3(256), dim3(512), 0, 0, m, v, iters, out, clk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  printf("%-44s %8.0f us   (s_memtime %llu ticks, wall clock %llu ticks of 100 MHz)\n", what, ms * 1e3f, h[0], h[1]);
}

int main() {
  int* out; hipMalloc(&out, 256 * 512 * 4);
  unsigned long long* clk; hipMalloc(&clk, 16);
  const int iters = 40000;
  printf("one block per CU; %d iterations of 4 MFMAs (32x32x32 i8: 4 x 32 clk)\n", iters);
  run<0>("two waves/SIMD: mfma wave alone", 1, 0, iters, out, clk);
  run<0>("two waves/SIMD: valu wave alone (32 instr/it)", 0, 1, iters, out, clk);
  run<0>("two waves/SIMD: both", 1, 1, iters, out, clk);
  run<1>("one wave/SIMD: mfma only", 1, 0, iters, out, clk);
  run<1>("one wave/SIMD: valu only (24 instr/it)", 0, 1, iters, out, clk);
  run<1>("one wave/SIMD: interleaved in one stream", 1, 1, iters, out, clk);
  run<2, 0>("one wave/SIMD, fenced: mfma only", 1, 1, iters, out, clk);
  run<2, 2>("one wave/SIMD, fenced: mfma + 2 valu each", 1, 1, iters, out, clk);
  run<3, 2>("one wave/SIMD, fenced: 2 valu per slot, no mfma", 1, 1, iters, out, clk);
  run<2, 4>("one wave/SIMD, fenced: mfma + 4 valu each", 1, 1, iters, out, clk);
  run<3, 4>("one wave/SIMD, fenced: 4 valu per slot, no mfma", 1, 1, iters, out, clk);
  run<2, 6>("one wave/SIMD, fenced: mfma + 6 valu each", 1, 1, iters, out, clk);
  run<3, 6>("one wave/SIMD, fenced: 6 valu per slot, no mfma", 1, 1, iters, out, clk);
  run<2, 8>("one wave/SIMD, fenced: mfma + 8 valu each", 1, 1, iters, out, clk);
  run<3, 8>("one wave/SIMD, fenced: 8 valu per slot, no mfma", 1, 1, iters, out, clk);
  run<2, 12>("one wave/SIMD, fenced: mfma + 12 valu each", 1, 1, iters, out, clk);
  run<3, 12>("one wave/SIMD, fenced: 12 valu per slot, no mfma", 1, 1, iters, out, clk);
  run<4, 0>("two waves/SIMD, fenced: mfma only", 1, 1, iters, out, clk);
  run<4, 2>("two waves/SIMD, fenced: mfma + 2 valu each", 1, 1, iters, out, clk);
  run<4, 4>("two waves/SIMD, fenced: mfma + 4 valu each", 1, 1, iters, out, clk);
  run<4, 5>("two waves/SIMD, fenced: mfma + 5 valu each", 1, 1, iters, out, clk);
  run<4, 6>("two waves/SIMD, fenced: mfma + 6 valu each", 1, 1, iters, out, clk);
  run<4, 8>("two waves/SIMD, fenced: mfma + 8 valu each", 1, 1, iters, out, clk);
  const double ops = 256.0 * 4 * iters * 4.0 * 65536.0;
  printf("(4 MFMA waves per CU: %.3g integer ops per run)\n", ops);
  return 0;
}
