// What does v_mfma_f64_16x16x4_f64 sustain on this chip, and what does feeding it cost?
//   variant 0: operands in registers, CH independent accumulator chains per wave
//   variant 1: B operand read from LDS before every MFMA (ds_read_b64, conflict-free), A in registers
//   variant 2: as 1, plus the A operand re-loaded from global memory (L2 resident) every KQ steps
// waves per SIMD = 1, 2, 4; reports TFLOP/s (2 * 16*16*4 flop per instruction per wave).
//   hipcc --offload-arch=gfx950 -O3 mfma_f64_rate.hip -o mfma_f64_rate && ./mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int CH, int VAR>
__global__ void k(const double* __restrict__ g, double* out, int iters) {
  extern __shared__ double sm[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 50 * 4 * 48; i += blockDim.x) sm[i] = 1e-3 * i;
  __syncthreads();
  constexpr int KQ = 50;
  double a[KQ];
#pragma unroll
  for (int q = 0; q < KQ; ++q) a[q] = g[(lane & 15) * 200 + 4 * q + (lane >> 4)];
  v4d acc[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) acc[c] = (v4d){0, 0, 0, 0};
  const double* bp = sm + (lane >> 4) * 48 + (lane & 15);
  for (int it = 0; it < iters; ++it) {
    if (VAR == 2) {
      const double* gp = g + ((it * 7 + blockIdx.x) & 1023) * 3200 + (lane & 15) * 200 + (lane >> 4);
#pragma unroll
      for (int q = 0; q < KQ; ++q) a[q] = gp[4 * q];
    }
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        // VAR 3: one LDS read feeds two chains, VAR 4: four chains (the b of chain c & ~1 / c & ~3: the compiler reads it once)
        if (VAR == 5 || VAR == 6) {      // B operand from global memory (a 25.6 KB strip: L1 resident), one load per MFMA (5) / per two MFMAs (6)
          const double bg = g[(size_t)(4 * q + (lane >> 4)) * 16 + (lane & 15) + 3200 * ((VAR == 5 ? c : (c >> 1)) & 1)];
          acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], bg, acc[c], 0, 0, 0);
          continue;
        }
        const double b = VAR == 3 ? bp[4 * q * 48 + 16 * ((c >> 1) & 1)] : VAR == 4 ? bp[4 * q * 48 + 16 * ((c >> 2) & 1)]
                         : VAR >= 1 ? bp[4 * q * 48 + 16 * (c & 1) + (c >> 1)] : a[(q + c + 1) % KQ];
        acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b, acc[c], 0, 0, 0);
      }
    }
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CH, int VAR>
static void run(int waves_per_simd, const double* g, double* out) {
  const int threads = 64 * 4 * waves_per_simd > 1024 ? 1024 : 64 * 4 * waves_per_simd;
  const int blocks_per_cu = (64 * 4 * waves_per_simd) / threads;
  const int iters = 200;
  dim3 grid(256 * blocks_per_cu * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const size_t smem = blocks_per_cu > 1 ? 60 * 1024 : 100 * 1024;     // keeps blocks_per_cu blocks on a CU, not more
  (void)hipFuncSetAttribute((const void*)k<CH, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL((k<CH, VAR>), grid, dim3(threads), smem, 0, g, out, iters);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<CH, VAR>), grid, dim3(threads), smem, 0, g, out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double flop = 2048.0 * 50 * CH * iters * (double)grid.x * (threads / 64);
  printf("variant %d  chains %d  waves/SIMD %d: %7.2f ms  %6.1f TFLOP/s\n", VAR, CH, waves_per_simd, ms, flop / (ms * 1e-3) / 1e12);
}

int main() {
  double *g, *out;
  (void)hipMalloc(&g, 1024 * 3200 * 8 + 4096); (void)hipMemset(g, 0, 1024 * 3200 * 8 + 4096);
  (void)hipMalloc(&out, 256 * 16 * 1024 * 8);
  for (int w : {1, 2, 4}) { run<2, 0>(w, g, out); run<4, 0>(w, g, out); }
  for (int w : {1, 2, 4}) { run<2, 1>(w, g, out); run<4, 1>(w, g, out); }
  for (int w : {2, 4}) { run<2, 2>(w, g, out); }
  for (int w : {1, 2, 4}) { run<2, 5>(w, g, out); run<4, 5>(w, g, out); run<4, 6>(w, g, out); }
  for (int w : {1, 2, 4}) { run<4, 3>(w, g, out); run<4, 4>(w, g, out); run<8, 4>(w, g, out); }
  return 0;
}
