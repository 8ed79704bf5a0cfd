// Does the ORDER of the (column, value) pairs of a neighbour row matter for the LDS scatter of k_nam_step_sparse
// (diffuse.hip)?  ds_add_f64 of ACT lanes into distinct columns of the wave's 200-double accumulator row, with the
// lanes' columns arranged four ways:
//   random      the order the first step leaves today (ascending column = random banks per lane)
//   contiguous  lane l -> column (base + l) mod 200: no bank conflict at all (the floor of the instruction)
//   grouped16   the same random columns, reordered so that aligned groups of 16 lanes hold distinct (col mod 16)
//   grouped32   ... aligned groups of 32 lanes hold distinct (col mod 32)
// (greedy: most frequent class first, each member to the emptiest group that does not hold the class yet; what does
// not fit goes anywhere).  Reports clocks per wave instruction and CU at an assumed 2.3 GHz.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_scatter_pattern.hip -o lds_scatter_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

template <int VAR>
__global__ __launch_bounds__(256) void k(const int* __restrict__ cols, double* out, int iters, int act) {
  __shared__ double sm[4 * 256];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double* acc = sm + wv * 256;
  for (int q = 0; q < 4; ++q) acc[lane + 64 * q] = 0.0;
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
        if (VAR == 4) acc[c[u]] = v;
      }
      if (VAR == 4) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
  }
  __syncthreads();
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc[lane] + acc[lane + 64] + acc[lane + 128] + acc[lane + 192];
}

template <int VAR>
static void run(const int* cols, double* out, int act, const char* what, const char* pat) {
  const int iters = 2000;
  dim3 grid(256 * 4 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<VAR>), grid, dim3(256), 0, 0, cols, out, iters, act);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<VAR>), grid, dim3(256), 0, 0, cols, out, iters, act);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_cu = 8.0 * iters * grid.x * 4 / 256.0;
  printf("%-14s %-11s active lanes %2d: %7.3f ms  %6.1f clk per wave instruction and CU\n", what, pat, act, ms,
         ms * 1e-3 * 2.3e9 / instr_per_cu);
}

static void arrange(int* p, int act, int G, int M) {        // groups of G lanes, classes col mod M
  const int ng = (act + G - 1) / G;
  std::vector<std::vector<int>> byc(M);
  for (int i = 0; i < act; ++i) byc[p[i] % M].push_back(p[i]);
  std::vector<int> order(M);
  for (int i = 0; i < M; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](int a, int b) { return byc[a].size() > byc[b].size(); });
  std::vector<std::vector<int>> grp(ng);
  std::vector<std::vector<bool>> has(ng, std::vector<bool>(M, false));
  std::vector<int> cap(ng);
  for (int g = 0; g < ng; ++g) cap[g] = std::min(G, act - g * G);
  std::vector<int> left;
  for (int cls : order)
    for (int col : byc[cls]) {
      int best = -1;
      for (int g = 0; g < ng; ++g)
        if (!has[g][cls] && (int)grp[g].size() < cap[g] && (best < 0 || grp[g].size() < grp[best].size())) best = g;
      if (best < 0) { left.push_back(col); continue; }
      grp[best].push_back(col); has[best][cls] = true;
    }
  for (int col : left)
    for (int g = 0; g < ng; ++g)
      if ((int)grp[g].size() < cap[g]) { grp[g].push_back(col); break; }
  int o = 0;
  for (int g = 0; g < ng; ++g) for (int col : grp[g]) p[o++] = col;
}

int main() {
  int* h = (int*)malloc(1024 * 512 * 4);
  int* cols; double* out;
  (void)hipMalloc(&cols, 1024 * 512 * 4);
  (void)hipMalloc(&out, (size_t)256 * 16 * 256 * 8);
  const char* names[4] = {"random", "contiguous", "grouped16", "grouped32"};
  for (int act : {35, 64}) {
    for (int pat = 0; pat < 4; ++pat) {
      uint64_t s = 88172645463325252ull;
      for (int w = 0; w < 1024 * 8; ++w) {
        int p[256];
        for (int i = 0; i < 256; ++i) p[i] = i;
        for (int i = 0; i < 64; ++i) {
          s ^= s << 13; s ^= s >> 7; s ^= s << 17;
          const int j = i + (int)(s % (uint64_t)(200 - i));
          const int t = p[i]; p[i] = p[j]; p[j] = t;
        }
        if (pat == 0) std::sort(p, p + act);                     // ascending column, as first_tail writes them
        if (pat == 1) { const int b = p[0]; for (int i = 0; i < 64; ++i) p[i] = (b + i) % 200; }
        if (pat == 2) arrange(p, act, 16, 16);
        if (pat == 3) arrange(p, act, 32, 32);
        for (int i = 0; i < 64; ++i) h[w * 64 + i] = p[i];
      }
      (void)hipMemcpy(cols, h, 1024 * 512 * 4, hipMemcpyHostToDevice);
      run<0>(cols, out, act, "ds_add_f64", names[pat]);
      run<4>(cols, out, act, "ds_write_b64", names[pat]);
    }
  }
  return 0;
}
