// Does the LDS f64 atomic add (ds_add_f64) round like v_add_f64?  One add per element on random
// operands of mixed magnitude; reports how often the bit patterns differ and in which direction.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_add_rounding.hip -o lds_add_rounding
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
#pragma clang diagnostic ignored "-Wunused-value"

__global__ void k(const double* a, const double* b, double* via_lds, double* via_valu, int n) {
  __shared__ double s[256];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  s[threadIdx.x] = a[i];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  unsafeAtomicAdd(&s[threadIdx.x], b[i]);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  via_lds[i] = s[threadIdx.x];
  via_valu[i] = __dadd_rn(a[i], b[i]);
}

// chains: acc = 0; acc += x_i for i < K, through the LDS atomic and through v_add_f64
__global__ void kchain(const double* x, int K, double* via_lds, double* via_valu, int n) {
  __shared__ double s[256];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  s[threadIdx.x] = 0.0;
  double r = 0.0;
  for (int k = 0; k < K; ++k) {
    const double v = x[(size_t)k * n + i];
    unsafeAtomicAdd(&s[threadIdx.x], v);
    r = __dadd_rn(r, v);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  via_lds[i] = s[threadIdx.x];
  via_valu[i] = r;
}

int main() {
  const int n = 1 << 22;
  std::vector<double> a(n), b(n);
  uint64_t st = 88172645463325252ull;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0; };
  for (int i = 0; i < n; ++i) {
    a[i] = rnd() * std::ldexp(1.0, (int)(rnd() * 20) - 10);
    b[i] = rnd() * std::ldexp(1.0, (int)(rnd() * 20) - 10) * (i % 3 == 0 ? -1.0 : 1.0);
  }
  double *da, *db, *dl, *dv;
  hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&dl, n * 8); hipMalloc(&dv, n * 8);
  hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice);
  hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, db, dl, dv, n);
  std::vector<double> l(n), v(n);
  hipMemcpy(l.data(), dl, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(v.data(), dv, n * 8, hipMemcpyDeviceToHost);
  long diff = 0, host_eq_valu = 0, lds_smaller_mag = 0, lds_larger_mag = 0;
  for (int i = 0; i < n; ++i) {
    const double h = a[i] + b[i];
    host_eq_valu += (h == v[i]);
    if (l[i] != v[i]) {
      ++diff;
      if (std::fabs(l[i]) < std::fabs(v[i])) ++lds_smaller_mag; else ++lds_larger_mag;
    }
  }
  printf("n = %d   v_add_f64 == host IEEE add: %ld   ds_add_f64 != v_add_f64: %ld (%.2f %%)   of those |lds| < |valu|: %ld, > : %ld\n",
         n, host_eq_valu, diff, 100.0 * diff / n, lds_smaller_mag, lds_larger_mag);
  {
    const int m = 1 << 18, K = 40;
    std::vector<double> x((size_t)m * K);
    for (auto& v : x) v = rnd() * 1e-5 * rnd();
    double *dx, *l2, *v2;
    hipMalloc(&dx, x.size() * 8); hipMalloc(&l2, m * 8); hipMalloc(&v2, m * 8);
    hipMemcpy(dx, x.data(), x.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(kchain, dim3(m / 256), dim3(256), 0, 0, dx, K, l2, v2, m);
    std::vector<double> ll(m), vv(m);
    hipMemcpy(ll.data(), l2, m * 8, hipMemcpyDeviceToHost);
    hipMemcpy(vv.data(), v2, m * 8, hipMemcpyDeviceToHost);
    long d2 = 0;
    for (int i = 0; i < m; ++i) d2 += ll[i] != vv[i];
    printf("chains of %d adds: ds_add_f64 result != v_add_f64 result in %ld of %d (%.2f %%)\n", K, d2, m, 100.0 * d2 / m);
  }
  return 0;
}
