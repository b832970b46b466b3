// Accuracy of v_rcp_f64 and of 1 / 2 Newton refinements against correctly rounded 1/x (host).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

__global__ void rcp_kernel(const double *x, double *r0, double *r1, double *r2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  double r = __builtin_amdgcn_rcp(v);
  r0[i] = r;
  double e = __builtin_fma(-v, r, 1.0);
  r = __builtin_fma(r, e, r);
  r1[i] = r;
  e = __builtin_fma(-v, r, 1.0);
  r = __builtin_fma(r, e, r);
  r2[i] = r;
}

int main() {
  const int n = 1 << 22;
  std::vector<double> x(n);
  std::mt19937_64 g(7);
  std::uniform_real_distribution<double> m(1.0, 2.0);
  std::uniform_int_distribution<int> ex(-40, 40);
  for (auto &v : x) v = std::ldexp(m(g), ex(g));
  double *dx, *d0, *d1, *d2;
  hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(rcp_kernel, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
  std::vector<double> r0(n), r1(n), r2(n);
  hipMemcpy(r0.data(), d0, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(r1.data(), d1, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(r2.data(), d2, n * 8, hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0, e2 = 0;
  for (int i = 0; i < n; ++i) {
    const long double ex = 1.0L / (long double)x[i];
    e0 = std::fmax(e0, (double)fabsl(((long double)r0[i] - ex) / ex));
    e1 = std::fmax(e1, (double)fabsl(((long double)r1[i] - ex) / ex));
    e2 = std::fmax(e2, (double)fabsl(((long double)r2[i] - ex) / ex));
  }
  printf("max relative error: v_rcp_f64 %.3e (2^%.1f), +1 Newton %.3e (%.2f ulp), +2 Newton %.3e (%.2f ulp)\n", e0,
         std::log2(e0), e1, e1 / 1.11e-16, e2, e2 / 1.11e-16);
  return 0;
}
