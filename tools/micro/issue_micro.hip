// How fast can ONE wavefront issue fp64 FMAs?  (background for the small-batch latency analysis)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/issue_micro.hip -o tools/micro/issue_micro
#include <hip/hip_runtime.h>
#include <cstdio>

template <int ILP>
__global__ void __launch_bounds__(64) fma_chain(double *out, int iters, long long *cycles) {
  double a[ILP];
#pragma unroll
  for (int k = 0; k < ILP; ++k) a[k] = threadIdx.x * 1e-3 + k;
  const double m = 1.0000001, c = 0.5;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int k = 0; k < ILP; ++k) a[k] = __builtin_fma(a[k], m, c);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
#pragma unroll
  for (int k = 0; k < ILP; ++k) s += a[k];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int ILP>
void run(int waves_per_block_unused) {
  double *out; long long *cyc;
  hipMalloc(&out, 64 * 8 * 1024); hipMalloc(&cyc, 8 * 1024);
  const int iters = 2000;
  hipLaunchKernelGGL((fma_chain<ILP>), dim3(1), dim3(64), 0, 0, out, iters, cyc);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((fma_chain<ILP>), dim3(1), dim3(64), 0, 0, out, iters, cyc);
  hipDeviceSynchronize();
  long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("ILP %d: %.2f cycles per fp64 FMA wave-instruction (one wave on the chip)\n", ILP, (double)h / (iters * 16.0 * ILP));
  hipFree(out); hipFree(cyc);
}

int main() {
  run<1>(0); run<2>(0); run<4>(0); run<8>(0);
  return 0;
}
