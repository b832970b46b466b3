// Issue rate of the fp64 vector instructions the pairwise pass is made of, with the chip full
// (4 waves per SIMD, 8 independent chains per lane): wave-instructions per SIMD cycle.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/f64_rate_micro.hip -o tools/micro/f64_rate_micro
#include <hip/hip_runtime.h>
#include <cstdio>

enum Op { ADD, MUL, FMA, MIN, ADD_SGPR, MIX7 };

template <int OP>
__global__ void __launch_bounds__(256, 4) rate_kernel(double *out, int iters, double sv) {
  constexpr int ILP = 8;
  double a[ILP];
#pragma unroll
  for (int k = 0; k < ILP; ++k) a[k] = threadIdx.x * 1e-3 + k + 1.0;
  const double m = 1.0000001, c = 0.5;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int k = 0; k < ILP; ++k) {
        if (OP == ADD) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[k]) : "v"(c));
        if (OP == MUL) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[k]) : "v"(m));
        if (OP == FMA) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
        if (OP == MIN) asm volatile("v_min_f64 %0, %0, %1" : "+v"(a[k]) : "v"(c));
        if (OP == ADD_SGPR) asm volatile("v_add_f64 %0, %1, -%0" : "+v"(a[k]) : "s"(sv));
        if (OP == MIX7) {   // the pass's mix per pair and sample: 3 differences, 1 product, 2 FMAs, 1 minimum
          double dx, dy, dz, p;
          asm volatile("v_add_f64 %0, %1, -%2" : "=v"(dx) : "s"(sv), "v"(a[k]));
          asm volatile("v_add_f64 %0, %1, -%2" : "=v"(dy) : "s"(sv), "v"(m));
          asm volatile("v_add_f64 %0, %1, -%2" : "=v"(dz) : "s"(sv), "v"(c));
          asm volatile("v_mul_f64 %0, %1, %1" : "=v"(p) : "v"(dx));
          asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(p) : "v"(dy));
          asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(p) : "v"(dz));
          asm volatile("v_min_f64 %0, %1, %0" : "+v"(a[k]) : "v"(p));
        }
      }
  }
  double s = 0;
#pragma unroll
  for (int k = 0; k < ILP; ++k) s += a[k];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
void run(const char *name, int per_iter) {
  int dev = 0, cus = 0, khz = 0;
  hipGetDevice(&dev);
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, dev);
  const int blocks = cus * 4;   // 4 workgroups of 4 waves per CU: 4 waves per SIMD
  double *out;
  hipMalloc(&out, (size_t)blocks * 256 * 8);
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((rate_kernel<OP>), dim3(blocks), dim3(256), 0, 0, out, iters, 0.25);
  hipEventRecord(e0);
  hipLaunchKernelGGL((rate_kernel<OP>), dim3(blocks), dim3(256), 0, 0, out, iters, 0.25);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double insts_per_simd = 4.0 * iters * 8.0 * 8.0 * per_iter;      // 4 waves per SIMD
  const double cycles = ms * 1e-3 * khz * 1e3;
  printf("%-28s %.2f cycles per wave-instruction per SIMD at the nominal %d MHz (%.3f ms)\n", name,
         cycles / insts_per_simd, khz / 1000, ms);
  hipFree(out);
}

int main() {
  run<ADD>("v_add_f64", 1);
  run<MUL>("v_mul_f64", 1);
  run<FMA>("v_fma_f64", 1);
  run<MIN>("v_min_f64", 1);
  run<ADD_SGPR>("v_add_f64 sgpr, -vgpr", 1);
  run<MIX7>("pairwise mix (7 instr)", 7);
  return 0;
}
