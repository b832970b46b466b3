// Store-path microbenchmark behind DESIGN.md's K1 analysis: how fast can a CU push the
// solve kernel's output pattern, as a function of waves per CU and of the access shape?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/store_micro.hip -o /tmp/store_micro && /tmp/store_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int kTileBytes = 40960;   // 16 drones x 10 segments x 256 B

// mode 0: tile written as 40 consecutive 1 KB instructions
// mode 1: K1 pattern: per segment (descending) 4 instructions, each 16 quads x 64 B at 2560 B stride
// mode 2: K1 LDS-transposed pattern: per segment 4 instructions, each 4 drones x 256 B
template <int MODE, int WORK>
__global__ void __launch_bounds__(64)
tile_store(double2 *out, int ntiles) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x;
  double acc = lane * 1.0;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    char *base = reinterpret_cast<char *>(out) + (size_t)tile * kTileBytes;
    for (int seg = 9; seg >= 0; --seg) {
      if (WORK > 0) {
#pragma unroll 8
        for (int w = 0; w < WORK; ++w) acc = __builtin_fma(acc, 1.0000001, 0.5);
      }
      const double2 v = make_double2(acc, (double)seg);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        size_t off;
        if (MODE == 0) off = (size_t)((9 - seg) * 4 + q) * 1024 + lane * 16;
        else if (MODE == 1) off = (size_t)(lane >> 2) * 2560 + seg * 256 + q * 64 + (lane & 3) * 16;
        else off = (size_t)(4 * q + (lane >> 4)) * 2560 + seg * 256 + (lane & 15) * 16;
        *reinterpret_cast<double2 *>(base + off) = v;
      }
    }
  }
  if (acc == 12345.0) lds[0] = acc;
}

__global__ void __launch_bounds__(256) fill16(double2 *out, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
    out[i] = make_double2(1.0, 2.0);
}

template <typename F>
float time_ms(F launch, int reps = 30) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 60; ++i) launch();   // ~30 ms of sustained load: the GPU's clocks have settled
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main() {
  const int ntiles = 65536;
  const size_t bytes = (size_t)ntiles * kTileBytes;
  double2 *out;
  CK(hipMalloc(&out, bytes));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int ncu = p.multiProcessorCount;
  printf("bytes %.3f GB, %d CUs\n", bytes / 1e9, ncu);
  float ms = time_ms([&] { hipLaunchKernelGGL(fill16, dim3(ncu * 8), dim3(256), 0, 0, out, bytes / 16); });
  printf("fill16 (8 x 256-thread blocks per CU): %.3f ms  %.2f TB/s\n", ms, bytes / ms / 1e9);
  const int waves[] = {4, 8, 16, 32};
  for (int w : waves) {
    const size_t lds = (160 * 1024 / w) & ~1023u;     // LDS request that admits exactly w single-wave blocks per CU
    CK(hipFuncSetAttribute((const void *)tile_store<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)tile_store<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)tile_store<2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)tile_store<1, 250>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)tile_store<1, 1000>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    float m0 = time_ms([&] { hipLaunchKernelGGL((tile_store<0, 0>), dim3(ncu * w), dim3(64), lds, 0, out, ntiles); });
    float m1 = time_ms([&] { hipLaunchKernelGGL((tile_store<1, 0>), dim3(ncu * w), dim3(64), lds, 0, out, ntiles); });
    float m2 = time_ms([&] { hipLaunchKernelGGL((tile_store<2, 0>), dim3(ncu * w), dim3(64), lds, 0, out, ntiles); });
    float m3 = time_ms([&] { hipLaunchKernelGGL((tile_store<1, 250>), dim3(ncu * w), dim3(64), lds, 0, out, ntiles); });
    float m4 = time_ms([&] { hipLaunchKernelGGL((tile_store<1, 1000>), dim3(ncu * w), dim3(64), lds, 0, out, ntiles); });
    printf("%2d waves/CU: seq-1KB %.3f ms | quad-64B %.3f ms | 4x256B %.3f ms | quad-64B + 250 FMA/seg %.3f ms | + 1000 FMA/seg %.3f ms\n",
           w, m0, m1, m2, m3, m4);
  }
  hipFree(out);
  return 0;
}
