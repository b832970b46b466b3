// Which ingredient of the solve kernel's tile loop costs what, on top of the bare output
// stores?  Persistent 8 waves/CU, 65536 tiles of 40 KB (16 drones x 10 segments x 256 B).
//   FEAT bit 0: read the tile's 7 KB of inputs (16 B per lane loads) and push them through LDS
//   FEAT bit 1: per segment 9 ds_write_b64 + 9 ds_read_b64 (the G stash traffic)
//   FEAT bit 2: per segment 64 32-bit DPP/select-like VALU ops (the quad transposition)
//   FEAT bit 3: per segment 260 fp64 FMAs with ILP 4 (the arithmetic's issue slots)
#include <hip/hip_runtime.h>
#include <cstdio>

// the GPU needs ~30 ms of sustained load to reach its steady clocks: warm up well before timing
constexpr int kWarm = 60, kTimed = 30;

constexpr int kTileBytes = 40960;
constexpr int kInBytes = 7040;

template <int FEAT>
__global__ void __launch_bounds__(64, 2)
tile_mix(const double2 *in, double2 *out, int ntiles) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x;
  double acc0 = lane * 1.0, acc1 = 2.0, acc2 = 3.0, acc3 = 4.0;
  int iacc = lane;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    if (FEAT & 1) {
      const double2 *src = in + (size_t)tile * (kInBytes / 16);
      double2 v[7];
#pragma unroll
      for (int u = 0; u < 7; ++u) { int e = u * 64 + lane; v[u] = src[e < 440 ? e : 439]; }
#pragma unroll
      for (int u = 0; u < 7; ++u) { int e = u * 64 + lane; if (e < 440) reinterpret_cast<double2 *>(lds)[e] = v[u]; }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();
      acc0 += lds[lane * 11] + lds[lane * 11 + 5];
    }
    char *base = reinterpret_cast<char *>(out) + (size_t)tile * kTileBytes;
    for (int seg = 9; seg >= 0; --seg) {
      if (FEAT & 2) {
#pragma unroll
        for (int q = 0; q < 9; ++q) lds[1024 + (seg * 9 + q) * 16 + (lane >> 2)] = acc0 + q;
#pragma unroll
        for (int q = 0; q < 9; ++q) acc1 += lds[1024 + (seg * 9 + q) * 16 + ((lane >> 2) ^ 1)];
      }
      if (FEAT & 4) {
#pragma unroll
        for (int q = 0; q < 32; ++q) {
          int t = __builtin_amdgcn_mov_dpp(iacc, 0xB1, 0xF, 0xF, true);
          iacc = (lane & 1) ? t + q : iacc ^ t;
        }
      }
      if (FEAT & 8) {
#pragma unroll 13
        for (int w = 0; w < 65; ++w) {
          acc0 = __builtin_fma(acc0, 1.0000001, 0.5);
          acc1 = __builtin_fma(acc1, 1.0000002, 0.25);
          acc2 = __builtin_fma(acc2, 0.9999999, 0.125);
          acc3 = __builtin_fma(acc3, 0.9999998, 0.0625);
        }
      }
      const double2 v = make_double2(acc0 + acc2, acc1 + acc3 + iacc);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const size_t off = (size_t)(lane >> 2) * 2560 + seg * 256 + q * 64 + (lane & 3) * 16;
        *reinterpret_cast<double2 *>(base + off) = v;
      }
    }
  }
}

template <int FEAT>
void run(const double2 *in, double2 *out, int ntiles, int ncu) {
  hipFuncSetAttribute((const void *)tile_mix<FEAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const size_t lds = 20 * 1024;   // 8 single-wave blocks per CU
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < kWarm; ++i) hipLaunchKernelGGL((tile_mix<FEAT>), dim3(ncu * 8), dim3(64), lds, 0, in, out, ntiles);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < kTimed; ++i) hipLaunchKernelGGL((tile_mix<FEAT>), dim3(ncu * 8), dim3(64), lds, 0, in, out, ntiles);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("features %c%c%c%c (in/lds/dpp/fma): %.3f ms\n", (FEAT & 1) ? 'I' : '-', (FEAT & 2) ? 'L' : '-', (FEAT & 4) ? 'D' : '-',
         (FEAT & 8) ? 'F' : '-', ms / kTimed);
}

int main() {
  const int ntiles = 65536;
  double2 *in, *out;
  hipMalloc(&in, (size_t)ntiles * kInBytes); hipMalloc(&out, (size_t)ntiles * kTileBytes);
  hipMemset(in, 0, (size_t)ntiles * kInBytes);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount;
  run<0>(in, out, ntiles, ncu); run<1>(in, out, ntiles, ncu); run<2>(in, out, ntiles, ncu); run<4>(in, out, ntiles, ncu);
  run<8>(in, out, ntiles, ncu); run<3>(in, out, ntiles, ncu); run<7>(in, out, ntiles, ncu); run<11>(in, out, ntiles, ncu);
  run<12>(in, out, ntiles, ncu); run<15>(in, out, ntiles, ncu);
  return 0;
}
