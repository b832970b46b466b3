// Does a thin read stream mixed into the output write stream cost more than its bytes?
// Skeleton of the shared-grid GEMM's memory behaviour: persistent waves, per iteration one
// 4-drone row tile = 1408 B read (3 x 8 B per lane, prefetched one iteration ahead with an
// exact vmcnt wait) and 10.5 KB written (10 stores of 16 B per lane, two 512-B runs each).
//   RD 0: no reads   RD 1: streaming reads (HBM)   RD 2: reads of one small region (L2 hits)
#include <hip/hip_runtime.h>
#include <cstdio>

// the GPU needs ~30 ms of sustained load to reach its steady clocks: warm up well before timing
constexpr int kWarm = 60, kTimed = 30;

constexpr int kM = 10, kNC = 8;
constexpr int kOutPerDrone = kM * 4 * kNC;    // doubles
constexpr int kInPerDrone = (kM + 1) * 4;

template <int RD>
__global__ void __launch_bounds__(64) rw_kernel(const double *in, double *out, int nrt) {
  const int lane = threadIdx.x, col = lane & 15, kq = lane >> 4;
  double a[3] = {0, 0, 0};
  auto load = [&](int rt) {
    const int t = RD == 2 ? (rt & 1023) : (RD == 3 ? (rt & 32767) : rt);   // RD 3: 46 MB window (MALL-sized)
    const double *w = in + ((size_t)t * 4 + (col >> 2)) * kInPerDrone + (col & 3);
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      const int j = 4 * ks + kq;
      const double *p = w + (j < kM + 1 ? j : kM) * 4;
      asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(a[ks]) : "v"(p) : "memory");
    }
  };
  bool first = true;
  if (RD) load(blockIdx.x);
  for (int rt = blockIdx.x; rt < nrt; rt += gridDim.x) {
    if (RD) {
      if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (RD == 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      first = false;
    }
    double s = 1.0;
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) { asm volatile("" : "+v"(a[ks])); s += a[ks]; }
    __builtin_amdgcn_sched_barrier(0);
    if (RD && rt + (int)gridDim.x < nrt) load(rt + gridDim.x);
    __builtin_amdgcn_sched_barrier(0);
    const bool odd = lane & 1;
#pragma unroll
    for (int ct = 0; ct < 5; ++ct) {
      const int c = 16 * ct + (col & ~1);
      const int seg = c / kNC, kc = c - seg * kNC;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int d = rt * 4 + 2 * h + (odd ? 1 : 0);
        if (RD != 5 || s == 12345.678)
          *reinterpret_cast<double2 *>(out + (((size_t)d * kM + seg) * 4 + kq) * kNC + kc) = make_double2(s, s + ct);
      }
    }
  }
}

// Burst variant: every wave reads the inputs of its next R tiles in one go, drains, then writes the
// R tiles.  All waves start together and run at the same rate, so the chip alternates (loosely)
// between a read phase and a write phase instead of mixing the two streams.
template <int R>
__global__ void __launch_bounds__(64) burst_kernel(const double *in, double *out, int nrt) {
  const int lane = threadIdx.x, col = lane & 15, kq = lane >> 4;
  for (int rt0 = blockIdx.x; rt0 < nrt; rt0 += gridDim.x * R) {
    double a[R][3];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int rt = rt0 + r * gridDim.x;
      const int t = rt < nrt ? rt : nrt - 1;
      const double *w = in + ((size_t)t * 4 + (col >> 2)) * kInPerDrone + (col & 3);
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        const int j = 4 * ks + kq;
        const double *p = w + (j < kM + 1 ? j : kM) * 4;
        asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(a[r][ks]) : "v"(p) : "memory");
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int rt = rt0 + r * gridDim.x;
      if (rt >= nrt) break;
      double s = 1.0;
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) { asm volatile("" : "+v"(a[r][ks])); s += a[r][ks]; }
      const bool odd = lane & 1;
#pragma unroll
      for (int ct = 0; ct < 5; ++ct) {
        const int c = 16 * ct + (col & ~1);
        const int seg = c / kNC, kc = c - seg * kNC;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int d = rt * 4 + 2 * h + (odd ? 1 : 0);
          *reinterpret_cast<double2 *>(out + (((size_t)d * kM + seg) * 4 + kq) * kNC + kc) = make_double2(s, s + ct);
        }
      }
    }
  }
}

template <int R>
void run_burst(const double *in, double *out, int nrt, int grid) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < kWarm; ++i) hipLaunchKernelGGL((burst_kernel<R>), dim3(grid), dim3(64), 0, 0, in, out, nrt);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < kTimed; ++i) hipLaunchKernelGGL((burst_kernel<R>), dim3(grid), dim3(64), 0, 0, in, out, nrt);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("burst R=%2d grid %6d: %.3f ms\n", R, grid, ms / kTimed);
}

template <int RD>
void run(const double *in, double *out, int nrt, int grid) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < kWarm; ++i) hipLaunchKernelGGL((rw_kernel<RD>), dim3(grid), dim3(64), 0, 0, in, out, nrt);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < kTimed; ++i) hipLaunchKernelGGL((rw_kernel<RD>), dim3(grid), dim3(64), 0, 0, in, out, nrt);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("reads %d grid %6d: %.3f ms\n", RD, grid, ms / kTimed);
}

int main() {
  const int N = 1 << 20, nrt = N / 4;
  double *in, *out;
  hipMalloc(&in, (size_t)N * kInPerDrone * 8); hipMalloc(&out, (size_t)N * kOutPerDrone * 8);
  hipMemset(in, 0, (size_t)N * kInPerDrone * 8);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount;
  for (int w : {8, 16, 32}) {
    run<0>(in, out, nrt, ncu * w); run<1>(in, out, nrt, ncu * w); run<2>(in, out, nrt, ncu * w);
    run<3>(in, out, nrt, ncu * w); run<5>(in, out, nrt, ncu * w);
  }
  run<0>(in, out, nrt, nrt); run<1>(in, out, nrt, nrt);
  for (int w : {8, 16}) {
    run_burst<4>(in, out, nrt, ncu * w); run_burst<8>(in, out, nrt, ncu * w); run_burst<16>(in, out, nrt, ncu * w);
    run_burst<32>(in, out, nrt, ncu * w);
  }
  return 0;
}
