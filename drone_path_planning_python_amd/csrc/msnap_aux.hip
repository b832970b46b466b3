// Kernels either side of the solve: the float32 pol-matrix pack (a7), the
// formation transform (a8), the piecewise sampler (a5) and the two collision
// passes (new capability).  gfx950, wave64.
#include <math.h>

#include <cstdlib>

#include "msnap_internal.h"

namespace msnap {

// ------------------------------------------------------------------------------------
// a7: matrix[M][1 + 4*ncoef] float32 = [T | x | y | z | yaw]
// (reference scripts/drones_pols_generator.py:63-77)
// ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
pack_kernel(const double *__restrict__ coef, const double *__restrict__ dur, float *__restrict__ out,
            size_t n_rows /* N*M */, int nc) {
  const int ncol = 1 + 4 * nc;
  const size_t total = n_rows * (size_t)ncol;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const size_t row = idx / ncol;
    const int col = (int)(idx - row * ncol);
    const double v = (col == 0) ? dur[row] : coef[row * (size_t)(4 * nc) + (col - 1)];
    out[idx] = (float)v;
  }
}

int launch_pack(msnap_ctx *ctx, int n_drones, int n_seg, const double *coef, const double *dur, float *out) {
  const size_t rows = (size_t)n_drones * n_seg;
  const int nc = ctx->order + 1;
  const size_t total = rows * (1 + 4 * nc);
  size_t blocks = (total + 255) / 256;
  if (blocks > (size_t)ctx->n_cu * 8) blocks = (size_t)ctx->n_cu * 8;
  hipLaunchKernelGGL(pack_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, coef, dur, out, rows, nc);
  MSNAP_HIP(ctx, hipGetLastError());
  return MSNAP_OK;
}

// ------------------------------------------------------------------------------------
// a8: p' = R(q_rb) p_k + t_rb ;  q' = quaternion of R(q_rb)
// (reference scripts/drones_traj_generator.py:67-82 through tf2_geometry_msgs
// do_transform_pose -> PyKDL Frame product; KDL is not vendored in the
// reference: Rotation::Quaternion / Rotation::GetQuaternion restated from the
// published orocos_kdl frames.cpp.  The drone poses carry identity orientation,
// drones_traj_generator.py:31,38.)
// ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
formation_kernel(const double *__restrict__ rb, const double *__restrict__ off, double *__restrict__ out,
                 int P, int Kn) {
#pragma clang fp contract(off)
  // A wave's 64 (offset, pose) items own 448 CONSECUTIVE output doubles; written per lane they would be
  // seven 8-B stores at a 56-B stride.  Through a per-wave LDS image (stride 7 is odd: conflict-free both
  // ways) they leave as seven stores of 512 contiguous bytes.
  __shared__ double image[4][kWave * 7];
  const int lane = threadIdx.x & (kWave - 1);
  double *img = image[threadIdx.x / kWave];
  const size_t total = (size_t)P * Kn;
  const size_t span = (size_t)gridDim.x * blockDim.x;
  const size_t trips = (total + span - 1) / span;      // the same for every wave: no partly exited wave below
  for (size_t trip = 0; trip < trips; ++trip) {
    const size_t wave_base = trip * span + (size_t)blockIdx.x * blockDim.x + (threadIdx.x - lane);
    const size_t idx = wave_base + lane;
    if (idx < total) {
    const int k = (int)(idx / P);
    const int p = (int)(idx - (size_t)k * P);
    const double *r = rb + (size_t)p * 7;
    const double tx = r[0], ty = r[1], tz = r[2];
    const double x = r[3], y = r[4], z = r[5], w = r[6];
    const double x2 = x * x, y2 = y * y, z2 = z * z, w2 = w * w;
    const double m00 = w2 + x2 - y2 - z2, m01 = 2 * x * y - 2 * w * z, m02 = 2 * x * z + 2 * w * y;
    const double m10 = 2 * x * y + 2 * w * z, m11 = w2 - x2 + y2 - z2, m12 = 2 * y * z - 2 * w * x;
    const double m20 = 2 * x * z - 2 * w * y, m21 = 2 * y * z + 2 * w * x, m22 = w2 - x2 - y2 + z2;
    const double ox = off[k * 3 + 0], oy = off[k * 3 + 1], oz = off[k * 3 + 2];
    double *o = img + lane * 7;
    o[0] = m00 * ox + m01 * oy + m02 * oz + tx;
    o[1] = m10 * ox + m11 * oy + m12 * oz + ty;
    o[2] = m20 * ox + m21 * oy + m22 * oz + tz;
    // Rotation::GetQuaternion
    const double trace = m00 + m11 + m22;
    double qx, qy, qz, qw;
    if (trace > 1e-12) {
      const double s = 0.5 / sqrt(trace + 1.0);
      qw = 0.25 / s;
      qx = (m21 - m12) * s;
      qy = (m02 - m20) * s;
      qz = (m10 - m01) * s;
    } else if (m00 > m11 && m00 > m22) {
      const double s = 2.0 * sqrt(1.0 + m00 - m11 - m22);
      qw = (m21 - m12) / s;
      qx = 0.25 * s;
      qy = (m01 + m10) / s;
      qz = (m02 + m20) / s;
    } else if (m11 > m22) {
      const double s = 2.0 * sqrt(1.0 + m11 - m00 - m22);
      qw = (m02 - m20) / s;
      qx = (m01 + m10) / s;
      qy = 0.25 * s;
      qz = (m12 + m21) / s;
    } else {
      const double s = 2.0 * sqrt(1.0 + m22 - m00 - m11);
      qw = (m10 - m01) / s;
      qx = (m02 + m20) / s;
      qy = (m12 + m21) / s;
      qz = 0.25 * s;
    }
    o[3] = qx;
    o[4] = qy;
    o[5] = qz;
    o[6] = qw;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    const size_t ebase = wave_base * 7, nelem = total * 7;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int e = j * kWave + lane;
      if (ebase + e < nelem) out[ebase + e] = img[e];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
  }
}

int launch_formation_transform(msnap_ctx *ctx, int n_poses, int n_offsets, const double *rb_pose,
                               const double *offsets, double *out) {
  const size_t total = (size_t)n_poses * n_offsets;
  size_t blocks = (total + 255) / 256;
  if (blocks > (size_t)ctx->n_cu * 8) blocks = (size_t)ctx->n_cu * 8;
  hipLaunchKernelGGL(formation_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, rb_pose, offsets,
                     out, n_poses, n_offsets);
  MSNAP_HIP(ctx, hipGetLastError());
  return MSNAP_OK;
}

// ------------------------------------------------------------------------------------
// a5: PiecewisePolynomial.eval on the grid t = s*dt
// (reference src/optimizations/uav_trajectory.py:154-169: strict '<' lookup,
// running sum of durations, last piece extrapolates; Horner of :17-22 with
// separate multiply and add, hence fp contract off for bit parity)
// ------------------------------------------------------------------------------------
constexpr int kRowBlockRows = 128;   // rows per row block of the pairwise pass (kRowBlock below): pitch granularity of its row image
__global__ void collide_transpose_kernel(const double *__restrict__ prow, int R, int Rp, int E, double *__restrict__ prow_t,
                                         int ny, int32_t *__restrict__ fill, size_t fill_n, const int32_t *__restrict__ perm,
                                         double *__restrict__ psorted);

// The sort key of the pairwise pass's broad phase (CollideCull below): Morton code of the centre of a drone's path box
// on a 1 m x 1 m x 4 m lattice (paths are metres; 11 + 11 + 10 bits around the origin, clamped beyond +-1 km: a swarm
// inside one cell, or far out, sorts arbitrarily and less is culled -- the result does not depend on the order).
// A drone without a finite sample (lo > hi) gets the largest key and sorts to the end.
__device__ __forceinline__ unsigned long long spread3(unsigned long long v) {      // 21 bits -> every third bit
  v &= 0x1fffffull;
  v = (v | (v << 32)) & 0x1f00000000ffffull;
  v = (v | (v << 16)) & 0x1f0000ff0000ffull;
  v = (v | (v << 8)) & 0x100f00f00f00f00full;
  v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
}
__device__ __forceinline__ unsigned drone_sort_key(const double (&lo)[3], const double (&hi)[3]) {
  unsigned long long kk = 0xffffffffull;
  if (lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2]) {
    const double cell[3] = {1.0, 1.0, 4.0}, half[3] = {1024.0, 1024.0, 512.0};
    unsigned long long q[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      double c = floor(0.5 * (lo[k] + hi[k]) / cell[k]) + half[k];
      c = c < 0.0 ? 0.0 : (c > 2.0 * half[k] - 1.0 ? 2.0 * half[k] - 1.0 : c);
      q[k] = (unsigned long long)c;
    }
    kk = spread3(q[0]) | (spread3(q[1]) << 1) | (spread3(q[2]) << 2);      // < 2^32 (z has 10 bits)
  }
  return (unsigned)kk;
}

__device__ __forceinline__ double shfl_xor_f64(double v, int mask);

// Generic form: one thread per (drone, sample), the reference's search loop as it stands.  Used for
// drones whose durations are not all >= 0 (the search is then not a partition into ranges), for
// dt == 0 and for paths whose samples do not fit the LDS image of the fast kernel.
template <int NC>
__device__ __forceinline__ void sample_point_generic(const double *__restrict__ coef, const double *__restrict__ dr,
                                                     int M, double t, int a, double &x, int seg_stride = 4 * NC,
                                                     int axis_stride = NC) {
#pragma clang fp contract(off)
  double acc = 0.0;
  int seg = M - 1;
  bool found = false;
  for (int i = 0; i < M; ++i) {
    const double Ti = dr[i];
    if (!found && t < acc + Ti) {
      seg = i;
      found = true;
    }
    if (!found && i < M - 1) acc = acc + Ti;
  }
  // not found: acc == sum(dur[:-1]) and seg == M-1 (uav_trajectory.py:161-163)
  const double tl = t - acc;
  const double *c = coef + (size_t)seg * seg_stride + (size_t)a * axis_stride;
  x = 0.0;
#pragma unroll
  for (int q = NC - 1; q >= 0; --q) x = x * tl + c[q];
}

template <int NC>
__global__ void __launch_bounds__(256)
sample_generic_kernel(const double *__restrict__ coef, const double *__restrict__ dur, double dt, int N, int M, int S,
                      int naxes, double *__restrict__ pos) {
  const size_t total = (size_t)N * S;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int d = (int)(idx / S);
    const int s = (int)(idx - (size_t)d * S);
    for (int a = 0; a < naxes; ++a) {
      double x;
      sample_point_generic<NC>(coef + (size_t)d * M * 4 * NC, dur + (size_t)d * M, M, (double)s * dt, a, x);
      pos[idx * naxes + a] = x;
    }
  }
}

// Fast form: one thread per (drone, piece, axis).  The reference's search `t < acc + T_i` over the
// running sum acc (uav_trajectory.py:157-165) partitions the sample times into one range per piece when
// every duration is >= 0: piece i owns the samples with b_i <= s*dt < b_{i+1}, b the running sum exactly
// as the reference accumulates it (the last piece also owns everything from b_M on and keeps its origin
// b_{M-1}: the extrapolation of :161-163).  The range's first sample is found from b_i / dt and then
// corrected against fl(s*dt) >= b_i itself, so the partition is the reference's bit for bit.  The thread
// keeps its coefficient row in registers over its samples (the (drone, sample) form re-reads a row per
// sample and gates the coefficient address behind M dependent duration loads: 34 % of the HBM rate).
// A workgroup owns DW whole drones; results go through an LDS image of their [S][naxes] blocks and
// leave as 16-byte-per-lane runs (direct 8-byte stores at a 24-byte stride reach L2 as 21-byte requests).
// the samples of one (drone, piece, axis): s_lo .. s_hi - 1 from the running sums, Horner on the piece's row (`load`
// fills it)
template <int NC, class Load>
__device__ __forceinline__ void sample_piece(double bi, double bn, int i, int M, int S, double dt, double *img,
                                             int naxes, int a, Load load) {
#pragma clang fp contract(off)
  // first sample with fl(s*dt) >= b: the quotient is a guess, the products decide
  auto first_at = [&](double b) -> int {
    const double x = b / dt;
    int c = x >= (double)S ? S : (int)x;
    while (c > 0 && (double)(c - 1) * dt >= b) --c;
    while (c < S && (double)c * dt < b) ++c;
    return c;
  };
  // (the row is fetched before the ranges are worked out -- two fp64 divisions and their corrections -- so that its
  // latency runs under them; a piece without samples fetches it for nothing)
  double c[NC];
  load(c);
  const int s_lo = (i == 0) ? 0 : first_at(bi);
  const int s_hi = (i == M - 1) ? S : first_at(bn);
  for (int sq = s_lo; sq < s_hi; ++sq) {
    const double tl = (double)sq * dt - bi;
    double x = 0.0;
#pragma unroll
    for (int q = NC - 1; q >= 0; --q) x = x * tl + c[q];
    img[(size_t)sq * naxes + a] = x;
  }
}

// the finished LDS image [nd][S][naxes] of a workgroup's drones d0 .. d0 + nd - 1 -> pos, and the second output
__device__ __forceinline__ void sample_image_out(const double *sImg, int tid, int nthreads, int nd, int d0,
                                                 size_t per_drone, int S, double *__restrict__ pos,
                                                 double *__restrict__ pos_t, int Rp, double *__restrict__ kbox,
                                                 unsigned *__restrict__ kkey) {
  // the nd drones' blocks are contiguous in pos: 16 bytes per lane (per_drone * nd doubles; odd tail by one lane)
  const size_t words = per_drone * nd;
  double *out = pos + (size_t)d0 * per_drone;
  const bool aligned = ((size_t)d0 * per_drone & 1) == 0;
  if (aligned) {
    for (size_t e = (size_t)tid * 2; e + 1 < words; e += (size_t)nthreads * 2)
      *reinterpret_cast<double2 *>(out + e) = *reinterpret_cast<const double2 *>(sImg + e);
    if ((words & 1) && tid == 0) out[words - 1] = sImg[words - 1];
  } else {
    for (size_t e = tid; e < words; e += nthreads) out[e] = sImg[e];
  }
  // second output for the pairwise pass (msnap_sample_collide): the same samples as the transposed row image
  // [sample][xyz][row] (row pitch Rp) that collide_span_kernel reads -- the workgroup's DW drones are DW
  // consecutive rows, so every (sample, axis) is one run of DW doubles -- instead of a transposition pass
  // over the finished positions
  if (pos_t != nullptr) {
    const int runs = (int)per_drone;              // (sample, axis) pairs; naxes == 3 (checked by the launcher)
    for (int e = tid; e < runs * nd; e += nthreads) {
      const int sk = e / nd, dl = e - sk * nd;
      pos_t[(size_t)sk * Rp + d0 + dl] = sImg[(size_t)dl * per_drone + sk];
    }
  }
  // third output, for a whole-swarm pass behind the exact broad phase: what collide_key_kernel would compute from
  // the finished positions -- the box of the drone's finite samples and its sort key -- while the samples sit in
  // the image: 16 threads per drone (DW <= 16, naxes == 3), folded inside their 16 lanes
  if (kbox != nullptr) {
    const int dl = tid >> 4, part = tid & 15;
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (dl < nd) {
      const double *img = sImg + (size_t)dl * per_drone;
      for (int sq = part; sq < S; sq += 16)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const double v = img[(size_t)sq * 3 + k];
          if (__builtin_isfinite(v)) {
            lo[k] = v < lo[k] ? v : lo[k];
            hi[k] = v > hi[k] ? v : hi[k];
          }
        }
    }
#pragma unroll
    for (int m = 1; m < 16; m <<= 1)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double ol = shfl_xor_f64(lo[k], m), oh = shfl_xor_f64(hi[k], m);
        lo[k] = ol < lo[k] ? ol : lo[k];
        hi[k] = oh > hi[k] ? oh : hi[k];
      }
    if (dl < nd && part == 0) {
      double *b = kbox + (size_t)(d0 + dl) * 6;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        b[k] = lo[k];
        b[3 + k] = hi[k];
      }
      kkey[d0 + dl] = drone_sort_key(lo, hi);
    }
  }
}

// x / d for 32-bit x through one v_mul_hi_u32: magic = floor((2^32 - 1) / d) + 1 is exact while x * d < 2^32 (items and
// their divisors here are a few thousand at most); d == 1 has no 32-bit magic and is passed as 0
__device__ __forceinline__ unsigned div_magic(unsigned d) { return d > 1 ? 0xffffffffu / d + 1u : 0u; }
__device__ __forceinline__ int div_by(int x, unsigned magic) { return magic ? (int)__umulhi((unsigned)x, magic) : x; }

template <int NC>
__global__ void __launch_bounds__(256)
sample_kernel(const double *__restrict__ coef, const double *__restrict__ dur, double dt, int N, int M, int S,
              int naxes, int DW, double *__restrict__ pos, double *__restrict__ pos_t, int Rp, double *__restrict__ kbox,
              unsigned *__restrict__ kkey) {
#pragma clang fp contract(off)
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double *sB = smem;                                    // [DW][M + 1] running sums, b_0 = 0
  int *sGen = reinterpret_cast<int *>(sB + (size_t)DW * (M + 1));   // [DW] 1: this drone takes the generic search
  double *sImg = reinterpret_cast<double *>(sGen + ((DW + 1) & ~1));  // [DW][S][naxes]
  const int tid = threadIdx.x;
  const size_t per_drone = (size_t)S * naxes;
  const unsigned inv_piece = div_magic((unsigned)(naxes * M)), inv_axes = div_magic((unsigned)naxes);
  for (int d0 = blockIdx.x * DW; d0 < N; d0 += gridDim.x * DW) {
    const int nd = N - d0 < DW ? N - d0 : DW;
    if (tid < nd) {
      const double *dr = dur + (size_t)(d0 + tid) * M;
      double acc = 0.0;
      bool ranges = dt > 0.0;
      sB[tid * (M + 1)] = 0.0;
      for (int i = 0; i < M; ++i) {
        const double Ti = dr[i];
        ranges = ranges && (Ti >= 0.0);
        acc = acc + Ti;
        sB[tid * (M + 1) + i + 1] = acc;
      }
      sGen[tid] = ranges ? 0 : 1;
    }
    __syncthreads();
    const int items = nd * M * naxes;
    for (int it = tid; it < items; it += blockDim.x) {
      const int dl = div_by(it, inv_piece), rem = it - dl * (naxes * M);
      const int i = div_by(rem, inv_axes), a = rem - i * naxes;
      const double *cbase = coef + (size_t)(d0 + dl) * M * 4 * NC;
      double *img = sImg + (size_t)dl * per_drone;
      if (sGen[dl]) {
        if (i == 0)       // rare: one thread per axis walks the whole path with the reference's own loop
          for (int sq = 0; sq < S; ++sq)
            sample_point_generic<NC>(cbase, dur + (size_t)(d0 + dl) * M, M, (double)sq * dt, a, img[(size_t)sq * naxes + a]);
        continue;
      }
      const double *crow = cbase + ((size_t)i * 4 + a) * NC;
      sample_piece<NC>(sB[dl * (M + 1) + i], sB[dl * (M + 1) + i + 1], i, M, S, dt, img, naxes, a, [&](double (&c)[NC]) {
#pragma unroll
        for (int q = 0; q < NC; q += 2) {
          const double2 v = *reinterpret_cast<const double2 *>(crow + q);
          c[q] = v.x;
          c[q + 1] = v.y;
        }
      });
    }
    __syncthreads();
    sample_image_out(sImg, tid, blockDim.x, nd, d0, per_drone, S, pos, pos_t, Rp, kbox, kkey);
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------
// Shared-grid solve and sampler in one launch (msnap_solve_grid_sample_device): a workgroup owns the sampler's DW
// drones, builds their coefficients as grid_gemm_kernel does -- the same fp64 MFMA chain over the same packed operator
// fragments, so the coefficients are its bit for bit -- into LDS, writes them out, and samples from LDS.  What it
// saves is the dependent launch and the coefficient read-back between the two kernels (4.6 us of a 15 us pair at
// 4096 drones x 10 segments); the waypoints [DW][M+1][4] are staged transposed through LDS as the A operand.
//   tasks = (row tile of 4 drones) x (column tile of 16 coefficients), wave w takes tasks w, w + 4, ...:
//   at most kFuseTasks per wave and kFuseKS k steps (M <= 11), or the launcher runs the two kernels.  Longer paths
//   gain nothing: the product grows with M^2 while the saved launch does not (4096 drones: 10 segments 17.2 -> 13.1 us,
//   15 segments 22.3 -> 24.7, 20 segments 20.1 -> 24.1 with six k steps in registers: 204 VGPRs, two workgroups per CU)
// ------------------------------------------------------------------------------------
#ifdef MSNAP_TOOLS_TIMELINE
// phase timestamps (s_memrealtime, 100 MHz) of the fused kernel: tools/grid_sample_timeline.py
__device__ unsigned long long g_gs_timeline[1024 * 8];
// (stamps go to LDS and leave at the end: a global store per stamp would sit in every later s_waitcnt vmcnt(0))
#define MSNAP_GSTL(k)                                              \
  do {                                                             \
    if (threadIdx.x == 0) s_gs_tl[(k)] = wall_clock64();           \
  } while (0)
#define MSNAP_GSTL_FLUSH()                                                                   \
  do {                                                                                       \
    lds_barrier();                                                                           \
    if (threadIdx.x < 8 && blockIdx.x < 1024) g_gs_timeline[blockIdx.x * 8 + threadIdx.x] = s_gs_tl[threadIdx.x]; \
  } while (0)
}  // namespace msnap
extern "C" int msnap_debug_read_gs_timeline(unsigned long long *out, int n_words) {
  if (hipDeviceSynchronize() != hipSuccess) return MSNAP_EHIP;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(msnap::g_gs_timeline), (size_t)n_words * 8) == hipSuccess ? MSNAP_OK : MSNAP_EHIP;
}
namespace msnap {
#else
#define MSNAP_GSTL(k) do { } while (0)
#define MSNAP_GSTL_FLUSH() do { } while (0)
#endif

typedef double v4f64s __attribute__((ext_vector_type(4)));
constexpr int kFuseKS = 3;
constexpr int kFuseTasks = 4;

template <int NC>
__global__ void __launch_bounds__(256, 2)
grid_sample_kernel(const double *__restrict__ wp, const double *__restrict__ frag, int ks_pitch,
                   const double *__restrict__ gdur, const int32_t *__restrict__ gstatus, double dt, int N, int M, int S,
                   int DW, double *__restrict__ coef, double *__restrict__ dur, int32_t *__restrict__ status,
                   double *__restrict__ pos, double *__restrict__ pos_t, int Rp, double *__restrict__ kbox,
                   unsigned *__restrict__ kkey) {
#pragma clang fp contract(off)
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int m = M + 1, ncols = M * NC, nct = (ncols + 15) >> 4, nks = (m + 3) >> 2;
  const int nrt = (DW * 4 + 15) >> 4, rows16 = nrt * 16, ntasks = nrt * nct;
  const int wpitch = m | 1, cpitch = ncols + 2;
  // t / nrt for t < 16, nrt <= 4, and e / (4 m) for e < 512, as multiplications (a dozen integer divisions by kernel
  // arguments were 490 scalar instructions in front of the first load)
  const int nrt_inv = nrt == 3 ? 86 : 256 >> (nrt >> 1);
  const unsigned m4_inv = (1u << 20) / (unsigned)(m * 4) + 1u;
  const unsigned inv_piece = div_magic((unsigned)(3 * M));
  double *sB = smem;                                        // [M + 1] running sums of the grid's durations
  double *sW = sB + ((m + 1) & ~1);                         // [rows16][wpitch] waypoints, row = 4 * drone + axis
  double *sC = sW + (((size_t)rows16 * wpitch + 1) & ~(size_t)1);   // [DW * 4][cpitch] coefficients
  double *sImg = sC + (size_t)DW * 4 * cpitch;              // [DW][S][3]
  int *sBad = reinterpret_cast<int *>(sImg + (size_t)DW * S * 3);   // [DW] non-finite waypoints; [DW]: generic search
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (uniform: the task tests become scalar branches)
  const int col = lane & 15, kq = lane >> 4;
  const int grid_st = gstatus[0];
  const size_t per_drone = (size_t)S * 3;
#ifdef MSNAP_TOOLS_TIMELINE
  __shared__ unsigned long long s_gs_tl[8];
#endif
  MSNAP_GSTL(0);

  if (tid == 0) {
    double acc = 0.0;
    bool ranges = dt > 0.0;
    sB[0] = 0.0;
    for (int i = 0; i < M; ++i) {
      const double Ti = gdur[i];
      ranges = ranges && (Ti >= 0.0);
      acc = acc + Ti;
      sB[i + 1] = acc;
    }
    sBad[DW] = ranges ? 0 : 1;
  }
  for (int e = tid; e < rows16 * wpitch; e += blockDim.x) sW[e] = 0.0;

  for (int d0 = blockIdx.x * DW; d0 < N; d0 += gridDim.x * DW) {
    const int nd = N - d0 < DW ? N - d0 : DW;
    const int words = nd * m * 4;                       // <= 512: two per thread
    const double *wsrc = wp + (size_t)d0 * m * 4;
    const int e0 = tid, e1 = tid + 256;
    const double w0 = wsrc[e0 < words ? e0 : 0], w1 = wsrc[e1 < words ? e1 : 0];
    // the operator fragments of this wave's tasks, in the same flight (loaded per pass over d0, so that they are not
    // live through the sampling loops: nearly every workgroup makes one pass)
    double bq[kFuseTasks][kFuseKS];
#pragma unroll
    for (int ti = 0; ti < kFuseTasks; ++ti) {
      // (a wave without a ti-th task repeats the last one: no branches around the loads and the MFMAs, one around
      // the write)
      const int t = wave + 4 * ti < ntasks ? wave + 4 * ti : ntasks - 1;
      const double *bsrc = frag + (size_t)((t * nrt_inv) >> 8) * ks_pitch * kWave + lane;
#pragma unroll
      for (int ks = 0; ks < kFuseKS; ++ks) bq[ti][ks] = bsrc[(ks < nks ? ks : 0) * kWave];
    }
    if (tid < DW) sBad[tid] = 0;
    lds_barrier();
    MSNAP_GSTL(1);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int e = h ? e1 : e0;
      const double w = h ? w1 : w0;
      if (e < words) {
        const int dl = (int)(((unsigned)e * m4_inv) >> 20), r = e - dl * (m * 4);
        sW[(size_t)(dl * 4 + (r & 3)) * wpitch + (r >> 2)] = w;
        if (!__builtin_isfinite(w)) sBad[dl] = 1;
      }
    }
    lds_barrier();
    MSNAP_GSTL(2);
    // the GEMM: C[row][c] = sum_j W[row][j] Gop[j][c], k ascending as in grid_gemm_kernel.  All A operands first (one
    // LDS flight), then the tasks' MFMA chains side by side, then the results (written as they are: a drone with a
    // non-finite waypoint is turned into NaN by the readers below)
    {
      double av[kFuseTasks][kFuseKS];
#pragma unroll
      for (int ti = 0; ti < kFuseTasks; ++ti) {
        const int t = wave + 4 * ti < ntasks ? wave + 4 * ti : ntasks - 1;
        const double *arow = sW + (size_t)((t - ((t * nrt_inv) >> 8) * nrt) * 16 + col) * wpitch;
#pragma unroll
        for (int ks = 0; ks < kFuseKS; ++ks) {
          const int j = 4 * ks + kq;
          const double v = arow[j < m ? j : m - 1];
          av[ti][ks] = j < m ? v : 0.0;
        }
      }
      v4f64s acc[kFuseTasks];
#pragma unroll
      for (int ti = 0; ti < kFuseTasks; ++ti) acc[ti] = v4f64s{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < kFuseKS; ++ks)
        if (ks < nks) {
#pragma unroll
          for (int ti = 0; ti < kFuseTasks; ++ti)
            acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ti][ks], bq[ti][ks], acc[ti], 0, 0, 0);
        }
#pragma unroll
      for (int ti = 0; ti < kFuseTasks; ++ti) {
        const int t = wave + 4 * ti;
        const int ct = (t * nrt_inv) >> 8, rt = t - ct * nrt;
        const int c = 16 * ct + col;
        if (t < ntasks && c < ncols) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int dl = rt * 4 + r;                  // row 4 * dl + kq: drone dl, axis kq
            if (dl < DW) sC[(size_t)(dl * 4 + kq) * cpitch + c] = acc[ti][r];
          }
        }
      }
    }
    MSNAP_GSTL(7);
    lds_barrier();
    MSNAP_GSTL(3);
    // coefficients, durations and status as msnap_solve_grid leaves them: a wave per drone, 16 bytes per lane
    {
      const int ppd = M * 4 * NC / 2;                     // double2 per drone
      for (int dl = wave; dl < nd; dl += 4) {
        const bool bad = grid_st != 0 || sBad[dl] != 0;
        double *cout = coef + (size_t)(d0 + dl) * M * 4 * NC;
        for (int e = lane; e < ppd; e += kWave) {
          const int g = 2 * e;
          const int kc = g % NC, q = g / NC;
          const int a = q & 3, seg = q >> 2;
          double2 v = *reinterpret_cast<const double2 *>(sC + (size_t)(dl * 4 + a) * cpitch + seg * NC + kc);
          if (bad) v = make_double2(__builtin_nan(""), __builtin_nan(""));
          *reinterpret_cast<double2 *>(cout + g) = v;
        }
      }
      for (int e = tid; e < nd * M; e += blockDim.x) dur[(size_t)d0 * M + e] = gdur[e % M];
      if (tid < nd) status[d0 + tid] = sBad[tid] ? MSNAP_ST_NONFINITE : grid_st;
    }
    MSNAP_GSTL(4);
    const int items = nd * M * 3;
    const bool generic = sBad[DW] != 0;
    for (int it = tid; it < items; it += blockDim.x) {
      const int dl = div_by(it, inv_piece), rem = it - dl * (3 * M);
      const int i = rem / 3, a = rem - i * 3;
      double *img = sImg + (size_t)dl * per_drone;
      const double *cbase = sC + (size_t)dl * 4 * cpitch;
      const bool bad = grid_st != 0 || sBad[dl] != 0;
      if (generic || bad) {
        // (a NaN row evaluates to NaN at every sample, whichever piece the search picks)
        if (i == 0)
          for (int sq = 0; sq < S; ++sq) {
            double x = __builtin_nan("");
            if (!bad) sample_point_generic<NC>(cbase, gdur, M, (double)sq * dt, a, x, NC, cpitch);
            img[(size_t)sq * 3 + a] = x;
          }
        continue;
      }
      const double *crow = cbase + (size_t)a * cpitch + i * NC;
      sample_piece<NC>(sB[i], sB[i + 1], i, M, S, dt, img, 3, a, [&](double (&c)[NC]) {
#pragma unroll
        for (int q = 0; q < NC; q += 2) {
          const double2 v = *reinterpret_cast<const double2 *>(crow + q);
          c[q] = v.x;
          c[q + 1] = v.y;
        }
      });
    }
    lds_barrier();
    MSNAP_GSTL(5);
    sample_image_out(sImg, tid, blockDim.x, nd, d0, per_drone, S, pos, pos_t, Rp, kbox, kkey);
    lds_barrier();
    MSNAP_GSTL(6);
  }
  MSNAP_GSTL_FLUSH();
}

__global__ void collide_key_kernel(const double *__restrict__ pos, int N, int S, double *__restrict__ box,
                                   unsigned *__restrict__ key);
constexpr int kKeyDrones = 4;      // drones (wavefronts) per workgroup of collide_key_kernel

// `pos_t`: the sampler's second output for the pairwise pass (msnap_sample_collide): with `keys_form` the per-drone
// boxes [N][6] followed by the sort keys [N] (uint32) of a whole-swarm pass behind the broad phase, otherwise the
// transposed row image
int launch_sample(msnap_ctx *ctx, int n_drones, int n_seg, const double *coef, const double *dur, double dt,
                  int n_samples, int n_axes, double *pos, double *pos_t, bool keys_form) {
  const int Rp = (n_drones + kRowBlockRows - 1) / kRowBlockRows * kRowBlockRows;
  double *kbox = keys_form ? pos_t : nullptr;
  unsigned *kkey = keys_form ? reinterpret_cast<unsigned *>(pos_t + (size_t)n_drones * 6) : nullptr;
  if (keys_form) pos_t = nullptr;
  // drones per workgroup: about 256 (piece, axis) threads, within 48 KB of LDS image
  const size_t img_per_drone = (size_t)n_samples * n_axes * sizeof(double);
  int dw = 256 / (n_seg * n_axes);
  if (dw < 1) dw = 1;
  if (dw > 16) dw = 16;
  while (dw > 1 && dw * img_per_drone > 48 * 1024) --dw;
  const size_t lds = ((size_t)dw * (n_seg + 1)) * sizeof(double) + (((size_t)dw + 1) & ~(size_t)1) * sizeof(int) +
                     dw * img_per_drone;
  if (lds <= 64 * 1024) {
    size_t blocks = ((size_t)n_drones + dw - 1) / dw;
    if (blocks > (size_t)ctx->n_cu * 64) blocks = (size_t)ctx->n_cu * 64;
    if (ctx->order == 7)
      hipLaunchKernelGGL((sample_kernel<8>), dim3((unsigned)blocks), dim3(256), lds, ctx->stream, coef, dur, dt,
                         n_drones, n_seg, n_samples, n_axes, dw, pos, pos_t, Rp, kbox, kkey);
    else
      hipLaunchKernelGGL((sample_kernel<10>), dim3((unsigned)blocks), dim3(256), lds, ctx->stream, coef, dur, dt,
                         n_drones, n_seg, n_samples, n_axes, dw, pos, pos_t, Rp, kbox, kkey);
  } else {
    // one drone's samples exceed the image: the (drone, sample) form
    const size_t total = (size_t)n_drones * n_samples;
    size_t blocks = (total + 255) / 256;
    if (blocks > (size_t)ctx->n_cu * 16) blocks = (size_t)ctx->n_cu * 16;
    if (ctx->order == 7)
      hipLaunchKernelGGL((sample_generic_kernel<8>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, coef, dur, dt,
                         n_drones, n_seg, n_samples, n_axes, pos);
    else
      hipLaunchKernelGGL((sample_generic_kernel<10>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, coef, dur, dt,
                         n_drones, n_seg, n_samples, n_axes, pos);
    if (pos_t != nullptr) {     // paths too long for the image: the transposition pass after all
      MSNAP_HIP(ctx, hipGetLastError());
      const int E = n_samples * 3;
      hipLaunchKernelGGL(collide_transpose_kernel, dim3(Rp / 64, (E + 31) / 32), dim3(256), 0, ctx->stream,
                         (const double *)pos, n_drones, Rp, E, pos_t, (E + 31) / 32, (int32_t *)nullptr, (size_t)0,
                         (const int32_t *)nullptr, (double *)nullptr);
    }
    if (kbox != nullptr) {      // ... or the key pass over the finished positions
      MSNAP_HIP(ctx, hipGetLastError());
      hipLaunchKernelGGL(collide_key_kernel, dim3((n_drones + kKeyDrones - 1) / kKeyDrones), dim3(kWave * kKeyDrones), 0,
                         ctx->stream, (const double *)pos, n_drones, n_samples, kbox, kkey);
    }
  }
  MSNAP_HIP(ctx, hipGetLastError());
  return MSNAP_OK;
}

// The fused launch, or the two kernels when the shape is outside its range (more than 11 segments, samples beyond the
// LDS image).  `pos_t` / `keys_form` as launch_sample.
int launch_grid_sample(msnap_ctx *ctx, int n_drones, const double *wp, double dt, int n_samples, double *coef,
                       double *dur, int32_t *status, double *pos, double *pos_t, bool keys_form) {
  const int M = ctx->grid_seg, m = M + 1, nc = ctx->order + 1;
  const int ks_pitch = grid_frag_ks_pitch(ctx, M);
  const size_t img_per_drone = (size_t)n_samples * 3 * sizeof(double);
  const int ncols = M * nc, nct = (ncols + 15) / 16, nks = (m + 3) / 4;
  int dw = 256 / (M * 3);
  if (dw > 16) dw = 16;
  auto lds_of = [&](int d) -> size_t {
    const size_t rows16 = (size_t)((d * 4 + 15) / 16) * 16;
    return (size_t)(((m + 1) & ~1) + ((rows16 * (m | 1) + 1) & ~(size_t)1) + (size_t)d * 4 * (ncols + 2)) * 8 +
           d * img_per_drone + (size_t)(d + 2) * 4;
  };
  while (dw > 1 && (dw * img_per_drone > 48 * 1024 || lds_of(dw) > 64 * 1024 ||
                    ((dw * 4 + 15) / 16) * nct > 4 * kFuseTasks))
    --dw;
  const bool fused = !ctx->no_grid_sample && ks_pitch > 0 && nks <= kFuseKS && dw >= 1 && lds_of(dw) <= 64 * 1024 &&
                     ((dw * 4 + 15) / 16) * nct <= 4 * kFuseTasks && dw * (size_t)m * 4 <= 512;
  if (!fused) {
    int rc = launch_solve_grid(ctx, n_drones, wp, coef, dur, status);
    if (rc) return rc;
    return launch_sample(ctx, n_drones, M, coef, dur, dt, n_samples, 3, pos, pos_t, keys_form);
  }
  if (stream_is_capturing(ctx))
    for (DevBuf *b : {&ctx->grid_t, &ctx->grid_frag, &ctx->grid_dur, &ctx->grid_status}) b->in_graph = true;
  const int Rp = (n_drones + kRowBlockRows - 1) / kRowBlockRows * kRowBlockRows;
  double *kbox = keys_form ? pos_t : nullptr;
  unsigned *kkey = keys_form ? reinterpret_cast<unsigned *>(pos_t + (size_t)n_drones * 6) : nullptr;
  if (keys_form) pos_t = nullptr;
  size_t blocks = ((size_t)n_drones + dw - 1) / dw;
  if (blocks > (size_t)ctx->n_cu * 64) blocks = (size_t)ctx->n_cu * 64;
  note_kernel(ctx, "msnap::grid_sample_kernel<%d>", nc);
#define MSNAP_GS_LAUNCH(NCV)                                                                                            \
  hipLaunchKernelGGL((grid_sample_kernel<NCV>), dim3((unsigned)blocks), dim3(256), lds_of(dw), ctx->stream, wp,          \
                     (const double *)ctx->grid_frag.p, ks_pitch, (const double *)ctx->grid_dur.p,                         \
                     (const int32_t *)ctx->grid_status.p, dt, n_drones, M, n_samples, dw, coef, dur, status, pos, pos_t,  \
                     Rp, kbox, kkey)
  if (nc == 8) MSNAP_GS_LAUNCH(8); else MSNAP_GS_LAUNCH(10);
#undef MSNAP_GS_LAUNCH
  MSNAP_HIP(ctx, hipGetLastError());
  return MSNAP_OK;
}

// ------------------------------------------------------------------------------------
// f1: Trajectory.eval / Polynomial4D.eval -- differential-flatness outputs
// (reference src/optimizations/uav_trajectory.py:64-85 and :119-127: piece lookup with
// '<=', derivative polynomials built as (i+1)*p[i+1], Horner with separate multiply and
// add, thrust = acc + (0,0,9.81), body axes from yaw, omega from the jerk).
//   out[d][s] = pos[3] vel[3] acc[3] omega[3] yaw ; NaN when t is outside [0, duration]
// ------------------------------------------------------------------------------------
template <int NC>
__device__ __forceinline__ void horner_derivs(const double *__restrict__ c, double t, double &p0, double &p1,
                                              double &p2, double &p3) {
#pragma clang fp contract(off)
  double d0[NC], d1[NC], d2[NC], d3[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) d0[i] = c[i];
#pragma unroll
  for (int i = 0; i < NC - 1; ++i) d1[i] = (double)(i + 1) * d0[i + 1];
#pragma unroll
  for (int i = 0; i < NC - 2; ++i) d2[i] = (double)(i + 1) * d1[i + 1];
#pragma unroll
  for (int i = 0; i < NC - 3; ++i) d3[i] = (double)(i + 1) * d2[i + 1];
  p0 = p1 = p2 = p3 = 0.0;
#pragma unroll
  for (int i = NC - 1; i >= 0; --i) p0 = p0 * t + d0[i];
#pragma unroll
  for (int i = NC - 2; i >= 0; --i) p1 = p1 * t + d1[i];
#pragma unroll
  for (int i = NC - 3; i >= 0; --i) p2 = p2 * t + d2[i];
#pragma unroll
  for (int i = NC - 4; i >= 0; --i) p3 = p3 * t + d3[i];
}

template <int NC>
__global__ void __launch_bounds__(256)
flat_eval_kernel(const double *__restrict__ coef, const double *__restrict__ dur, const double *__restrict__ ts,
                 int N, int M, int S, double *__restrict__ out) {
#pragma clang fp contract(off)
  // A workgroup's 256 (drone, instant) items own 256 * 13 CONSECUTIVE output doubles: they go through an LDS
  // image (stride 13 is odd: conflict-free) and leave as contiguous 8-byte-per-lane runs instead of 13 stores
  // at a 104-byte stride per lane.
  __shared__ double image[256 * 13];
  const size_t total = (size_t)N * S;
  const size_t span = (size_t)gridDim.x * blockDim.x;
  const size_t trips = (total + span - 1) / span;      // the same for every thread: the barriers below are uniform
  for (size_t trip = 0; trip < trips; ++trip) {
    const size_t block_base = trip * span + (size_t)blockIdx.x * blockDim.x;
    const size_t idx = block_base + threadIdx.x;
    double *o = image + threadIdx.x * 13;
    if (idx < total) {
    const int d = (int)(idx / S);
    const int s = (int)(idx - (size_t)d * S);
    const double t = ts[s];
    const double *dr = dur + (size_t)d * M;
    double acc_t = 0.0;
    int seg = -1;
    for (int i = 0; i < M; ++i) {
      const double Ti = dr[i];
      if (seg < 0) {
        if (t <= acc_t + Ti) seg = i;
        else acc_t = acc_t + Ti;
      }
    }
    if (!(t >= 0.0) || seg < 0) {
#pragma unroll
      for (int q = 0; q < 13; ++q) o[q] = __builtin_nan("");
    } else {
    const double tl = t - acc_t;
    const double *c = coef + ((size_t)d * M + seg) * 4 * NC;
    double px, vx, ax, jx, py, vy, ay, jy, pz, vz, az, jz, yaw, dyaw, q2, q3;
    horner_derivs<NC>(c + 0 * NC, tl, px, vx, ax, jx);
    horner_derivs<NC>(c + 1 * NC, tl, py, vy, ay, jy);
    horner_derivs<NC>(c + 2 * NC, tl, pz, vz, az, jz);
    horner_derivs<NC>(c + 3 * NC, tl, yaw, dyaw, q2, q3);
    const double thx = ax + 0.0, thy = ay + 0.0, thz = az + 9.81;
    const double tn = sqrt(thx * thx + thy * thy + thz * thz);
    const double zbx = thx / tn, zby = thy / tn, zbz = thz / tn;
    const double xwx = cos(yaw), xwy = sin(yaw), xwz = 0.0;
    // y_body = normalize(z_body x x_world)
    double ybx = zby * xwz - zbz * xwy, yby = zbz * xwx - zbx * xwz, ybz = zbx * xwy - zby * xwx;
    const double yn = sqrt(ybx * ybx + yby * yby + ybz * ybz);
    ybx = ybx / yn; yby = yby / yn; ybz = ybz / yn;
    // x_body = y_body x z_body
    const double xbx = yby * zbz - ybz * zby, xby = ybz * zbx - ybx * zbz, xbz = ybx * zby - yby * zbx;
    const double jd = jx * zbx + jy * zby + jz * zbz;
    const double hx = (jx - jd * zbx) / tn, hy = (jy - jd * zby) / tn, hz = (jz - jd * zbz) / tn;
    o[0] = px; o[1] = py; o[2] = pz;
    o[3] = vx; o[4] = vy; o[5] = vz;
    o[6] = ax; o[7] = ay; o[8] = az;
    o[9] = -(hx * ybx + hy * yby + hz * ybz);
    o[10] = hx * xbx + hy * xby + hz * xbz;
    o[11] = zbz * dyaw;
    o[12] = yaw;
    }
    }
    __syncthreads();
    const size_t ebase = block_base * 13, nelem = total * 13;
#pragma unroll
    for (int j = 0; j < 13; ++j) {
      const size_t e = (size_t)j * blockDim.x + threadIdx.x;
      if (ebase + e < nelem) out[ebase + e] = image[e];
    }
    __syncthreads();
  }
}

int launch_eval_flat(msnap_ctx *ctx, int n_drones, int n_seg, const double *coef, const double *dur,
                     int n_samples, const double *ts, double *out) {
  const size_t total = (size_t)n_drones * n_samples;
  size_t blocks = (total + 255) / 256;
  if (blocks > (size_t)ctx->n_cu * 16) blocks = (size_t)ctx->n_cu * 16;
  if (ctx->order == 7)
    hipLaunchKernelGGL((flat_eval_kernel<8>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, coef, dur, ts,
                       n_drones, n_seg, n_samples, out);
  else
    hipLaunchKernelGGL((flat_eval_kernel<10>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, coef, dur, ts,
                       n_drones, n_seg, n_samples, out);
  MSNAP_HIP(ctx, hipGetLastError());
  return MSNAP_OK;
}

// ------------------------------------------------------------------------------------
// Snap cost J = sum_seg int_0^T (p^(k)(t))^2 dt per (drone, axis), k = (order+1)/2: the quantity
// the trajectory minimises (the reference never evaluates it; it is the objective of the QP
// whose KKT system its collocation rows encode).  Monomial Hessian
//   Q[m][n] = m!/(m-k)! * n!/(n-k)! * T^(m+n-2k+1) / (m+n-2k+1),  m, n >= k.
// ------------------------------------------------------------------------------------
template <int NC>
__global__ void __launch_bounds__(256)
snap_cost_kernel(const double *__restrict__ coef, const double *__restrict__ dur, int N, int M,
                 double *__restrict__ cost) {
  constexpr int K = NC / 2;
  const int total = N * 4;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int d = idx >> 2, a = idx & 3;
    double J = 0.0;
    for (int i = 0; i < M; ++i) {
      const double *c = coef + (((size_t)d * M + i) * 4 + a) * NC;
      const double T = dur[(size_t)d * M + i];
      double f[K];                      // f[q] = (k+q)!/q! * c[k+q]
#pragma unroll
      for (int q = 0; q < K; ++q) {
        double ff = 1.0;
#pragma unroll
        for (int r = q + 1; r <= K + q; ++r) ff *= (double)r;
        f[q] = ff * c[K + q];
      }
      double tp[2 * K];                 // T^e, e = 1 .. 2k-1
      tp[0] = 1.0;
#pragma unroll
      for (int e = 1; e < 2 * K; ++e) tp[e] = tp[e - 1] * T;
      double acc = 0.0;
#pragma unroll
      for (int p = 0; p < K; ++p)
#pragma unroll
        for (int q = 0; q < K; ++q) acc += f[p] * f[q] * (tp[p + q + 1] / (double)(p + q + 1));
      J += acc;
    }
    cost[idx] = J;
  }
}

int launch_snap_cost(msnap_ctx *ctx, int n_drones, int n_seg, const double *coef, const double *dur, double *cost) {
  const int total = n_drones * 4;
  int blocks = (total + 255) / 256;
  if (blocks > ctx->n_cu * 8) blocks = ctx->n_cu * 8;
  if (ctx->order == 7)
    hipLaunchKernelGGL((snap_cost_kernel<8>), dim3(blocks), dim3(256), 0, ctx->stream, coef, dur, n_drones, n_seg, cost);
  else
    hipLaunchKernelGGL((snap_cost_kernel<10>), dim3(blocks), dim3(256), 0, ctx->stream, coef, dur, n_drones, n_seg, cost);
  MSNAP_HIP(ctx, hipGetLastError());
  return MSNAP_OK;
}

// ------------------------------------------------------------------------------------
// Formation pass: for every owned drone i the minimum over all other drones j and all common
// samples s of |p_i(s) - p_j(s)|.  Semantics are this repo's: no reference
// implementation exists.
//
// Arithmetic.  One lane per TWO row drones (row blocks of 128: rows lane and lane + 64); the column
// drone is wave-uniform, its samples arrive through scalar loads and are SGPR operands of the 7
// operations per pair and sample: 3 differences, d2 = fma(dz, dz, fma(dy, dy, dx * dx)) -- the
// definition of include/msnap.h, restated bit for bit by both oracles, which is what decides ties
// between equidistant formation neighbours -- and the minimum.
// The running minima of a block of 8 columns stay in registers over all samples.  A scalar load has
// only an all-or-nothing wait, so a wave has ONE column fetch (6 samples, 2 x 42 operations) in
// flight while it computes the previous one; the other waves of the SIMD (4 fit) cover the rest of
// the latency.  The rows are read from a transposed image [sample][xyz][row] written once per call
// (coalesced 512-byte loads; drone-major row loads would saturate the texture addresser).
//
// Work.  Columns that are also rows of this call (a single GPU: all of them; a shard: its own
// 1/G) are evaluated ONCE per unordered pair: row block I meets the own-range columns from its
// own first column on -- one-sidedly inside its diagonal block, two-sidedly behind it: d2 is
// bitwise symmetric, so after such a block the per-column minima over the 128 rows (through a
// per-wave LDS image) are stored as partial results of the COLUMN drones.  The (row block,
// column) units of the whole launch form one line -- row block after row block, the columns each
// still has to meet -- and every wave takes an equal contiguous share of it: the waves finish
// together, no SIMD idles while another still has tiles queued (a triangular grid of whole tiles
// leaves 2.03 tiles per wave: a third of the chip waits for the rest).  A merge kernel takes the
// minimum over a drone's row-side shares and column-side row blocks (lowest partner index wins
// ties on both sides).
// ------------------------------------------------------------------------------------
constexpr int kRowsPerLane = 2;   // (3 rows x 4-column blocks, 165 VGPRs, 3 waves per SIMD: 4096 x 91 in 346 us against 230)
constexpr int kRowBlock = kWave * kRowsPerLane;
static_assert(kRowBlock == kRowBlockRows, "the sampler's row image uses the pairwise pass's row pitch");
constexpr int kColBlock = 8;      // column drones whose running minima a lane keeps in registers (per row)
constexpr int kSampleChunk = 6;   // samples per scalar fetch

struct CollideGeom {
  int R, ro, Cn, S;         // rows, global index of row 0, columns, samples
  int Rp;                   // rows of the transposed row image (R rounded up to whole row blocks)
  int os, oe;               // own range: the columns [os, oe) that are this call's rows (os == oe: none)
  int sym;                  // own-range pairs evaluated once
  int n_rb;                 // row blocks
  int upw;                  // (row block, column) units per share
  int upw_tail;             // units per share behind `split` (smaller shares even out the end of the launch)
  long long split;          // first unit of the tail shares (a multiple of upw)
  int sparts;               // sample parts: a share of the line is taken by `sparts` waves, each a range of chunks
  long long total;          // units of the whole line
  // The launch walks the units [u_lo, u_lo + u_n) of the line, cut into its own shares (`split` is relative to
  // u_lo); they touch the row blocks I_lo .. I_hi.  A whole pass: 0, total, 0, n_rb - 1.  One rank's part of a
  // pass over the whole swarm (msnap_formation_collide_part, `part` = 1): a contiguous 1/P of the line; the
  // transposed row image and the partial buffers are indexed relative to I_lo, and the merge writes the squared
  // minimum of EVERY drone (+inf / -1 where this part met none of its pairs) instead of distances.
  long long u_lo, u_n;
  int I_lo, I_hi;
  int part;
};

// units of the row blocks before I: without the own-range shortcut every row block meets all Cn columns;
// with it row block I skips the own-range columns before its own first one
__device__ __host__ __forceinline__ long long collide_ustart(const CollideGeom &g, int I) {
  return g.sym ? (long long)I * g.Cn - (long long)kRowBlock * I * (I - 1) / 2 : (long long)I * g.Cn;
}

// share w covers the units [collide_share_begin(w), collide_share_begin(w + 1)) of the line
__device__ __host__ __forceinline__ long long collide_share_begin(const CollideGeom &g, long long w) {
  const long long w1 = g.split / g.upw;                  // shares of the head
  const long long u = w <= w1 ? w * g.upw : g.split + (w - w1) * g.upw_tail;
  return g.u_lo + (u < g.u_n ? u : g.u_n);
}
__device__ __host__ __forceinline__ long long collide_share_of(const CollideGeom &g, long long u) {   // u: a unit of the launch
  const long long ul = u - g.u_lo;
  return ul < g.split ? ul / g.upw : g.split / g.upw + (ul - g.split) / g.upw_tail;
}

// 6 samples of a column drone = 18 contiguous doubles in scalar registers.  The loads are issued by
// hand: next to LDS fences the compiler can no longer prove that the position arrays are not written
// and would fall back to vector loads of a uniform address.  SMEM returns out of order, so the only
// wait is lgkmcnt(0); it carries the registers as operands so that no use is scheduled above it.
typedef unsigned int u32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));
struct ColChunk {
  u32x16 a, b;
  u32x4s c;
  __device__ __forceinline__ void fetch(const double *p) {
    asm volatile("s_load_dwordx16 %0, %3, 0x0\n\ts_load_dwordx16 %1, %3, 0x40\n\ts_load_dwordx4 %2, %3, 0x80"
                 : "=&s"(a), "=&s"(b), "=&s"(c)
                 : "s"(p));
  }
  // `after` (a value the preceding arithmetic produces) pins the wait behind that arithmetic: without
  // it the compiler may sink the other register set's VALU work below this wait and lose the overlap
  __device__ __forceinline__ void wait(double &after) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b), "+s"(c), "+v"(after));
  }
  // element i (a compile-time constant after unrolling) minus v, exactly rounded.  The difference is issued
  // by hand with the scalar register pair as the first operand: a lane owns two rows, and left to the
  // compiler a scalar value with two VALU users is first copied into vector registers (36 extra moves per
  // fetch).
  __device__ __forceinline__ double minus(int i, double v) const {
    const unsigned long long x =
        i < 8 ? ((unsigned long long)a[2 * i + 1] << 32) | a[2 * i]
              : i < 16 ? ((unsigned long long)b[2 * (i - 8) + 1] << 32) | b[2 * (i - 8)]
                       : ((unsigned long long)c[2 * (i - 16) + 1] << 32) | c[2 * (i - 16)];
    double d;
    asm("v_add_f64 %0, %1, -%2" : "=v"(d) : "s"(x), "v"(v));
    return d;
  }
};

__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
  const int lo = __shfl_xor(__double2loint(v), mask);
  const int hi = __shfl_xor(__double2hiint(v), mask);
  return __hiloint2double(hi, lo);
}

// What a lane carries through a share: its kRowsPerLane rows (lane, lane + 64, ...)
struct RowSet {
  bool live[kRowsPerLane];       // row exists (rows past the batch end replay row R - 1 and must not win)
  int grow[kRowsPerLane];        // global index (to exclude the drone itself)
  double best[kRowsPerLane];     // row-side minimum over the share's columns
  int bestj[kRowsPerLane];
};

// The exact broad phase of a whole-swarm pass (launch_formation_collide, "cull" path).  Rows and columns are walked in
// a spatially sorted order (Morton order of the drones' path boxes); every aligned group of 8 drones of that order has
// the bounding box of its drones' finite samples and the largest of its drones' BOUNDS -- a squared distance each drone
// is known to attain to some other drone (its sorted neighbours, collide_bound_kernel).  A two-sided share (128 rows x
// 8 columns) is evaluated unless, for every one of the row block's 16 groups, the box distance to the column group
// exceeds both groups' largest bounds: then no pair of the share can lower any of its rows' or columns' minima -- nor tie
// them: the test is strict, and the box distance is formed with the pass's own fma formula, so it never exceeds the
// squared distance of any pair of the two boxes.  Minima, partners (compared by ORIGINAL index, `oid`) and hits are
// those of the full pass, whatever the order and whatever is skipped.
struct CollideCull {
  const double *colbox;    // [ceil(N / 8)][6] lo x,y,z / hi x,y,z per aligned group of 8 sorted drones
};

__device__ __forceinline__ double box_box_lb2(const double *__restrict__ a, const double *__restrict__ b) {
#pragma clang fp contract(off)
  double gp[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double g1 = a[k] - b[3 + k], g2 = b[k] - a[3 + k];     // lo_a - hi_b, lo_b - hi_a
    gp[k] = fmax(0.0, fmax(g1, g2));
  }
  return __builtin_fma(gp[2], gp[2], __builtin_fma(gp[1], gp[1], gp[0] * gp[0]));
}

// One block of NC (even, <= kColBlock) consecutive columns [cj, cj + ncols) against the wave's
// kRowBlock rows: straight-line code over the columns -- with a branch inside the column loop the
// scalar register sets cross basic blocks and the compiler copies every fetched value into vector
// registers (36 extra VALU moves per fetch) -- so a short block takes the next instance up and
// re-reads its last column instead of branching.
template <int NC, bool CULL = false>
__device__ __forceinline__ void collide_block(const CollideGeom &g, const double *__restrict__ prowT,
                                              const double *__restrict__ pcol, int cj, int ncols, bool two_sided,
                                              RowSet &rs, int I, int h, int crow, int lane, double *sFold, int *sFoldI,
                                              double *__restrict__ cpart_d2, int32_t *__restrict__ cpart_i,
                                              const int32_t *__restrict__ oid = nullptr) {
#pragma clang fp contract(off)
  constexpr int CH = kSampleChunk, RPL = kRowsPerLane;
  const int S = g.S;
  const int stride = S * 3;
  double acc[RPL][NC];
#pragma unroll
  for (int rr = 0; rr < RPL; ++rr)
#pragma unroll
    for (int jj = 0; jj < NC; ++jj) acc[rr][jj] = INFINITY;
  // One or two samples behind the last whole chunk (91 = 15 x 6 + 1) go through a plain loop at the end;
  // a longer remainder is a last chunk moved back to overlap its predecessor (a minimum does not mind seeing
  // a sample twice), so that every chunk takes the wide scalar loads.
  const int rem = S % CH;
  const int Sw = (rem == 1 || rem == 2) ? S - rem : S;
  // sample part h of g.sparts takes its range of whole chunks (the last part also the plain remainder)
  const int nch = (Sw + CH - 1) / CH;
  const int sc_begin = (int)((long long)nch * h / g.sparts) * CH, sc_end = (int)((long long)nch * (h + 1) / g.sparts) * CH;
  for (int sc = sc_begin; sc < sc_end; sc += CH) {
    const int s0 = (Sw - sc < CH) ? Sw - CH : sc;
    double row[RPL][CH][3];
    // the rows come from the transposed image [sample][xyz][row]: the 64 lanes of a load read 512
    // contiguous bytes (from the drone-major layout every lane would touch its own cache line, and with
    // several rows per lane the texture addresser, not the VALU, would set the pace: TA_BUSY 79 %)
    // (uniform base per load, lane offset in one register: no per-lane 64-bit address arithmetic)
    const double *pt = prowT + (size_t)s0 * 3 * g.Rp;
#pragma unroll
    for (int q = 0; q < CH; ++q)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double *pk = pt + (size_t)(3 * q + k) * g.Rp;
#pragma unroll
        for (int rr = 0; rr < RPL; ++rr) row[rr][q][k] = pk[lane + rr * kWave];
      }
    // one running pointer walks the block's columns; `nvalid` is made opaque per chunk so that the
    // per-column strides are not hoisted out of the sample loop into spilled scalar registers
    int nvalid = ncols;
    asm volatile("" : "+s"(nvalid));
    const double *pc = pcol + ((size_t)cj * S + s0) * 3;
    auto consume = [&](int jj, const ColChunk &k) {
#pragma unroll
      for (int q = 0; q < CH; ++q)
#pragma unroll
        for (int rr = 0; rr < RPL; ++rr) {
          const double dx = k.minus(3 * q + 0, row[rr][q][0]), dy = k.minus(3 * q + 1, row[rr][q][1]),
                       dz = k.minus(3 * q + 2, row[rr][q][2]);
          const double d2 = __builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx));
          // the minimum by hand: behind the wait's register tie the compiler no longer knows the accumulator
          // to be canonical and would put a v_max in front of every fmin.  Neither operand can be a signalling
          // NaN (d2 comes out of arithmetic, the accumulator out of earlier minima), and a quiet NaN loses, as
          // fmin's would.
          asm("v_min_f64 %0, %1, %0" : "+v"(acc[rr][jj]) : "v"(d2));
        }
    };
    // two register sets alternate: the loads of column j+1 are issued right after the wait for
    // column j and fly during its RPL x 6 x 7 VALU operations
    ColChunk ca, cb2;
    ca.fetch(pc);
#pragma unroll
    for (int jj = 0; jj < NC; jj += 2) {
      pc += (jj + 1 < nvalid) ? stride : 0;
      ca.wait(acc[RPL - 1][jj > 0 ? jj - 1 : 0]);
      cb2.fetch(pc);
      consume(jj, ca);
      pc += (jj + 2 < nvalid) ? stride : 0;
      cb2.wait(acc[RPL - 1][jj]);
      if (jj + 2 < NC) ca.fetch(pc);
      consume(jj + 1, cb2);
    }
  }
  for (int s1 = (h == g.sparts - 1) ? Sw : S; s1 < S; ++s1) {
    const double *pt = prowT + (size_t)s1 * 3 * g.Rp;
    const double *px = pt, *py = pt + g.Rp, *pz = pt + 2 * (size_t)g.Rp;
    double rx[RPL], ry[RPL], rz[RPL];
#pragma unroll
    for (int rr = 0; rr < RPL; ++rr) {
      rx[rr] = px[lane + rr * kWave];
      ry[rr] = py[lane + rr * kWave];
      rz[rr] = pz[lane + rr * kWave];
    }
#pragma unroll
    for (int jj = 0; jj < NC; ++jj) {
      const double *pcs = pcol + ((size_t)(cj + (jj < ncols ? jj : ncols - 1)) * S + s1) * 3;
      const double cx = pcs[0], cy = pcs[1], cz = pcs[2];
#pragma unroll
      for (int rr = 0; rr < RPL; ++rr) {
        const double dx = cx - rx[rr], dy = cy - ry[rr], dz = cz - rz[rr];
        acc[rr][jj] = __builtin_fmin(__builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx)), acc[rr][jj]);
      }
    }
  }
  // row side: columns ascend, so the lowest index wins a tie (cull path: sorted order, ties by the ORIGINAL index)
#pragma unroll
  for (int jj = 0; jj < NC; ++jj) {
    const int j = cj + jj;
    const int oj = CULL ? oid[jj < ncols ? j : cj] : j;
#pragma unroll
    for (int rr = 0; rr < RPL; ++rr) {
      acc[rr][jj] = (j == rs.grow[rr] || jj >= ncols) ? INFINITY : acc[rr][jj];
      const bool take = CULL ? (acc[rr][jj] < rs.best[rr]) | ((acc[rr][jj] == rs.best[rr]) & (oj < rs.bestj[rr]))
                             : (acc[rr][jj] < rs.best[rr]);
      if (take) {
        rs.best[rr] = acc[rr][jj];
        rs.bestj[rr] = oj;
      }
    }
  }
  if (two_sided) {
    // column side: min over the kRowBlock rows of every column of the block, with the lowest row.  Each
    // lane first folds its own rows (the lower row wins a tie); the 64 candidates of a column go through the
    // LDS image [column][lane]; lane = kColBlock * part + column then scans its part of the column and the
    // parts are folded with cross-lane exchanges.  Rows past the batch end replay row R-1 and must not win.
    // (The lane index is taken from an opaque copy: left visible, the fold's lane-derived addresses are hoisted
    // out of the share's loops and kept -- in scratch -- across the column loop, whose 128 registers are spoken for.)
    asm volatile("" : "+v"(lane));
    int orow[RPL];       // cull path: the rows' ORIGINAL indices, fetched here rather than held through the column loop
#pragma unroll
    for (int rr = 0; rr < RPL; ++rr) orow[rr] = CULL ? oid[rs.live[rr] ? I * kRowBlock + rr * kWave + lane : g.R - 1] : 0;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      double v = rs.live[0] ? acc[0][c] : INFINITY;
      int vi = CULL ? orow[0] : lane;
#pragma unroll
      for (int rr = 1; rr < RPL; ++rr) {
        const double o = rs.live[rr] ? acc[rr][c] : INFINITY;
        const int oi = CULL ? orow[rr] : lane + rr * kWave;
        if (o < v || (CULL && o == v && oi < vi)) {
          v = o;
          vi = oi;
        }
      }
      sFold[c * kWave + lane] = v;
      sFoldI[c * kWave + lane] = vi;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    constexpr int PER = kColBlock;                 // candidates a lane scans: 64 lanes / (64 / kColBlock) parts
    const int c = lane & (kColBlock - 1), part = lane / kColBlock;
    double cm = INFINITY;
    int ci = 0;
    if (c < NC) {
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const double v = sFold[c * kWave + part * PER + k];
        const int vi = sFoldI[c * kWave + part * PER + k];
        if (v < cm || (v == cm && vi < ci)) {
          cm = v;
          ci = vi;
        }
      }
    }
#pragma unroll
    for (int mask = kColBlock; mask <= 32; mask <<= 1) {
      const double other = shfl_xor_f64(cm, mask);
      const int oi = __shfl_xor(ci, mask);
      const bool take = (other < cm) | ((other == cm) & (oi < ci));
      cm = take ? other : cm;
      ci = take ? oi : ci;
    }
    if (part == 0 && c < ncols) {
      const size_t slot = (size_t)crow * g.R + (size_t)(cj + c - g.os);      // crow: the (row block, sample part) slot row
      cpart_d2[slot] = cm;
      cpart_i[slot] = (cm == INFINITY) ? -1 : (CULL ? ci : g.ro + I * kRowBlock + ci);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
  }
}

// The rows of a call as [sample][xyz][row] (row pitch Rp; the rows behind R replay row R - 1): 64 x 64
// tiles through LDS, read along a drone's samples and written along the rows.
// Rows of workgroups behind the `ny` of the transposition mark the column-side partner slots of a part launch
// as empty (-1): a part covers its first and last row block only partly, and the merge skips empty slots.
// `perm` (cull path): row r of the image is drone perm[r] of prow, and `psorted` receives the same rows drone-major
// (the column array of the sorted pass), both from one read of the tile.
__global__ void __launch_bounds__(256)
collide_transpose_kernel(const double *__restrict__ prow, int R, int Rp, int E, double *__restrict__ prow_t, int ny,
                         int32_t *__restrict__ fill, size_t fill_n, const int32_t *__restrict__ perm,
                         double *__restrict__ psorted) {
  constexpr int TE = 32;      // elements of a drone per tile (x 64 rows): 2.4 workgroups per CU at 4096 x 91
  __shared__ double tile[64][TE + 1];
  if ((int)blockIdx.y >= ny) {
    const size_t stride = (size_t)(gridDim.y - ny) * gridDim.x * 256;
    for (size_t i = ((size_t)(blockIdx.y - ny) * gridDim.x + blockIdx.x) * 256 + threadIdx.x; i < fill_n; i += stride)
      fill[i] = -1;
    return;
  }
  const int r0 = blockIdx.x * 64, e0 = blockIdx.y * TE;
  {
    const int tx = threadIdx.x & (TE - 1), ty = threadIdx.x / TE;
#pragma unroll
    for (int i = ty; i < 64; i += 256 / TE) {
      const int r = min(r0 + i, R - 1), e = e0 + tx;
      const double v = e < E ? prow[(size_t)(perm ? perm[r] : r) * E + e] : 0.0;
      tile[i][tx] = v;
      if (psorted && e < E && r0 + i < R) psorted[(size_t)r * E + e] = v;
    }
  }
  __syncthreads();
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int i = ty; i < TE; i += 4) {
    const int e = e0 + i;
    if (e < E) prow_t[(size_t)e * Rp + r0 + tx] = tile[tx][i];
  }
}

__device__ __forceinline__ void collide_span_body(const double *__restrict__ prow_t, const double *__restrict__ pcol,
                                                  const CollideGeom &g, double *__restrict__ part_d2,
                                                  int32_t *__restrict__ part_j, double *__restrict__ cpart_d2,
                                                  int32_t *__restrict__ cpart_i) {
  constexpr int CB = kColBlock;
  __shared__ double sFold[CB * kWave];
  __shared__ int sFoldI[CB * kWave];
  const int lane = threadIdx.x;
  const int w = blockIdx.x / g.sparts, h = blockIdx.x - w * g.sparts;
  long long u = collide_share_begin(g, w);
  const long long u_end = collide_share_begin(g, (long long)w + 1);
  if (u >= u_end) return;
  // the row block the share starts in
  int I = 0;
  while (I + 1 < g.n_rb && collide_ustart(g, I + 1) <= u) ++I;
  for (; u < u_end; ++I) {
    const long long ub = collide_ustart(g, I), un = collide_ustart(g, I + 1);
    const int ua = (int)(u - ub);                                         // first unit inside row block I
    const int ue = (int)((u_end < un ? u_end : un) - ub);                 // one past the last
    u = ub + ue;
    // unit -> column: the columns left of the own range, then from the row block's own first column on
    const int diag0 = g.os + I * kRowBlock;                               // only meaningful with g.sym
    const int skip = g.sym ? I * kRowBlock : 0;                           // own-range columns not met
    RowSet rs;
#pragma unroll
    for (int rr = 0; rr < kRowsPerLane; ++rr) {
      const int raw = I * kRowBlock + rr * kWave + lane;
      rs.live[rr] = raw < g.R;
      rs.grow[rr] = g.ro + (rs.live[rr] ? raw : g.R - 1);
      rs.best[rr] = INFINITY;
      rs.bestj[rr] = -1;
    }
    const double *prowT = prow_t + (size_t)(I - g.I_lo) * kRowBlock;
    const int crow = (I - g.I_lo) * g.sparts + h;
    for (int ux = ua; ux < ue;) {
      // a block: up to CB consecutive columns that do not straddle a boundary of the line
      const int cj = (g.sym && ux >= g.os) ? ux + skip : ux;
      int lim = ue - ux;                                                  // columns left in the share
      bool two_sided = false;
      if (g.sym) {
        if (cj < g.os) lim = lim < g.os - cj ? lim : g.os - cj;                               // left of the own range
        else if (cj < diag0 + kRowBlock) lim = lim < diag0 + kRowBlock - cj ? lim : diag0 + kRowBlock - cj;   // diagonal block
        else if (cj < g.oe) {                                                                // behind it, still own
          lim = lim < g.oe - cj ? lim : g.oe - cj;
          two_sided = true;
        }
      }
      const int ncols = lim < CB ? lim : CB;
      ux += ncols;
      if (ncols <= 2)
        collide_block<2>(g, prowT, pcol, cj, ncols, two_sided, rs, I, h, crow, lane, sFold, sFoldI, cpart_d2, cpart_i);
      else if (ncols <= kColBlock / 2)
        collide_block<kColBlock / 2>(g, prowT, pcol, cj, ncols, two_sided, rs, I, h, crow, lane, sFold, sFoldI, cpart_d2, cpart_i);
      else
        collide_block<kColBlock>(g, prowT, pcol, cj, ncols, two_sided, rs, I, h, crow, lane, sFold, sFoldI, cpart_d2, cpart_i);
    }
    // one partial entry per (wave, row block): w + I is unique (a later wave starts in a later or the same
    // row block) and the entries of row block I are the contiguous ids of the waves that meet it
    const size_t id = ((size_t)w + (I - g.I_lo)) * g.sparts + h;
#pragma unroll
    for (int rr = 0; rr < kRowsPerLane; ++rr) {
      part_d2[id * kRowBlock + rr * kWave + lane] = rs.best[rr];
      part_j[id * kRowBlock + rr * kWave + lane] = (rs.best[rr] == INFINITY) ? -1 : rs.bestj[rr];
    }
  }
}

__global__ void __launch_bounds__(kWave, 4)
collide_span_kernel(const double *__restrict__ prow_t, const double *__restrict__ pcol, CollideGeom g,
                    double *__restrict__ part_d2, int32_t *__restrict__ part_j, double *__restrict__ cpart_d2,
                    int32_t *__restrict__ cpart_i) {
  collide_span_body(prow_t, pcol, g, part_d2, part_j, cpart_d2, cpart_i);
}

// ---- the whole-swarm pass with the exact broad phase (CollideCull) ----
// Sorted order; row block I (rows 128 I ..) meets the columns from its own first one on, in aligned groups of 8:
// share k of row block I is the columns 128 I + 8 k .. (+ 8); its first 16 shares are the diagonal block (one-sided
// among the block's own rows, never culled), the others are two-sided and culled by the box test.

// Sample parts per surviving share.  A share is kSampleChunk-sample chunks of VALU work (1.7 us of SIMD time each
// at 8 columns x 128 rows) plus some 1.5 us of its own per item -- list entry, first row loads, the fold; the items are
// spread over the SIMDs, which run their waves one instruction at a time: the estimate is the number of items a SIMD
// works off times the length of one, and the part count with the smallest estimate wins (a part keeps two chunks).
// The fixture's 1393 survivors take 95-96 us per pass with 2, 3 or 4 parts alike (1: 106); cutting some shares into one
// part more than the others so that the items are exactly the wave slots changed nothing there and cost a sparse
// swarm 10 us (its uncut shares become the long poles) -- CullSplit keeps that form, x = 0.
constexpr int kCullMaxParts = 8;
constexpr int kCullMaxDrones = 16384;      // largest whole swarm that takes the broad phase (the rank count's words per lane)
constexpr int kCullMinDrones = 3072;       // smallest, by default ("collide_cull_min_drones")
constexpr int kCullGroupMaxDrones = 8192;  // largest whose group pairs are all given a list slot (524 800): the group evaluator
constexpr int kGroupCapLarge = 1 << 18;    // list slots of larger swarms (16 384 drones have 2.1 M group pairs): the group
                                           // evaluator runs while the survivors fit, the share evaluator behind it otherwise
struct CullSplit {
  int lo, hi, x;      // x shares in `hi` parts, the others in `lo`
  __device__ __host__ __forceinline__ int items(int tot) const { return x * hi + (tot - x) * lo; }
  // first item of list position f
  __device__ __host__ __forceinline__ int item_of(int f) const { return f <= x ? f * hi : x * hi + (f - x) * lo; }
};
// `force`: bits 0..7 a part count to use whatever the estimate says (0: none), bits 8.. the largest part count the
// column-side slots of this launch were laid out for (0: kCullMaxParts)
__device__ __host__ __forceinline__ CullSplit cull_split(int tot, int slots, int nch, int force) {
  CullSplit c;
  c.x = 0;
  int maxp = (force >> 8) > 0 && (force >> 8) < kCullMaxParts ? (force >> 8) : kCullMaxParts;
  force &= 0xff;
  if (force > 0) {
    c.lo = c.hi = force < maxp ? force : maxp;
    return c;
  }
  // the row-side entries of a launch are laid out for max(shares, 2 x slots) items (launch_formation_collide): a share
  // is cut only while the items stay below twice the wave slots
  const int fit = tot > 0 ? 2 * slots / tot : maxp;
  maxp = fit < maxp ? (fit > 1 ? fit : 1) : maxp;
  const int simds = slots / 4 > 0 ? slots / 4 : 1;
  int best = 1;
  long long best_cost = -1;
  for (int sp = 1; sp <= maxp && (sp == 1 || nch / sp >= 2); ++sp) {
    const long long serial = ((long long)tot * sp + simds - 1) / simds;
    const long long cost = serial * (17 * ((nch + sp - 1) / sp) + 15);      // 0.1 us
    if (best_cost < 0 || cost < best_cost) {
      best_cost = cost;
      best = sp;
    }
  }
  c.lo = c.hi = best;
  return c;
}
__device__ __host__ __forceinline__ int cull_nch(int S) {
  const int rem = S % kSampleChunk;
  const int Sw = (rem == 1 || rem == 2) ? S - rem : S;
  return (Sw + kSampleChunk - 1) / kSampleChunk;
}

// what the broad-phase kernels hand each other (int32 words in device memory)
enum : int {
  kMetaStart = 0,                           // [n_rb <= 128] first list position of row block I's survivors
  kMetaTotal = MSNAP_COLLIDE_META_SHARES,   // surviving shares (the list's length; zeroed by collide_key_kernel, reserved atomically by the selection)
  kMetaParts = kMetaTotal + 1,              // CullSplit lo, hi, x (the share evaluator, for the merge)
  kMetaGroups = MSNAP_COLLIDE_META_GROUPS,  // surviving GROUP PAIRS (zeroed and reserved like kMetaTotal)
  kMetaWords = kMetaGroups + 2
};

// The second granularity of the broad phase: pairs of GROUPS (8 x 8 drones of the sorted order).  Of a surviving share
// (128 rows x 8 columns) usually one or two of its 16 row groups are what kept it; the group pairs that pass the same
// test are a few per cent of all (fixture: 3919 of 131 328, 0.25 M pairs against the surviving shares' 1.43 M).  They
// are evaluated by collide_eval_groups_kernel with the samples across the lanes, every item leaving 16 candidates -- one
// per row drone and one per column drone.  The selection leaves, besides the list itself, what lets
// collide_finish_groups_kernel fold a group's candidates without searching: the items of group g as the row side are the
// contiguous list range [astart[g], + acnt[g]) (the diagonal item (g, g) included), the items (a, g) with g as the column
// side are the non-zero entries of blist[g][a < g] = list position + 1.
struct CullGroups {
  int32_t *glist;                 // [cap] (a << 16 | b), a <= b: surviving group pairs, a-major, ascending b per a
  int32_t *astart, *acnt;         // [nG] list range of the items (g, b >= g)
  int32_t *blist;                 // [nG][nG]: list position + 1 of the item (a, g) at [g][a] (0: none; zeroed per pass)
  double *cand_d2;                // [cap][16] what item `it` found for its 8 row drones and its 8 column drones
  int32_t *cand_j;
  int cap;                        // list capacity (all group pairs when this path is taken: it cannot overflow)
  int nG;
};
// Same arithmetic per pair and sample on both paths; the group kernel spends about 2.2 x as many vector instructions per
// pair-sample (91 samples on 128 lanes, the cross-lane folds), the share kernel runs a single short round at 0.7 of its
// pace: groups when 64 x 2.2 x (group pairs) < 1024 x 1.45 x (shares).
__device__ __host__ __forceinline__ bool cull_groups_cheaper(long long shares, long long groups) {
  return groups * 141 < shares * 1485;
}
// what a pass leaves for the next one's choice of evaluator (one 64-bit word in page-locked host memory, written by
// the evaluator's first wave: the host reads it without synchronising): swarm size, surviving shares, surviving
// group pairs
__device__ __host__ __forceinline__ unsigned long long cull_hint_pack(int N, int shares, long long groups) {
  return ((unsigned long long)(N & 0x7fff) << 48) | ((unsigned long long)(shares & 0xffffff) << 24) |
         (unsigned long long)(groups < 0xffffff ? groups : 0xffffff);
}

// One workgroup per row block: its threads test the shares (at most 1024 of them: kCullMaxDrones / 8); the survivors are
// written, in ascending order, to a range of the list that the workgroup reserves with one atomic add -- the row
// blocks' ranges come in any order, each is contiguous: list[start[I] .. + cnt[I]) = (I << 16 | k).
// (1024 threads: with 256 a wavefront sat alone on its SIMD and walked 32 dependent box tests -- 12 us of an 8-launch
// pass's shortest chain; sixteen wavefronts per workgroup hide each other's latencies)
constexpr int kSelThreads = 1024;
constexpr int kSelGroups = 8;      // group-pair selection: a's per workgroup
// kSelTrips (template parameter): trips of a workgroup over a row block's shares / a group's partners: 1 up to 8192
// drones, 2 up to kCullMaxDrones (16 384 / 8 = 2 x kSelThreads)
template <int kSelTrips>
__global__ void __launch_bounds__(kSelThreads)
collide_select_kernel(int N, int n_rb, CollideCull cu, const double *__restrict__ bound, int32_t *__restrict__ list,
                      int32_t *__restrict__ cnt, int32_t *__restrict__ meta, CullGroups cg) {
  // Every workgroup first stages what all its tests read: the largest bound of every group of 8 sorted drones (the
  // bounds were finished by the previous launch: atomic minima of the gather tiles over the sorted neighbours; 0 for a
  // drone without a finite sample) and -- swarms up to 8192 drones, 56 KB -- the groups' boxes: one memory round trip,
  // then the box tests run out of LDS.
  constexpr bool kStage = kSelTrips == 1;
  constexpr int kStageGroups = kStage ? kCullGroupMaxDrones / kColBlock : 1;
  __shared__ double sCmax[kCullMaxDrones / kColBlock];
  __shared__ double sBox[kStageGroups * 6];
  {
    const int nG = (N + kColBlock - 1) / kColBlock;
    for (int gq = threadIdx.x; gq < nG; gq += kSelThreads) {
      double m = 0.0;
#pragma unroll
      for (int d = 0; d < kColBlock; ++d) {
        const int r = gq * kColBlock + d;
        m = fmax(m, bound[r < N ? r : N - 1]);
      }
      sCmax[gq] = m;
    }
    if constexpr (kStage) {      // (one flight: six loads per thread at most)
      double bv[kStageGroups * 6 / kSelThreads];
#pragma unroll
      for (int u = 0; u < kStageGroups * 6 / kSelThreads; ++u) {
        const int e = threadIdx.x + u * kSelThreads;
        bv[u] = e < nG * 6 ? cu.colbox[e] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < kStageGroups * 6 / kSelThreads; ++u) {
        const int e = threadIdx.x + u * kSelThreads;
        if (e < nG * 6) sBox[e] = bv[u];
      }
    }
    __syncthreads();
  }
  auto load_box = [&](int gq, double(&B)[6]) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      if constexpr (kStage) B[k] = sBox[gq * 6 + k];
      else B[k] = cu.colbox[(size_t)gq * 6 + k];
    }
  };
  if ((int)blockIdx.x >= n_rb) {
    // group pairs (a, b), a <= b: a workgroup takes kSelGroups a's and tests each against every b (the same strict test
    // on the two groups' boxes and bounds); its survivors are appended with ONE atomic reservation -- a reservation per
    // a was 512 atomics on one word, 20 ns apiece -- the workgroups in no particular order
    constexpr int NW = kSelThreads / kWave, NS = kSelGroups * kSelTrips;
    __shared__ int gsum[NS * NW + 1];
    __shared__ int gbase, gtot;
    const int a0 = (blockIdx.x - n_rb) * kSelGroups, nG = (N + kColBlock - 1) / kColBlock;
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    unsigned long long kept = 0;      // bit (ai * kSelTrips + t): pair (a0 + ai, a + t * kSelThreads + thread) survives
#pragma unroll
    for (int ai = 0; ai < kSelGroups; ++ai) {
      const int a = a0 + ai;
      double A[7], A6[6];
      load_box(a < nG ? a : nG - 1, A6);
#pragma unroll
      for (int k = 0; k < 6; ++k) A[k] = A6[k];
      A[6] = sCmax[a < nG ? a : nG - 1];
#pragma unroll
      for (int t = 0; t < kSelTrips; ++t) {      // (at most kCullMaxDrones / 8 groups)
        bool keep = false;
        if (a + t * kSelThreads < nG) {      // (wave-uniform)
          const int b = a + t * kSelThreads + threadIdx.x, bc = b < nG ? b : nG - 1;
          double B6[6];
          load_box(bc, B6);
          const double lb2 = box_box_lb2(A, B6);
          keep = b < nG && (b == a || !((lb2 > A[6]) & (lb2 > sCmax[bc])));
        }
        const unsigned long long m = __ballot(keep);
        kept |= keep ? 1ull << (ai * kSelTrips + t) : 0ull;
        if (lane == 0) gsum[(ai * kSelTrips + t) * NW + w] = __popcll(m);
      }
    }
    __syncthreads();
    // exclusive scan of the NS x NW wave counts (two wavefronts' worth)
    if (threadIdx.x < kWave) {
      int run = 0;
      for (int i0 = 0; i0 < NS * NW; i0 += kWave) {
        const int v = i0 + lane < NS * NW ? gsum[i0 + lane] : 0;
        int inc = v;
#pragma unroll
        for (int m = 1; m < kWave; m <<= 1) {
          const int o = __shfl_up(inc, m);
          inc += lane >= m ? o : 0;
        }
        if (i0 + lane < NS * NW) gsum[i0 + lane] = run + inc - v;
        run += __shfl(inc, kWave - 1);
      }
      if (lane == 0) {
        gbase = atomicAdd(&meta[kMetaGroups], run);
        gtot = run;
      }
    }
    __syncthreads();
    // the survivors of one a are contiguous (a-major order): its row-side range for the evaluator's finish
    if (threadIdx.x < kSelGroups && a0 + (int)threadIdx.x < nG) {
      const int ai = threadIdx.x;
      const int s0 = gsum[ai * kSelTrips * NW], s1 = ai + 1 < kSelGroups ? gsum[(ai + 1) * kSelTrips * NW] : gtot;
      cg.astart[a0 + ai] = gbase + s0;
      cg.acnt[a0 + ai] = s1 - s0;
    }
#pragma unroll
    for (int ai = 0; ai < kSelGroups; ++ai)
#pragma unroll
      for (int t = 0; t < kSelTrips; ++t) {
        const bool keep = (kept >> (ai * kSelTrips + t)) & 1ull;
        const unsigned long long m = __ballot(keep);
        const int pos = gbase + gsum[(ai * kSelTrips + t) * NW + w] + __popcll(m & ((1ull << lane) - 1ull));
        if (keep && pos < cg.cap) {
          const int a = a0 + ai, b = a + t * kSelThreads + threadIdx.x;
          cg.glist[pos] = (a << 16) | b;
          if (b != a) cg.blist[(size_t)b * cg.nG + a] = pos + 1;      // the column side's reverse list (0: none)
        }
      }
    return;
  }
  constexpr int GPB = kRowBlock / kColBlock;      // groups of 8 per row block
  constexpr int NW = kSelThreads / kWave;
  __shared__ int wsum[kSelTrips][NW];
  __shared__ int sStart;
  __shared__ double sRow[GPB][8];      // the row block's own groups: box, largest bound
  __shared__ double sAll[8];           // their union, the largest of their bounds
  const int I = blockIdx.x, lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
  const int nsh = (N - I * kRowBlock + kColBlock - 1) / kColBlock;      // <= kSelTrips * kSelThreads (kCullMaxDrones)
  const int ng = nsh < GPB ? nsh : GPB;
  if (threadIdx.x < 7 * GPB) {
    const int q = threadIdx.x / 7, k = threadIdx.x % 7;
    if (q < ng) sRow[q][k] = k < 6 ? (kStage ? sBox[(I * GPB + q) * 6 + k] : cu.colbox[(size_t)(I * GPB + q) * 6 + k]) : sCmax[I * GPB + q];
  }
  __syncthreads();
  // the union of the row block's groups: a share that fails against it fails against every group (most do, and whole
  // wavefronts of them: the group-by-group test below is 16 box distances per share)
  if (threadIdx.x < 7) {
    const int k = threadIdx.x;
    double v = sRow[0][k];
    for (int q = 1; q < ng; ++q) v = k < 3 ? fmin(v, sRow[q][k]) : fmax(v, sRow[q][k]);
    sAll[k] = v;
  }
  __syncthreads();
  unsigned long long mk[kSelTrips];
  bool keep[kSelTrips];
  // (the column groups' boxes of all trips are fetched before the first test: one memory round trip, not four)
  double cbx[kSelTrips][6], cmx[kSelTrips];
#pragma unroll
  for (int t = 0; t < kSelTrips; ++t) {
    if (t * kSelThreads >= nsh) continue;      // (workgroup-uniform)
    const int k = t * kSelThreads + threadIdx.x, J = I * GPB + (k < nsh ? k : nsh - 1);
    load_box(J, cbx[t]);
    cmx[t] = sCmax[J];
  }
#pragma unroll
  for (int t = 0; t < kSelTrips; ++t) {
    const int k = t * kSelThreads + threadIdx.x;
    keep[t] = false;
    if (k < nsh) {
      keep[t] = true;
      if (k >= GPB) {
        // Skip the share unless some pair of it could reach (or tie) a minimum of its row or its column: the rows are
        // taken group by group -- a block of 128 consecutive drones of the sorted order can straddle a jump of the
        // curve, its groups of 8 hardly ever do.
        const double cm = cmx[t];
        bool any = false;
        const double lb0 = box_box_lb2(sAll, cbx[t]);
        if (!((lb0 > sAll[6]) & (lb0 > cm))) {
          for (int q = 0; q < ng; ++q) {
            const double lb2 = box_box_lb2(sRow[q], cbx[t]);
            any |= !((lb2 > sRow[q][6]) & (lb2 > cm));
          }
        }
        keep[t] = any;
      }
    }
    mk[t] = __ballot(keep[t]);
    if (lane == 0) wsum[t][w] = __popcll(mk[t]);
  }
  __syncthreads();
  int all = 0, off[kSelTrips];
#pragma unroll
  for (int t = 0; t < kSelTrips; ++t) {
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      if (q == w) off[t] = all;      // (all: the survivors with a lower share number so far)
      all += wsum[t][q];
    }
  }
  if (threadIdx.x == 0) {
    const int start = atomicAdd(&meta[kMetaTotal], all);
    sStart = start;
    meta[kMetaStart + I] = start;
    cnt[I] = all;
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < kSelTrips; ++t)
    if (keep[t]) list[sStart + off[t] + __popcll(mk[t] & ((1ull << lane) - 1ull))] = (I << 16) | (t * kSelThreads + threadIdx.x);
}

// The surviving shares, walked by a fixed grid of waves: item it = (survivor it / sp, sample part it % sp) of the
// list; its row-side partial entry is entry `it` of part_d2 / part_j, the column side goes to the (row block,
// sample part) slots as in the plain pass.  The next item's list entry is fetched before the current item runs.
__device__ __forceinline__ void
collide_span_list_body(const double *__restrict__ prow_t, const double *__restrict__ pcol, CollideGeom g,
                       double *__restrict__ part_d2, int32_t *__restrict__ part_j, double *__restrict__ cpart_d2,
                       int32_t *__restrict__ cpart_i, const int32_t *__restrict__ oid, const int32_t *__restrict__ list,
                       int sp_force, int slots, int32_t *__restrict__ meta) {
  constexpr int CB = kColBlock;
  __shared__ double sFold[CB * kWave];
  __shared__ int sFoldI[CB * kWave];
  const int lane = threadIdx.x;
  const int tot = meta[kMetaTotal];
  const CullSplit sp = cull_split(tot, slots, cull_nch(g.S), sp_force);
  if (blockIdx.x == 0 && lane == 0) {
    meta[kMetaParts] = sp.lo;
    meta[kMetaParts + 1] = sp.hi;
    meta[kMetaParts + 2] = sp.x;
  }
  const int items = sp.items(tot);      // (at most 33280 shares x kCullMaxParts)
  const int xi = sp.x * sp.hi;          // items of the shares cut into `hi` parts
  auto share_of = [&](int it) { return it < xi ? it / sp.hi : sp.x + (it - xi) / sp.lo; };
  int it = blockIdx.x;
  int entry = it < items ? list[share_of(it)] : 0;
  while (it < items) {
    const int f = share_of(it);
    g.sparts = it < xi ? sp.hi : sp.lo;
    const int h = it - sp.item_of(f);
    const int I = entry >> 16, k = entry & 0xffff;
    const int nxt = it + gridDim.x;
    entry = nxt < items ? list[share_of(nxt)] : 0;
    const int crow = I * sp.hi + h;      // column-side slot row: `hi` rows per row block
    const int cj = I * kRowBlock + k * CB;
    const int ncols = g.Cn - cj < CB ? g.Cn - cj : CB;
    const bool two_sided = k >= kRowBlock / CB;
    RowSet rs;
#pragma unroll
    for (int rr = 0; rr < kRowsPerLane; ++rr) {
      const int raw = I * kRowBlock + rr * kWave + lane;
      rs.live[rr] = raw < g.R;
      rs.grow[rr] = rs.live[rr] ? raw : g.R - 1;
      rs.best[rr] = INFINITY;
      rs.bestj[rr] = -1;
    }
    const double *prowT = prow_t + (size_t)I * kRowBlock;
    if (ncols <= 2)
      collide_block<2, true>(g, prowT, pcol, cj, ncols, two_sided, rs, I, h, crow, lane, sFold, sFoldI, cpart_d2, cpart_i, oid);
    else if (ncols <= CB / 2)
      collide_block<CB / 2, true>(g, prowT, pcol, cj, ncols, two_sided, rs, I, h, crow, lane, sFold, sFoldI, cpart_d2, cpart_i, oid);
    else
      collide_block<CB, true>(g, prowT, pcol, cj, ncols, two_sided, rs, I, h, crow, lane, sFold, sFoldI, cpart_d2, cpart_i, oid);
#pragma unroll
    for (int rr = 0; rr < kRowsPerLane; ++rr) {
      part_d2[(size_t)it * kRowBlock + rr * kWave + lane] = rs.best[rr];
      part_j[(size_t)it * kRowBlock + rr * kWave + lane] = (rs.best[rr] == INFINITY) ? -1 : rs.bestj[rr];
    }
    it = nxt;
  }
}

// One wavefront per surviving group pair (a, b): 32 SAMPLES x the two halves of the column group across the lanes
// (lane = 32 half + sample; three trips for 65..96 samples), the 8 rows x 4 columns of the half in registers -- both
// drones' positions come as coalesced loads of the sorted drone-major copy, no scalar or LDS operand traffic.  After the
// samples a reduce-scatter butterfly inside each half (32 -> 16 -> ... -> 1 value per lane, halving the lane span each
// time) leaves pair (r, c) = ((lane >> 2) & 7, lane & 3) of the half in its lane; row-side (over c, then over the halves)
// and column-side (over r) candidates follow with lexicographic (distance, ORIGINAL index) folds and go to the item's
// 16 candidate slots.
constexpr int kGroupHalf = kColBlock / 2;
constexpr int kGroupLanes = kWave / 2;  // samples per trip

// The pass's last launch on the group path: one wavefront per group of 8 sorted drones folds the candidates of all the
// group's items -- row-side slots of the items (g, b) (a contiguous list range), column-side slots of the items (a, g)
// (the non-zero entries of the group's reverse-list row, fetched in one flight and compacted through LDS) -- a lane per
// item, then across the lanes drone by drone: the minimum, and the lowest ORIGINAL partner index among the lanes that hold
// it (as in the all-pairs pass); distance, partner and hit leave through the sort permutation.
// (Built first as the tail of the evaluator -- the wave completing a group's last item, found by an arrival counter,
// finished the group: 36 us where evaluator + this launch take 27.  Inside one launch the candidates cross XCDs through
// device-coherent stores the writer has to wait for, a returning atomic and coherent loads: five memory-side round trips
// of ~2 us behind every item, against one kernel boundary.)
constexpr int kFinishWaves = 4;      // groups (wavefronts) per workgroup
// kChunks (template parameter): 64-entry chunks of a reverse-list row: 8 up to 4096 drones, 16 up to 8192, 32 up to 16 384
template <int kChunks>
__global__ void __launch_bounds__(kWave * kFinishWaves)
collide_finish_groups_kernel(int N, const int32_t *__restrict__ oid, const int32_t *__restrict__ meta, CullGroups cg,
                             double radius, double *__restrict__ min_dist, int32_t *__restrict__ partner,
                             int32_t *__restrict__ hit) {
  __shared__ int sItAll[kFinishWaves][kChunks * kWave];
  const int lane = threadIdx.x & (kWave - 1), w = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  int *sIt = sItAll[w];
  const int gq = blockIdx.x * kFinishWaves + w;
  if (gq >= cg.nG || meta[kMetaGroups] > cg.cap) return;      // (more survivors than list slots: the share evaluator runs)
  int pl[kChunks];
#pragma unroll
  for (int c = 0; c < kChunks; ++c)
    pl[c] = (c * kWave < gq && c * kWave + lane < gq) ? cg.blist[(size_t)gq * cg.nG + c * kWave + lane] : 0;
  const int nA = cg.acnt[gq], sA = cg.astart[gq];
  // lane = 8 e + k: entry slot e, drone k of the group
  const int k = lane & (kColBlock - 1), e = lane >> 3;
  const int r = gq * kColBlock + k;
  const int out = r < N ? oid[r] : 0;      // (fetched under the candidates)
  int nB = 0;
#pragma unroll
  for (int c = 0; c < kChunks; ++c) {
    const unsigned long long m = __ballot(pl[c] > 0);      // (0 behind the group's own row)
    if (pl[c] > 0) sIt[nB + __popcll(m & ((1ull << lane) - 1ull))] = pl[c] - 1;
    nB += __popcll(m);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
  double best = INFINITY;
  int bj = -1;
  constexpr int U = 8;      // entries of a lane in flight: 64 items of the group per round (a round is a dependent round
                            // trip: configs[3] fixture 12.5 -> 11.2 us against U = 4; 16 gives no more)
  for (int q0 = 0; q0 < nA + nB; q0 += U * (kWave / kColBlock)) {
    double d[U];
    int j[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q = q0 + u * (kWave / kColBlock) + e;
      d[u] = INFINITY;
      j[u] = -1;
      if (q < nA + nB) {
        const size_t slot = (q < nA ? (size_t)(sA + q) * 16 : (size_t)sIt[q - nA] * 16 + kColBlock) + k;
        d[u] = cg.cand_d2[slot];
        j[u] = cg.cand_j[slot];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (j[u] >= 0 && (bj < 0 || d[u] < best || (d[u] == best && j[u] < bj))) {
        best = d[u];
        bj = j[u];
      }
  }
  // the eight entry lanes of a drone folded: lowest distance, then lowest ORIGINAL partner index
#pragma unroll
  for (int m = kColBlock; m < kWave; m <<= 1) {
    const double o = shfl_xor_f64(best, m);
    const int oj = __shfl_xor(bj, m);
    if (oj >= 0 && (bj < 0 || o < best || (o == best && oj < bj))) {
      best = o;
      bj = oj;
    }
  }
  if (e == 0 && r < N) {
    const double dist = sqrt(bj >= 0 ? best : INFINITY);
    min_dist[out] = dist;
    partner[out] = bj;
    hit[out] = (dist < 2.0 * radius) ? 1 : 0;
  }
  // the row is left clean for the next pass (here, not where it is read: the counter the candidate loads wait on
  // counts stores too, and they would wait for these)
#pragma unroll
  for (int c = 0; c < kChunks; ++c)
    if (pl[c] > 0) cg.blist[(size_t)gq * cg.nG + c * kWave + lane] = 0;
}

// The evaluator of the surviving group pairs: a fixed grid of waves walks the list, one item per wave slot at the
// fixture's 3919 survivors; collide_finish_groups_kernel follows.
constexpr int kGroupWaves = 4;      // waves per SIMD the group evaluator is built for (128 registers)
__global__ void __launch_bounds__(kWave, kGroupWaves)
collide_eval_groups_kernel(const double *__restrict__ pcol, int N, int S, const int32_t *__restrict__ oid,
                           const int32_t *__restrict__ meta, CullGroups cg, unsigned long long *__restrict__ hint) {
#pragma clang fp contract(off)
  const int lane = threadIdx.x;
  int tot = meta[kMetaGroups];
  if (blockIdx.x == 0 && lane == 0 && hint != nullptr)
    __hip_atomic_store(hint, cull_hint_pack(N, meta[kMetaTotal], tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (tot > cg.cap) tot = 0;      // (more survivors than list slots: the share evaluator behind this launch runs instead)
  const unsigned stride = (unsigned)S * 3u;
  // XCD-aware item order.  Blocks are dealt round-robin over the 8 XCDs (observed; a speed matter only), so the blocks
  // b, b + 8, ... share an XCD and its L2: they walk one contiguous eighth of the list.  The list is a-major -- the ~8
  // items of a group sit side by side and each reads the group's 8 drone rows (17 KB), their column groups are
  // neighbours in the sorted order -- so most of an item's 35 KB are then hits in its XCD's 4 MB L2; in list order the
  // eight items of a group went to eight XCDs and every one of them fetched the rows from beyond L2.
  const int per = (tot + 7) >> 3;
  auto item_of = [&](int v) { return (v >> 3) < per ? (v & 7) * per + (v >> 3) : tot; };      // (>= tot: none)
  // (the first list entry is fetched beside the survivor count, the next one under the current item)
  int v = blockIdx.x, it = item_of(v);
  int entry = (it < tot && it < cg.cap) ? cg.glist[it] : 0;
  for (; (v >> 3) < per; v += gridDim.x, it = item_of(v)) {
    const int it_next = item_of(v + (int)gridDim.x);
    const int ent_raw = entry;
    entry = it_next < tot ? cg.glist[it_next] : 0;
    if (it >= tot) continue;
    // (the entry is wave-uniform: as a scalar the drones' row bases are scalar too, and every load is base + one lane offset)
    const int ent = __builtin_amdgcn_readfirstlane(ent_raw);
    const int a = ent >> 16, b = ent & 0xffff;
    const bool diag = a == b;

    // (everything derived from the lane index is rebuilt per phase from an opaque copy: left visible, the values the
    // candidate folds need are computed up front and held through the sample loop, whose 128 registers are spoken for)
    int lane_s = threadIdx.x;
    asm volatile("" : "+v"(lane_s));
    const int half = lane_s >> 5, ls = lane_s & (kGroupLanes - 1);
    double acc[kColBlock * kGroupHalf];
#pragma unroll
    for (int p = 0; p < kColBlock * kGroupHalf; ++p) acc[p] = INFINITY;
#pragma unroll 1
    for (int s0 = 0; s0 < S; s0 += kGroupLanes) {
      const int sq = s0 + ls < S ? s0 + ls : S - 1;      // (a sample seen twice does not change a minimum)
      // one 32-bit lane offset for all loads of the trip: scalar drone base + zero-extended lane offset + immediate
      const unsigned voff = (unsigned)sq * 24u;
      double cx[kGroupHalf], cy[kGroupHalf], cz[kGroupHalf];
#pragma unroll
      for (int c = 0; c < kGroupHalf; ++c) {
        const int d = b * kColBlock + half * kGroupHalf + c;      // (a last group may be short: clamped, masked below)
        const double *p = reinterpret_cast<const double *>(
            reinterpret_cast<const char *>(pcol + (size_t)(d < N ? d : N - 1) * stride) + (size_t)voff);
        cx[c] = p[0];
        cy[c] = p[1];
        cz[c] = p[2];
      }
      // The row drones' samples are fetched two rows ahead of the arithmetic, through a ring of three register sets.
      // Left to the compiler every row was load, wait, 28 operations -- 24 dependent L2 round trips per item, the whole
      // of the kernel's 21 us; the compiler barrier pins each fetch in front of the arithmetic two rows earlier.
      double rw[3][3];
      auto fetch = [&](int r, double(&dst)[3]) {
        const int d = a * kColBlock + r;
        const double *p = reinterpret_cast<const double *>(
            reinterpret_cast<const char *>(pcol + (size_t)(d < N ? d : N - 1) * stride) + (size_t)voff);
        dst[0] = p[0];
        dst[1] = p[1];
        dst[2] = p[2];
      };
      fetch(0, rw[0]);
      fetch(1, rw[1]);
#pragma unroll
      for (int r = 0; r < kColBlock; ++r) {
        if (r + 2 < kColBlock) fetch(r + 2, rw[(r + 2) % 3]);
        asm volatile("" ::: "memory");
        const double x = rw[r % 3][0], y = rw[r % 3][1], z = rw[r % 3][2];
#pragma unroll
        for (int c = 0; c < kGroupHalf; ++c) {
          const double dx = cx[c] - x, dy = cy[c] - y, dz = cz[c] - z;
          const double d2 = __builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx));
          // (the minimum by hand, as in collide_block: fmin would canonicalise the accumulator with a v_max first --
          // an eighth instruction per pair and sample; a quiet NaN loses either way)
          asm("v_min_f64 %0, %1, %0" : "+v"(acc[r * kGroupHalf + c]) : "v"(d2));
        }
      }
    }
    // reduce-scatter over the 32 lanes of the half: 32 values per lane -> 1, pair p = lane & 31
#pragma unroll
    for (int n = 16, m = 16; n >= 1; n >>= 1, m >>= 1) {
      const bool up = (lane_s & m) != 0;
#pragma unroll
      for (int i = 0; i < n; ++i) {
        const double keep = up ? acc[i + n] : acc[i], send = up ? acc[i] : acc[i + n];
        acc[i] = __builtin_fmin(keep, __shfl_xor(send, m));
      }
    }
    double v = acc[0];
    // this lane's pair after the butterfly
    int lane_c = threadIdx.x;
    asm volatile("" : "+v"(lane_c));
    const int pr = (lane_c & (kGroupLanes - 1)) >> 2, pc = lane_c & (kGroupHalf - 1), half_c = lane_c >> 5;
    const int ra = a * kColBlock + pr, cb = b * kColBlock + half_c * kGroupHalf + pc;
    const int oi = oid[ra < N ? ra : N - 1], oj = oid[cb < N ? cb : N - 1];
    if (ra >= N || cb >= N || (diag && ra == cb)) v = INFINITY;
    // row side: over the 4 columns of the half (lane bits 0, 1), then over the halves (bit 5)
    {
      double w = v;
      int wj = oj;
#pragma unroll
      for (int m = 1; m <= 32; m = (m == 2 ? 32 : m << 1)) {
        const double o = __shfl_xor(w, m);
        const int ojx = __shfl_xor(wj, m);
        const bool take = (o < w) | ((o == w) & (ojx < wj));
        w = take ? o : w;
        wj = take ? ojx : wj;
      }
      if ((lane & 35) == 0) {      // half 0, pc == 0: lane = 4 pr
        cg.cand_d2[(size_t)it * 16 + pr] = w;
        cg.cand_j[(size_t)it * 16 + pr] = w < INFINITY ? wj : -1;
      }
    }
    // column side: over the 8 rows (lane bits 2..4); the diagonal group's columns are its rows
    {
      double w = v;
      int wi = oi;
#pragma unroll
      for (int m = 4; m <= 16; m <<= 1) {
        const double o = __shfl_xor(w, m);
        const int oix = __shfl_xor(wi, m);
        const bool take = (o < w) | ((o == w) & (oix < wi));
        w = take ? o : w;
        wi = take ? oix : wi;
      }
      if ((lane_c & (kGroupLanes - 1)) < kGroupHalf) {      // pr == 0: lane = 32 half + pc
        const int k = kColBlock + half_c * kGroupHalf + pc;
        cg.cand_d2[(size_t)it * 16 + k] = w;
        cg.cand_j[(size_t)it * 16 + k] = (!diag && w < INFINITY) ? wi : -1;
      }
    }
  }
}

// The evaluator of the surviving shares (collide_span_list_body above); collide_merge_kernel follows it.
__global__ void __launch_bounds__(kWave, 4)
collide_eval_shares_kernel(const double *__restrict__ prow_t, const double *__restrict__ pcol, CollideGeom g,
                           double *__restrict__ part_d2, int32_t *__restrict__ part_j, double *__restrict__ cpart_d2,
                           int32_t *__restrict__ cpart_i, const int32_t *__restrict__ oid, const int32_t *__restrict__ list,
                           int sp_force, int slots, int32_t *__restrict__ meta, unsigned long long *__restrict__ hint,
                           int groups_cap) {
  // (launched behind the group evaluator of a large swarm: only if the survivors did not fit its list)
  if (groups_cap > 0 && meta[kMetaGroups] <= groups_cap) return;
  if (blockIdx.x == 0 && threadIdx.x == 0 && hint != nullptr)
    __hip_atomic_store(hint, cull_hint_pack(g.R, meta[kMetaTotal], meta[kMetaGroups]), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  collide_span_list_body(prow_t, pcol, g, part_d2, part_j, cpart_d2, cpart_i, oid, list, sp_force, slots, meta);
}

// paths shorter than one sample chunk: plain loops, one-sided
__global__ void __launch_bounds__(kWave)
collide_short_kernel(const double *__restrict__ prow, const double *__restrict__ pcol, int R, int ro, int Cn, int S,
                     double radius, double *__restrict__ min_dist, int32_t *__restrict__ partner,
                     int32_t *__restrict__ hit, int raw) {
#pragma clang fp contract(off)
  const int r = blockIdx.x * kWave + threadIdx.x;
  if (r >= R) return;
  const int grow = ro + r;
  const double *pr = prow + (size_t)r * S * 3;
  double best = INFINITY;
  int bestj = -1;
  for (int j = 0; j < Cn; ++j) {
    const double *pc = pcol + (size_t)j * S * 3;
    double m = INFINITY;
    for (int sq = 0; sq < S; ++sq) {
      const double dx = pc[(size_t)sq * 3 + 0] - pr[(size_t)sq * 3 + 0];
      const double dy = pc[(size_t)sq * 3 + 1] - pr[(size_t)sq * 3 + 1];
      const double dz = pc[(size_t)sq * 3 + 2] - pr[(size_t)sq * 3 + 2];
      const double d2 = __builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx));
      m = __builtin_fmin(d2, m);
    }
    if (j == grow) m = INFINITY;
    if (m < best) {
      best = m;
      bestj = j;
    }
  }
  if (raw) {      // a part of a pass (msnap_formation_collide_part): squared minimum, no hit flag
    min_dist[r] = best;
    partner[r] = (best == INFINITY) ? -1 : bestj;
    return;
  }
  const double dist = sqrt(best);
  min_dist[r] = dist;
  partner[r] = (best == INFINITY) ? -1 : bestj;
  hit[r] = (dist < 2.0 * radius) ? 1 : 0;
}

// One workgroup per kMergeRows rows, kMergeParts sub-groups: sub-group q sweeps every kMergeParts-th partial
// entry of its rows (a drone has a few hundred of them: swept by one thread the kernel is 28 us of dependent
// loads at 4096 drones; with 64 rows per workgroup only 64 of the 256 CUs had work), the sub-groups'
// candidates are folded through LDS.  16 consecutive rows are 128 contiguous bytes of a partial entry.
// (64 sub-groups with 8 loads per round: 12 us instead of 9 -- the sequential fold and the big workgroups cost
// more than the shorter sweeps save.)
constexpr int kMergeRows = 16;
constexpr int kMergeParts = 16;
__global__ void __launch_bounds__(kMergeRows * kMergeParts)
collide_merge_kernel(const double *__restrict__ part_d2, const int32_t *__restrict__ part_j, CollideGeom g,
                     const double *__restrict__ cpart_d2, const int32_t *__restrict__ cpart_i, double radius,
                     double *__restrict__ min_dist, int32_t *__restrict__ partner, int32_t *__restrict__ hit,
                     const int32_t *__restrict__ oid, const int32_t *__restrict__ cnt, const int32_t *__restrict__ meta,
                     int groups_cap = 0) {
  if (groups_cap > 0 && meta[kMetaGroups] <= groups_cap) return;      // (the group evaluator's fold wrote the results)
  __shared__ double sD[kMergeParts][kMergeRows];
  __shared__ int sJ[kMergeParts][kMergeRows];
  const int lr = threadIdx.x & (kMergeRows - 1), q = threadIdx.x / kMergeRows;
  const int r_raw = blockIdx.x * kMergeRows + lr;
  const int r = r_raw < g.R ? r_raw : g.R - 1;
  const int out = oid ? oid[r] : r;      // cull path: row r of the sorted pass is drone oid[r] (fetched under the sweeps)
  double best = INFINITY;
  int bj = -1;
  constexpr int U = 8;      // entries fetched per round: the loads of a round are independent
  auto sweep = [&](const double *pd, const int32_t *pj, size_t pitch, int n) {
    for (int s0 = q; s0 < n; s0 += U * kMergeParts) {
      double v[U];
      int j[U];
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int sidx = s0 + k * kMergeParts;
        const int sc = sidx < n ? sidx : n - 1;
        v[k] = pd[(size_t)sc * pitch];
        j[k] = sidx < n ? pj[(size_t)sc * pitch] : -1;
      }
#pragma unroll
      for (int k = 0; k < U; ++k) {
        if (j[k] >= 0 && (v[k] < best || (v[k] == best && j[k] < bj))) {
          best = v[k];
          bj = j[k];
        }
      }
    }
  };
  if (cnt) {
    // broad-phase pass through the surviving shares (collide_span_list_body): they are consecutive entries per row block
    const int I = __builtin_amdgcn_readfirstlane(r / kRowBlock);      // (kMergeRows divides kRowBlock)
    CullSplit sp;
    sp.lo = meta[kMetaParts];
    sp.hi = meta[kMetaParts + 1];
    sp.x = meta[kMetaParts + 2];
    const int f0 = meta[kMetaStart + I], first_item = sp.item_of(f0), n_items = sp.item_of(f0 + cnt[I]) - first_item;
    const size_t first = (size_t)first_item * kRowBlock + (r - I * kRowBlock);
    sweep(part_d2 + first, part_j + first, kRowBlock, n_items);
    if (I > 0) sweep(cpart_d2 + r, cpart_i + r, (size_t)g.R, I * sp.hi);
  } else if (g.total > 0) {
    // row side: the shares of this launch that met this drone's row block
    const int I = r / kRowBlock;
    long long ua = collide_ustart(g, I), ub = collide_ustart(g, I + 1) - 1;      // the row block's units ...
    ua = ua < g.u_lo ? g.u_lo : ua;                                                // ... that this launch walks
    ub = ub >= g.u_lo + g.u_n ? g.u_lo + g.u_n - 1 : ub;
    if (ub >= ua) {
      const long long wf = collide_share_of(g, ua), wl = collide_share_of(g, ub);
      const size_t first = ((size_t)wf + (I - g.I_lo)) * g.sparts * kRowBlock + (r - I * kRowBlock);
      sweep(part_d2 + first, part_j + first, kRowBlock, (int)(wl - wf + 1) * g.sparts);
    }
    // column side: the row blocks of this launch before this drone's own (each left `sparts` entries per
    // column; a part launch leaves the slots it did not reach marked empty)
    const int Ib = I < g.I_hi + 1 ? I : g.I_hi + 1;
    if (g.sym && Ib > g.I_lo) sweep(cpart_d2 + r, cpart_i + r, (size_t)g.R, (Ib - g.I_lo) * g.sparts);
  }
  sD[q][lr] = best;
  sJ[q][lr] = bj;
  __syncthreads();
  if (q == 0 && r_raw < g.R) {
#pragma unroll
    for (int k = 1; k < kMergeParts; ++k) {
      const double v = sD[k][lr];
      const int j = sJ[k][lr];
      if (j >= 0 && (v < best || (v == best && j < bj))) {
        best = v;
        bj = j;
      }
    }
    if (g.part) {     // one part of a pass: squared minima, folded over the parts by collide_finish_kernel
      min_dist[r] = best;
      partner[r] = bj;
    } else {
      const double dist = sqrt(best);
      min_dist[out] = dist;
      partner[out] = bj;
      hit[out] = (dist < 2.0 * radius) ? 1 : 0;
    }
  }
}

// The parts of a pass folded: minimum over the parts' squared minima (lowest partner wins a tie), distance, hit.
__global__ void __launch_bounds__(256)
collide_finish_kernel(const unsigned char *__restrict__ parts, size_t part_stride, int n_parts, int N, int row_offset,
                      int n_rows, double radius, double *__restrict__ min_dist, int32_t *__restrict__ partner,
                      int32_t *__restrict__ hit) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_rows) return;
  const int r = row_offset + i;
  double best = INFINITY;
  int bj = -1;
  for (int p = 0; p < n_parts; ++p) {
    const unsigned char *base = parts + (size_t)p * part_stride;
    const double v = reinterpret_cast<const double *>(base)[r];
    const int j = reinterpret_cast<const int32_t *>(base + (size_t)N * sizeof(double))[r];
    if (j >= 0 && (v < best || (v == best && j < bj))) {
      best = v;
      bj = j;
    }
  }
  const double dist = sqrt(best);
  min_dist[i] = dist;
  partner[i] = bj;
  hit[i] = (dist < 2.0 * radius) ? 1 : 0;
}

// rows [r0, r1) of a part's output marked empty (paths shorter than one sample chunk: the plain kernel fills its rows)
__global__ void __launch_bounds__(256) collide_part_clear_kernel(double *__restrict__ d2, int32_t *__restrict__ pj, int N) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < N) {
    d2[i] = INFINITY;
    pj[i] = -1;
  }
}

// fp64 min / max over the 64 lanes of a wave (every lane gets the result): four DPP stages inside
// the rows of 16 (lane xor 1, xor 2, mirror of 8, mirror of 16), two exchanges across the rows
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
template <bool MAX>
__device__ __forceinline__ double wave_minmax_f64(double v) {
  auto fold = [](double a, double b) { return MAX ? ((b > a) ? b : a) : ((b < a) ? b : a); };
  v = fold(v, dpp_f64<0xB1>(v));     // quad_perm [1,0,3,2]
  v = fold(v, dpp_f64<0x4E>(v));     // quad_perm [2,3,0,1]
  v = fold(v, dpp_f64<0x141>(v));    // row_half_mirror
  v = fold(v, dpp_f64<0x140>(v));    // row_mirror
  v = fold(v, __shfl_xor(v, 16));
  v = fold(v, __shfl_xor(v, 32));
  return v;
}
__device__ __forceinline__ double wave_min_f64(double v) { return wave_minmax_f64<false>(v); }
__device__ __forceinline__ double wave_max_f64(double v) { return wave_minmax_f64<true>(v); }
// The same butterflies with IEEE minNum / maxNum: a NaN operand is ignored, so a wave that mixes NaN and finite
// values ends with the extreme of the finite ones in EVERY lane (with the compare-and-select fold a lane holding
// NaN keeps it and its partner drops that subtree: the lanes would disagree); all-NaN stays NaN.
template <bool MAX>
__device__ __forceinline__ double wave_minmax_num_f64(double v) {
  auto fold = [](double a, double b) { return MAX ? __builtin_fmax(a, b) : __builtin_fmin(a, b); };
  v = fold(v, dpp_f64<0xB1>(v));
  v = fold(v, dpp_f64<0x4E>(v));
  v = fold(v, dpp_f64<0x141>(v));
  v = fold(v, dpp_f64<0x140>(v));
  v = fold(v, __shfl_xor(v, 16));
  v = fold(v, __shfl_xor(v, 32));
  return v;
}

// the value of lane 0 as a compiler-visible wave-uniform value (after a wave reduction every lane holds
// the same number, but only this makes the branches and triangle loads that depend on it scalar)
__device__ __forceinline__ double uniform_f64(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)),
                          __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// ------------------------------------------------------------------------------------
// The exact broad phase of a whole-swarm pass (CollideCull above): sort keys, sort, bounds and boxes.
// ------------------------------------------------------------------------------------

// per drone (one wavefront each, four to a workgroup: the drone's S x 3 doubles are one coalesced sweep): the box of its
// finite samples (lo = +inf, hi = -inf when it has none) and the sort key (drone_sort_key).  Only for callers whose
// positions do not come from this library's sampler -- msnap_sample_collide_device computes both while the samples sit
// in its LDS image.
__global__ void __launch_bounds__(kWave * kKeyDrones)
collide_key_kernel(const double *__restrict__ pos, int N, int S, double *__restrict__ box, unsigned *__restrict__ key) {
  const int lane = threadIdx.x & (kWave - 1);
  const int d = blockIdx.x * kKeyDrones + __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  if (d >= N) return;
  const double *p = pos + (size_t)d * S * 3;
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  const int E = S * 3;
  for (int e0 = 0; e0 < E; e0 += 3 * kWave) {
    // three elements per lane and trip, 64 apart: element e is coordinate e % 3, and 64 % 3 == 1
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int e = e0 + q * kWave + lane;
      const double v = e < E ? p[e] : __builtin_nan("");
      const int k = (lane + q) % 3;           // (e0 is a multiple of 3)
      if (__builtin_isfinite(v)) {
#pragma unroll
        for (int kk = 0; kk < 3; ++kk)
          if (k == kk) {
            lo[kk] = v < lo[kk] ? v : lo[kk];
            hi[kk] = v > hi[kk] ? v : hi[kk];
          }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    lo[k] = uniform_f64(wave_minmax_f64<false>(lo[k]));
    hi[k] = uniform_f64(wave_minmax_f64<true>(hi[k]));
  }
  if (lane < 3) {
    box[(size_t)d * 6 + lane] = lane == 0 ? lo[0] : lane == 1 ? lo[1] : lo[2];
    box[(size_t)d * 6 + 3 + lane] = lane == 0 ? hi[0] : lane == 1 ? hi[1] : hi[2];
  }
  if (lane == 0) key[d] = drone_sort_key(lo, hi);
}

// The sort, as a rank count spread over the chip: drones are ordered by (key, index) -- all distinct -- so the sorted
// position of drone i is the number of drones below it.  One workgroup of 16 wavefronts per 16 drones: the lanes hold
// the (key << 32 | index) words of ALL drones in registers (wavefront w the w-th sixteenth, up to 8 per lane), the 16
// drones of the tile are wave-uniform: one 64-bit compare per 64 pairs, the count is the population of its lane mask
// (scalar unit), the sixteen wavefronts' counts are added through LDS.  N^2 / 64 vector compares -- 260 k at 4096
// drones -- over 256 workgroups; counting in the lanes (a compare and an add-with-carry per pair, 64 drones per
// workgroup) took 8-10 us, a 78-stage bitonic network in one workgroup 36 us.
// perm[sorted position] = original index.  The kernel is also the pass's first launch: it zeroes the counters the later
// launches add to (survivor totals, per-group item counts and arrivals) and starts every drone's BOUND -- a squared
// distance the drone is known to attain, lowered by the gather tiles with atomic minima -- at +inf, or at 0 for a drone
// without a finite sample (its own result is +inf / -1 whatever is evaluated, it is invisible to the others, and it
// must not keep its group from being culled).
constexpr int kRankWaves = 16;
constexpr int kRankTile = 16;
// kRankKeys (template parameter): words per lane, N / 1024 rounded up to 4, 8 or 16 (kCullMaxDrones)
template <int kRankKeys>
__global__ void __launch_bounds__(kWave * kRankWaves)
collide_rank_kernel(const unsigned *__restrict__ key, int N, int32_t *__restrict__ perm, const double *__restrict__ box,
                    unsigned long long *__restrict__ bound, int32_t *__restrict__ zero, int n_zero,
                    int32_t *__restrict__ meta) {
  __shared__ int cnt[kRankWaves][kRankTile];
  {
    const int gid = blockIdx.x * (kWave * kRankWaves) + threadIdx.x;
    for (int z = gid; z < n_zero; z += gridDim.x * (kWave * kRankWaves)) zero[z] = 0;
    if (gid == 0) {      // (the selection adds its survivors)
      meta[kMetaTotal] = 0;
      meta[kMetaGroups] = 0;
    }
  }
  const int lane = threadIdx.x & (kWave - 1), w = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int nq = (N + kRankWaves * kWave - 1) / (kRankWaves * kWave);      // words per lane
  // All of a lane's words and the tile's 16 keys are ONE flight of loads (unconditional, clamped addresses; masked
  // afterwards): written with the bounds tests around the loads, every word and every tile key was its own load-and-
  // wait -- 20 dependent round trips, two thirds of the kernel's 7.5 us.  The tile keys travel as one vector load and
  // are handed out with readlane.
  const int i0 = blockIdx.x * kRankTile;
  unsigned kraw[kRankKeys];
#pragma unroll
  for (int q = 0; q < kRankKeys; ++q) {
    const int j = (w * nq + q) * kWave + lane;
    kraw[q] = key[(q < nq && j < N) ? j : 0];
  }
  const unsigned ktile = key[i0 + (lane & (kRankTile - 1)) < N ? i0 + (lane & (kRankTile - 1)) : N - 1];
  // (a tile drone's box travels in the same flight: it only decides where the drone's bound starts)
  double bx[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (threadIdx.x < kRankTile && i0 + (int)threadIdx.x < N) {
#pragma unroll
    for (int k = 0; k < 6; ++k) bx[k] = box[(size_t)(i0 + threadIdx.x) * 6 + k];
  }
  unsigned long long mine[kRankKeys];
#pragma unroll
  for (int q = 0; q < kRankKeys; ++q) {
    const int j = (w * nq + q) * kWave + lane;
    mine[q] = (q < nq && j < N) ? ((unsigned long long)kraw[q] << 32) | (unsigned)j : ~0ull;      // (~0: below nothing)
  }
#pragma unroll
  for (int t = 0; t < kRankTile; ++t) {
    const int i = i0 + t < N ? i0 + t : N - 1;      // wave-uniform
    const unsigned long long ki = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)ktile, t) << 32) | (unsigned)i;
    int c = 0;
#pragma unroll
    for (int q = 0; q < kRankKeys; ++q)
      if (q < nq) c += __popcll(__ballot(mine[q] < ki));
    if (lane == 0) cnt[w][t] = c;
  }
  lds_barrier();      // (the zeroing stores above need not have landed)
  if (threadIdx.x < kRankTile && i0 + (int)threadIdx.x < N) {
    int r = 0;
#pragma unroll
    for (int q = 0; q < kRankWaves; ++q) r += cnt[q][threadIdx.x];
    const int i = i0 + threadIdx.x;
    perm[r] = i;
    const bool has = (bx[0] <= bx[3]) & (bx[1] <= bx[4]) & (bx[2] <= bx[5]);
    bound[r] = has ? 0x7ff0000000000000ull : 0ull;
  }
}

// The pass's second launch, three kinds of workgroups behind the sort:
//  * TILES (64 sorted rows x 5 samples, through LDS, as collide_transpose_kernel): the sorted row image
//    [sample][xyz][row] and the sorted drone-major copy from one read of pos through the permutation -- and, while the
//    tile (with the 4 rows behind it) sits in LDS, the drones' BOUNDS: for sorted drone r the minimum over its sorted
//    neighbours r +- 1..4 and the tile's samples of the pass's own squared distance (fma(dz, dz, fma(dy, dy, dx dx));
//    non-finite samples never win), folded into bound[r] with an atomic minimum on the bit pattern (squared distances
//    are non-negative doubles: ordered like their patterns).  ANY subset of a drone's pair-samples bounds its final
//    minimum from above; with every tile contributing, all samples and the pairs across tile boundaries count.
//  * BOXES: per aligned group of 8 sorted drones the union of their path boxes (the group's bound is the maximum of
//    its drones' bounds, formed by the selection once the atomics are complete).
//  * FILL: the column-side partner slots of the share evaluator marked empty (share path only).
#ifndef MSNAP_TILE_E
#define MSNAP_TILE_E 15      // (tools: A/B builds; 4096 x 91: 33 -> 55.5 us per pass, 24 -> 54.1, 18 -> 53.1, 15 -> 52.7, 12 -> 52.4, 9 -> 52.6)
#endif
constexpr int kTileE = MSNAP_TILE_E, kTileRows = 64, kTileHalo = 4;
constexpr int kTilePitch = kTileE + 1 + (kTileE & 1);      // an odd number of doubles: conflict-free columns (33 -> 35)
static_assert(kTileE % 3 == 0, "whole samples per tile");
constexpr int kBoxGroups = 32;      // groups per BOXES workgroup: 8 lanes each
__global__ void __launch_bounds__(256)
collide_gather_kernel(const double *__restrict__ pos, int N, int Rp, int E, double *__restrict__ prow_t,
                      double *__restrict__ psorted, const int32_t *__restrict__ perm, unsigned long long *__restrict__ bound,
                      const double *__restrict__ box, double *__restrict__ colbox, int nx, int ny, int n_box,
                      int32_t *__restrict__ fill, size_t fill_n) {
#pragma clang fp contract(off)
  __shared__ double tile[(kTileRows + kTileHalo) * kTilePitch];
  __shared__ double sF[kTileHalo][kTileRows];
  const int tid = threadIdx.x;
  const int nb = (int)blockIdx.x - nx * ny;
  if (nb >= n_box) {
    const size_t stride = (size_t)(gridDim.x - nx * ny - n_box) * 256;
    for (size_t i = (size_t)(nb - n_box) * 256 + tid; i < fill_n; i += stride) fill[i] = -1;
    return;
  }
  if (nb >= 0) {
    const int gq = nb * kBoxGroups + (tid >> 3), r = gq * kColBlock + (tid & 7);
    double g6[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) g6[k] = k < 3 ? INFINITY : -INFINITY;
    if (r < N) {
      const double *bx = box + (size_t)perm[r] * 6;
#pragma unroll
      for (int k = 0; k < 6; ++k) g6[k] = bx[k];
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      auto fold = [k](double a, double c) { return k < 3 ? fmin(a, c) : fmax(a, c); };
      g6[k] = fold(g6[k], dpp_f64<0xB1>(g6[k]));      // lane xor 1
      g6[k] = fold(g6[k], dpp_f64<0x4E>(g6[k]));      // lane xor 2
      g6[k] = fold(g6[k], dpp_f64<0x141>(g6[k]));     // mirror inside the half-row of 8
    }
    if ((tid & 7) == 0 && gq * kColBlock < N) {
#pragma unroll
      for (int k = 0; k < 6; ++k) colbox[(size_t)gq * 6 + k] = g6[k];
    }
    return;
  }
  const int bx = blockIdx.x % nx, by = blockIdx.x / nx;
  const int r0 = bx * kTileRows, e0 = by * kTileE;
  const int ne = E - e0 < kTileE ? E - e0 : kTileE;      // whole samples: E and kTileE are multiples of 3
  // the tile's rows of the permutation first (one round trip), then every thread's elements in ONE flight: written as
  // a plain loop each element was permutation load, wait, position load, wait -- eight dependent round trips per thread
  __shared__ int sPerm[kTileRows + kTileHalo];
  if (tid < kTileRows + kTileHalo) sPerm[tid] = perm[r0 + tid < N ? r0 + tid : N - 1];
  __syncthreads();
  constexpr int kElems = (kTileRows + kTileHalo) * kTileE, kPer = (kElems + 255) / 256;
  double val[kPer];
#pragma unroll
  for (int u = 0; u < kPer; ++u) {
    const int idx = tid + u * 256, i = idx / kTileE, tx = idx - i * kTileE;
    val[u] = (idx < kElems && tx < ne) ? pos[(size_t)sPerm[i] * E + e0 + tx] : 0.0;
  }
#pragma unroll
  for (int u = 0; u < kPer; ++u) {
    const int idx = tid + u * 256, i = idx / kTileE, tx = idx - i * kTileE;
    if (idx < kElems) {
      tile[i * kTilePitch + tx] = val[u];
      if (i < kTileRows && tx < ne && r0 + i < N) psorted[(size_t)(r0 + i) * E + e0 + tx] = val[u];
    }
  }
  lds_barrier();      // (not __syncthreads(): that would sit out the round trip of the stores above)
  {
    const int tx = tid & 63, ty = tid >> 6;
    for (int i = ty; i < ne; i += 4) prow_t[(size_t)(e0 + i) * Rp + r0 + tx] = tile[tx * kTilePitch + i];
  }
  {
    // pair (row, row + k) over the tile's samples, one thread each
    const int row = tid & 63, k = (tid >> 6) + 1;
    const double *pa = tile + row * kTilePitch, *pb = tile + (row + k) * kTilePitch;
    double f = INFINITY;
    for (int q = 0; q < ne; q += 3) {
      const double dx = pb[q] - pa[q], dy = pb[q + 1] - pa[q + 1], dz = pb[q + 2] - pa[q + 2];
      f = __builtin_fmin(__builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx)), f);
    }
    sF[k - 1][row] = r0 + row + k < N ? f : INFINITY;      // (rows past the end replay row N - 1)
  }
  lds_barrier();
  if (tid < kTileRows + kTileHalo) {
    // row t of the tile (the halo rows too): its pairs with the rows after it and before it
    double b = INFINITY;
#pragma unroll
    for (int k = 1; k <= kTileHalo; ++k) {
      if (tid < kTileRows) b = __builtin_fmin(b, sF[k - 1][tid]);
      if (tid >= k && tid - k < kTileRows) b = __builtin_fmin(b, sF[k - 1][tid - k]);
    }
    if (b < INFINITY && r0 + tid < N) atomicMin(&bound[r0 + tid], (unsigned long long)__double_as_longlong(b));
  }
}

// what the cost model makes of a pass's survivor counts (the choice the NEXT pass of this swarm takes from the hint)
bool collide_counts_by_groups(const msnap_ctx *ctx, int n_drones, int shares_surviving, int group_pairs_surviving) {
  if (ctx->collide_cull_mode == 1) return false;
  if (n_drones > kCullGroupMaxDrones && 2LL * group_pairs_surviving > kGroupCapLarge) return false;
  if (ctx->collide_cull_mode == 2) return true;
  return cull_groups_cheaper(shares_surviving, group_pairs_surviving);
}

// whether a pass with these arguments runs behind the exact broad phase (which builds its own, spatially sorted, row
// image: a caller-provided one is then not read -- msnap_formation_collide_reads_rows_t)
bool formation_collide_takes_broad_phase(const msnap_ctx *ctx, int n_rows, int row_offset, int n_cols, int n_samples) {
  if (n_samples < kSampleChunk || n_rows != n_cols || row_offset != 0 || ctx->collide_no_cull || ctx->collide_no_sym) return false;
  const int cull_min = ctx->collide_cull_min_drones > 0 ? ctx->collide_cull_min_drones : kCullMinDrones;
  return n_rows >= cull_min && n_rows >= 2 * kRowBlock && n_rows <= kCullMaxDrones;
}

// `rows_t`: the rows' transposed image [n_samples][3][row pitch] when the caller already has it (the sampler's
// second output, msnap_sample_collide); nullptr: built here from pos_rows
int launch_formation_collide(msnap_ctx *ctx, int n_rows, int row_offset, int n_cols, int n_samples,
                             const double *pos_rows, const double *pos_cols, double radius, double *min_dist,
                             int32_t *partner, int32_t *hit, const double *rows_t_in) {
  CollideGeom g;
  g.R = n_rows;
  g.ro = row_offset;
  g.Cn = n_cols;
  g.S = n_samples;
  g.n_rb = (n_rows + kRowBlock - 1) / kRowBlock;
  g.Rp = g.n_rb * kRowBlock;
  g.sparts = 1;
  g.u_lo = 0;
  g.u_n = 0;
  g.I_lo = 0;
  g.I_hi = g.n_rb - 1;
  g.part = 0;
  if (n_cols == 0) {
    // nobody to collide with: the merge of nothing writes inf / -1 / 0
    g.os = g.oe = 0;
    g.sym = 0;
    g.upw = g.upw_tail = 1;
    g.split = 0;
    g.total = 0;
    hipLaunchKernelGGL(collide_merge_kernel, dim3((n_rows + kMergeRows - 1) / kMergeRows), dim3(kMergeRows * kMergeParts), 0, ctx->stream,
                       (const double *)nullptr, (const int32_t *)nullptr, g, (const double *)nullptr,
                       (const int32_t *)nullptr, radius, min_dist, partner, hit, (const int32_t *)nullptr,
                       (const int32_t *)nullptr, (const int32_t *)nullptr);
    MSNAP_HIP(ctx, hipGetLastError());
    return MSNAP_OK;
  }
  if (n_samples < kSampleChunk) {
    hipLaunchKernelGGL(collide_short_kernel, dim3((n_rows + kWave - 1) / kWave), dim3(kWave), 0, ctx->stream,
                       pos_rows, pos_cols, n_rows, row_offset, n_cols, n_samples, radius, min_dist, partner, hit, 0);
    MSNAP_HIP(ctx, hipGetLastError());
    return MSNAP_OK;
  }
  // the rows are the columns [row_offset, row_offset + n_rows) when that range exists: pairs inside it
  // are evaluated once (unless the column-side partial buffer would be unreasonable)
  const bool rows_in_cols = (long long)row_offset + n_rows <= n_cols;
  g.os = rows_in_cols ? row_offset : n_cols;
  g.oe = rows_in_cols ? row_offset + n_rows : n_cols;
  const size_t cpart_entries = (size_t)g.n_rb * n_rows;
  // (the column-side partial buffer is bounded at 2 GB: 170 k rows on one GPU; beyond that, and for callers whose
  // rows are not the slice of the columns -- "collide_no_sym" -- every pair of the range is evaluated from both
  // sides; msnap_get_option("collide_last_sym") reports which way the last pass went)
  g.sym = (rows_in_cols && n_rows > kRowBlock && !ctx->collide_no_sym && cpart_entries * 12 <= ((size_t)2 << 30)) ? 1 : 0;
  ctx->collide_last_sym = g.sym;
  g.upw = 1;
  g.total = collide_ustart(g, g.n_rb);
  g.u_n = g.total;
  // Equal contiguous shares of the line, one 8-column block (x 128 rows) each.  Many small shares beat one
  // share per resident wave: the dispatcher hands the next share to whichever SIMD frees a slot, which
  // evens out the speed differences between SIMDs; shares of 4 columns everywhere lose more to the
  // per-block prologue and fold than they gain (4096 x 91: 288 -> 340 us); a share that ends inside a
  // block pays for the whole block.  "collide_waves_per_cu" (tuning tools) asks for fewer, longer shares.
  long long upw = kColBlock;
  const long long slots = (long long)ctx->n_cu * 4 * 4;      // resident waves: 4 per SIMD
  if (ctx->collide_waves_per_cu > 0) {
    upw = g.total / ((long long)ctx->n_cu * ctx->collide_waves_per_cu) / kColBlock * kColBlock;
    if (upw < kColBlock) upw = kColBlock;
  }
  long long waves;
  // the row-side partial buffer holds one 128-row entry per (wave, row block) pair: bound it
  while (((g.total + upw - 1) / upw + g.n_rb) * kRowBlock * 12 > ((long long)512 << 20)) upw *= 2;
  g.upw = (int)upw;
  // A launch that does not even fill the wave slots once (4 per SIMD: <= 2730 drones on one GPU) sends the
  // last quarter of the line out in half-size shares, so that the SIMDs finish within half a share of each
  // other (2048 x 91: 90-95 -> 81-89 us; a shard's rows against all columns -- equal one-sided blocks -- lose by
  // it: 1024 of 4096 rows 109-117 -> 120-133 us, so whole swarms only).  With more shares than slots the dispatcher evens things out by
  // itself and the 4-column blocks only cost (3072: 144 -> 136 us, 4096: 230-250 -> 222-229, 6144: 477-505 -> 448).
  g.upw_tail = (int)upw;
  g.split = (g.total + upw - 1) / upw * upw;
  const long long shares = (g.total + upw - 1) / upw;
  if (upw == kColBlock && ctx->collide_waves_per_cu == 0 && shares <= slots && n_rows == n_cols) {
    g.upw_tail = kColBlock / 2;
    g.split = shares * 3 / 4 * upw;
  } else if (upw == kColBlock && ctx->collide_waves_per_cu == 0 && shares % slots != 0 && shares % slots <= slots / 8) {
    // a few shares more than whole rounds of the slots (4096 drones: 8448 on 4096): left whole they would run
    // as a last round of their own; as 2-column shares they are one short round spread over all SIMDs
    // (4096 x 91: 217 -> 207 us; with 704 of 4800 shares beyond the round at 3072 drones the 2-column blocks
    // cost more than they even out: 136 -> 153 us, hence the limit of an eighth of the slots)
    g.upw_tail = 2;
    g.split = (shares - shares % slots) * upw;
  }
  waves = g.split / upw + (g.total - g.split + g.upw_tail - 1) / g.upw_tail;
  // A small launch (a small swarm, or one of many shards) is as long as ONE share takes -- 16 sample chunks x 8
  // columns are one dependent chain of scalar fetches, 43 us at 512 drones whatever the arithmetic.  While the
  // shares do not fill a quarter of the wave slots (half-full launches lose: 512 of 4096 rows 71 -> 84 us), each
  // is taken by 2, 4 or 8 waves with a range of the sample chunks each (narrower column blocks instead would repeat the row loads per block: 1024 drones 46 -> 59 us).
  // the exact broad phase (CollideCull): a whole swarm, sorted on the GPU
  // (below some 3000 drones the six small launches in front of the pass cost more than a sparse swarm saves:
  // 2048 x 91 dense 91 -> 121 us, sparse 92 -> 84; 4096 x 91 dense 243 -> 282, sparse 243 -> 122, the formation
  // fixture 243 -> 105)
  const bool cull = g.sym && formation_collide_takes_broad_phase(ctx, n_rows, row_offset, n_cols, n_samples);
  // (what the last pass did: the broad-phase fields are set together, once its buffer exists)
  ctx->collide_last_cull = 0;
  ctx->collide_meta = nullptr;
  ctx->collide_last_shares = (int)(waves < 0x7fffffff ? waves : 0x7fffffff);
  const int E = n_samples * 3;
  if (cull) {
    // Launches: [key, unless the sampler left boxes and keys] -> rank -> gather -> select -> evaluator -> fold, the last
    // two either
    //   * the surviving GROUP PAIRS (8 x 8 drones) and the per-group fold of their candidates, or
    //   * the surviving SHARES (128 x 8) and the merge of their entries.
    // Which one is a HOST decision (the launch sequences differ): "collide_cull_mode" 1 / 2 force it; otherwise the
    // survivor counts the previous pass of this context left in page-locked memory (cull_hint_pack; read without
    // synchronising, so possibly a few passes old) are put through the cost model, and a pass without such a hint
    // for its swarm size takes the shares.  Both evaluators are exact; the choice only moves time.
    const int N = n_rows;
    const size_t nJ = ((size_t)N + kColBlock - 1) / kColBlock;
    long long shares = 0;
    for (int I = 0; I < g.n_rb; ++I) shares += (N - I * kRowBlock + kColBlock - 1) / kColBlock;
    const long long all_groups = (long long)nJ * (nJ + 1) / 2;
    // Up to kCullGroupMaxDrones every group pair has a list slot; larger swarms get kGroupCapLarge slots and BOTH
    // evaluators are launched -- the share evaluator and its merge return at once unless the survivors overflowed the
    // list (then the group launches did nothing): two empty launches on a pass of a quarter millisecond.
    const bool all_fit = N <= kCullGroupMaxDrones;
    bool by_groups = false;
    if (ctx->collide_cull_mode == 2) {
      by_groups = true;
    } else if (ctx->collide_cull_mode != 1 && ctx->cull_hint) {
      const unsigned long long h = *(volatile unsigned long long *)ctx->cull_hint;
      const int hN = (int)(h >> 48) & 0x7fff;
      const long long hs = (long long)((h >> 24) & 0xffffff), hg = (long long)(h & 0xffffff);
      by_groups = h != 0 && hN == (N & 0x7fff) && hg != 0xffffff && cull_groups_cheaper(hs, hg) &&
                  (all_fit || 2 * hg <= kGroupCapLarge);
    }
    const bool both = by_groups && !all_fit;
    // sample parts a share may be cut into: the column-side slots are n_rb x spmax x N entries, pre-filled per call
    const int spmax = N > 8192 ? 2 : kCullMaxParts;
    ctx->collide_last_shares = (int)shares;
    const int sp_force = (ctx->collide_sample_parts > 0 ? (ctx->collide_sample_parts < spmax ? ctx->collide_sample_parts : spmax) : 0) |
                         (spmax << 8);
    // share evaluator: row-side entries, one per item (cull_split cuts shares only while the items stay below twice the
    // wave slots), column-side slots.  Group evaluator: list, 16 candidate slots per item, reverse lists
    const long long items_max = (sp_force & 0xff) ? shares * (sp_force & 0xff) : (shares > 2 * slots ? shares : 2 * slots);
    const size_t entries = (by_groups && !both) ? 0 : (size_t)items_max * kRowBlock;
    const size_t centries = (by_groups && !both) ? 0 : (size_t)g.n_rb * spmax * N;
    const size_t gcap = !by_groups ? 0 : all_fit ? (size_t)all_groups : (size_t)kGroupCapLarge;
    // the sampler's hand-over (msnap_sample_collide_device: boxes [N][6], then keys [N]) saves the key launch
    const bool have_keys = handover_form(ctx, rows_t_in, N, n_samples) == 2;
    // Buffers (doubles, then ints): sorted row image [E][Rp] | sorted columns [N][E] | box [N][6] | colbox [nJ][6] |
    // bound [N] | row-side entries | column-side slots | candidates [gcap][16] || entries (j) | slots (j, pre-filled -1) |
    // keys [N] | perm [N] | survivor list [shares] | cnt [n_rb] | meta | acnt, astart [nJ] | blist [nJ][nJ] (zeroed per
    // pass) | glist [gcap] | cand_j [gcap][16]
    const size_t doubles = (size_t)g.Rp * E + (size_t)N * E + (size_t)N * 6 + nJ * 6 + (size_t)N + entries + centries + gcap * 16;
    const size_t ints = entries + centries + (size_t)N + (size_t)N + (size_t)shares + g.n_rb + kMetaWords + 2 * nJ + gcap +
                        gcap * 16 + (by_groups ? nJ * nJ : 0);
    int rc = ensure(ctx, ctx->stage[7], doubles * sizeof(double) + ints * sizeof(int32_t) + 64);
    if (rc) return rc;
    double *rows_t = (double *)ctx->stage[7].p;
    double *psorted = rows_t + (size_t)g.Rp * E, *box_own = psorted + (size_t)N * E, *colbox = box_own + (size_t)N * 6;
    unsigned long long *bound = (unsigned long long *)(colbox + nJ * 6);
    double *pd = (double *)(bound + N), *cd = pd + entries;
    double *cand_d2 = cd + centries;
    int32_t *pj = (int32_t *)(cand_d2 + gcap * 16), *ci = pj + entries;
    unsigned *key_own = (unsigned *)(ci + centries);
    int32_t *perm = (int32_t *)(key_own + N), *surv = perm + N, *cnt = surv + shares, *meta = cnt + g.n_rb;
    int32_t *acnt = meta + kMetaWords, *astart = acnt + nJ;
    int32_t *blist = astart + nJ, *glist = blist + (by_groups ? nJ * nJ : 0), *cand_j = glist + gcap;
    const double *box = have_keys ? rows_t_in : box_own;
    const unsigned *key = have_keys ? reinterpret_cast<const unsigned *>(rows_t_in + (size_t)N * 6) : key_own;
    CullGroups cg{glist, astart, acnt, blist, cand_d2, cand_j, (int)gcap, (int)nJ};
    // the reverse lists are all-zero between passes (the fold zeroes what it read): they are cleared only when this
    // block has held something else since -- another layout, another evaluator, any other pass of the context -- and,
    // so that a graph replays whatever ran between its replays, always under stream capture
    // (nor is a block trusted that a graph may replay on between two eager passes)
    const bool trust = by_groups && !stream_is_capturing(ctx) && !ctx->stage[7].in_graph;
    const bool blist_clean = trust && ctx->blist_clean == (const void *)blist && ctx->blist_clean_n == (int)nJ;
    ctx->blist_clean = trust ? (const void *)blist : nullptr;
    ctx->blist_clean_n = (int)nJ;
    ctx->collide_meta = meta;
    ctx->collide_last_cull = 1;
    ctx->collide_last_by_groups = by_groups ? (all_fit ? 1 : 2) : 0;      // (2: while the survivors fit kGroupCapLarge)
    ctx->collide_last_n = N;
    if (!have_keys) {
      hipLaunchKernelGGL(collide_key_kernel, dim3((N + kKeyDrones - 1) / kKeyDrones), dim3(kWave * kKeyDrones), 0,
                         ctx->stream, pos_cols, N, n_samples, box_own, key_own);
      MSNAP_HIP(ctx, hipGetLastError());
    }
    hipLaunchKernelGGL((N <= 4096 ? collide_rank_kernel<4> : N <= 8192 ? collide_rank_kernel<8> : collide_rank_kernel<16>),
                       dim3((N + kRankTile - 1) / kRankTile), dim3(kWave * kRankWaves), 0, ctx->stream, key, N, perm, box,
                       bound, blist, (int)(by_groups && !blist_clean ? nJ * nJ : 0), meta);
    MSNAP_HIP(ctx, hipGetLastError());
    const int nx = g.Rp / kTileRows, ny = (E + kTileE - 1) / kTileE;
    const int n_box = (int)((nJ + kBoxGroups - 1) / kBoxGroups), n_fill = centries ? 4 * nx : 0;
    hipLaunchKernelGGL(collide_gather_kernel, dim3((unsigned)(nx * ny + n_box + n_fill)), dim3(256), 0, ctx->stream, pos_cols, N,
                       g.Rp, E, rows_t, psorted, (const int32_t *)perm, bound, box, colbox, nx, ny, n_box, ci, centries);
    MSNAP_HIP(ctx, hipGetLastError());
    CollideCull cu{colbox};
    hipLaunchKernelGGL((N <= kCullGroupMaxDrones ? collide_select_kernel<1> : collide_select_kernel<2>),
                       dim3(g.n_rb + (unsigned)((nJ + kSelGroups - 1) / kSelGroups)), dim3(kSelThreads), 0, ctx->stream, N, g.n_rb, cu,
                       (const double *)bound, surv, cnt, meta, cg);
    MSNAP_HIP(ctx, hipGetLastError());
    if (by_groups) {
      hipLaunchKernelGGL(collide_eval_groups_kernel, dim3((unsigned)(ctx->n_cu * 4 * kGroupWaves)), dim3(kWave), 0, ctx->stream,
                         (const double *)psorted, N, n_samples, (const int32_t *)perm, (const int32_t *)meta, cg, ctx->cull_hint);
      MSNAP_HIP(ctx, hipGetLastError());
      hipLaunchKernelGGL((N <= 4096 ? collide_finish_groups_kernel<8> : N <= 8192 ? collide_finish_groups_kernel<16> : collide_finish_groups_kernel<32>),
                         dim3((unsigned)((nJ + kFinishWaves - 1) / kFinishWaves)),
                         dim3(kWave * kFinishWaves), 0, ctx->stream, N, (const int32_t *)perm, (const int32_t *)meta, cg, radius,
                         min_dist, partner, hit);
      MSNAP_HIP(ctx, hipGetLastError());
      if (!both) return MSNAP_OK;
    }
    hipLaunchKernelGGL(collide_eval_shares_kernel, dim3((unsigned)slots), dim3(kWave), 0, ctx->stream, (const double *)rows_t,
                       (const double *)psorted, g, pd, pj, cd, ci, (const int32_t *)perm, (const int32_t *)surv, sp_force,
                       (int)slots, meta, ctx->cull_hint, both ? (int)gcap : 0);
    MSNAP_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(collide_merge_kernel, dim3((N + kMergeRows - 1) / kMergeRows), dim3(kMergeRows * kMergeParts), 0,
                       ctx->stream, pd, pj, g, cd, ci, radius, min_dist, partner, hit, (const int32_t *)perm,
                       (const int32_t *)cnt, (const int32_t *)meta, both ? (int)gcap : 0);
    MSNAP_HIP(ctx, hipGetLastError());
    return MSNAP_OK;
  }
  // (a row image is only what the sampler's record says it is: a hand-over in the keys form -- written for a whole-
  // swarm pass that, by the options now in force, is not taken --, for other rows, or not on record is not one)
  if (handover_form(ctx, rows_t_in, n_rows, n_samples) != 1) rows_t_in = nullptr;
  if (ctx->collide_sample_parts > 0) {
    g.sparts = ctx->collide_sample_parts < 8 ? ctx->collide_sample_parts : 8;
  } else if (ctx->collide_waves_per_cu == 0) {
    const int nch = (n_samples + kSampleChunk - 1) / kSampleChunk;
    while (g.sparts < 8 && waves * g.sparts < slots / 4 && nch / (g.sparts * 2) >= 2) g.sparts *= 2;
  }
  if (upw > 0x3fffffff || waves * g.sparts > 0x7fffffff) return MSNAP_EINVAL;
  const size_t part_entries = ((size_t)waves + g.n_rb) * g.sparts * kRowBlock;
  const size_t centries = g.sym ? cpart_entries * g.sparts : 0;
  const size_t t_entries = rows_t_in ? 0 : (size_t)g.Rp * E;
  ctx->blist_clean = nullptr;      // (the block is about to hold this pass's buffers)
  int rc = ensure(ctx, ctx->stage[7],
                  t_entries * sizeof(double) + (part_entries + centries) * (sizeof(double) + sizeof(int32_t)) + 64);
  if (rc) return rc;
  double *rows_t = (double *)ctx->stage[7].p;
  double *pd = rows_t + t_entries;
  double *cd = pd + part_entries;
  int32_t *pj = (int32_t *)(cd + centries);
  int32_t *ci = pj + part_entries;
  if (!rows_t_in) {
    hipLaunchKernelGGL(collide_transpose_kernel, dim3(g.Rp / 64, (E + 31) / 32), dim3(256), 0, ctx->stream, pos_rows,
                       n_rows, g.Rp, E, rows_t, (E + 31) / 32, (int32_t *)nullptr, (size_t)0, (const int32_t *)nullptr,
                       (double *)nullptr);
    MSNAP_HIP(ctx, hipGetLastError());
  }
  hipLaunchKernelGGL(collide_span_kernel, dim3((unsigned)(waves * g.sparts)), dim3(kWave), 0, ctx->stream,
                     rows_t_in ? rows_t_in : (const double *)rows_t, pos_cols, g, pd, pj, cd, ci);
  MSNAP_HIP(ctx, hipGetLastError());
  hipLaunchKernelGGL(collide_merge_kernel, dim3((n_rows + kMergeRows - 1) / kMergeRows), dim3(kMergeRows * kMergeParts), 0,
                     ctx->stream, pd, pj, g, cd, ci, radius, min_dist, partner, hit, (const int32_t *)nullptr,
                     (const int32_t *)nullptr, (const int32_t *)nullptr);
  MSNAP_HIP(ctx, hipGetLastError());
  return MSNAP_OK;
}

// One rank's part of the pass over the WHOLE swarm: every unordered pair of the swarm is a (row block, column)
// unit of one triangular line -- the line a single-GPU launch walks -- and part p of P takes the p-th of P equal
// contiguous ranges of its 8-column shares.  It evaluates each of its pairs once, credits both drones, and leaves
// the squared minimum and partner of EVERY drone (+inf / -1 where it met none of the drone's pairs) in
// out_d2 [N] / out_j [N]; collide_finish_kernel folds the P parts.
int launch_formation_collide_part(msnap_ctx *ctx, int N, int n_samples, const double *pos_all, int part, int n_parts,
                                  double *out_d2, int32_t *out_j) {
  if (n_samples < kSampleChunk) {
    // short paths: the plain kernel on the part's block of rows, one-sidedly against all columns
    const int r0 = (int)((long long)N * part / n_parts), r1 = (int)((long long)N * (part + 1) / n_parts);
    hipLaunchKernelGGL(collide_part_clear_kernel, dim3((N + 255) / 256), dim3(256), 0, ctx->stream, out_d2, out_j, N);
    MSNAP_HIP(ctx, hipGetLastError());
    if (r1 > r0) {
      hipLaunchKernelGGL(collide_short_kernel, dim3((r1 - r0 + kWave - 1) / kWave), dim3(kWave), 0, ctx->stream,
                         pos_all + (size_t)r0 * n_samples * 3, pos_all, r1 - r0, r0, N, n_samples, 0.0, out_d2 + r0,
                         out_j + r0, (int32_t *)nullptr, 1);
      MSNAP_HIP(ctx, hipGetLastError());
    }
    return MSNAP_OK;
  }
  CollideGeom g;
  g.R = N;
  g.ro = 0;
  g.Cn = N;
  g.S = n_samples;
  g.n_rb = (N + kRowBlock - 1) / kRowBlock;
  g.os = 0;
  g.oe = N;
  g.sym = 1;
  g.total = collide_ustart(g, g.n_rb);
  // part p walks the units [total * p / P, total * (p + 1) / P) of the line, the cuts rounded to whole 8-column blocks
  auto cut = [&](int p) {
    const long long u = g.total * p / n_parts / kColBlock * kColBlock;
    return p >= n_parts ? g.total : u;
  };
  g.u_lo = cut(part);
  g.u_n = cut(part + 1) - g.u_lo;
  g.part = 1;
  g.sparts = 1;
  g.upw = g.upw_tail = kColBlock;
  g.I_lo = 0;
  g.I_hi = -1;
  const long long shares = (g.u_n + kColBlock - 1) / kColBlock;
  g.split = shares * kColBlock;
  long long waves = shares;
  if (shares > 0) {
    const long long u0 = g.u_lo, u1 = g.u_lo + g.u_n - 1;
    while (g.I_lo + 1 < g.n_rb && collide_ustart(g, g.I_lo + 1) <= u0) ++g.I_lo;
    g.I_hi = g.I_lo;
    while (g.I_hi + 1 < g.n_rb && collide_ustart(g, g.I_hi + 1) <= u1) ++g.I_hi;
    // Filling the wave slots (4 per SIMD) ONCE is what a part launch is tuned for -- nothing refills a slot that
    // frees early, and a launch that exceeds the slots by a few shares runs those as a round of their own
    // (4096 x 91 in two parts: 4224 shares on 4096 slots took 133 us against 112 for half of the single launch).
    //  * fewer shares than slots: every share is taken by floor(slots / shares) waves, each a range of the sample
    //    chunks (a share alone is one chain of 16 x 8 dependent fetches; an eighth of the 4096-drone line,
    //    1056 shares, span kernel alone / merge under rocprofv3: 54 / 8 us whole, 43 / 11 in halves, 40 / 17 in
    //    quarters -- 4224 waves, 128 of them queued -- 46 / 31 in eighths; every split multiplies the partial
    //    entries the merge sweeps)
    //  * a few shares more than whole rounds of the slots: the excess goes out as 2-column shares (the whole-pass
    //    rule above)
    const long long slots = (long long)ctx->n_cu * 4 * 4;
    const int nch = (n_samples + kSampleChunk - 1) / kSampleChunk;
    if (ctx->collide_sample_parts > 0) {
      g.sparts = ctx->collide_sample_parts < 8 ? ctx->collide_sample_parts : 8;
    } else if (shares * 2 <= slots) {
      long long sp = slots / shares;
      sp = sp > 8 ? 8 : sp;
      while (sp > 1 && nch / sp < 2) --sp;
      g.sparts = (int)sp;
    } else if (shares % slots != 0 && shares % slots <= slots / 8 && shares > slots) {
      g.upw_tail = 2;
      g.split = (shares - shares % slots) * kColBlock;
    }
    waves = g.split / kColBlock + (g.u_n - g.split + g.upw_tail - 1) / g.upw_tail;
  }
  const int nrb = g.I_hi - g.I_lo + 1;                 // row blocks this part touches (0: an empty part)
  g.Rp = (nrb > 0 ? nrb : 1) * kRowBlock;
  if (waves * g.sparts > 0x7fffffff) return MSNAP_EINVAL;
  const size_t part_entries = ((size_t)waves + (size_t)(nrb > 0 ? nrb : 0)) * g.sparts * kRowBlock;
  const size_t centries = (size_t)(nrb > 0 ? nrb : 0) * g.sparts * (size_t)N;
  const int E = n_samples * 3;
  const size_t t_entries = (size_t)g.Rp * E;
  if ((part_entries + centries) * 12 > ((size_t)16 << 30)) return MSNAP_ENOMEM;
  ctx->collide_last_cull = 0;      // (the buffer the last broad-phase pass left its counts in is reused)
  ctx->collide_meta = nullptr;
  ctx->blist_clean = nullptr;      // (the block is about to hold this pass's buffers)
  int rc = ensure(ctx, ctx->stage[7],
                  t_entries * sizeof(double) + (part_entries + centries) * (sizeof(double) + sizeof(int32_t)) + 64);
  if (rc) return rc;
  double *rows_t = (double *)ctx->stage[7].p;
  double *pd = rows_t + t_entries;
  double *cd = pd + part_entries;
  int32_t *pj = (int32_t *)(cd + centries);
  int32_t *ci = pj + part_entries;
  if (waves > 0) {
    const int r_first = g.I_lo * kRowBlock;
    const int r_cnt = (N - r_first) < g.Rp ? (N - r_first) : g.Rp;
    const int ny = (E + 31) / 32;
    hipLaunchKernelGGL(collide_transpose_kernel, dim3(g.Rp / 64, ny + 4), dim3(256), 0, ctx->stream,
                       pos_all + (size_t)r_first * E, r_cnt, g.Rp, E, rows_t, ny, ci, centries, (const int32_t *)nullptr,
                       (double *)nullptr);
    MSNAP_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(collide_span_kernel, dim3((unsigned)(waves * g.sparts)), dim3(kWave), 0, ctx->stream,
                       (const double *)rows_t, pos_all, g, pd, pj, cd, ci);
    MSNAP_HIP(ctx, hipGetLastError());
  } else {
    g.total = 0;   // the merge of nothing: +inf / -1 everywhere
  }
  hipLaunchKernelGGL(collide_merge_kernel, dim3((N + kMergeRows - 1) / kMergeRows), dim3(kMergeRows * kMergeParts), 0,
                     ctx->stream, pd, pj, g, cd, ci, 0.0, out_d2, out_j, (int32_t *)nullptr, (const int32_t *)nullptr,
                     (const int32_t *)nullptr, (const int32_t *)nullptr);
  MSNAP_HIP(ctx, hipGetLastError());
  return MSNAP_OK;
}

int launch_formation_collide_finish(msnap_ctx *ctx, int N, int n_parts, const void *parts, size_t part_stride,
                                    int row_offset, int n_rows, double radius, double *min_dist, int32_t *partner,
                                    int32_t *hit) {
  hipLaunchKernelGGL(collide_finish_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, ctx->stream,
                     (const unsigned char *)parts, part_stride, n_parts, N, row_offset, n_rows, radius, min_dist, partner,
                     hit);
  MSNAP_HIP(ctx, hipGetLastError());
  return MSNAP_OK;
}

// ------------------------------------------------------------------------------------
// Mesh sweep: min over samples and triangles of the point-triangle distance
// (closest-point regions, Ericson 5.1.5).  One workgroup per drone, one lane per
// sample; the triangle is wave-uniform (scalar loads).  Semantics are this
// repo's; the reference only has a boolean FCL mesh-mesh test in the
// planner (src/RigidBodyPlanners/fcl_checker.py:93-100).
// ------------------------------------------------------------------------------------
__device__ __forceinline__ double pt_tri_d2(double px, double py, double pz, const double *__restrict__ t) {
#pragma clang fp contract(off)
  const double ax = t[0], ay = t[1], az = t[2];
  const double bx = t[3], by = t[4], bz = t[5];
  const double cx = t[6], cy = t[7], cz = t[8];
  const double abx = bx - ax, aby = by - ay, abz = bz - az;
  const double acx = cx - ax, acy = cy - ay, acz = cz - az;
  const double apx = px - ax, apy = py - ay, apz = pz - az;
  const double d1 = abx * apx + aby * apy + abz * apz;
  const double d2 = acx * apx + acy * apy + acz * apz;
  double qx, qy, qz;
  if (d1 <= 0.0 && d2 <= 0.0) {
    qx = ax; qy = ay; qz = az;
  } else {
    const double bpx = px - bx, bpy = py - by, bpz = pz - bz;
    const double d3 = abx * bpx + aby * bpy + abz * bpz;
    const double d4 = acx * bpx + acy * bpy + acz * bpz;
    if (d3 >= 0.0 && d4 <= d3) {
      qx = bx; qy = by; qz = bz;
    } else {
      const double vc = d1 * d4 - d3 * d2;
      if (vc <= 0.0 && d1 >= 0.0 && d3 <= 0.0) {
        const double v = d1 / (d1 - d3);
        qx = ax + v * abx; qy = ay + v * aby; qz = az + v * abz;
      } else {
        const double cpx = px - cx, cpy = py - cy, cpz = pz - cz;
        const double d5 = abx * cpx + aby * cpy + abz * cpz;
        const double d6 = acx * cpx + acy * cpy + acz * cpz;
        if (d6 >= 0.0 && d5 <= d6) {
          qx = cx; qy = cy; qz = cz;
        } else {
          const double vb = d5 * d2 - d1 * d6;
          if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) {
            const double w = d2 / (d2 - d6);
            qx = ax + w * acx; qy = ay + w * acy; qz = az + w * acz;
          } else {
            const double va = d3 * d6 - d5 * d4;
            if (va <= 0.0 && (d4 - d3) >= 0.0 && (d5 - d6) >= 0.0) {
              const double w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
              qx = bx + w * (cx - bx); qy = by + w * (cy - by); qz = bz + w * (cz - bz);
            } else {
              const double denom = 1.0 / (va + vb + vc);
              const double v = vb * denom, w = vc * denom;
              qx = ax + abx * v + acx * w;
              qy = ay + aby * v + acy * w;
              qz = az + abz * v + acz * w;
            }
          }
        }
      }
    }
  }
  const double ex = px - qx, ey = py - qy, ez = pz - qz;
  return ex * ex + ey * ey + ez * ez;
}

// squared distance between the box [lo, hi] and the bounding box of triangle t: a lower bound of every
// point-triangle distance between them
__device__ __forceinline__ double box_tri_lb2(const double (&lo)[3], const double (&hi)[3], const double *__restrict__ t) {
#pragma clang fp contract(off)
  double lb2 = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double tmin = fmin(t[k], fmin(t[3 + k], t[6 + k])), tmax = fmax(t[k], fmax(t[3 + k], t[6 + k]));
    const double gap = fmax(0.0, fmax(lo[k] - tmax, tmin - hi[k]));
    lb2 = lb2 + gap * gap;
  }
  return lb2;
}

// One workgroup per drone, one lane per sample (ceil(S / 64) waves, at most 16); the triangle under
// test is wave-uniform: its vertices are scalar loads and scalar operands.
// Exact culling per wave.  The squared distance between a triangle's bounding box and the bounding box
// of the wave's stretch of path is a lower bound of every point-triangle distance of that pair, so a
// triangle whose bound is not below the best distance found so far cannot lower the minimum.  Lane t
// of the wave holds the bound of triangle t (groups of 64 triangles); the wave repeatedly takes the
// triangle with the smallest bound, tests it against its 64 samples, and stops the group as soon as the
// smallest remaining bound is not below the best distance: a path far from the scene tests one or
// two triangles, a path through a wall all of the wall's.  min_dist stays the exact minimum over all
// samples and triangles -- a skipped test could only have returned something larger.
__global__ void __launch_bounds__(1024)
mesh_sweep_kernel(const double *__restrict__ pos, int N, int S, const double *__restrict__ tris, int n_tris,
                  double radius, double *__restrict__ min_dist, int32_t *__restrict__ hit,
                  unsigned long long *__restrict__ tests_done) {
  __shared__ double sBest[16];
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave, nw = blockDim.x / kWave;
  unsigned long long done = 0;
  // one drone per workgroup, or ("mesh_waves_per_cu") a smaller grid whose workgroups walk the drones: a sweep that
  // runs beside another stream's kernels then leaves them wave slots and registers for as long as it runs
  for (int d = blockIdx.x; d < N; d += gridDim.x) {
  double best = INFINITY;
  for (int s0 = wv * kWave; s0 < S; s0 += nw * kWave) {      // one trip unless S > 1024
    const int s = s0 + lane;
    const double *p = pos + ((size_t)d * S + (s < S ? s : S - 1)) * 3;   // lanes past the end replay the last sample
    const double px = p[0], py = p[1], pz = p[2];
    // the box of the wave's FINITE samples, wave-uniform (non-finite samples never win a minimum, include/msnap.h:
    // they must not distort the cull of the finite ones either; a stretch without a finite coordinate gives a
    // NaN box, whose bound is 0 for every triangle: nothing is culled, nothing can win)
    double lo[3], hi[3];
    const double fx = __builtin_isfinite(px) ? px : __builtin_nan(""), fy = __builtin_isfinite(py) ? py : __builtin_nan(""),
                 fz = __builtin_isfinite(pz) ? pz : __builtin_nan("");
    lo[0] = uniform_f64(wave_minmax_num_f64<false>(fx)); lo[1] = uniform_f64(wave_minmax_num_f64<false>(fy));
    lo[2] = uniform_f64(wave_minmax_num_f64<false>(fz));
    hi[0] = uniform_f64(wave_minmax_num_f64<true>(fx)); hi[1] = uniform_f64(wave_minmax_num_f64<true>(fy));
    hi[2] = uniform_f64(wave_minmax_num_f64<true>(fz));
    double wbest = uniform_f64(wave_min_f64(best));            // wave-uniform bound: min over the lanes so far
    for (int t0 = 0; t0 < n_tris; t0 += kWave) {
      const int tl = t0 + lane;
      double lb = (tl < n_tris) ? box_tri_lb2(lo, hi, tris + (size_t)(tl < n_tris ? tl : 0) * 9) : INFINITY;
      // the triangle with the smallest bound first: it usually sets the distance the others have to beat
      const double m = uniform_f64(wave_min_f64(lb));
      if (!(m < wbest)) continue;                              // (also when the group has no triangle: every bound inf)
      {
        const int sel = __builtin_ctzll(__ballot(lb == m));    // wave-uniform: the ballot is a scalar
        lb = (lane == sel) ? INFINITY : lb;
        const double v = pt_tri_d2(px, py, pz, tris + (size_t)(t0 + sel) * 9);
        best = (v < best) ? v : best;
        wbest = uniform_f64(wave_min_f64(best));
        done += 1;
      }
      // then every triangle whose bound is still below the best distance, in lane order; the best distance (two
      // cross-lane reductions: as long as a third of a triangle's evaluation) is refreshed every fourth triangle only --
      // a superset of what the one-by-one order evaluates, the same minimum
      unsigned long long cand = __ballot(lb < wbest);
      for (int k = 1; cand; ++k) {
        const int sel = __builtin_ctzll(cand);
        cand &= cand - 1;
        const double v = pt_tri_d2(px, py, pz, tris + (size_t)(t0 + sel) * 9);
        best = (v < best) ? v : best;
        done += 1;
        if ((k & 3) == 0 && cand) {
          wbest = uniform_f64(wave_min_f64(best));
          cand &= __ballot(lb < wbest);
        }
      }
      wbest = uniform_f64(wave_min_f64(best));
    }
  }
  best = wave_min_f64(best);
  if (lane == 0) sBest[wv] = best;
  lds_barrier();
  if (threadIdx.x == 0) {
    for (int k = 1; k < nw; ++k) best = (sBest[k] < best) ? sBest[k] : best;
    const double dist = sqrt(best);
    min_dist[d] = dist;
    hit[d] = (dist < radius) ? 1 : 0;
  }
  lds_barrier();      // (sBest is free for the next drone)
  }
  if (tests_done && lane == 0) atomicAdd(tests_done, done * kWave);
}

int launch_mesh_sweep(msnap_ctx *ctx, int n_drones, int n_samples, const double *pos, int n_tris,
                      const double *tris, double radius, double *min_dist, int32_t *hit) {
  int waves = (n_samples + kWave - 1) / kWave;
  if (waves > 16) waves = 16;
  int blocks = n_drones;
  if (ctx->mesh_waves_per_cu > 0) {      // msnap_set_option: room for other streams' workgroups beside the sweep
    const int cap = ctx->n_cu * ctx->mesh_waves_per_cu / waves;
    blocks = blocks < cap ? blocks : (cap > 0 ? cap : 1);
  }
  hipLaunchKernelGGL(mesh_sweep_kernel, dim3(blocks), dim3(waves * kWave), 0, ctx->stream, pos, n_drones,
                     n_samples, tris, n_tris, radius, min_dist, hit, (unsigned long long *)ctx->mesh_tests);
  MSNAP_HIP(ctx, hipGetLastError());
  return MSNAP_OK;
}

// ------------------------------------------------------------------------------------
// f4: rigid-body state validity, batched -- the OMPL validity callback of the planner
// (reference src/RigidBodyPlanners/RB_planning_sep_coll_check.py:208-226: the robot mesh
// is placed at (x, y, z) with quaternion_from_euler(0, 0, yaw) and fcl.collide is asked
// whether it touches the environment mesh, src/RigidBodyPlanners/fcl_checker.py:93-100).
// FCL is not vendored in the reference (parity unpinned): the predicate here is "some
// robot triangle and some environment triangle intersect as closed sets", decided by the
// 17-axis separating-axis test (2 face normals, 9 edge x edge, 6 edge x normal for the
// coplanar case).  One wavefront per state, lanes stride the triangle pairs.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ bool sat_separates(const double (&P)[3][3], const double (&Q)[3][3], double lx, double ly,
                                              double lz) {
#pragma clang fp contract(off)
  double p0 = P[0][0] * lx + P[0][1] * ly + P[0][2] * lz;
  double p1 = P[1][0] * lx + P[1][1] * ly + P[1][2] * lz;
  double p2 = P[2][0] * lx + P[2][1] * ly + P[2][2] * lz;
  double q0 = Q[0][0] * lx + Q[0][1] * ly + Q[0][2] * lz;
  double q1 = Q[1][0] * lx + Q[1][1] * ly + Q[1][2] * lz;
  double q2 = Q[2][0] * lx + Q[2][1] * ly + Q[2][2] * lz;
  const double pmin = fmin(p0, fmin(p1, p2)), pmax = fmax(p0, fmax(p1, p2));
  const double qmin = fmin(q0, fmin(q1, q2)), qmax = fmax(q0, fmax(q1, q2));
  return (pmin > qmax) || (pmax < qmin);
}

__device__ __forceinline__ bool tri_tri_intersect(const double (&P)[3][3], const double (&Q)[3][3]) {
#pragma clang fp contract(off)
  double e[3][3], f[3][3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    e[0][c] = P[1][c] - P[0][c]; e[1][c] = P[2][c] - P[1][c]; e[2][c] = P[0][c] - P[2][c];
    f[0][c] = Q[1][c] - Q[0][c]; f[1][c] = Q[2][c] - Q[1][c]; f[2][c] = Q[0][c] - Q[2][c];
  }
  const double n1x = e[0][1] * e[1][2] - e[0][2] * e[1][1];
  const double n1y = e[0][2] * e[1][0] - e[0][0] * e[1][2];
  const double n1z = e[0][0] * e[1][1] - e[0][1] * e[1][0];
  if (sat_separates(P, Q, n1x, n1y, n1z)) return false;
  const double n2x = f[0][1] * f[1][2] - f[0][2] * f[1][1];
  const double n2y = f[0][2] * f[1][0] - f[0][0] * f[1][2];
  const double n2z = f[0][0] * f[1][1] - f[0][1] * f[1][0];
  if (sat_separates(P, Q, n2x, n2y, n2z)) return false;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const double lx = e[i][1] * f[j][2] - e[i][2] * f[j][1];
      const double ly = e[i][2] * f[j][0] - e[i][0] * f[j][2];
      const double lz = e[i][0] * f[j][1] - e[i][1] * f[j][0];
      if (sat_separates(P, Q, lx, ly, lz)) return false;
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    double lx = n1y * e[i][2] - n1z * e[i][1], ly = n1z * e[i][0] - n1x * e[i][2], lz = n1x * e[i][1] - n1y * e[i][0];
    if (sat_separates(P, Q, lx, ly, lz)) return false;
    lx = n2y * f[i][2] - n2z * f[i][1]; ly = n2z * f[i][0] - n2x * f[i][2]; lz = n2x * f[i][1] - n2y * f[i][0];
    if (sat_separates(P, Q, lx, ly, lz)) return false;
  }
  return true;
}

__global__ void __launch_bounds__(kWave)
mesh_validity_kernel(const double *__restrict__ states, int N, const double *__restrict__ rtris, int R,
                     const double *__restrict__ etris, int E, int32_t *__restrict__ valid) {
#pragma clang fp contract(off)
  const int sidx = blockIdx.x;
  const int lane = threadIdx.x;
  const double tx = states[(size_t)sidx * 4 + 0], ty = states[(size_t)sidx * 4 + 1], tz = states[(size_t)sidx * 4 + 2];
  const double yaw = states[(size_t)sidx * 4 + 3];
  // quaternion_from_euler(0, 0, yaw) = (0, 0, sin(yaw/2), cos(yaw/2)); its rotation matrix
  const double qz = sin(0.5 * yaw), qw = cos(0.5 * yaw);
  const double c = 1.0 - 2.0 * (qz * qz), s2 = 2.0 * (qz * qw);
  bool hit = false;
  const int pairs = R * E;
  for (int p0 = 0; p0 < pairs; p0 += kWave) {
    const int p = p0 + lane;
    if (p < pairs) {
      const int rt = p / E, et = p - rt * E;
      double P[3][3], Q[3][3];
#pragma unroll
      for (int v = 0; v < 3; ++v) {
        const double x = rtris[(size_t)rt * 9 + v * 3 + 0], y = rtris[(size_t)rt * 9 + v * 3 + 1];
        P[v][0] = (c * x - s2 * y) + tx;
        P[v][1] = (s2 * x + c * y) + ty;
        P[v][2] = rtris[(size_t)rt * 9 + v * 3 + 2] + tz;
#pragma unroll
        for (int k = 0; k < 3; ++k) Q[v][k] = etris[(size_t)et * 9 + v * 3 + k];
      }
      hit = hit || tri_tri_intersect(P, Q);
    }
    if (__ballot(hit) != 0ULL) break;   // wave-uniform early exit
  }
  const unsigned long long any = __ballot(hit);
  if (lane == 0) valid[sidx] = (any == 0ULL) ? 1 : 0;
}

int launch_mesh_validity(msnap_ctx *ctx, int n_states, const double *states, int n_rtris, const double *rtris,
                         int n_etris, const double *etris, int32_t *valid) {
  hipLaunchKernelGGL(mesh_validity_kernel, dim3(n_states), dim3(kWave), 0, ctx->stream, states, n_states, rtris,
                     n_rtris, etris, n_etris, valid);
  MSNAP_HIP(ctx, hipGetLastError());
  return MSNAP_OK;
}

}  // namespace msnap
