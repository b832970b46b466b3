// K2 -- shared-time-grid fast path: one dense fp64 GEMM on the matrix cores.
//
// When every drone of a batch uses the same time grid -- the reference's own
// usage: path_to_pol gives all paths the uniform grid t_i = i*10/n
// (scripts/drones_pols_generator.py:44-46,56) -- the collocation matrix A of
// calculate_trajectory1D (src/optimizations/calculatingTrajectories.py:48-131)
// is the same for every drone and axis, and b = S w is linear in the waypoint
// values, so all coefficients are
//        C[(drone,axis)][(seg,coef)] = sum_j  W[(drone,axis)][j] * Gop[j][(seg,coef)]
// with Gop = (A^-1 S)^T of shape (M+1) x (M*ncoef).  msnap_grid_prepare computes
// Gop ON THE GPU by running the K1 solve on the M+1 unit waypoint vectors (so it
// inherits K1's parity), and msnap_solve_grid is then a single
// v_mfma_f64_16x16x4_f64 GEMM -- the one genuinely dense contraction of the path,
// and the only place MFMA is used.
//
// Tiling (wave64): an MFMA row tile is 16 (drone,axis) rows = 4 drones x 4 axes,
// a column tile 16 flat (segment,coef) columns, the k dimension the M+1 waypoints
// in steps of 4.  f64 operand maps (cdna_hip_programming.md 3): A[row=l&15][k=l>>4],
// B[k=l>>4][col=l&15], D row=(l>>4)+4*reg, col=l&15 -- with row = 4*drone+axis the
// accumulator register index is the drone and lane>>4 the axis, so one store
// instruction (fixed reg) covers [2 segments][4 axes][8 coefs] = 512 contiguous
// bytes of one drone at order 7.  Every wave keeps the whole Gop in registers
// (B fragments) and streams row tiles: the kernel is bound by the output stores.
//
// Operators too large for the registers (16..63 segments at order 7: the reference's own
// 50-pose paths are 49 segments, scripts/drones_pols_generator.py:44-46 with
// src/RigidBodyPlanners/RB_planning_sep_coll_check.py:164) take grid_gemm_stream_kernel: the
// wave keeps the A fragments of RT row tiles in registers and streams the B fragments column
// tile by column tile from L2 (one fragment feeds RT MFMAs); small batches spread the column
// tiles over blockIdx.y so that a 1-drone call is a few dozen short waves, not one long one.
#include <cstdlib>

#include "msnap_internal.h"

namespace msnap {

typedef double v4f64 __attribute__((ext_vector_type(4)));

constexpr int kGridMaxCT = 8;   // column tiles kept in registers (16 columns each)
constexpr int kGridMaxKS = 4;   // k steps (4 waypoints each): M + 1 <= 16

// wp'[p][i][a] = 1 if i == 4p + a else 0 : the M+1 unit waypoint vectors, 4 per pseudo-drone
__global__ void __launch_bounds__(256)
unit_wp_kernel(double *__restrict__ wp, int P, int m) {
  const int total = P * m * 4;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int a = idx & 3;
    const int i = (idx >> 2) % m;
    const int p = (idx >> 2) / m;
    wp[idx] = (i == 4 * p + a) ? 1.0 : 0.0;
  }
}

// Gop in MFMA B-fragment order: frag[(ct*nks + ks)*64 + lane] = Gop[j = 4*ks + lane>>4][c = 16*ct + lane&15]
// (zero outside), so a fragment is one coalesced 512-byte load with no index arithmetic.
__global__ void __launch_bounds__(kWave)
pack_gop_kernel(const double *__restrict__ gop /* [P][M][4][NC] */, int M, int NC, int nks,
                double *__restrict__ frag) {
  const int lane = threadIdx.x;
  const int ct = blockIdx.x / nks, ks = blockIdx.x % nks;
  const int j = 4 * ks + (lane >> 4);
  const int c = 16 * ct + (lane & 15);
  double v = 0.0;
  if (j < M + 1 && c < M * NC) {
    const int seg = c / NC, kc = c - seg * NC;
    v = gop[(((size_t)(j >> 2) * M + seg) * 4 + (j & 3)) * NC + kc];
  }
  frag[(size_t)blockIdx.x * kWave + lane] = v;
}

template <int CTRL>
__device__ __forceinline__ double dpp_quad_f64(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

#ifdef MSNAP_TOOLS_TIMELINE
// phase timestamps (s_memrealtime, 100 MHz) of the GEMM kernel: tools/grid_timeline.py
__device__ unsigned long long g_grid_timeline[1024 * 8];
#define MSNAP_GTL(k)                                                                     \
  do {                                                                                   \
    if (lane == 0 && blockIdx.x < 1024) g_grid_timeline[blockIdx.x * 8 + (k)] = wall_clock64(); \
  } while (0)
extern "C" int msnap_debug_read_grid_timeline(unsigned long long *out, int n_words) {
  if (hipDeviceSynchronize() != hipSuccess) return MSNAP_EHIP;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_grid_timeline), (size_t)n_words * 8) == hipSuccess ? MSNAP_OK : MSNAP_EHIP;
}
#else
#define MSNAP_GTL(k) do { } while (0)
#endif

// One instance per (ncoef, segment count): tile counts are constants, so there are no per-tile
// branches, all column tiles' MFMA chains are issued back to back (independent accumulators keep
// the matrix pipe full) and only then exchanged and stored.
template <int NC, int M>
__global__ void __launch_bounds__(kWave)
grid_gemm_kernel(const double *__restrict__ wp, const double *__restrict__ gop /* packed B fragments */,
                 const double *__restrict__ gdur /* [M] */, const int32_t *__restrict__ gstatus, int N,
                 double *__restrict__ coef, double *__restrict__ dur, int32_t *__restrict__ status, int nrt) {
  constexpr int m = M + 1;
  constexpr int ncols = M * NC;
  constexpr int NCT = (ncols + 15) / 16;
  constexpr int NKS = (m + 3) / 4;
  static_assert(NCT <= kGridMaxCT && NKS <= kGridMaxKS && 4 * M <= kWave, "operator does not fit the registers");
  const int lane = threadIdx.x;
  MSNAP_GTL(0);
  const int col = lane & 15;
  const int kq = lane >> 4;          // k offset inside a step (A/B operands), axis (D)
  const int grid_st = gstatus[0];
  const bool odd = (lane & 1) != 0;

  // B fragments (pre-packed by pack_gop_kernel): Gop[j = 4*ks + kq][c = 16*ct + col]
  double bf[NCT][NKS];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) bf[ct][ks] = gop[(size_t)(ct * kGridMaxKS + ks) * kWave + lane];

  // this lane's slot of the tile's [4 drones][M] durations
  const int r_dur = (lane < 4 * M ? lane : 0) / M;
  const double my_dur = gdur[(lane < 4 * M ? lane : 0) - r_dur * M];

  // A fragments of a row tile: W[row = lane&15 -> (drone 4*rt + row>>2, axis row&3)][j = 4*ks + kq]
  auto load_a = [&](int rt, double (&af)[NKS]) {
    const int drow = rt * 4 + (col >> 2);
    const int dclamp = drow < N ? drow : N - 1;
    const double *wrow = wp + (size_t)dclamp * m * 4 + (col & 3);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int j = 4 * ks + kq;
      const double v = wrow[(size_t)(j < m ? j : m - 1) * 4];   // clamped: always a valid address
      af[ks] = (4 * ks + 3 < m || j < m) ? v : 0.0;             // only the last k step can run past m
    }
  };

  double anext[NKS];
  if ((int)blockIdx.x < nrt) load_a(blockIdx.x, anext);
  for (int rt = blockIdx.x; rt < nrt; rt += gridDim.x) {
    const int d0 = rt * 4;
    double af[NKS];
    bool nonfin = false;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      af[ks] = anext[ks];
      nonfin = nonfin || !__builtin_isfinite(af[ks]);
    }
    MSNAP_GTL(1);
    // the next tile's waypoints are in flight while this tile's MFMAs and stores run
    if (rt + (int)gridDim.x < nrt) load_a(rt + gridDim.x, anext);
    // per-drone non-finite flag: rows 4*dl .. 4*dl+3 of the tile, any k
    const unsigned long long bal = __ballot(nonfin);

    v4f64 acc[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) acc[ct] = v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
        acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[ks], bf[ct][ks], acc[ct], 0, 0, 0);

    // drones 2h + odd of the tile: bad flags once per tile
    bool badh[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = 2 * h + (odd ? 1 : 0);
      const unsigned long long dm = 0x000F000F000F000FULL << (4 * r);   // lanes of drone r as A rows
      badh[h] = (grid_st != 0) || ((bal & dm) != 0ULL);
    }
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      // pair exchange (lanes l, l^1 hold adjacent columns): afterwards an even lane owns
      // columns (c, c+1) of drones 0 and 2, an odd lane the same columns of drones 1 and 3,
      // so every lane issues two 16-byte stores instead of four 8-byte ones
      const double s0 = dpp_quad_f64<0xB1>(odd ? acc[ct][0] : acc[ct][1]);
      const double s1 = dpp_quad_f64<0xB1>(odd ? acc[ct][2] : acc[ct][3]);
      const double lo0 = odd ? s0 : acc[ct][0], hi0 = odd ? acc[ct][1] : s0;   // drone (odd ? 1 : 0)
      const double lo1 = odd ? s1 : acc[ct][2], hi1 = odd ? acc[ct][3] : s1;   // drone (odd ? 3 : 2)
      const int c = 16 * ct + (col & ~1);
      const int seg = c / NC, kc = c - seg * NC;     // NC is even: a pair never straddles a segment
      if (16 * ct + 15 < ncols || c < ncols) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int d = d0 + 2 * h + (odd ? 1 : 0);   // kq = axis
          double2 v = h == 0 ? make_double2(lo0, hi0) : make_double2(lo1, hi1);
          if (badh[h]) v = make_double2(__builtin_nan(""), __builtin_nan(""));
          if (d < N) *reinterpret_cast<double2 *>(coef + (((size_t)d * M + seg) * 4 + kq) * NC + kc) = v;
        }
      }
    }
    MSNAP_GTL(2);
    // durations and status of the 4 drones
    if (lane < 4 * M && d0 + r_dur < N) dur[(size_t)d0 * M + lane] = my_dur;
    if (lane < 4 && d0 + lane < N) {
      const unsigned long long dm = 0x000F000F000F000FULL << (4 * lane);
      status[d0 + lane] = (bal & dm) ? MSNAP_ST_NONFINITE : grid_st;
    }
    MSNAP_GTL(3);
  }
  MSNAP_GTL(4);
}


// ------------------------------------------------------------------------------------
// streaming variant: B fragments from L2, A fragments of RT row tiles in registers
// ------------------------------------------------------------------------------------
constexpr int kStreamMaxKS = 16;   // k steps: M + 1 <= 64

template <int NC, int NKS, int RT>
__global__ void __launch_bounds__(kWave)
grid_gemm_stream_kernel(const double *__restrict__ wp, const double *__restrict__ gop /* [nct][NKS][64] */,
                        const double *__restrict__ gdur /* [M] */, const int32_t *__restrict__ gstatus, int N,
                        int M, double *__restrict__ coef, double *__restrict__ dur, int32_t *__restrict__ status,
                        int nrg /* row groups of RT tiles */, int nct, int cts /* column tiles per slice */) {
  const int m = M + 1;
  const int ncols = M * NC;
  const int lane = threadIdx.x;
  const int col = lane & 15;
  const int kq = lane >> 4;          // k offset inside a step (A/B operands), axis (D)
  const bool odd = (lane & 1) != 0;
  const int grid_st = gstatus[0];
  const int ct0 = blockIdx.y * cts;
  const int ct1 = ct0 + cts < nct ? ct0 + cts : nct;

  auto load_b = [&](int ct, double (&bf)[NKS]) {
    const double *src = gop + (size_t)ct * NKS * kWave + lane;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) bf[ks] = src[ks * kWave];
  };

  for (int g = blockIdx.x; g < nrg; g += gridDim.x) {
    double af[RT][NKS];
    unsigned long long bal[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int drow = (g * RT + rt) * 4 + (col >> 2);
      const int dclamp = drow < N ? drow : N - 1;
      const double *wrow = wp + (size_t)dclamp * m * 4 + (col & 3);
      bool nonfin = false;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const int j = 4 * ks + kq;
        const double v = wrow[(size_t)(j < m ? j : m - 1) * 4];   // clamped: always a valid address
        af[rt][ks] = (j < m) ? v : 0.0;
        nonfin = nonfin || !__builtin_isfinite(af[rt][ks]);
      }
      bal[rt] = __ballot(nonfin);
    }

    double bnext[NKS];
    load_b(ct0, bnext);
    for (int ct = ct0; ct < ct1; ++ct) {
      double bf[NKS];
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) bf[ks] = bnext[ks];
      load_b(ct + 1 < ct1 ? ct + 1 : ct, bnext);   // the next column tile's fragments fly during the MFMAs

      v4f64 acc[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt] = v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[rt][ks], bf[ks], acc[rt], 0, 0, 0);

      const int c = 16 * ct + (col & ~1);
      const int seg = c / NC, kc = c - seg * NC;     // NC is even: a pair never straddles a segment
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const int d0 = (g * RT + rt) * 4;
        // pair exchange as in grid_gemm_kernel: even lanes end up with columns (c, c+1) of drones
        // 0 and 2, odd lanes with those of drones 1 and 3 -> 16-byte stores
        const double s0 = dpp_quad_f64<0xB1>(odd ? acc[rt][0] : acc[rt][1]);
        const double s1 = dpp_quad_f64<0xB1>(odd ? acc[rt][2] : acc[rt][3]);
        const double lo0 = odd ? s0 : acc[rt][0], hi0 = odd ? acc[rt][1] : s0;
        const double lo1 = odd ? s1 : acc[rt][2], hi1 = odd ? acc[rt][3] : s1;
        if (c < ncols) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int r = 2 * h + (odd ? 1 : 0);
            const int d = d0 + r;
            const unsigned long long dm = 0x000F000F000F000FULL << (4 * r);   // lanes of drone r as A rows
            const bool bad = (grid_st != 0) || ((bal[rt] & dm) != 0ULL);
            double2 v = h == 0 ? make_double2(lo0, hi0) : make_double2(lo1, hi1);
            if (bad) v = make_double2(__builtin_nan(""), __builtin_nan(""));
            if (d < N) *reinterpret_cast<double2 *>(coef + (((size_t)d * M + seg) * 4 + kq) * NC + kc) = v;
          }
        }
      }
    }

    if (blockIdx.y == 0) {   // durations and status once per row group
      const int d0 = g * RT * 4;
      const int nd = N - d0 < RT * 4 ? N - d0 : RT * 4;
      for (int e = lane; e < nd * M; e += kWave) dur[(size_t)d0 * M + e] = gdur[e % M];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        if (lane < 4 && d0 + 4 * rt + lane < N) {
          const unsigned long long dm = 0x000F000F000F000FULL << (4 * lane);
          status[d0 + 4 * rt + lane] = (bal[rt] & dm) ? MSNAP_ST_NONFINITE : grid_st;
        }
      }
    }
  }
}

static bool grid_gemm_reg_supported(const msnap_ctx *ctx, int n_seg) {
  const int nc = ctx->order + 1;
  const int nct = (n_seg * nc + 15) / 16;
  const int nks = (n_seg + 1 + 3) / 4;
  return nct <= kGridMaxCT && nks <= kGridMaxKS;
}

bool grid_gemm_supported(const msnap_ctx *ctx, int n_seg) {
  (void)ctx;
  return (n_seg + 1 + 3) / 4 <= kStreamMaxKS;
}

int grid_frag_ks_pitch(const msnap_ctx *ctx, int n_seg) {
  if (!grid_gemm_supported(ctx, n_seg)) return 0;
  return grid_gemm_reg_supported(ctx, n_seg) ? kGridMaxKS : (n_seg + 1 + 3) / 4;
}

int launch_grid_prepare(msnap_ctx *ctx, int n_seg, const double *t, int t_on_device) {
  const int m = n_seg + 1;
  const int nc = ctx->order + 1;
  const int P = (m + 3) / 4;
  int rc;
  if ((rc = ensure(ctx, ctx->grid_t, (size_t)m * 8))) return rc;
  ctx->grid_ready = 0;
  MSNAP_HIP(ctx, hipMemcpyAsync(ctx->grid_t.p, t, (size_t)m * 8,
                                t_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
  if (grid_gemm_supported(ctx, n_seg)) {
    // the operator: the K1 solve on the M+1 unit waypoint vectors, then packed as B fragments
    if ((rc = ensure(ctx, ctx->grid_wp, (size_t)P * m * 4 * 8))) return rc;
    if ((rc = ensure(ctx, ctx->grid_op, (size_t)P * n_seg * 4 * nc * 8))) return rc;
    if ((rc = ensure(ctx, ctx->grid_dur, (size_t)P * n_seg * 8))) return rc;
    if ((rc = ensure(ctx, ctx->grid_status, (size_t)P * 4))) return rc;
    hipLaunchKernelGGL(unit_wp_kernel, dim3((P * m * 4 + 255) / 256), dim3(256), 0, ctx->stream,
                       (double *)ctx->grid_wp.p, P, m);
    MSNAP_HIP(ctx, hipGetLastError());
    rc = launch_solve(ctx, P, n_seg, (const double *)ctx->grid_wp.p, (const double *)ctx->grid_t.p, 1,
                      (double *)ctx->grid_op.p, (double *)ctx->grid_dur.p, (int32_t *)ctx->grid_status.p);
    if (rc) return rc;
    const bool reg = grid_gemm_reg_supported(ctx, n_seg);
    const int nks = reg ? kGridMaxKS : P;
    const int nct = reg ? kGridMaxCT : (n_seg * nc + 15) / 16;
    if ((rc = ensure(ctx, ctx->grid_frag, (size_t)nct * nks * kWave * 8))) return rc;
    hipLaunchKernelGGL(pack_gop_kernel, dim3(nct * nks), dim3(kWave), 0, ctx->stream,
                       (const double *)ctx->grid_op.p, n_seg, nc, nks, (double *)ctx->grid_frag.p);
    MSNAP_HIP(ctx, hipGetLastError());
  }
  // longer paths: msnap_solve_grid runs the K1 solve on the stored grid, nothing to build
  ctx->grid_seg = n_seg;
  ctx->grid_ready = 1;
  return MSNAP_OK;
}

// Launch geometry of the streaming variant.  Large batches: RT = 4 row tiles (16 drones) per wave,
// every wave walks all column tiles, persistent over the row groups.  Small batches: one row tile
// per wave and the column tiles sliced over blockIdx.y until the chip has ~16 waves per CU.
template <int NC, int NKS>
static void launch_stream_nks(msnap_ctx *ctx, int N, int M, const double *wp, double *coef, double *dur,
                              int32_t *status) {
  const int nct = (M * NC + 15) / 16;
  // (16 waves per CU: 4096 drones x 20 segments 9.7 -> 8.1 us against 4, 8192 drones 13.3 -> 12.0; one row group
  // and the reference's 2 x 49 shape are sliced down to single column tiles either way)
  const int target = ctx->n_cu * (ctx->gemm_stream_waves_per_cu > 0 ? ctx->gemm_stream_waves_per_cu : 16);
  const double *frag = (const double *)ctx->grid_frag.p, *gdur = (const double *)ctx->grid_dur.p;
  const int32_t *gst = (const int32_t *)ctx->grid_status.p;
  if (N >= 16 * ctx->n_cu * 4) {
    constexpr int RT = 4;
    const int nrg = (N + 4 * RT - 1) / (4 * RT);
    int grid = ctx->n_cu * 16;
    if (ctx->gemm_grid_waves > 0) grid = ctx->gemm_grid_waves;
    if (grid > nrg) grid = nrg;
    note_kernel(ctx, "msnap::grid_gemm_stream_kernel<%d, %d, %d>", NC, NKS, RT);
    hipLaunchKernelGGL((grid_gemm_stream_kernel<NC, NKS, RT>), dim3(grid, 1), dim3(kWave), 0, ctx->stream, wp, frag,
                       gdur, gst, N, M, coef, dur, status, nrg, nct, nct);
  } else {
    constexpr int RT = 1;
    const int nrg = (N + 3) / 4;
    int slices = (target + nrg - 1) / nrg;
    if (slices > nct) slices = nct;
    if (slices < 1) slices = 1;
    const int cts = (nct + slices - 1) / slices;
    slices = (nct + cts - 1) / cts;
    int grid = nrg;
    if (ctx->gemm_grid_waves > 0 && ctx->gemm_grid_waves < grid) grid = ctx->gemm_grid_waves;
    note_kernel(ctx, "msnap::grid_gemm_stream_kernel<%d, %d, %d>", NC, NKS, RT);
    hipLaunchKernelGGL((grid_gemm_stream_kernel<NC, NKS, RT>), dim3(grid, slices), dim3(kWave), 0, ctx->stream, wp,
                       frag, gdur, gst, N, M, coef, dur, status, nrg, nct, cts);
  }
}

template <int NC>
static int launch_stream_nc(msnap_ctx *ctx, int N, int M, const double *wp, double *coef, double *dur,
                            int32_t *status) {
  switch ((M + 1 + 3) / 4) {
#define MSNAP_STREAM_CASE(KS) case KS: launch_stream_nks<NC, KS>(ctx, N, M, wp, coef, dur, status); break;
    MSNAP_STREAM_CASE(4) MSNAP_STREAM_CASE(5) MSNAP_STREAM_CASE(6) MSNAP_STREAM_CASE(7) MSNAP_STREAM_CASE(8)
    MSNAP_STREAM_CASE(9) MSNAP_STREAM_CASE(10) MSNAP_STREAM_CASE(11) MSNAP_STREAM_CASE(12) MSNAP_STREAM_CASE(13)
    MSNAP_STREAM_CASE(14) MSNAP_STREAM_CASE(15) MSNAP_STREAM_CASE(16)
#undef MSNAP_STREAM_CASE
    default: return MSNAP_EINVAL;   // unreachable: grid_gemm_supported / grid_gemm_reg_supported
  }
  MSNAP_HIP(ctx, hipGetLastError());
  return MSNAP_OK;
}

static int launch_solve_grid_stream(msnap_ctx *ctx, int n_drones, const double *wp, double *coef, double *dur,
                                    int32_t *status) {
  if (ctx->order == 7) return launch_stream_nc<8>(ctx, n_drones, ctx->grid_seg, wp, coef, dur, status);
  return launch_stream_nc<10>(ctx, n_drones, ctx->grid_seg, wp, coef, dur, status);
}

int launch_solve_grid(msnap_ctx *ctx, int n_drones, const double *wp, double *coef, double *dur,
                      int32_t *status) {
  const int M = ctx->grid_seg;
  if (stream_is_capturing(ctx))      // the captured launch holds pointers into the grid's blocks (ensure(), msnap.h)
    for (DevBuf *b : {&ctx->grid_t, &ctx->grid_frag, &ctx->grid_dur, &ctx->grid_status}) b->in_graph = true;
  if (!grid_gemm_supported(ctx, M)) {
    // more than 63 segments: the K1 solve on the shared grid
    return launch_solve(ctx, n_drones, M, wp, (const double *)ctx->grid_t.p, 1, coef, dur, status);
  }
  const int nc = ctx->order + 1;
  const int nrt = (n_drones + 3) / 4;
  if (!grid_gemm_reg_supported(ctx, M)) return launch_solve_grid_stream(ctx, n_drones, wp, coef, dur, status);
  // 8x the resident set: the hardware starts waves as others retire, which staggers their phases
  // (0.655 -> 0.641 ms at 2^20 drones; one wave per tile loses the operator's register residency: 0.76 ms)
  int grid = ctx->n_cu * 64;
  if (ctx->gemm_grid_waves > 0) grid = ctx->gemm_grid_waves;   // msnap_set_option
  if (grid > nrt) grid = nrt;
#define MSNAP_GRID_CASE(NCV, MM)                                                                       \
  case MM:                                                                                             \
    note_kernel(ctx, "msnap::grid_gemm_kernel<%d, %d>", NCV, MM);                                      \
    hipLaunchKernelGGL((grid_gemm_kernel<NCV, MM>), dim3(grid), dim3(kWave), 0, ctx->stream, wp,       \
                       (const double *)ctx->grid_frag.p, (const double *)ctx->grid_dur.p,              \
                       (const int32_t *)ctx->grid_status.p, n_drones, coef, dur, status, nrt);         \
    break;
  if (nc == 8) {
    switch (M) {   // grid_gemm_supported: M <= 15
      MSNAP_GRID_CASE(8, 1) MSNAP_GRID_CASE(8, 2) MSNAP_GRID_CASE(8, 3) MSNAP_GRID_CASE(8, 4)
      MSNAP_GRID_CASE(8, 5) MSNAP_GRID_CASE(8, 6) MSNAP_GRID_CASE(8, 7) MSNAP_GRID_CASE(8, 8)
      MSNAP_GRID_CASE(8, 9) MSNAP_GRID_CASE(8, 10) MSNAP_GRID_CASE(8, 11) MSNAP_GRID_CASE(8, 12)
      MSNAP_GRID_CASE(8, 13) MSNAP_GRID_CASE(8, 14) MSNAP_GRID_CASE(8, 15)
      default: return MSNAP_EINVAL;
    }
  } else {
    switch (M) {   // grid_gemm_supported: M <= 12
      MSNAP_GRID_CASE(10, 1) MSNAP_GRID_CASE(10, 2) MSNAP_GRID_CASE(10, 3) MSNAP_GRID_CASE(10, 4)
      MSNAP_GRID_CASE(10, 5) MSNAP_GRID_CASE(10, 6) MSNAP_GRID_CASE(10, 7) MSNAP_GRID_CASE(10, 8)
      MSNAP_GRID_CASE(10, 9) MSNAP_GRID_CASE(10, 10) MSNAP_GRID_CASE(10, 11) MSNAP_GRID_CASE(10, 12)
      default: return MSNAP_EINVAL;
    }
  }
#undef MSNAP_GRID_CASE
  MSNAP_HIP(ctx, hipGetLastError());
  return MSNAP_OK;
}

}  // namespace msnap
