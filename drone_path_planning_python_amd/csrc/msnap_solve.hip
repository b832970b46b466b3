// K1 -- batched minimum-snap solve, one lane per (drone, axis), gfx950.
//
// Replaces calculate_trajectory1D / calculate_trajectory4D
// (reference src/optimizations/calculatingTrajectories.py:37-213) for a batch.
//
// Algorithm (DESIGN.md "K1"): the reference's square 8M x 8M collocation system
// has the unique solution of the minimum-snap QP.  Writing every segment in
// Hermite form (endpoint position + k-1 derivatives, k = 4 for order 7)
// satisfies the waypoint, endpoint and C^1..C^(k-1) rows identically; the
// remaining rows (continuity of d^k..d^(2k-2), calculatingTrajectories.py:115-119)
// are, up to sign, the stationarity conditions dJ/d(d_i[n]) = 0 of the snap
// cost J = sum_seg T^-(2k-1) e^T H e.  That reduced KKT system is symmetric
// positive definite and block tridiagonal with (k-1) x (k-1) blocks:
//     O_{i-1}^T u_{i-1} + D_i u_i + O_i u_{i+1} = r_i ,   i = 1..M-1
//     D_i = T_{i-1}^(n+m-K) HEE + T_i^(n+m-K) HSS ,  O_i = T_i^(n+m-K) HSE
// It is solved by a block LDL^T sweep (forward: S_i, G_i = S_i^-1 O_i,
// z_i = S_i^-1 y_i; backward: u_i = z_i - G_i u_{i+1}) and each segment's
// monomial coefficients are recovered from its two endpoint states with the
// constant matrices CS / CE.
//
// Mapping: lane = 4*dl + axis; a wavefront carries 16 drones.  The matrix part
// (S_i, G_i) depends on the time grid only and is recomputed by the 4 axis
// lanes of a drone (no cross-lane traffic, no divergence: every lane runs the
// same M-step recurrence).  Per-knot G_i (once per drone) and z_i (per lane)
// live in LDS; when n_seg is too large for 160 KiB the same code runs on a
// global-memory scratch slab (GS = true).
#include "msnap_consts.h"
#include "msnap_internal.h"

namespace msnap {

template <int NU>
__device__ __forceinline__ constexpr int sidx(int r, int c) {  // r >= c
  return r * (r + 1) / 2 + c;
}

__device__ __forceinline__ double rcp64(double v) {
#ifdef MSNAP_EXACT_DIV
  return 1.0 / v;
#else
  // v_rcp_f64 seed + two Newton steps; inputs are durations / SPD pivots in a
  // sane range (status flags catch the rest), so no denormal/overflow fix-up.
  double r = __builtin_amdgcn_rcp(v);
  double e = __builtin_fma(-v, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-v, r, 1.0);
  r = __builtin_fma(r, e, r);
  return r;
#endif
}

__device__ __forceinline__ bool finite64(double v) { return __builtin_isfinite(v); }

template <int K, bool GS>
__global__ void __launch_bounds__(kWave)
solve_kernel(const double *__restrict__ wp, const double *__restrict__ tt, int shared_times,
             int N, int M, double *__restrict__ coef, double *__restrict__ dur,
             int32_t *__restrict__ status, double *__restrict__ gscratch, int ntiles) {
  constexpr int NU = K - 1;           // unknown derivatives per interior knot
  constexpr int NC = 2 * K;           // coefficients per segment
  constexpr int KK = 2 * K - 1;       // polynomial order
  constexpr int NS = NU * (NU + 1) / 2;
  constexpr int PM = 2 * K - 2;       // highest power of 1/T needed in the sweep
  using C = HermiteConsts<K>;

  extern __shared__ __attribute__((aligned(16))) double lds[];

  const int lane = threadIdx.x;
  const int dl = lane >> 2;
  const int a = lane & 3;
  const int knots = M - 1;

  // scratch carve (doubles): W[(M+1)][64] | T[M][16] | X[M][16] | G[knots][NU*NU][16] | Z[knots][NU][64]
  double *scr;
  if constexpr (GS) {
    scr = gscratch + (size_t)blockIdx.x *
                         (size_t)(64 * (M + 1) + 32 * M + 16 * NU * NU * knots + 64 * NU * knots);
  } else {
    scr = lds;
  }
  double *sW = scr;
  double *sT = sW + 64 * (M + 1);
  double *sX = sT + 16 * M;
  double *sG = sX + 16 * M;
  double *sZ = sG + 16 * NU * NU * knots;

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int d_raw = tile * kDronesPerWave + dl;
    const bool live = d_raw < N;
    const int d = live ? d_raw : N - 1;
    const double *wrow = wp + (size_t)d * (size_t)(M + 1) * 4 + a;
    const double *trow = shared_times ? tt : tt + (size_t)d * (size_t)(M + 1);
    double *drow = dur + (size_t)d * M;

    // ---------------- segment 0: only its end side feeds knot 1 ----------------
    const double t0 = trow[0];
    double tcur = trow[1];
    double wprev = wrow[0];
    double wcur = wrow[4];
    bool nonfinite = !(finite64(t0) && finite64(tcur) && finite64(wprev) && finite64(wcur));
    double T = tcur - t0;
    // Appendix-A quirk of the reference (calculatingTrajectories.py:59,65-73):
    // the start rows are evaluated at local time t[0], the end of segment 0 at
    // t[1]-t[0]; segment 0 is therefore a Hermite piece of length T - t0 in the
    // shifted variable s - t0 (identity when t[0] == 0).
    double Teff = T - t0;
    bool badtime = !(T > 0.0) || !(Teff > 0.0) || (t0 < 0.0);
    bool singular = false;
    if (live && a == 0) drow[0] = T;
    double x = rcp64(Teff);
    sW[0 * 64 + lane] = wprev;
    sW[1 * 64 + lane] = wcur;
    sT[0 * 16 + dl] = Teff;
    sX[0 * 16 + dl] = x;

    double xp[PM + 1];
    xp[0] = 1.0;
    xp[1] = x;
#pragma unroll
    for (int p = 2; p <= PM; ++p) xp[p] = xp[p - 1] * x;

    double dw = wcur - wprev;
    double E[NS], re[NU];
#pragma unroll
    for (int n = 1; n <= NU; ++n) {
#pragma unroll
      for (int m = 1; m <= n; ++m) E[sidx<NU>(n - 1, m - 1)] = C::HEE[n][m] * xp[KK - n - m];
      re[n - 1] = (C::HEE[n][0] * xp[KK - n]) * dw;
    }

    double Op[NU][NU], Gp[NU][NU], zp[NU];
#pragma unroll
    for (int r = 0; r < NU; ++r) {
      zp[r] = 0.0;
#pragma unroll
      for (int c = 0; c < NU; ++c) {
        Op[r][c] = 0.0;
        Gp[r][c] = 0.0;
      }
    }

    // ---------------- forward block LDL^T sweep over interior knots ----------------
    for (int i = 1; i < M; ++i) {
      const double tnext = trow[i + 1];
      const double wnext = wrow[(size_t)(i + 1) * 4];
      nonfinite = nonfinite || !finite64(tnext) || !finite64(wnext);
      T = tnext - tcur;
      badtime = badtime || !(T > 0.0);
      if (live && a == 0) drow[i] = T;
      x = rcp64(T);
      sW[(i + 1) * 64 + lane] = wnext;
      sT[i * 16 + dl] = T;
      sX[i * 16 + dl] = x;
      xp[1] = x;
#pragma unroll
      for (int p = 2; p <= PM; ++p) xp[p] = xp[p - 1] * x;
      dw = wnext - wcur;

      // S_i = D_i - O_{i-1}^T G_{i-1}   (symmetric, lower triangle)
      double S[NS], y[NU];
#pragma unroll
      for (int n = 1; n <= NU; ++n) {
#pragma unroll
        for (int m = 1; m <= n; ++m) {
          double v = __builtin_fma(C::HSS[n][m], xp[KK - n - m], E[sidx<NU>(n - 1, m - 1)]);
#pragma unroll
          for (int q = 0; q < NU; ++q) v = __builtin_fma(-Op[q][n - 1], Gp[q][m - 1], v);
          S[sidx<NU>(n - 1, m - 1)] = v;
        }
        double rv = -__builtin_fma(C::HSE[n][0] * xp[KK - n], dw, re[n - 1]);
#pragma unroll
        for (int q = 0; q < NU; ++q) rv = __builtin_fma(-Op[q][n - 1], zp[q], rv);
        y[n - 1] = rv;
      }

      // LDL^T of S: L unit lower (stored in S's strict lower part), dinv
      double dinv[NU];
#pragma unroll
      for (int j = 0; j < NU; ++j) {
        double dj = S[sidx<NU>(j, j)];
#pragma unroll
        for (int p = 0; p < j; ++p) {
          // S(j,p) holds w_jp = L_jp * d_p until scaled below
          dj = __builtin_fma(-S[sidx<NU>(j, p)] * dinv[p], S[sidx<NU>(j, p)], dj);
        }
        singular = singular || !(dj > 0.0) || !finite64(dj);
        dinv[j] = rcp64(dj);
#pragma unroll
        for (int r = j + 1; r < NU; ++r) {
          double v = S[sidx<NU>(r, j)];
#pragma unroll
          for (int p = 0; p < j; ++p) v = __builtin_fma(-S[sidx<NU>(r, p)] * dinv[p], S[sidx<NU>(j, p)], v);
          S[sidx<NU>(r, j)] = v;  // w_rj = L_rj d_j
        }
      }
      // convert w -> L
#pragma unroll
      for (int r = 1; r < NU; ++r)
#pragma unroll
        for (int p = 0; p < r; ++p) S[sidx<NU>(r, p)] *= dinv[p];

      // O_i and the NU+1 solves  G_i = S^-1 O_i ,  z_i = S^-1 y
      double O[NU][NU], G[NU][NU], z[NU];
#pragma unroll
      for (int n = 1; n <= NU; ++n)
#pragma unroll
        for (int m = 1; m <= NU; ++m) O[n - 1][m - 1] = C::HSE[n][m] * xp[KK - n - m];

#pragma unroll
      for (int c = 0; c <= NU; ++c) {
        double v[NU];
#pragma unroll
        for (int r = 0; r < NU; ++r) v[r] = (c < NU) ? O[r][c < NU ? c : 0] : y[r];
        // forward  L v' = v
#pragma unroll
        for (int r = 1; r < NU; ++r)
#pragma unroll
          for (int p = 0; p < r; ++p) v[r] = __builtin_fma(-S[sidx<NU>(r, p)], v[p], v[r]);
#pragma unroll
        for (int r = 0; r < NU; ++r) v[r] *= dinv[r];
        // backward L^T v'' = v'
#pragma unroll
        for (int r = NU - 2; r >= 0; --r)
#pragma unroll
          for (int p = r + 1; p < NU; ++p) v[r] = __builtin_fma(-S[sidx<NU>(p, r)], v[p], v[r]);
#pragma unroll
        for (int r = 0; r < NU; ++r) {
          if (c < NU) G[r][c < NU ? c : 0] = v[r];
          else z[r] = v[r];
        }
      }

      // stash for the backward sweep
      double *g = sG + (size_t)(i - 1) * (NU * NU * 16) + dl;
      double *zz = sZ + (size_t)(i - 1) * (NU * 64) + lane;
#pragma unroll
      for (int r = 0; r < NU; ++r) {
#pragma unroll
        for (int c = 0; c < NU; ++c) g[(r * NU + c) * 16] = G[r][c];
        zz[r * 64] = z[r];
      }

      // carry: end side of segment i feeds knot i+1
#pragma unroll
      for (int n = 1; n <= NU; ++n) {
#pragma unroll
        for (int m = 1; m <= n; ++m) E[sidx<NU>(n - 1, m - 1)] = C::HEE[n][m] * xp[KK - n - m];
        re[n - 1] = (C::HEE[n][0] * xp[KK - n]) * dw;
      }
#pragma unroll
      for (int r = 0; r < NU; ++r) {
        zp[r] = z[r];
#pragma unroll
        for (int c = 0; c < NU; ++c) {
          Op[r][c] = O[r][c];
          Gp[r][c] = G[r][c];
        }
      }
      tcur = tnext;
      wcur = wnext;
    }

    // ---------------- per-drone status (combine the 4 axis lanes) ----------------
    int flags = (nonfinite ? 4 : 0) | (badtime ? 2 : 0) | (singular ? 1 : 0);
    flags |= __shfl_xor(flags, 1);
    flags |= __shfl_xor(flags, 2);
    const int st = (flags & 4) ? MSNAP_ST_NONFINITE
                   : (flags & 2) ? MSNAP_ST_TIMES
                   : (flags & 1) ? MSNAP_ST_SINGULAR
                                 : MSNAP_ST_OK;
    if (live && a == 0) status[d] = st;
    const bool bad = st != 0;

    if constexpr (!GS) __syncthreads();  // one-wave workgroup: orders the LDS stash

    // ---------------- backward sweep + coefficient recovery ----------------
    double un[NU];
#pragma unroll
    for (int r = 0; r < NU; ++r) un[r] = 0.0;
    double wn = wcur;  // w_M
    for (int i = M - 1; i >= 0; --i) {
      double u[NU];
      if (i >= 1) {
        const double *zz = sZ + (size_t)(i - 1) * (NU * 64) + lane;
#pragma unroll
        for (int r = 0; r < NU; ++r) u[r] = zz[r * 64];
        if (i < M - 1) {
          const double *g = sG + (size_t)(i - 1) * (NU * NU * 16) + dl;
#pragma unroll
          for (int r = 0; r < NU; ++r)
#pragma unroll
            for (int c = 0; c < NU; ++c) u[r] = __builtin_fma(-g[(r * NU + c) * 16], un[c], u[r]);
        }
      } else {
#pragma unroll
        for (int r = 0; r < NU; ++r) u[r] = 0.0;
      }
      const double wi = sW[i * 64 + lane];
      const double Ti = sT[i * 16 + dl];
      const double xi = sX[i * 16 + dl];
      const double dwi = wn - wi;

      double tp[K];        // Ti^n, n < K
      tp[0] = 1.0;
#pragma unroll
      for (int n = 1; n < K; ++n) tp[n] = tp[n - 1] * Ti;
      double xq[NC];       // xi^m
      xq[0] = 1.0;
#pragma unroll
      for (int m = 1; m < NC; ++m) xq[m] = xq[m - 1] * xi;

      double c[NC];
      c[0] = wi;
#pragma unroll
      for (int n = 1; n < K; ++n) c[n] = u[n - 1] * C::INVFACT[n];
      double es[K], ee[K];
#pragma unroll
      for (int n = 1; n < K; ++n) {
        es[n] = tp[n] * u[n - 1];
        ee[n] = tp[n] * un[n - 1];
      }
#pragma unroll
      for (int m = 0; m < K; ++m) {
        double acc = C::CE[m][0] * dwi;
#pragma unroll
        for (int n = 1; n < K; ++n) {
          acc = __builtin_fma(C::CS[m][n], es[n], acc);
          acc = __builtin_fma(C::CE[m][n], ee[n], acc);
        }
        c[K + m] = acc * xq[K + m];
      }

      if (i == 0 && t0 != 0.0) {
        // p(s) = q(s - t0): Taylor shift of segment 0 (reference quirk, see above)
        const double h = -t0;
#pragma unroll
        for (int j = 0; j < NC - 1; ++j)
#pragma unroll
          for (int q = NC - 2; q >= j; --q) c[q] = __builtin_fma(h, c[q + 1], c[q]);
      }

      if (bad) {
#pragma unroll
        for (int m = 0; m < NC; ++m) c[m] = __builtin_nan("");
      }
      if (live) {
        double *o = coef + (((size_t)d * M + i) * 4 + a) * NC;
#pragma unroll
        for (int m = 0; m < NC; m += 2) {
          double2 v2 = make_double2(c[m], c[m + 1]);
          *reinterpret_cast<double2 *>(o + m) = v2;
        }
      }
#pragma unroll
      for (int r = 0; r < NU; ++r) un[r] = u[r];
      wn = wi;
    }
    if constexpr (!GS) __syncthreads();  // scratch is reused by the next tile
  }
}

template <int K>
static int launch_solve_k(msnap_ctx *ctx, int N, int M, const double *wp, const double *t, int shared,
                          double *coef, double *dur, int32_t *status) {
  const int ntiles = (N + kDronesPerWave - 1) / kDronesPerWave;
  const size_t words = solve_scratch_words(K, M);
  const size_t bytes = words * sizeof(double);
  if (bytes <= kMaxLdsBytes) {
    hipLaunchKernelGGL((solve_kernel<K, false>), dim3(ntiles), dim3(kWave), bytes, ctx->stream, wp, t,
                       shared, N, M, coef, dur, status, (double *)nullptr, ntiles);
  } else {
    // too many segments for LDS: same recurrence on a global scratch slab,
    // persistent grid so the slab stays bounded
    int grid = ctx->n_cu * 8;
    if (grid > ntiles) grid = ntiles;
    int rc = ensure(ctx, ctx->scratch, (size_t)grid * bytes);
    if (rc) return rc;
    hipLaunchKernelGGL((solve_kernel<K, true>), dim3(grid), dim3(kWave), 0, ctx->stream, wp, t, shared,
                       N, M, coef, dur, status, (double *)ctx->scratch.p, ntiles);
  }
  MSNAP_HIP(ctx, hipGetLastError());
  return MSNAP_OK;
}

int solve_kernel_setup(msnap_ctx *ctx) {
  // allow the full 160 KiB of dynamic LDS
  MSNAP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&solve_kernel<4, false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes));
  MSNAP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&solve_kernel<5, false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes));
  return MSNAP_OK;
}

int launch_solve(msnap_ctx *ctx, int n_drones, int n_seg, const double *wp, const double *t,
                 int shared_times, double *coef, double *dur, int32_t *status) {
  if (n_drones == 0) return MSNAP_OK;
  if (ctx->khalf == 4)
    return launch_solve_k<4>(ctx, n_drones, n_seg, wp, t, shared_times, coef, dur, status);
  return launch_solve_k<5>(ctx, n_drones, n_seg, wp, t, shared_times, coef, dur, status);
}

}  // namespace msnap
