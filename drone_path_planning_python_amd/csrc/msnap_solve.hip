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
// same M-step recurrence).  The tile's inputs are staged in LDS by one
// coalesced sweep.  The variants share the arithmetic (Sweep<K>::knot_geom / chain, recover_segment)
// and are chosen per launch by launch_solve_k:
//   solve_kernel_twist<K,H,M>  2 <= n_seg <= 24 (order 7) / 12 (order 9), batch up to one 8-drone
//                              wavefront per SIMD: two-sided sweep (halves the dependent chain)
//                              meeting at a shared knot, one straight-line instance per
//                              (order, n_seg)                                (latency path)
//   solve_kernel_reg<K,10|20>  n_seg <= 20: knot loops unrolled, path data and z_i in
//                              registers, G_i in LDS, persistent waves with cross-tile
//                              input prefetch                               (throughput path)
//   solve_kernel<K, false>     any n_seg whose stash fits 160 KiB: rolled loops, LDS stash
//   solve_kernel<K, true>      longer paths: stash on a global scratch slab
#include "msnap_consts.h"
#include <cstdlib>

#include "msnap_internal.h"

namespace msnap {

__device__ __forceinline__ constexpr int sidx(int r, int c) {  // symmetric lower, r >= c
  return r * (r + 1) / 2 + c;
}

__device__ __forceinline__ double rcp64(double v) {
  // v_rcp_f64 seed (2^-24) + two Newton steps (1 ulp, tools/micro/rcp_micro.hip); inputs are
  // durations / SPD pivots in a sane range (status flags catch the rest), so no denormal/overflow fix-up.
  double r = __builtin_amdgcn_rcp(v);
  double e = __builtin_fma(-v, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-v, r, 1.0);
  r = __builtin_fma(r, e, r);
  return r;
}

__device__ __forceinline__ bool finite64(double v) { return __builtin_isfinite(v); }

// ------------------------------------------------------------------------------------
// the forward recurrence carried from knot to knot
// ------------------------------------------------------------------------------------
template <int K>
struct Sweep {
  static constexpr int NU = K - 1;            // unknown derivatives per interior knot
  static constexpr int NC = 2 * K;            // coefficients per segment
  static constexpr int KK = 2 * K - 1;        // polynomial order
  static constexpr int NS = NU * (NU + 1) / 2;
  static constexpr int PM = 2 * K - 2;        // highest power of 1/T in the sweep
  using C = HermiteConsts<K>;

  double E[NS], re[NU];                       // end side of the previous segment
  double OtG[NS], Otz[NU];                    // O_{i-1}^T G_{i-1} (symmetric) and O_{i-1}^T z_{i-1}
  bool singular;

  __device__ __forceinline__ static void powers(double x, double (&xp)[PM + 1]) {
    xp[0] = 1.0;
    xp[1] = x;
#pragma unroll
    for (int p = 2; p <= PM; ++p) xp[p] = xp[p - 1] * x;
  }

  __device__ __forceinline__ void end_side(const double (&xp)[PM + 1], double dw) {
#pragma unroll
    for (int n = 1; n <= NU; ++n) {
#pragma unroll
      for (int m = 1; m <= n; ++m) E[sidx(n - 1, m - 1)] = C::HEE[n][m] * xp[KK - n - m];
      re[n - 1] = (C::HEE[n][0] * xp[KK - n]) * dw;
    }
  }

  // segment 0 (x = 1/T_0, dw = w_1 - w_0): only its end side feeds knot 1
  __device__ __forceinline__ void init(double x, double dw) {
    double xp[PM + 1];
    powers(x, xp);
    end_side(xp, dw);
    singular = false;
#pragma unroll
    for (int r = 0; r < NU; ++r) Otz[r] = 0.0;
#pragma unroll
    for (int e = 0; e < NS; ++e) OtG[e] = 0.0;
  }

  // Everything of knot i that depends on segment lengths and waypoints only (not on the
  // recurrence): diagonal block D_i, coupling block O_i, the right-hand side before the Schur
  // correction, and the end side of segment i for knot i+1.
  struct Knot {
    double D[NS], O[NU][NU], yb[NU], E[NS], re[NU];
  };

  // from segment i (x = 1/T_i, dw = w_{i+1} - w_i) and the end side (Eprev, reprev) of segment i-1
  __device__ __forceinline__ static void knot_geom(double x, double dw, const double (&Eprev)[NS],
                                                   const double (&reprev)[NU], Knot &k) {
    double xp[PM + 1];
    powers(x, xp);
    knot_geom_xp(xp, dw, Eprev, reprev, k);
  }

  // the same from the powers of x = 1/T_i (for callers that prepare them ahead of the recurrence)
  __device__ __forceinline__ static void knot_geom_xp(const double (&xp)[PM + 1], double dw,
                                                      const double (&Eprev)[NS], const double (&reprev)[NU],
                                                      Knot &k) {
#pragma unroll
    for (int n = 1; n <= NU; ++n) {
#pragma unroll
      for (int m = 1; m <= n; ++m) {
        k.D[sidx(n - 1, m - 1)] = __builtin_fma(C::HSS[n][m], xp[KK - n - m], Eprev[sidx(n - 1, m - 1)]);
        k.E[sidx(n - 1, m - 1)] = C::HEE[n][m] * xp[KK - n - m];
      }
      k.yb[n - 1] = __builtin_fma(C::HSE[n][0] * xp[KK - n], dw, reprev[n - 1]);
      k.re[n - 1] = (C::HEE[n][0] * xp[KK - n]) * dw;
#pragma unroll
      for (int m = 1; m <= NU; ++m) k.O[n - 1][m - 1] = C::HSE[n][m] * xp[KK - n - m];
    }
  }

  // LDL^T of the symmetric block S in place (strict lower part -> L, dinv -> 1/d); returns true
  // if a pivot is not positive and finite
  __device__ __forceinline__ static bool ldl_factor(double (&S)[NS], double (&dinv)[NU]) {
    bool bad = false;
#pragma unroll
    for (int j = 0; j < NU; ++j) {
      double dj = S[sidx(j, j)];
#pragma unroll
      for (int p = 0; p < j; ++p) dj = __builtin_fma(-S[sidx(j, p)] * dinv[p], S[sidx(j, p)], dj);
      bad |= !(dj > 0.0) | !finite64(dj);   // bitwise: no short-circuit branches
      dinv[j] = rcp64(dj);
#pragma unroll
      for (int r = j + 1; r < NU; ++r) {
        double v = S[sidx(r, j)];
#pragma unroll
        for (int p = 0; p < j; ++p) v = __builtin_fma(-S[sidx(r, p)] * dinv[p], S[sidx(j, p)], v);
        S[sidx(r, j)] = v;     // w_rj = L_rj d_j until scaled below
      }
    }
#pragma unroll
    for (int r = 1; r < NU; ++r)
#pragma unroll
      for (int p = 0; p < r; ++p) S[sidx(r, p)] *= dinv[p];
    return bad;
  }

  // v <- S^-1 v with the factor of ldl_factor
  __device__ __forceinline__ static void ldl_solve(const double (&S)[NS], const double (&dinv)[NU], double (&v)[NU]) {
#pragma unroll
    for (int r = 1; r < NU; ++r)
#pragma unroll
      for (int p = 0; p < r; ++p) v[r] = __builtin_fma(-S[sidx(r, p)], v[p], v[r]);
#pragma unroll
    for (int r = 0; r < NU; ++r) v[r] *= dinv[r];
#pragma unroll
    for (int r = NU - 2; r >= 0; --r)
#pragma unroll
      for (int p = r + 1; p < NU; ++p) v[r] = __builtin_fma(-S[sidx(p, r)], v[p], v[r]);
  }

  // the recurrence proper: S_i = D_i - O_{i-1}^T G_{i-1}, LDL^T, G_i = S_i^-1 O_i, z_i = S_i^-1 y_i,
  // and the Schur terms carried to knot i+1.  Returns false if a pivot is not positive and finite.
  __device__ __forceinline__ bool chain(const Knot &k, double (&G)[NU][NU], double (&z)[NU]) {
    double S[NS], y[NU];
#pragma unroll
    for (int n = 0; n < NU; ++n) {
#pragma unroll
      for (int m = 0; m <= n; ++m) S[sidx(n, m)] = k.D[sidx(n, m)] - OtG[sidx(n, m)];
      y[n] = -k.yb[n] - Otz[n];
    }

    double dinv[NU];
    const bool bad = ldl_factor(S, dinv);

    // NU + 1 solves with the factor: columns of O_i, then y
#pragma unroll
    for (int c = 0; c <= NU; ++c) {
      double v[NU];
#pragma unroll
      for (int r = 0; r < NU; ++r) v[r] = (c < NU) ? k.O[r][c < NU ? c : 0] : y[r];
      ldl_solve(S, dinv, v);
#pragma unroll
      for (int r = 0; r < NU; ++r) {
        if (c < NU) G[r][c < NU ? c : 0] = v[r];
        else z[r] = v[r];
      }
    }

    // carry to knot i+1: the Schur terms O_i^T G_i, O_i^T z_i
#pragma unroll
    for (int n = 0; n < NU; ++n) {
#pragma unroll
      for (int m = 0; m <= n; ++m) {
        double v = 0.0;
#pragma unroll
        for (int q = 0; q < NU; ++q) v = __builtin_fma(k.O[q][n], G[q][m], v);
        OtG[sidx(n, m)] = v;
      }
      double w = 0.0;
#pragma unroll
      for (int q = 0; q < NU; ++q) w = __builtin_fma(k.O[q][n], z[q], w);
      Otz[n] = w;
    }
    return !bad;
  }

  // knot i with segment i (x = 1/T_i, dw = w_{i+1} - w_i) on its right: geometry, then recurrence
  __device__ __forceinline__ void step(double x, double dw, double (&G)[NU][NU], double (&z)[NU]) {
    Knot k;
    knot_geom(x, dw, E, re, k);
    singular |= !chain(k, G, z);
#pragma unroll
    for (int e = 0; e < NS; ++e) E[e] = k.E[e];
#pragma unroll
    for (int r = 0; r < NU; ++r) re[r] = k.re[r];
  }
};

// monomial coefficients of one segment from its endpoint states
// (u = d_i[1..k-1], un = d_{i+1}[1..k-1], xi = 1/T_i, dwi = w_{i+1} - w_i):
//   c_{K+m} = x^(K+m) [ CE_m0 dw + sum_n T^n (CS_mn u_n + CE_mn un_n) ]
//           = x^(m+1) [ CE_m0 dw x^(K-1) + sum_n (CS_mn x^(K-1-n) u_n + CE_mn x^(K-1-n) un_n) ]
// -- powers of x only, so callers need not keep T_i.
template <int K>
__device__ __forceinline__ void recover_segment(double wi, double dwi, double xi, const double (&u)[K - 1],
                                                const double (&un)[K - 1], double (&c)[2 * K]) {
  using C = HermiteConsts<K>;
  double xq[K + 1];
  xq[0] = 1.0;
#pragma unroll
  for (int m = 1; m <= K; ++m) xq[m] = xq[m - 1] * xi;
  c[0] = wi;
#pragma unroll
  for (int n = 1; n < K; ++n) c[n] = u[n - 1] * C::INVFACT[n];
  double es[K], ee[K];
#pragma unroll
  for (int n = 1; n < K; ++n) {
    es[n] = (n == K - 1) ? u[n - 1] : xq[K - 1 - n] * u[n - 1];
    ee[n] = (n == K - 1) ? un[n - 1] : xq[K - 1 - n] * un[n - 1];
  }
  const double dwx = dwi * xq[K - 1];
#pragma unroll
  for (int m = 0; m < K; ++m) {
    double acc = C::CE[m][0] * dwx;
#pragma unroll
    for (int n = 1; n < K; ++n) {
      acc = __builtin_fma(C::CS[m][n], es[n], acc);
      acc = __builtin_fma(C::CE[m][n], ee[n], acc);
    }
    c[K + m] = acc * xq[m + 1];
  }
}

// p(s) = q(s - t0): the reference evaluates the start rows of segment 0 at local
// time t[0] (calculatingTrajectories.py:59,65-73), so that piece is a Hermite
// segment of length T_0 - t0 in the shifted variable (identity for t[0] == 0).
template <int NC>
__device__ __forceinline__ void taylor_shift(double (&c)[NC], double h) {
#pragma unroll
  for (int j = 0; j < NC - 1; ++j)
#pragma unroll
    for (int q = NC - 2; q >= j; --q) c[q] = __builtin_fma(h, c[q + 1], c[q]);
}

template <int NC>
__device__ __forceinline__ void store_segment(double *__restrict__ o, double (&c)[NC], bool bad, bool live) {
  if (bad) {
#pragma unroll
    for (int m = 0; m < NC; ++m) c[m] = __builtin_nan("");
  }
  if (live) {
#pragma unroll
    for (int m = 0; m < NC; m += 2) *reinterpret_cast<double2 *>(o + m) = make_double2(c[m], c[m + 1]);
  }
}

// Full-line output stores.  A lane owns the NC coefficients of one (drone, axis):
// stored directly, a wave instruction would scatter 64 x 16 B over 64 different
// 64-byte segments.  Instead the segment's 64 x NC doubles take a round trip
// through an LDS image [NC/2 rows][68 slots of 16 B] (row pitch 68 keeps
// ds_read_b128 conflict-free for NC = 8) and leave as 16-B-per-lane stores that
// are contiguous over each drone's 4*NC*8-byte block (256 B = two full lines).
// LDS is in-order within a wave; the wavefront-scope fences only pin the
// compiler's ordering (no vmcnt wait: output stores stay in flight).
constexpr int kTrPitch = 68;
#define MSNAP_SEG_BASE(coef, tile, M, i, NC) ((coef) + ((size_t)(tile) * kDronesPerWave * (M) + (i)) * (4 * (NC)))
#define MSNAP_SEG_STRIDE(M, NC) ((size_t)(M) * 4 * (NC))
template <int NC>
__device__ __forceinline__ void store_segment_coalesced(double2 *sTr, double *__restrict__ seg_base,
                                                        size_t drone_stride, int nvalid, int lane,
                                                        double (&c)[NC], bool bad) {
  constexpr int NJ = NC / 2;
  // a failed drone is rare: one wave-uniform test instead of 2 * NC selects per segment
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(bad) != 0, 0)) {
    asm volatile("" ::: "memory");   // keep the block a branch: the compiler would flatten it into selects again
#pragma unroll
    for (int m = 0; m < NC; ++m) c[m] = bad ? __builtin_nan("") : c[m];
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) sTr[j * kTrPitch + lane] = make_double2(c[2 * j], c[2 * j + 1]);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
#pragma unroll
  for (int q = 0; q < NJ; ++q) {
    const int s = q * kWave + lane;          // flat 16-byte slot of the tile-segment
    const int drone = s / (4 * NJ);
    const int within = s - drone * (4 * NJ);
    const int a2 = within / NJ;
    const int j2 = within - a2 * NJ;
    const double2 v = sTr[j2 * kTrPitch + drone * 4 + a2];
    // The lanes past the batch end replay the tile's last valid drone (same inputs, same instruction
    // stream, bitwise the same coefficients), so their slots are stored ON TOP of that drone's instead
    // of being masked off: no exec-mask region and branch pair per store.
    const int dst = drone < nvalid ? drone : nvalid - 1;
    *reinterpret_cast<double2 *>(seg_base + (size_t)dst * drone_stride + within * 2) = v;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
}

// Order-7 variant of the same idea without LDS: the 4 axis lanes of a drone hold a
// 4 x 4 grid of 16-byte pieces (lane = axis, piece = coefficient pair) of the drone's
// 256-byte block.  Two butterfly stages of quad-local exchanges transpose the grid, after
// which store q of lane a carries (axis q, pair a): the quad writes 64 contiguous bytes per
// instruction (whole 64-byte segments), with no LDS round trip and no wait on the critical
// path.  A stage exchanges (A, B) pairs between lanes that differ in one bit of the axis:
// lanes with the bit keep B and take the partner's B as their A, lanes without keep A and
// take the partner's A as their B.  Written as select-with-DPP-source, one instruction per
// dword and side:   B' = bit ? B : dpp(A)      (v_cndmask_b32_dpp, vcc = bit)
//                   A' = !bit ? A : dpp(B)     (v_cndmask_b32_dpp, vcc = !bit)
// -- 32 vector instructions per segment; select / v_mov_dpp / select, as the compiler emits
// the same exchange from C++, takes 64 (a quarter of the order-7 backward sweep).
// The leading s_nop covers the 2 wait states a DPP read needs after a VALU write.
#define MSNAP_QCND(d, s0, s1, QP) \
  "v_cndmask_b32_dpp %" #d ", %" #s0 ", %" #s1 ", vcc quad_perm:" QP " row_mask:0xf bank_mask:0xf\n\t"
#define MSNAP_QSTAGE(QP, out, src, keep, mask)                                                                     \
  asm("s_nop 1\n\ts_mov_b64 vcc, %16\n\t" MSNAP_QCND(0, 8, 17, QP) MSNAP_QCND(1, 9, 18, QP) MSNAP_QCND(2, 10, 19, QP) \
          MSNAP_QCND(3, 11, 20, QP) MSNAP_QCND(4, 12, 21, QP) MSNAP_QCND(5, 13, 22, QP) MSNAP_QCND(6, 14, 23, QP)      \
              MSNAP_QCND(7, 15, 24, QP)                                                                            \
      : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3]), "=&v"(out[4]), "=&v"(out[5]), "=&v"(out[6]),    \
        "=&v"(out[7])                                                                                              \
      : "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(src[3]), "v"(src[4]), "v"(src[5]), "v"(src[6]), "v"(src[7]),     \
        "s"(mask), "v"(keep[0]), "v"(keep[1]), "v"(keep[2]), "v"(keep[3]), "v"(keep[4]), "v"(keep[5]), "v"(keep[6]), \
        "v"(keep[7])                                                                                               \
      : "vcc")
// a, b: 4 doubles each as dwords (lo, hi); `bit` as a lane mask
template <int STAGE>   // 1: partner = lane ^ 1, 2: partner = lane ^ 2
__device__ __forceinline__ void quad_stage(uint32_t (&a)[8], uint32_t (&b)[8], unsigned long long bit) {
  uint32_t na[8], nb[8];
  const unsigned long long nbit = ~bit;
  if constexpr (STAGE == 1) {
    MSNAP_QSTAGE("[1,0,3,2]", nb, a, b, bit);
    MSNAP_QSTAGE("[1,0,3,2]", na, b, a, nbit);
  } else {
    MSNAP_QSTAGE("[2,3,0,1]", nb, a, b, bit);
    MSNAP_QSTAGE("[2,3,0,1]", na, b, a, nbit);
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    a[k] = na[k];
    b[k] = nb[k];
  }
}
#undef MSNAP_QSTAGE
#undef MSNAP_QCND

// `blk` is this lane's drone-segment block (4 axes x 8 coefficients); all 64 lanes must be active
// (the exchanges are quad-local and read their partners through DPP).
__device__ __forceinline__ void store_quad8_at(double *__restrict__ blk, int a, const double (&c)[8], bool bad) {
  double p[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) p[k] = c[k];
  // a failed drone is rare: one wave-uniform test instead of 16 selects per segment
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(bad) != 0, 0)) {
    asm volatile("" ::: "memory");   // keep the block a branch: the compiler would flatten it into selects again
#pragma unroll
    for (int k = 0; k < 8; ++k) p[k] = bad ? __builtin_nan("") : p[k];
  }
  auto lo = [](double v) { return (uint32_t)__double2loint(v); };
  auto hi = [](double v) { return (uint32_t)__double2hiint(v); };
  // pieces P0..P3 = coefficient pairs (0,1) (2,3) (4,5) (6,7).  Stage 1 pairs (P0,P1) and (P2,P3):
  uint32_t A[8] = {lo(p[0]), hi(p[0]), lo(p[1]), hi(p[1]), lo(p[4]), hi(p[4]), lo(p[5]), hi(p[5])};   // P0 | P2
  uint32_t B[8] = {lo(p[2]), hi(p[2]), lo(p[3]), hi(p[3]), lo(p[6]), hi(p[6]), lo(p[7]), hi(p[7])};   // P1 | P3
  quad_stage<1>(A, B, __builtin_amdgcn_ballot_w64((a & 1) != 0));
  // stage 2 pairs (P0,P2) and (P1,P3)
  uint32_t A2[8] = {A[0], A[1], A[2], A[3], B[0], B[1], B[2], B[3]};   // P0 | P1
  uint32_t B2[8] = {A[4], A[5], A[6], A[7], B[4], B[5], B[6], B[7]};   // P2 | P3
  quad_stage<2>(A2, B2, __builtin_amdgcn_ballot_w64((a & 2) != 0));
  // now piece q = coefficient pair `a` of axis q
  uint4 *o = reinterpret_cast<uint4 *>(blk + a * 2);
  o[0] = make_uint4(A2[0], A2[1], A2[2], A2[3]);
  o[4] = make_uint4(A2[4], A2[5], A2[6], A2[7]);
  o[8] = make_uint4(B2[0], B2[1], B2[2], B2[3]);
  o[12] = make_uint4(B2[4], B2[5], B2[6], B2[7]);
}

// The quads past the batch end replay the tile's last valid drone (same inputs, same instruction
// stream, bitwise the same coefficients): they store on top of that drone's block instead of being
// masked off.
__device__ __forceinline__ void store_segment_quad8(double *__restrict__ seg_base, size_t drone_stride,
                                                    int nvalid, int lane, const double (&c)[8], bool bad) {
  const int dl = lane >> 2;
  store_quad8_at(seg_base + (size_t)(dl < nvalid ? dl : nvalid - 1) * drone_stride, lane & 3, c, bad);
}

// dur[d][i] = t[d][i+1] - t[d][i] for the whole tile, one contiguous sweep
template <int MAXCNT = 0>   // compile-time bound of drones x segments per tile (0: not known, scalar loop)
__device__ __forceinline__ void store_durations(const double *sTraw, int shared_times, int tpitch, int M,
                                                int nvalid, int lane, double *__restrict__ dur_tile) {
  // The trip count is WAVE-UNIFORM (a scalar loop around a predicated body), not `for (e = lane; e < cnt; e += 64)`:
  // a loop the lanes leave one by one ends with exec == 0, and the compiler put register-pressure copies of values
  // that live across it (v_accvgpr_write_b32 of lane + 64 / lane + 128, solve_kernel_twin<5, 20>; the scratch spills of
  // round 3's two-sided 16-segment instance) into that exit block IN FRONT of the instruction that restores exec --
  // they wrote no lane, the next tile's prefetch indices were garbage: "Memory access fault by GPU" (DESIGN.md 9.3;
  // tools/check_exec_isa.py now refuses a build with such a copy under a reduced exec mask)
  const int cnt = nvalid * M;
  auto one = [&](int e) {
    if (e < cnt) {
      const int dl = e / M;
      const int i = e - dl * M;
      const double *lt = sTraw + (shared_times ? 0 : dl * tpitch);
      dur_tile[e] = lt[i + 1] - lt[i];
    }
  };
  if constexpr (MAXCNT > 0) {      // straight-line instances: two or three predicated rounds, no loop at all
#pragma unroll
    for (int e0 = 0; e0 < MAXCNT; e0 += kWave) one(e0 + lane);
  } else {
    for (int e0 = 0; e0 < cnt; e0 += kWave) one(e0 + lane);
  }
}

__device__ __forceinline__ int drone_status(bool nonfinite, bool badtime, bool singular) {
  int flags = (nonfinite ? 4 : 0) | (badtime ? 2 : 0) | (singular ? 1 : 0);
  flags |= __shfl_xor(flags, 1);   // combine the 4 axis lanes of the drone
  flags |= __shfl_xor(flags, 2);
  return (flags & 4) ? MSNAP_ST_NONFINITE : (flags & 2) ? MSNAP_ST_TIMES : (flags & 1) ? MSNAP_ST_SINGULAR : MSNAP_ST_OK;
}

// one coalesced sweep of the tile's waypoints and times into LDS, all loads in flight
__device__ __forceinline__ void stage_inputs(const double *__restrict__ wp, const double *__restrict__ tt,
                                             int shared_times, int tile, int nvalid, int wpitch, int tpitch,
                                             double *sWraw, double *sTraw, int lane) {
  const double2 *wsrc = reinterpret_cast<const double2 *>(wp + (size_t)tile * kDronesPerWave * wpitch);
  double2 *wdst = reinterpret_cast<double2 *>(sWraw);
  const int wcnt = nvalid * wpitch / 2;   // wpitch is a multiple of 4
  const double *tsrc = shared_times ? tt : tt + (size_t)tile * kDronesPerWave * tpitch;
  const int tcnt = shared_times ? tpitch : nvalid * tpitch;
  constexpr int UW = 8, UT = 4;           // the first round covers n_seg <= 14
  for (int e0 = 0, f0 = 0; e0 < wcnt || f0 < tcnt; e0 += UW * kWave, f0 += UT * kWave) {
    double2 vw[UW];
    double vt[UT];
#pragma unroll
    for (int u = 0; u < UW; ++u) {
      const int e = e0 + u * kWave + lane;
      vw[u] = wsrc[e < wcnt ? e : wcnt - 1];
    }
#pragma unroll
    for (int u = 0; u < UT; ++u) {
      const int f = f0 + u * kWave + lane;
      vt[u] = tsrc[f < tcnt ? f : tcnt - 1];
    }
    // pin: keeps the compiler from sinking each load next to its guarded store
#pragma unroll
    for (int u = 0; u < UW; ++u) asm volatile("" : "+v"(vw[u].x), "+v"(vw[u].y));
#pragma unroll
    for (int u = 0; u < UT; ++u) asm volatile("" : "+v"(vt[u]));
#pragma unroll
    for (int u = 0; u < UW; ++u) {
      const int e = e0 + u * kWave + lane;
      if (e < wcnt) wdst[e] = vw[u];
    }
#pragma unroll
    for (int u = 0; u < UT; ++u) {
      const int f = f0 + u * kWave + lane;
      if (f < tcnt) sTraw[f] = vt[u];
    }
  }
}

// the same sweep split in two for software pipelining across tiles (n_seg <= 12: one round):
// the loads of tile k+1 are issued near the end of tile k and land in LDS at the top of k+1
template <int MAXM>
struct StageRegs {
  static constexpr int UW = (MAXM + 2) / 2;   // ceil(16*(MAXM+1)*4/2 / 64) 16-byte loads per lane
  static constexpr int UT = (MAXM + 4) / 4;   // ceil(16*(MAXM+1) / 64) 8-byte loads per lane
  double2 vw[UW];
  double vt[UT];
};

template <int MAXM>
__device__ __forceinline__ void stage_load_once(const double *__restrict__ wp, const double *__restrict__ tt,
                                                int shared_times, int tile, int nvalid, int wpitch, int tpitch,
                                                int lane, StageRegs<MAXM> &r) {
  constexpr int kStageUW = StageRegs<MAXM>::UW, kStageUT = StageRegs<MAXM>::UT;
  const double2 *wsrc = reinterpret_cast<const double2 *>(wp + (size_t)tile * kDronesPerWave * wpitch);
  const int wcnt = nvalid * wpitch / 2;
  const double *tsrc = shared_times ? tt : tt + (size_t)tile * kDronesPerWave * tpitch;
  const int tcnt = shared_times ? tpitch : nvalid * tpitch;
#pragma unroll
  for (int u = 0; u < kStageUW; ++u) {
    const int e = u * kWave + lane;
    r.vw[u] = wsrc[e < wcnt ? e : wcnt - 1];
  }
#pragma unroll
  for (int u = 0; u < kStageUT; ++u) {
    const int f = u * kWave + lane;
    r.vt[u] = tsrc[f < tcnt ? f : tcnt - 1];
  }
}

template <int MAXM>
__device__ __forceinline__ void stage_store_once(int shared_times, int nvalid, int wpitch, int tpitch,
                                                 double *sWraw, double *sTraw, int lane, StageRegs<MAXM> &r) {
  constexpr int kStageUW = StageRegs<MAXM>::UW, kStageUT = StageRegs<MAXM>::UT;
  double2 *wdst = reinterpret_cast<double2 *>(sWraw);
  const int wcnt = nvalid * wpitch / 2;
  const int tcnt = shared_times ? tpitch : nvalid * tpitch;
#pragma unroll
  for (int u = 0; u < kStageUW; ++u) asm volatile("" : "+v"(r.vw[u].x), "+v"(r.vw[u].y));
#pragma unroll
  for (int u = 0; u < kStageUT; ++u) asm volatile("" : "+v"(r.vt[u]));
#pragma unroll
  for (int u = 0; u < kStageUW; ++u) {
    const int e = u * kWave + lane;
    if (e < wcnt) wdst[e] = r.vw[u];
  }
#pragma unroll
  for (int u = 0; u < kStageUT; ++u) {
    const int f = u * kWave + lane;
    if (f < tcnt) sTraw[f] = r.vt[u];
  }
}

// Hand-managed variant of the same prefetch for the persistent kernel.  hipcc waits for a
// prefetched load with vmcnt(0) once the wait sits behind the loop back-edge, which also
// drains the tile's 40 KB of output stores at every tile boundary.  The loads are therefore
// issued from inline asm (invisible to the compiler's wait-count pass) and retired with an
// exact s_waitcnt vmcnt(N), N = the store instructions issued after them, so the previous
// tile's stores stay in flight while the next tile starts.  (cdna_hip_programming.md 5.7:
// loads inside asm are counted and waited for by hand; the wait carries the registers as
// "+v" operands so no consumer can be scheduled above it.)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int MAXM>
struct StageRegsAsm {
  static constexpr int UW = (MAXM + 2) / 2;
  static constexpr int UT = (MAXM + 4) / 4;
  u32x4 vw[UW];
  double vt[UT];
};

template <int MAXM>
__device__ __forceinline__ void stage_load_asm(const double *__restrict__ wp, const double *__restrict__ tt,
                                               int shared_times, int tile, int nvalid, int wpitch, int tpitch,
                                               int lane, StageRegsAsm<MAXM> &r) {
  const double2 *wsrc = reinterpret_cast<const double2 *>(wp + (size_t)tile * kDronesPerWave * wpitch);
  const int wcnt = nvalid * wpitch / 2;
  const double *tsrc = shared_times ? tt : tt + (size_t)tile * kDronesPerWave * tpitch;
  const int tcnt = shared_times ? tpitch : nvalid * tpitch;
#pragma unroll
  for (int u = 0; u < StageRegsAsm<MAXM>::UW; ++u) {
    const int e = u * kWave + lane;
    const double2 *p = wsrc + (e < wcnt ? e : wcnt - 1);
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r.vw[u]) : "v"(p) : "memory");
  }
#pragma unroll
  for (int u = 0; u < StageRegsAsm<MAXM>::UT; ++u) {
    const int f = u * kWave + lane;
    const double *p = tsrc + (f < tcnt ? f : tcnt - 1);
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(r.vt[u]) : "v"(p) : "memory");
  }
}

// retire the prefetch: all but the YOUNGER most recent vector-memory operations are complete
template <int MAXM, int YOUNGER>
__device__ __forceinline__ void stage_wait_asm(StageRegsAsm<MAXM> &r) {
  wait_vmcnt<YOUNGER>();
  // tie the registers to this point so that no use is scheduled above the wait
#pragma unroll
  for (int u = 0; u < StageRegsAsm<MAXM>::UW; ++u) asm volatile("" : "+v"(r.vw[u]));
#pragma unroll
  for (int u = 0; u < StageRegsAsm<MAXM>::UT; ++u) asm volatile("" : "+v"(r.vt[u]));
}

template <int MAXM>
__device__ __forceinline__ void stage_store_asm(int shared_times, int nvalid, int wpitch, int tpitch,
                                                double *sWraw, double *sTraw, int lane, StageRegsAsm<MAXM> &r) {
  u32x4 *wdst = reinterpret_cast<u32x4 *>(sWraw);
  const int wcnt = nvalid * wpitch / 2;
  const int tcnt = shared_times ? tpitch : nvalid * tpitch;
#pragma unroll
  for (int u = 0; u < StageRegsAsm<MAXM>::UW; ++u) {
    const int e = u * kWave + lane;
    if (e < wcnt) wdst[e] = r.vw[u];
  }
#pragma unroll
  for (int u = 0; u < StageRegsAsm<MAXM>::UT; ++u) {
    const int f = u * kWave + lane;
    if (f < tcnt) sTraw[f] = r.vt[u];
  }
}

// one-wave workgroups: LDS is in-order within the wave, only the compiler must be held back
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

// ------------------------------------------------------------------------------------
// generic variant: rolled loops, G_i / z_i stashed in LDS (GS = false) or on a
// global slab (GS = true)
// ------------------------------------------------------------------------------------
template <int K, bool GS>
__global__ void __launch_bounds__(kWave)
solve_kernel(const double *__restrict__ wp, const double *__restrict__ tt, int shared_times,
             int N, int M, double *__restrict__ coef, double *__restrict__ dur,
             int32_t *__restrict__ status, double *__restrict__ gscratch, int ntiles) {
  using SW = Sweep<K>;
  constexpr int NU = SW::NU, NC = SW::NC;

  extern __shared__ __attribute__((aligned(16))) double lds[];

  const int lane = threadIdx.x;
  const int dl = lane >> 2;
  const int a = lane & 3;
  const int knots = M - 1;
  const int wpitch = (M + 1) * 4;
  const int tpitch = M + 1;

  // LDS: output transpose image Tr[NC/2][68] (16-B slots) | then, LDS variant only,
  //      inputs Wraw[16][(M+1)*4] | Traw[16][M+1]
  // stash (LDS or global slab): X[M][16] | G[knots][NU*NU][16] | Z[knots][NU][64]
  double2 *sTr = reinterpret_cast<double2 *>(lds);
  double *sWraw = lds + (NC / 2) * kTrPitch * 2;
  double *sTraw = sWraw + 16 * wpitch;
  double *scr;
  if constexpr (GS) {
    scr = gscratch + (size_t)blockIdx.x * (size_t)(16 * M + 16 * NU * NU * knots + 64 * NU * knots);
  } else {
    scr = sTraw + 16 * tpitch;
  }
  double *sX = scr;
  double *sG = sX + 16 * M;
  double *sZ = sG + 16 * NU * NU * knots;

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int d_raw = tile * kDronesPerWave + dl;
    const bool live = d_raw < N;
    const int d = live ? d_raw : N - 1;
    const double *wrow = wp + (size_t)d * (size_t)wpitch + a;
    const double *trow = shared_times ? tt : tt + (size_t)d * (size_t)tpitch;
    double *drow = dur + (size_t)d * M;

    const int left = N - tile * kDronesPerWave;
    const int nvalid = left < kDronesPerWave ? left : kDronesPerWave;
    if constexpr (!GS) {
      if (tile != (int)blockIdx.x) __syncthreads();  // LDS is reused by this block's next tile
      stage_inputs(wp, tt, shared_times, tile, nvalid, wpitch, tpitch, sWraw, sTraw, lane);
      __syncthreads();
      store_durations(sTraw, shared_times, tpitch, M, nvalid, lane, dur + (size_t)tile * kDronesPerWave * M);
    }
    // lanes past the end of the batch replay the last valid drone (and store on top of its results)
    const int dloc = live ? dl : (N - 1 - tile * kDronesPerWave);
    const double *lw = sWraw + dloc * wpitch + a;
    const double *lt = sTraw + (shared_times ? 0 : dloc * tpitch);
    auto Wv = [&](int i) -> double { if constexpr (GS) return wrow[(size_t)i * 4]; else return lw[i * 4]; };
    auto Tv = [&](int i) -> double { if constexpr (GS) return trow[i]; else return lt[i]; };

    // ---------------- segment 0 ----------------
    const double t0 = Tv(0);
    double tcur = Tv(1);
    const double w0 = Wv(0);
    double wcur = Wv(1);
    bool nonfinite = !(finite64(t0) & finite64(tcur) & finite64(w0) & finite64(wcur));
    double T = tcur - t0;
    const double Teff = T - t0;   // Appendix-A quirk, see taylor_shift()
    bool badtime = !(T > 0.0) | !(Teff > 0.0) | (t0 < 0.0);
    if constexpr (GS) {
      if (live && a == 0) drow[0] = T;
    }
    double x = rcp64(Teff);
    sX[0 * 16 + dl] = x;
    SW sw;
    sw.init(x, wcur - w0);

    // ---------------- forward block LDL^T sweep over interior knots ----------------
    // operands of knot i+1 are fetched one iteration ahead (LDS latency off the chain)
    double tpre = Tv(M >= 2 ? 2 : M);
    double wpre = Wv(M >= 2 ? 2 : M);
    for (int i = 1; i < M; ++i) {
      const double tnext = tpre;
      const double wnext = wpre;
      {
        const int ip = (i + 2 <= M) ? i + 2 : M;
        tpre = Tv(ip);
        wpre = Wv(ip);
      }
      nonfinite |= !finite64(tnext) | !finite64(wnext);
      T = tnext - tcur;
      badtime |= !(T > 0.0);
      if constexpr (GS) {
        if (live && a == 0) drow[i] = T;
      }
      x = rcp64(T);
      sX[i * 16 + dl] = x;
      double G[NU][NU], z[NU];
      sw.step(x, wnext - wcur, G, z);
      double *g = sG + (size_t)(i - 1) * (NU * NU * 16) + dl;
      double *zz = sZ + (size_t)(i - 1) * (NU * 64) + lane;
#pragma unroll
      for (int r = 0; r < NU; ++r) {
#pragma unroll
        for (int c = 0; c < NU; ++c) g[(r * NU + c) * 16] = G[r][c];
        zz[r * 64] = z[r];
      }
      tcur = tnext;
      wcur = wnext;
    }

    const int st = drone_status(nonfinite, badtime, sw.singular);
    if (live && a == 0) status[d] = st;
    const bool bad = st != 0;

    // ---------------- backward sweep + coefficient recovery ----------------
    // operands of segment i-1 are fetched while segment i is being recovered
    double un[NU], zq[NU], gq[NU][NU];
#pragma unroll
    for (int r = 0; r < NU; ++r) un[r] = 0.0;
    double wn = wcur;   // w_M
    double wq = Wv(M - 1), xq1 = sX[(M - 1) * 16 + dl];
    {
      const int kq = (M >= 2) ? M - 2 : 0;  // stash slot of knot M-1
      const double *zz = sZ + (size_t)kq * (NU * 64) + lane;
      const double *g = sG + (size_t)kq * (NU * NU * 16) + dl;
#pragma unroll
      for (int r = 0; r < NU; ++r) {
        zq[r] = (M >= 2) ? zz[r * 64] : 0.0;
#pragma unroll
        for (int c = 0; c < NU; ++c) gq[r][c] = (M >= 2) ? g[(r * NU + c) * 16] : 0.0;
      }
    }
    for (int i = M - 1; i >= 0; --i) {
      double u[NU];
      const double wi = wq;
      const double xi = xq1;
#pragma unroll
      for (int r = 0; r < NU; ++r) {
        double v = zq[r];
        if (i < M - 1) {
#pragma unroll
          for (int c = 0; c < NU; ++c) v = __builtin_fma(-gq[r][c], un[c], v);
        }
        u[r] = (i >= 1) ? v : 0.0;
      }
      if (i >= 1) {
        wq = Wv(i - 1);
        xq1 = sX[(i - 1) * 16 + dl];
        if (i >= 2) {
          const double *zz = sZ + (size_t)(i - 2) * (NU * 64) + lane;
          const double *g = sG + (size_t)(i - 2) * (NU * NU * 16) + dl;
#pragma unroll
          for (int r = 0; r < NU; ++r) {
            zq[r] = zz[r * 64];
#pragma unroll
            for (int c = 0; c < NU; ++c) gq[r][c] = g[(r * NU + c) * 16];
          }
        }
      }
      double c[NC];
      recover_segment<K>(wi, wn - wi, xi, u, un, c);
      if (i == 0 && t0 != 0.0) taylor_shift<NC>(c, -t0);
      if constexpr (NC == 8)
        store_segment_quad8(MSNAP_SEG_BASE(coef, tile, M, i, NC), MSNAP_SEG_STRIDE(M, NC), nvalid, lane, c, bad);
      else
        store_segment_coalesced<NC>(sTr, MSNAP_SEG_BASE(coef, tile, M, i, NC), MSNAP_SEG_STRIDE(M, NC), nvalid,
                                    lane, c, bad);
#pragma unroll
      for (int r = 0; r < NU; ++r) un[r] = u[r];
      wn = wi;
    }
  }
}

// ------------------------------------------------------------------------------------
// fast variant for n_seg <= MAXM: knot loops unrolled, z_i in registers, G_i in LDS.
// LDS per wave = inputs + 1/T + G  (18.7 KB at n_seg = 10, order 7) -> 8 waves / CU.
// ------------------------------------------------------------------------------------
template <int K, int MAXM>
__global__ void __launch_bounds__(kWave, ((K <= 4 && MAXM <= 10) ? 2 : 1))   // order 9 / 20 segments need > 256 VGPRs
solve_kernel_reg(const double *__restrict__ wp, const double *__restrict__ tt, int shared_times,
                 int N, int M, double *__restrict__ coef, double *__restrict__ dur,
                 int32_t *__restrict__ status, int ntiles) {
  using SW = Sweep<K>;
  constexpr int NU = SW::NU, NC = SW::NC;

  extern __shared__ __attribute__((aligned(16))) double lds[];

  const int lane = threadIdx.x;
  const int dl = lane >> 2;
  const int a = lane & 3;
  const int wpitch = (M + 1) * 4;
  const int tpitch = M + 1;
  // LDS: [ inputs Wraw | Traw ] -- dead after the forward sweep, then reused as the output
  //      transpose image -- followed by G[M-2][NU*NU][16] (G of the last knot is never read)
  double *sWraw = lds;
  double *sTraw = sWraw + 16 * wpitch;
  double2 *sTr = reinterpret_cast<double2 *>(lds);
  const int in_words = 16 * wpitch + 16 * tpitch;
  const int tr_words = (NC / 2) * kTrPitch * 2;
  double *sG = lds + (in_words > tr_words ? in_words : tr_words);

  auto tile_valid = [&](int tl) {
    const int left = N - tl * kDronesPerWave;
    return left < kDronesPerWave ? left : kDronesPerWave;
  };
  StageRegsAsm<MAXM> pre;
  if ((int)blockIdx.x < ntiles)
    stage_load_asm(wp, tt, shared_times, blockIdx.x, tile_valid(blockIdx.x), wpitch, tpitch, lane, pre);
  wait_vmcnt<0>();   // the first tile's inputs (nothing to overlap them with)
  // Prefetch distance: the next tile's inputs are requested two segments before the end of the
  // backward sweep, when most of this tile's registers are dead.  (Requesting them right after the
  // forward sweep hides more latency on paper but measured 4 % slower at saturation)
  // The wait at the top of a tile leaves the 2 x kStoresPerSeg stores of the two segments issued after the
  // loads in flight (this kernel is launched for n_seg >= 2 only, so both always run and the immediate is a
  // constant; waiting for the first of the two segments' stores as well costs 6 % at 65 536 x 10, order 9).
  // tools/check_prefetch_isa.py counts the stores on every path of the code object; the multi-tile tests
  // of tests/test_solve_gpu.py compare a persistent wave's tiles with the same tiles solved alone.
  constexpr int kStoresPerSeg = (NC == 8 ? 4 : NC / 2);

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int d_raw = tile * kDronesPerWave + dl;
    const bool live = d_raw < N;
    const int d = live ? d_raw : N - 1;
    const int nvalid = tile_valid(tile);
    const int next = tile + gridDim.x;

    // tile top: a marker instruction for tools/check_prefetch_isa.py (priority 0 is the default: no effect)
    asm volatile("s_setprio 0" ::: "memory");
    wave_lds_fence();   // the previous tile's LDS reads are done (in-order LDS, one wave)
    stage_wait_asm<MAXM, 2 * kStoresPerSeg>(pre);
    stage_store_asm(shared_times, nvalid, wpitch, tpitch, sWraw, sTraw, lane, pre);
    wave_lds_fence();
    store_durations(sTraw, shared_times, tpitch, M, nvalid, lane, dur + (size_t)tile * kDronesPerWave * M);
    const int dloc = live ? dl : (N - 1 - tile * kDronesPerWave);
    const double *lw = sWraw + dloc * wpitch + a;
    const double *lt = sTraw + (shared_times ? 0 : dloc * tpitch);

    // per-lane copies of the path: waypoints, segment lengths and their reciprocals stay in
    // registers for the backward sweep (knot loops are unrolled, so the indices are static)
    double wreg[MAXM + 1], xreg[MAXM], zreg[MAXM > 1 ? MAXM - 1 : 1][NU];

    const double t0 = lt[0];
    double tcur = lt[1];
    wreg[0] = lw[0];
    wreg[1] = lw[4];
    bool nonfinite = !(finite64(t0) & finite64(tcur) & finite64(wreg[0]) & finite64(wreg[1]));
    const double T0 = tcur - t0;
    const double T0q = T0 - t0;   // Appendix-A quirk: segment 0 has length T_0 - t0 in s - t0
    bool badtime = !(T0 > 0.0) | !(T0q > 0.0) | (t0 < 0.0);
    xreg[0] = rcp64(T0q);
    SW sw;
    sw.init(xreg[0], wreg[1] - wreg[0]);

    double tpre = lt[M >= 2 ? 2 : M];
    double wpre = lw[(M >= 2 ? 2 : M) * 4];
#pragma unroll
    for (int i = 1; i < MAXM; ++i) {
      if (i < M) {
        const double tnext = tpre;
        wreg[i + 1] = wpre;
        {
          const int ip = (i + 2 <= M) ? i + 2 : M;
          tpre = lt[ip];
          wpre = lw[ip * 4];
        }
        nonfinite |= !finite64(tnext) | !finite64(wreg[i + 1]);
        const double Ti = tnext - tcur;
        badtime |= !(Ti > 0.0);
        xreg[i] = rcp64(Ti);
        double G[NU][NU], z[NU];
        sw.step(xreg[i], wreg[i + 1] - wreg[i], G, z);
        {   // the checks of this knot settled here, in one register: left as booleans the compares are sunk behind the
            // sweep and every |w|, T and pivot waits for them in registers (see solve_kernel_twin)
          int f = (nonfinite ? 4 : 0) | (badtime ? 2 : 0) | (sw.singular ? 1 : 0);
          asm volatile("" : "+v"(f));
          nonfinite = (f & 4) != 0;
          badtime = (f & 2) != 0;
          sw.singular = (f & 1) != 0;
        }
        if (i < M - 1) {
          double *g = sG + (i - 1) * (NU * NU * 16) + dl;
#pragma unroll
          for (int r = 0; r < NU; ++r)
#pragma unroll
            for (int c = 0; c < NU; ++c) g[(r * NU + c) * 16] = G[r][c];
        }
#pragma unroll
        for (int r = 0; r < NU; ++r) zreg[i - 1][r] = z[r];
        tcur = tnext;
      }
    }

    const int st = drone_status(nonfinite, badtime, sw.singular);
    if (live && a == 0) status[d] = st;
    const bool bad = st != 0;

    // ---------------- backward sweep + recovery: registers + one G block per knot ----------------
    double un[NU], gq[NU][NU];
#pragma unroll
    for (int r = 0; r < NU; ++r) {
      un[r] = 0.0;
#pragma unroll
      for (int c = 0; c < NU; ++c) gq[r][c] = 0.0;
    }
#pragma unroll
    for (int i = MAXM - 1; i >= 0; --i) {
      if (i == (MAXM >= 2 ? 1 : 0)) {
        // software pipelining across tiles: with two segments left most of this tile's registers
        // are dead, so the next tile's inputs start their trip from HBM now.  Unconditional
        // (clamped to the last tile) and at a static point of the unrolled loop: a conditional
        // definition would keep `pre` live -- and spilled -- across the whole tile.
        __builtin_amdgcn_sched_barrier(0);
        const int nx = next < ntiles ? next : ntiles - 1;
        stage_load_asm(wp, tt, shared_times, nx, tile_valid(nx), wpitch, tpitch, lane, pre);
      }
      if (i <= 1 || i < M) {   // n_seg >= 2 (launch_solve_k): segments 0 and 1 always exist
        double u[NU];
#pragma unroll
        for (int r = 0; r < NU; ++r) {
          double v = (i >= 1) ? zreg[i >= 1 ? i - 1 : 0][r] : 0.0;
          if (i >= 1) {
            // at the last knot (i == M - 1, the first iteration that runs) gq and un are still zero and
            // the products add -0.0: bitwise z itself, without a select on the runtime segment count
#pragma unroll
            for (int c = 0; c < NU; ++c) v = __builtin_fma(-gq[r][c], un[c], v);
          }
          u[r] = v;
        }
        if (i >= 2) {   // G of knot i-1 (slot i-2), used by the next iteration (i-1 < M-1 always)
          const double *g = sG + (i - 2) * (NU * NU * 16) + dl;
#pragma unroll
          for (int r = 0; r < NU; ++r)
#pragma unroll
            for (int c = 0; c < NU; ++c) gq[r][c] = g[(r * NU + c) * 16];
        }
        double c[NC];
        recover_segment<K>(wreg[i], wreg[i + 1] - wreg[i], xreg[i], u, un, c);
        if (i == 0 && t0 != 0.0) taylor_shift<NC>(c, -t0);
        if constexpr (NC == 8)
          store_segment_quad8(MSNAP_SEG_BASE(coef, tile, M, i, NC), MSNAP_SEG_STRIDE(M, NC), nvalid, lane, c, bad);
        else
          store_segment_coalesced<NC>(sTr, MSNAP_SEG_BASE(coef, tile, M, i, NC), MSNAP_SEG_STRIDE(M, NC), nvalid,
                                      lane, c, bad);
#pragma unroll
        for (int r = 0; r < NU; ++r) un[r] = u[r];
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// order-9 throughput variant, two-sided AND column-split ("twin"): lane = 8*drone + 4*side + axis,
// 8 drones per wavefront.
//  * Two sides.  The per-lane state a solve keeps between its two sweeps (z_i, the waypoints, 1/T) is what
//    pushes the one-sided order-9 kernel to 307 registers = one wave per SIMD; with the knots of a path
//    shared between two sides (the merge at the meeting knot and the reversed-time bookkeeping of side 1 are
//    those of solve_kernel_twist below) every lane keeps half of it and the dependent chain is half as long.
//  * Column split.  Order 9 has NU = 4 unknown derivatives per knot = the 4 axis lanes of a drone: lane a
//    solves column a of G_i = S_i^-1 O_i and forms column a of the Schur term O_i^T G_i (plus its own axis'
//    z_i and O_i^T z_i) instead of all four lanes carrying all 16 entries; the symmetric Schur block of the
//    next knot is gathered with quad-broadcast DPP moves, and the G columns go to LDS as they are made.
//    What a lane carries of the previous segment is its 8 powers of 1/T (the end-side block E = HEE * powers
//    is folded into the FMAs that build S) and 4 right-hand terms.
//  * All inputs of the tile go to registers at the tile top (the reciprocals run side by side, off the
//    recurrence), so the input stage is dead before the first G column is stashed and aliases the rest.
// 256 registers without scratch and 13.6 KB of LDS -> two waves per SIMD.  One instance per even segment
// count 4..10 (both sides own (M-2)/2 >= 1 knots and M/2 segments); odd counts keep solve_kernel_reg<5,10>.
// (A one-sided column-split kernel was tried first: 16 drones per wave need 33 doubles of scratch at 256
// registers and ran at 83 us against solve_kernel_reg's 75 and this kernel's 57-60 at 65 536 x 10.)
// ------------------------------------------------------------------------------------
template <int SRC>
__device__ __forceinline__ double quad_bcast(double v) {   // lane SRC of the quad to all four
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, SRC * 0x55, 0xf, 0xf, true);
  hi = __builtin_amdgcn_mov_dpp(hi, SRC * 0x55, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// Output transposition image of the twin kernel: 5 rows (coefficient pairs) of kTwinTrPitch 16-byte slots.  A lane
// writes piece p at slot p * pitch + lane (8 contiguous lanes per LDS cycle: conflict-free for any pitch) and reads
// back, with store q, piece (a2, j2) = ((4q + j) / 5, (4q + j) % 5) of its own block at slot j2 * pitch + 4 * blk + a2.
// ds_read_b128 serves the lane groups {0-3, 12-15, 20-27}, ... (blocks 0, 3, 5, 6: all residues mod 4) and a slot
// covers 4 of the 64 banks, so the 16 slots of a group must differ mod 16: with pitch = 1 (mod 16) they are
// a2 + j2 + 4 * blk, distinct for every q (pitch 68 = 4 (mod 16), the one-sided kernels' choice for their flat
// read-back order, made 4 of them collide here: SQ_LDS_BANK_CONFLICT 2.0 of 9.9 M LDS cycles per launch).
constexpr int kTwinTrPitch = 65;
constexpr int kTwinTrWords = 5 * kTwinTrPitch * 2;   // in doubles
constexpr int kTwinDrones = 8;

// Output stores of the twin kernel.  The 4 axis lanes of a (drone, side) block hold the block's 320 bytes as
// 4 x 5 pieces of 16 bytes (lane = axis, piece = coefficient pair).  Through the LDS image the pieces are
// redistributed so that store q of lane j carries piece 4q + j of the lane's OWN block: every quad writes one
// whole 64-byte segment per instruction (the pattern of store_quad8_at), the address is the block base plus an
// immediate, and the only per-lane plan is where in the image the five pieces sit (set up once per tile).
struct TwinStorePlan {
  int ridx[5];   // double2 index into the image of the piece this lane stores with instruction q
};

__device__ __forceinline__ void twin_store_plan(int lane, TwinStorePlan &pl) {
  const int j = lane & 3, blk4 = lane & ~3;
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    const int within = q * 4 + j;            // 16-byte piece of the block: (axis a2, pair j2)
    const int a2 = within / 5;
    const int j2 = within - a2 * 5;
    pl.ridx[q] = j2 * kTwinTrPitch + blk4 + a2;
  }
}

// `blkp`: this lane's block of the segment + (lane & 3) * 2 doubles
__device__ __forceinline__ void store_twin_coalesced(double2 *sTr, double *__restrict__ blkp, int lane,
                                                     const TwinStorePlan &pl, const double (&c)[10]) {
  constexpr int NJ = 5;
#pragma unroll
  for (int j = 0; j < NJ; ++j) sTr[j * kTwinTrPitch + lane] = make_double2(c[2 * j], c[2 * j + 1]);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
#pragma unroll
  for (int q = 0; q < NJ; ++q) *reinterpret_cast<double2 *>(blkp + q * 8) = sTr[pl.ridx[q]];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
}

inline size_t twin_lds_bytes(int khalf, int n_seg) {
  const size_t h = (size_t)(n_seg - 2) - (size_t)(n_seg - 2) / 2;     // knots of the longer side
  size_t g_words = h * (size_t)(khalf - 1) * kWave;           // h knots x NU rows x 16 (drone, side) blocks x 4 column slots
  if (khalf == 4 && n_seg >= 17) g_words += 4 * (size_t)(khalf - 1) * kWave;   // z of the first knots (kTwinZInLds)
  const size_t in_words = (size_t)kTwinDrones * (n_seg + 1) * 5;
  const size_t body = (khalf == 5 ? (size_t)kTwinTrWords : 0) + g_words;
  return (in_words > body ? in_words : body) * sizeof(double);
}
// waves per SIMD an instance is built for (registers: 512 / waves per lane)
// (registers as built: order 7: 81 / 96 / 124 / 148 / 175 / 202 / 233 / 256 at 4 / 6 / ... / 18 segments; order 9:
//  92 / 123 / 164 / 201 / 242 at 4 / 6 / 8 / 10 / 12)
// knots per side whose z lives in LDS (1.5 KB per knot at order 7): the 17- to 20-segment instances -- 19 and 20
// segments would otherwise keep 23 / 8 dwords in scratch at the 256 registers of two waves per SIMD, 17 and 18 sit
// exactly at 256 (no kernel of the library uses scratch: tests/test_abi.py)
template <int K, int M>
constexpr int kTwinZInLds = (K == 4 && M >= 17) ? 4 : 0;
template <int K, int M>
constexpr int kTwinWaves = (K == 5 && M >= 13) ? 1 : 2;   // (order 9, 13..20 segments: 512 registers per lane, see kTwinMaxSeg)

// cross-tile input prefetch of the twin kernel: the hand-issued loads and exact wait of solve_kernel_reg
// (stage_load_asm / stage_wait_asm), for 8 drones per tile and a compile-time segment count
template <int M>
struct TwinStage {
  static constexpr int UW = (kTwinDrones * (M + 1) * 4 / 2 + kWave - 1) / kWave;   // 16-byte loads per lane
  static constexpr int UT = (kTwinDrones * (M + 1) + kWave - 1) / kWave;           // 8-byte loads per lane
  u32x4 vw[UW];
  double vt[UT];
};

template <int M>
__device__ __forceinline__ void twin_stage_load(const double *__restrict__ wp, const double *__restrict__ tt,
                                                int shared_times, int tile, int nvalid, int lane, TwinStage<M> &r) {
  constexpr int wpitch = (M + 1) * 4, tpitch = M + 1;
  const double2 *wsrc = reinterpret_cast<const double2 *>(wp + (size_t)tile * kTwinDrones * wpitch);
  const int wcnt = nvalid * wpitch / 2;
  const double *tsrc = shared_times ? tt : tt + (size_t)tile * kTwinDrones * tpitch;
  const int tcnt = shared_times ? tpitch : nvalid * tpitch;
#pragma unroll
  for (int u = 0; u < TwinStage<M>::UW; ++u) {
    const int e = u * kWave + lane;
    const double2 *p = wsrc + (e < wcnt ? e : wcnt - 1);
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r.vw[u]) : "v"(p) : "memory");
  }
#pragma unroll
  for (int u = 0; u < TwinStage<M>::UT; ++u) {
    const int f = u * kWave + lane;
    const double *p = tsrc + (f < tcnt ? f : tcnt - 1);
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(r.vt[u]) : "v"(p) : "memory");
  }
}

template <int M, int YOUNGER>
__device__ __forceinline__ void twin_stage_wait(TwinStage<M> &r) {
  wait_vmcnt<YOUNGER>();
#pragma unroll
  for (int u = 0; u < TwinStage<M>::UW; ++u) asm volatile("" : "+v"(r.vw[u]));
#pragma unroll
  for (int u = 0; u < TwinStage<M>::UT; ++u) asm volatile("" : "+v"(r.vt[u]));
}

template <int M>
__device__ __forceinline__ void twin_stage_store(int shared_times, int nvalid, double *sWraw, double *sTraw, int lane,
                                                 TwinStage<M> &r) {
  constexpr int wpitch = (M + 1) * 4, tpitch = M + 1;
  u32x4 *wdst = reinterpret_cast<u32x4 *>(sWraw);
  const int wcnt = nvalid * wpitch / 2;
  const int tcnt = shared_times ? tpitch : nvalid * tpitch;
#pragma unroll
  for (int u = 0; u < TwinStage<M>::UW; ++u) {
    const int e = u * kWave + lane;
    if (e < wcnt) wdst[e] = r.vw[u];
  }
#pragma unroll
  for (int u = 0; u < TwinStage<M>::UT; ++u) {
    const int f = u * kWave + lane;
    if (f < tcnt) sTraw[f] = r.vt[u];
  }
}

// gather the symmetric Schur block from the lanes' columns: entry (n, m), m <= n, is element n of lane m's column
template <int NU, int NS, int MCOL = 0>
__device__ __forceinline__ void twin_gather(const double (&col)[NU], double (&OtG)[NS]) {
  if constexpr (MCOL < NU) {
#pragma unroll
    for (int n = MCOL; n < NU; ++n) OtG[sidx(n, MCOL)] = quad_bcast<MCOL>(col[n]);
    twin_gather<NU, NS, MCOL + 1>(col, OtG);
  }
}

template <int K, int M>
__global__ void __launch_bounds__(kWave, (kTwinWaves<K, M>))
solve_kernel_twin(const double *__restrict__ wp, const double *__restrict__ tt, int shared_times,
                  int N, double *__restrict__ coef, double *__restrict__ dur,
                  int32_t *__restrict__ status, int ntiles) {
  using SW = Sweep<K>;
  using C = HermiteConsts<K>;
  constexpr int NU = SW::NU, NC = SW::NC, NS = SW::NS, KK = SW::KK, PM = SW::PM;
  static_assert(NU <= kAxes && NU >= 3, "one column of the knot blocks per axis lane (order 7: the fourth lane idles through the column work)");
  static_assert(M >= 4, "both sides own at least one knot");
  constexpr int kSlotWords = NU * kWave;      // one knot's G: [row][16 (drone, side) blocks][4 column slots]
  // M - 1 interior knots = nL (side 0) + the meeting knot + nR (side 1), nL <= nR <= nL + 1.  With an odd segment
  // count side 1 owns one knot and one segment more: all lanes run H = nR knot steps -- side 0's last one works on
  // valid data of the other side's territory and its effect on the carried state is undone (solve_kernel_twist) --
  // and side 0 sits out the first segment of the backward sweep.
  constexpr int nL = (M - 2) / 2, nR = (M - 2) - nL;
  constexpr int H = nR;
  constexpr bool kAsym = nL != nR;
  constexpr int HA = H > 0 ? H : 1;

  extern __shared__ __attribute__((aligned(16))) double lds[];

  const int lane0 = threadIdx.x;
  constexpr int wpitch = (M + 1) * 4;
  constexpr int tpitch = M + 1;
  // LDS: [ transposition image (order 9 only) | G: [knot][row][16 blocks][column] ]; the input stage aliases both
  // (dead before the first use)
  double *sWraw = lds;
  double *sTraw = sWraw + kTwinDrones * wpitch;
  double2 *sTr = reinterpret_cast<double2 *>(lds);
  double *sG = lds + (NC == 10 ? kTwinTrWords : 0);
  // the longest instances keep z of their first ZL knots (the ones that wait longest for the backward sweep) in LDS
  // behind the G slots instead of in registers: [knot][row][64 lanes]
  constexpr int ZL = kTwinZInLds<K, M>;
  double *sZ = sG + H * kSlotWords;
  double dsg[NU];
#pragma unroll
  for (int r = 0; r < NU; ++r) dsg[r] = (r & 1) ? 1.0 : -1.0;

  auto tile_valid = [&](int tl) {
    const int left = N - tl * kTwinDrones;
    return left < kTwinDrones ? left : kTwinDrones;
  };
  // cross-tile prefetch as in solve_kernel_reg: the next tile's inputs are requested before the last two
  // segments of the backward sweep and retired at the tile top by an exact vmcnt that leaves those two
  // segments' 2 x kStoresPerSeg stores in flight (every instance has H >= 1, so both always run)
  static_assert(nL >= 1, "the prefetch sits in front of segments 1 and 0 of the backward sweep, which both sides own");
  constexpr int kStoresPerSeg = (NC == 8 ? 4 : NC / 2);
  TwinStage<M> pre;
  if ((int)blockIdx.x < ntiles)
    twin_stage_load<M>(wp, tt, shared_times, blockIdx.x, tile_valid(blockIdx.x), lane0, pre);
  wait_vmcnt<0>();

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    // Everything derived from the lane index is rebuilt per tile from an opaque copy: left to the compiler,
    // two dozen loop-invariant addresses are hoisted out of the tile loop and live -- spilled -- across the
    // whole tile, which costs more than recomputing them once per ~2 500 instructions.
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int a = lane & 3;
    const int side = (lane >> 2) & 1;
    const int blk = lane >> 2;                  // (drone, side)
    const int dl = lane >> 3;
    double ca[NU];
#pragma unroll
    for (int n = 0; n < NU; ++n)
      ca[n] = a == 0 ? C::HSE[n + 1][1] : a == 1 ? C::HSE[n + 1][2] : a == 2 ? C::HSE[n + 1][3] : C::HSE[n + 1][NU];
    const int d_raw = tile * kTwinDrones + dl;
    const bool live = d_raw < N;
    const int d = live ? d_raw : N - 1;
    const int nvalid = tile_valid(tile);
    const int next = tile + gridDim.x;

    asm volatile("s_setprio 0" ::: "memory");   // tile-top marker for tools/check_prefetch_isa.py
    wave_lds_fence();   // the previous tile's LDS reads are done (in-order LDS, one wave)
    twin_stage_wait<M, 2 * kStoresPerSeg>(pre);
    twin_stage_store<M>(shared_times, nvalid, sWraw, sTraw, lane, pre);
    wave_lds_fence();
    // (the scalar-loop form: with the rounds unrolled -- store_durations<kTwinDrones * M> -- the compiler put accumulation-register
    //  copies of the 17..20-segment order-9 instances INSIDE a round's `if (e < cnt)`; tools/check_exec_isa.py refused that build)
    store_durations(sTraw, shared_times, tpitch, M, nvalid, lane, dur + (size_t)tile * kTwinDrones * M);

    const int dloc = live ? dl : (N - 1 - tile * kTwinDrones);
    const double *lw = sWraw + dloc * wpitch + a;
    const double *lt = sTraw + (shared_times ? 0 : dloc * tpitch);
    // own-coordinate accessors: side 1 walks the path backwards
    auto Wown = [&](int i) -> double { return lw[(side ? M - i : i) * 4]; };
    auto Town = [&](int i) -> double {
      const int j = side ? M - 1 - i : i;
      return lt[j + 1] - lt[j];
    };

    // ---- this side's half of the path into registers: waypoints 0..H+1, 1/T of segments 0..H ----
    double wreg[HA + 2], xreg[HA + 1], zreg[HA][NU];
    const double t0 = lt[0];
    bool nonfinite = !finite64(t0);
    bool badtime = t0 < 0.0;
#pragma unroll
    for (int i = 0; i <= H + 1; ++i) {
      wreg[i] = Wown(i);
      nonfinite |= !finite64(wreg[i]);
    }
#pragma unroll
    for (int i = 0; i <= H; ++i) {
      double T = Town(i);
      nonfinite |= !finite64(T);
      badtime |= !(T > 0.0);
      if (i == 0) {
        T = side ? T : T - t0;   // Appendix-A quirk lives on the start side only
        badtime |= !(T > 0.0);
      }
      xreg[i] = rcp64(T);
    }
    // The input checks are settled HERE, in one register: left as booleans the compares are sunk to where the status
    // is formed -- behind both sweeps -- and every |w|, |T| and T waits there in registers (32 of them at 10 segments).
    int in_flags = (nonfinite ? 4 : 0) | (badtime ? 2 : 0);
    asm volatile("" : "+v"(in_flags));
    wave_lds_fence();   // the input stage is dead: the image and the G slots alias it

    // ---- forward sweep over this side's knots (column split) ----
    // carried from knot to knot: the powers of the previous own segment's 1/T (xpp[p-1] = x^p; the end-side block
    // E = HEE * powers is folded into the FMAs that build S), rz = that segment's end-side right-hand term +
    // O^T z of the previous knot, and the Schur block O^T G
    double xpp[PM], rz[NU], OtG[NS];
    {
      double xp[PM + 1];
      SW::powers(xreg[0], xp);
      const double dw0 = wreg[1] - wreg[0];
#pragma unroll
      for (int p = 1; p <= PM; ++p) xpp[p - 1] = xp[p];
#pragma unroll
      for (int n = 1; n <= NU; ++n) rz[n - 1] = C::HEE[n][0] * (xp[KK - n] * dw0);
#pragma unroll
      for (int e = 0; e < NS; ++e) OtG[e] = 0.0;
    }
#pragma unroll
    for (int it = 1; it <= H; ++it) {
      double xpp0[PM], rz0[NU], OtG0[NS];     // side 0's carried state in front of its phantom step
      if (kAsym && it == H) {
#pragma unroll
        for (int p = 0; p < PM; ++p) xpp0[p] = xpp[p];
#pragma unroll
        for (int r = 0; r < NU; ++r) rz0[r] = rz[r];
#pragma unroll
        for (int e = 0; e < NS; ++e) OtG0[e] = OtG[e];
      }
      double xp[PM + 1];
      SW::powers(xreg[it], xp);
      const double dw = wreg[it + 1] - wreg[it];
      double S[NS], y[NU];
#pragma unroll
      for (int n = 1; n <= NU; ++n)
#pragma unroll
        for (int m = 1; m <= n; ++m)
          S[sidx(n - 1, m - 1)] =
              __builtin_fma(C::HSS[n][m], xp[KK - n - m],
                            __builtin_fma(C::HEE[n][m], xpp[KK - n - m - 1], -OtG[sidx(n - 1, m - 1)]));
#pragma unroll
      for (int n = 1; n <= NU; ++n) {
        const double tdw = xp[KK - n] * dw;
        y[n - 1] = __builtin_fma(-C::HSE[n][0], tdw, -rz[n - 1]);
        rz[n - 1] = C::HEE[n][0] * tdw;
      }
      double dinv[NU];
      in_flags |= (SW::ldl_factor(S, dinv) && (!kAsym || it <= nL || side)) ? 1 : 0;   // (not side 0's phantom step)
      asm volatile("" : "+v"(in_flags));      // (the pivot checks settled per knot, for the same reason)
      // (order 7: lane 3 has no column; it runs the same instructions on column 3's constants and lands in the
      //  unused fourth column slot of the stash)
      const double pa = a == 0 ? xp[NU] : a == 1 ? xp[NU - 1] : a == 2 ? xp[NU - 2] : xp[NU - 3];
      double g[NU];
#pragma unroll
      for (int n = 0; n < NU; ++n) g[n] = ca[n] * (n == NU - 1 ? pa : pa * xp[NU - 1 - n]);
      SW::ldl_solve(S, dinv, g);
      SW::ldl_solve(S, dinv, y);
      {
        // [knot][row r][block][column a]: the 64 lanes of a store write 64 consecutive doubles
        double *gs = sG + (it - 1) * kSlotWords + lane;
#pragma unroll
        for (int r = 0; r < NU; ++r) gs[r * kWave] = g[r];
      }
      if (it <= ZL) {
#pragma unroll
        for (int r = 0; r < NU; ++r) sZ[((it - 1) * NU + r) * kWave + lane] = y[r];
      } else {
#pragma unroll
        for (int r = 0; r < NU; ++r) zreg[it - 1][r] = y[r];
      }
#pragma unroll
      for (int q = 0; q < NU - 1; ++q) {
        g[q] *= xp[NU - 1 - q];
        y[q] *= xp[NU - 1 - q];
      }
      double col[NU];
#pragma unroll
      for (int n = 0; n < NU; ++n) {
        double sg = C::HSE[1][n + 1] * g[0], sz = C::HSE[1][n + 1] * y[0];
#pragma unroll
        for (int q = 1; q < NU; ++q) {
          sg = __builtin_fma(C::HSE[q + 1][n + 1], g[q], sg);
          sz = __builtin_fma(C::HSE[q + 1][n + 1], y[q], sz);
        }
        col[n] = sg * xp[NU - n];
        rz[n] = __builtin_fma(sz, xp[NU - n], rz[n]);
      }
      twin_gather<NU, NS>(col, OtG);
#pragma unroll
      for (int p = 1; p <= PM; ++p) xpp[p - 1] = xp[p];
      if (kAsym && it == H) {       // side 0 keeps what it carried in front of the phantom step
#pragma unroll
        for (int p = 0; p < PM; ++p) xpp[p] = side ? xpp[p] : xpp0[p];
#pragma unroll
        for (int r = 0; r < NU; ++r) rz[r] = side ? rz[r] : rz0[r];
#pragma unroll
        for (int e = 0; e < NS; ++e) OtG[e] = side ? OtG[e] : OtG0[e];
      }
    }

    // ---- the meeting knot (solve_kernel_twist): S = (E - O^T G)_own + D (E - O^T G)_other D ----
    double Sm[NS], um[NU];
#pragma unroll
    for (int r = 0; r < NU; ++r) {
      const double q = rz[r];
      um[r] = -q - dsg[r] * __shfl_xor(q, 4);
#pragma unroll
      for (int c = 0; c <= r; ++c) {
        const double p = __builtin_fma(C::HEE[r + 1][c + 1], xpp[KK - r - c - 3], -OtG[sidx(r, c)]);
        Sm[sidx(r, c)] = p + (dsg[r] * dsg[c]) * __shfl_xor(p, 4);
      }
    }
    {
      double dinv[NU];
      in_flags |= SW::ldl_factor(Sm, dinv) ? 1 : 0;
      SW::ldl_solve(Sm, dinv, um);
    }

    const int gsh = lane & ~7;
    const bool f_nonfinite = ((__ballot((in_flags & 4) != 0) >> gsh) & 0xFFull) != 0;
    const bool f_time = ((__ballot((in_flags & 2) != 0) >> gsh) & 0xFFull) != 0;
    const bool f_sing = ((__ballot((in_flags & 1) != 0) >> gsh) & 0xFFull) != 0;
    const int st = f_nonfinite ? MSNAP_ST_NONFINITE : f_time ? MSNAP_ST_TIMES : f_sing ? MSNAP_ST_SINGULAR : MSNAP_ST_OK;
    if (live && (lane & 7) == 0) status[d] = st;
    const bool bad = st != 0;

    // ---- outward back-substitution + recovery: every side owns its segments 0 .. H ----
    const double qnan = __builtin_nan("");
    const double zero_or_nan = bad ? qnan : 0.0;
    // drones past the batch end replay the tile's last valid one (bitwise the same values) and store on top of it
    TwinStorePlan plan;
    if constexpr (NC == 10) twin_store_plan(lane, plan);
    double *blkp = coef + ((size_t)d * M + (side ? M - 1 - H : H)) * (4 * NC) + (NC == 10 ? a * 2 : 0);
    const int blkstep = side ? 4 * NC : -(4 * NC);   // side 0 walks its segments down, side 1 up
#pragma unroll
    for (int i = 0; i < H + 2; ++i) wreg[i] = bad ? qnan : wreg[i];
    double un[NU];
#pragma unroll
    for (int r = 0; r < NU; ++r) un[r] = bad ? qnan : um[r];
#pragma unroll
    for (int it = H; it >= 0; --it) {
      // one segment at a time: left free, the scheduler issues the G reads of ALL knots at the top of the sweep and
      // holds them (12 registers per knot: 85 / 128 / 166 / 202 / 242 registers at 4..12 segments, order 7)
      if (it != 1) __builtin_amdgcn_sched_barrier(0);
      if (it == 1) {
        // with two segments left most of this tile's registers are dead: the next tile's inputs start now
        // (unconditional, clamped to the last tile, at a static point of the unrolled loop)
        __builtin_amdgcn_sched_barrier(0);
        const int nx = next < ntiles ? next : ntiles - 1;
        twin_stage_load<M>(wp, tt, shared_times, nx, tile_valid(nx), lane, pre);
      }
      const bool own = !kAsym || it <= nL || side != 0;     // side 0 sits out segment nR of an odd path
      double u[NU];
      if (it >= 1) {
        const double2 *gsl = reinterpret_cast<const double2 *>(sG + (it - 1) * kSlotWords + blk * kAxes);
#pragma unroll
        for (int r = 0; r < NU; ++r) {
          const double2 g01 = gsl[r * (kWave / 2)], g23 = gsl[r * (kWave / 2) + 1];
          double v = (it <= ZL) ? sZ[((it >= 1 ? it - 1 : 0) * NU + r) * kWave + lane] : zreg[it >= 1 ? it - 1 : 0][r];
          v = __builtin_fma(-g01.x, un[0], v);
          v = __builtin_fma(-g01.y, un[1], v);
          v = __builtin_fma(-g23.x, un[2], v);
          if constexpr (NU == 4) v = __builtin_fma(-g23.y, un[3], v);
          u[r] = v;
        }
      } else {
#pragma unroll
        for (int r = 0; r < NU; ++r) u[r] = zero_or_nan;
      }
      double ua[NU], ub[NU];
#pragma unroll
      for (int r = 0; r < NU; ++r) {
        ua[r] = side ? dsg[r] * un[r] : u[r];      // state at the forward start of the piece
        ub[r] = side ? dsg[r] * u[r] : un[r];      // state at its forward end
      }
      const double wa = side ? wreg[it + 1] : wreg[it];
      const double wb = side ? wreg[it] : wreg[it + 1];
      // 1/T through an opaque copy: left visible, the powers of 1/T the recovery needs are recognised as the forward
      // sweep's and kept from there to here -- 6-8 registers per knot and side (85 / 128 / 166 / 202 / 242 registers at
      // 4..12 segments, order 7, before; three or four multiplies per segment instead)
      double xi = xreg[it];
      asm volatile("" : "+v"(xi));
      double c[NC];
      recover_segment<K>(wa, wb - wa, xi, ua, ub, c);
      if (side == 0 && it == 0 && t0 != 0.0) taylor_shift<NC>(c, -t0);
      if (own) {      // (quads are side-uniform: whole quads store or sit out)
        if constexpr (NC == 10) store_twin_coalesced(sTr, blkp, lane, plan, c);
        else store_quad8_at(blkp, a, c, false);      // order 7: the quad transposes its 4 x 4 pieces with DPP, no LDS
      }
      blkp += blkstep;
#pragma unroll
      for (int r = 0; r < NU; ++r) un[r] = own ? u[r] : un[r];
    }
  }
}

// ------------------------------------------------------------------------------------
// small-batch variant (2 <= n_seg <= kTwistMaxSeg): two-sided ("twisted") sweep.
// With few drones a launch is one wavefront's dependent chain, so the chain is halved:
// lane = 8*drone + 4*side + axis.  Side 0 sweeps the knots from the start of the path,
// side 1 runs THE SAME recurrence on the time-reversed path (p'(s) = p(T_total - s):
// waypoints and segment lengths reversed, odd derivatives change sign, D = diag(-1,+1,-1,..)).
// Both stop one knot short of the MEETING knot in the middle.  What each side carries towards
// it -- the end-side block of its last segment and its Schur terms -- adds up (the other
// side's part conjugated by D) to the meeting knot's own symmetric positive definite block, so
// the merge is one cross-lane swap and one more LDL^T step; then both sides back-substitute
// outwards.  Side 1 obtains forward-time coefficients by recovering each piece with its two
// end states swapped and the odd derivatives negated (p(t) = q(T - t)).
// ------------------------------------------------------------------------------------
constexpr int kTwistDrones = 8;
constexpr int kTwistFenceHalf = 9;   // instances with this many knots per side fence the scheduler per knot

#ifdef MSNAP_TOOLS_TIMELINE
// phase timestamps of the twisted kernel (s_memrealtime, 100 MHz, and s_memtime): tools/twist_timeline.py
__device__ unsigned long long g_timeline[1024 * 32];
#define MSNAP_TL(k)                                                                       \
  do {                                                                                    \
    const int it__ = (tl_tile - (int)blockIdx.x) / (int)gridDim.x;                        \
    if (lane == 0 && blockIdx.x < 1024 && it__ < 2) {                                     \
      g_timeline[blockIdx.x * 32 + 16 * it__ + 2 * (k)] = wall_clock64();                 \
      g_timeline[blockIdx.x * 32 + 16 * it__ + 2 * (k) + 1] = clock64();                  \
    }                                                                                     \
  } while (0)
#else
#define MSNAP_TL(k) do { } while (0)
#endif

// One instance per (order, segment count M); MAXH = knots of the longer side.  Every loop bound is
// a constant, so the whole solve is straight-line code that the scheduler interleaves across knots
// (a run-time M costs block boundaries with dozens of register copies each: 6.2 vs 5.3 us at M = 10).
template <int K, int MAXH, int M>
__global__ void __launch_bounds__(kWave)
solve_kernel_twist(const double *__restrict__ wp, const double *__restrict__ tt, int shared_times,
                   int N, double *__restrict__ coef, double *__restrict__ dur,
                   int32_t *__restrict__ status, int ntiles) {
  using SW = Sweep<K>;
  constexpr int NU = SW::NU, NC = SW::NC, NS = SW::NS;
  // M - 1 interior knots = nL (side 0) + the meeting knot + nR (side 1), nL <= nR <= nL + 1
  constexpr int nL = (M - 2) / 2, nR = (M - 2) - nL;
  static_assert(M >= 2 && MAXH == nR, "MAXH is the knot count of the longer side");
  constexpr int HA = MAXH > 0 ? MAXH : 1;     // array extent (M = 2 has no side knots)

  extern __shared__ __attribute__((aligned(16))) double lds[];

  const int lane = threadIdx.x;
  const int a = lane & 3;
  const int side = (lane >> 2) & 1;
  const int dl = lane >> 3;           // drone inside the tile
  constexpr int wpitch = (M + 1) * 4;
  constexpr int tpitch = M + 1;
  const int mside = side ? nR : nL;
  double *sWraw = lds;
  double *sTraw = sWraw + kTwistDrones * wpitch;
  // D = diag(-1, +1, -1, ...): derivative r+1 of the time-reversed path
  double dsg[NU];
#pragma unroll
  for (int r = 0; r < NU; ++r) dsg[r] = (r & 1) ? 1.0 : -1.0;

  // ONE tile per wave, no tile loop: the launcher starts exactly one wave per tile (launch_solve_k: grid = nt8).  With
  // a loop around the body the compiler hoisted ~130 loop-invariant instructions (constants, lane-derived indices) into
  // its preheader -- IN FRONT of the tile's input loads, a quarter of a microsecond of a 3.6 us kernel (round 4)
  // (and no `tile < ntiles` test: it made the kernel fetch `ntiles` alone, wait, branch, and only then fetch its other
  //  arguments -- two dependent scalar-cache round trips in front of the first load; gridDim.x == ntiles by construction)
  const int tile = blockIdx.x;
  (void)ntiles;
  // (pinning the three output pointers in SGPRs at the top -- the compiler re-fetches each right in front of its first
  //  use, a scalar-cache hit on the dependent chain -- was measured and LOST: 4.52 against 4.38 us per step; the kernel
  //  holds ~60 fp64 constants in scalar registers and six more live ones push some of them out)
  {
#ifdef MSNAP_TOOLS_TIMELINE
    const int tl_tile = tile;
#endif
    MSNAP_TL(0);
    const int d_raw = tile * kTwistDrones + dl;
    const bool live = d_raw < N;
    const int d = live ? d_raw : N - 1;
    const int left = N - tile * kTwistDrones;
    const int nvalid = left < kTwistDrones ? left : kTwistDrones;

    {   // stage the 8 drones' inputs (one flight)
      const double2 *wsrc = reinterpret_cast<const double2 *>(wp + (size_t)tile * kTwistDrones * wpitch);
      double2 *wdst = reinterpret_cast<double2 *>(sWraw);
      const int wcnt = nvalid * wpitch / 2;
      const double *tsrc = shared_times ? tt : tt + (size_t)tile * kTwistDrones * tpitch;
      const int tcnt = shared_times ? tpitch : nvalid * tpitch;
      constexpr int UW = (kTwistDrones * wpitch / 2 + kWave - 1) / kWave;
      constexpr int UT = (kTwistDrones * tpitch + kWave - 1) / kWave;
      double2 vw[UW];
      double vt[UT];
#pragma unroll
      for (int u = 0; u < UW; ++u) {
        const int e = u * kWave + lane;
        vw[u] = wsrc[e < wcnt ? e : wcnt - 1];
      }
#pragma unroll
      for (int u = 0; u < UT; ++u) {
        const int f = u * kWave + lane;
        vt[u] = tsrc[f < tcnt ? f : tcnt - 1];
      }
#pragma unroll
      for (int u = 0; u < UW; ++u) asm volatile("" : "+v"(vw[u].x), "+v"(vw[u].y));
#pragma unroll
      for (int u = 0; u < UT; ++u) asm volatile("" : "+v"(vt[u]));
#pragma unroll
      for (int u = 0; u < UW; ++u) {
        const int e = u * kWave + lane;
        if (e < wcnt) wdst[e] = vw[u];
      }
#pragma unroll
      for (int u = 0; u < UT; ++u) {
        const int f = u * kWave + lane;
        if (f < tcnt) sTraw[f] = vt[u];
      }
    }
    __syncthreads();
    MSNAP_TL(1);

    const int dloc = live ? dl : (N - 1 - tile * kTwistDrones);
    const double *lw = sWraw + dloc * wpitch + a;
    const double *lt = sTraw + (shared_times ? 0 : dloc * tpitch);
    // own-coordinate accessors: side 1 walks the path backwards
    auto Wown = [&](int i) -> double { return lw[(side ? M - i : i) * 4]; };
    auto Town = [&](int i) -> double {
      const int j = side ? M - 1 - i : i;
      return lt[j + 1] - lt[j];
    };

    // long paths keep z_i in LDS (one slot per lane) so the instance fits the register file
    constexpr bool kZReg = MAXH * NU < 3 * kTwistFenceHalf;
    double wreg[HA + 2], xreg[HA + 1], zreg[kZReg ? HA : 1][NU];
    double *sZ = sTraw + kTwistDrones * tpitch + lane;      // [knot][r][64 lanes]
    // a launch of this kernel never has more than two waves per CU: the G_i blocks stay in
    // registers (overflowing into AGPRs on long paths) -- no LDS round trip on the dependent chain
    double Greg[HA][NU][NU];
    const double t0 = lt[0];
    wreg[0] = Wown(0);
    wreg[1] = Wown(1);
    const double T0 = Town(0);
    const double T0q = side ? T0 : T0 - t0;   // Appendix-A quirk lives on the start side only
    bool nonfinite = !(finite64(t0) & finite64(T0) & finite64(wreg[0]) & finite64(wreg[1]));
    bool badtime = !(T0 > 0.0) | !(T0q > 0.0) | (t0 < 0.0);
    xreg[0] = rcp64(T0q);
    SW sw;
    sw.init(xreg[0], wreg[1] - wreg[0]);

    // Per knot, the long dependent prefix that does not involve the recurrence -- inputs from LDS,
    // T, 1/T (reciprocal + Newton), its powers -- is prepared one knot AHEAD, inside the same
    // basic block as the previous knot's recurrence, where it fills that chain's latency bubbles.
    // All lanes run all nR steps; when side 0 owns one knot less its extra step works on valid
    // data and its effect on the carried state is undone.
    double xpc[SW::PM + 1], dwc = 0.0;     // powers of 1/T_it and w_{it+1} - w_it of the current knot
    auto ahead = [&](int it, double (&xp)[SW::PM + 1], double &dw) {
      wreg[it + 1] = Wown(it + 1);
      const double Tit = Town(it);
      nonfinite |= !finite64(Tit) | !finite64(wreg[it + 1]);
      badtime |= !(Tit > 0.0);
      xreg[it] = rcp64(Tit);
      SW::powers(xreg[it], xp);
      dw = wreg[it + 1] - wreg[it];
    };
    if (MAXH >= 1) ahead(1, xpc, dwc);
    bool singular = false;
    auto knot_step = [&](int it) {
      double G[NU][NU], z[NU];
      typename SW::Knot k;
      SW::knot_geom_xp(xpc, dwc, sw.E, sw.re, k);
      const bool ok = sw.chain(k, G, z);
      singular |= !ok & (it <= mside);
#pragma unroll
      for (int e = 0; e < NS; ++e) sw.E[e] = k.E[e];
#pragma unroll
      for (int r = 0; r < NU; ++r) {
        sw.re[r] = k.re[r];
        if constexpr (kZReg) zreg[it - 1][r] = z[r];
        else sZ[((it - 1) * NU + r) * kWave] = z[r];
#pragma unroll
        for (int c = 0; c < NU; ++c) Greg[it - 1][r][c] = G[r][c];
      }
    };
#pragma unroll
    for (int it = 1; it <= MAXH; ++it) {
      if (it < MAXH) {             // knot it, and the prefix of knot it+1
        double xpn[SW::PM + 1], dwn;
        ahead(it + 1, xpn, dwn);
        knot_step(it);
        dwc = dwn;
#pragma unroll
        for (int q = 0; q <= SW::PM; ++q) xpc[q] = xpn[q];
        // long paths: keep the scheduler from hoisting several knots' inputs at once (register
        // pressure beyond the 512-entry file would spill to scratch)
        if constexpr (MAXH * NU >= 3 * kTwistFenceHalf) __builtin_amdgcn_sched_barrier(0);
      } else if (nL == nR) {       // last knot, owned by both sides
        knot_step(it);
      } else {                     // last knot of side 1 only: side 0 keeps its carried state
        double E0[NS], G0[NS], r0[NU], z0[NU];
#pragma unroll
        for (int e = 0; e < NS; ++e) { E0[e] = sw.E[e]; G0[e] = sw.OtG[e]; }
#pragma unroll
        for (int r = 0; r < NU; ++r) { r0[r] = sw.re[r]; z0[r] = sw.Otz[r]; }
        knot_step(it);
#pragma unroll
        for (int e = 0; e < NS; ++e) { sw.E[e] = side ? sw.E[e] : E0[e]; sw.OtG[e] = side ? sw.OtG[e] : G0[e]; }
#pragma unroll
        for (int r = 0; r < NU; ++r) { sw.re[r] = side ? sw.re[r] : r0[r]; sw.Otz[r] = side ? sw.Otz[r] : z0[r]; }
      }
    }

    MSNAP_TL(2);
    // ---- the meeting knot: both sides' carried blocks add up to ONE symmetric positive definite
    // system,  S = (E - O^T G)_own + D (E - O^T G)_other D ,  y = -(re + O^T z)_own - D (re + O^T z)_other
    // (each side in its own orientation; D conjugates the other side's blocks).  One cross-lane
    // swap, then the same LDL^T as every other knot.
    double Sm[NS], um[NU];
#pragma unroll
    for (int r = 0; r < NU; ++r) {
      const double q = sw.re[r] + sw.Otz[r];
      um[r] = -q - dsg[r] * __shfl_xor(q, 4);
#pragma unroll
      for (int c = 0; c <= r; ++c) {
        const double p = sw.E[sidx(r, c)] - sw.OtG[sidx(r, c)];
        Sm[sidx(r, c)] = p + (dsg[r] * dsg[c]) * __shfl_xor(p, 4);
      }
    }
    {
      double dinv[NU];
      singular |= SW::ldl_factor(Sm, dinv);
      SW::ldl_solve(Sm, dinv, um);
    }

    // per-drone status over the 8 lanes (2 sides x 4 axes): wave ballots, no cross-lane round trips
    const int gsh = lane & ~7;
    const bool f_nonfinite = ((__ballot(nonfinite) >> gsh) & 0xFFull) != 0;
    const bool f_time = ((__ballot(badtime) >> gsh) & 0xFFull) != 0;
    const bool f_sing = ((__ballot(singular) >> gsh) & 0xFFull) != 0;
    const int st = f_nonfinite ? MSNAP_ST_NONFINITE : f_time ? MSNAP_ST_TIMES : f_sing ? MSNAP_ST_SINGULAR : MSNAP_ST_OK;
    if (live && (lane & 7) == 0) status[d] = st;      // (moved to the end with the durations: no gain, 4.30 against 4.28 us)
    const bool bad = st != 0;

    MSNAP_TL(3);
    // ---- outward back-substitution + recovery: side s owns its segments 0 .. mside ----
    // A failed drone's outputs are NaN: poison what every coefficient is computed from once
    // (waypoints -> c0 and the end-side block, knot states -> c1..) instead of selecting per piece.
    const double qnan = __builtin_nan("");
    const double zero_or_nan = bad ? qnan : 0.0;
#pragma unroll
    for (int i = 0; i < MAXH + 2; ++i) wreg[i] = bad ? qnan : wreg[i];
    double un[NU];
#pragma unroll
    for (int r = 0; r < NU; ++r) un[r] = bad ? qnan : um[r];
#pragma unroll
    for (int it = MAXH; it >= 0; --it) {
      if (it <= nL || side) {    // only side 1 owns segment nR when nL < nR
        double u[NU];
        if (it >= 1) {
#pragma unroll
          for (int r = 0; r < NU; ++r) {
            double v;
            if constexpr (kZReg) v = zreg[it >= 1 ? it - 1 : 0][r];
            else v = sZ[((it >= 1 ? it - 1 : 0) * NU + r) * kWave];
#pragma unroll
            for (int c = 0; c < NU; ++c) v = __builtin_fma(-Greg[it >= 1 ? it - 1 : 0][r][c], un[c], v);
            u[r] = v;
          }
        } else {
#pragma unroll
          for (int r = 0; r < NU; ++r) u[r] = zero_or_nan;
        }
        // Side 1 holds the piece in reversed time, q(s) with p(t) = q(T - t).  Its endpoint states in
        // forward time are the reversed ones with the odd derivatives negated, so the forward
        // coefficients come from the same recovery with the two ends swapped -- no Taylor shift.
        double ua[NU], ub[NU];
#pragma unroll
        for (int r = 0; r < NU; ++r) {
          ua[r] = side ? dsg[r] * un[r] : u[r];      // state at the forward start of the piece
          ub[r] = side ? dsg[r] * u[r] : un[r];      // state at its forward end
        }
        const double wa = side ? wreg[it + 1] : wreg[it];
        const double wb = side ? wreg[it] : wreg[it + 1];
        double c[NC];
        recover_segment<K>(wa, wb - wa, xreg[it], ua, ub, c);
        if (side == 0 && it == 0 && t0 != 0.0) taylor_shift<NC>(c, -t0);
        const int seg = side ? M - 1 - it : it;
        // a batch this small is latency bound, not store bound: plain per-lane stores.  Lanes past
        // the batch end mirror drone N-1 (same inputs, same values, same addresses), so the stores
        // need no predicate and the sweep stays one basic block.
        {
          // (a lane base formed early and pinned in two vector registers, so that the fetch of `coef` leaves the
          //  dependent chain, was measured: 4.54 against 4.28 us per step -- not shipped)
          double *o = coef + (((size_t)d * M + seg) * 4 + a) * NC;
#pragma unroll
          for (int m = 0; m < NC; m += 2) *reinterpret_cast<double2 *>(o + m) = make_double2(c[m], c[m + 1]);
        }
#pragma unroll
        for (int r = 0; r < NU; ++r) un[r] = u[r];
      }
      if constexpr (MAXH * NU >= 3 * kTwistFenceHalf) __builtin_amdgcn_sched_barrier(0);
    }
    // the durations leave LAST: in front of the sweeps they cost the fetch of their pointer and ~40 instructions on the
    // dependent chain; the staged times are still in LDS (the z stash lives behind them, nothing aliases the inputs)
    __builtin_amdgcn_sched_barrier(0);
    store_durations<kTwistDrones * M>(sTraw, shared_times, tpitch, M, nvalid, lane, dur + (size_t)tile * kTwistDrones * M);
    MSNAP_TL(4);
  }
}

constexpr int kTwistMaxSeg = 24;    // twisted variant, order 7: one instance per n_seg in 2..24
constexpr int kTwistMaxSeg9 = 12;   // order 9: 2..12

constexpr int kTwinMaxSeg = 20;    // order 9: solve_kernel_twin for 4..20 segments.  13..20 segments do not fit the 256 registers of two
                                   // waves per SIMD (290..402 as built) and are instances for ONE wave per SIMD: the compiler keeps the
                                   // excess in accumulation registers (10..146; no scratch).  Against solve_kernel_reg<5, 20> (also one wave
                                   // per SIMD): 65 536 x 20: 136-143 against 187 us, 2^20 x 20: 1.80 against 2.32 ms = 54 % against 42 % of
                                   // the HBM peak, 8192 x 20: 19.6 against 32.1 us (profiles/r04_order9_long.txt); 13 segments tie at 2^20
constexpr int kTwinMaxSeg7 = 20;   // order 7 (19 and 20 segments: z of the first knots in LDS instead of registers)
constexpr int kRegMaxSeg = 10;    // n_seg <= 10 takes the register-resident variant (2 waves per SIMD) ...
constexpr int kRegMaxSeg2 = 20;   // ... 11 <= n_seg <= 20 a second instance at one wave per SIMD

template <int K>
static int launch_solve_k(msnap_ctx *ctx, int N, int M, const double *wp, const double *t, int shared,
                          double *coef, double *dur, int32_t *status) {
  const int ntiles = (N + kDronesPerWave - 1) / kDronesPerWave;
  const size_t tr_bytes = (size_t)K * kTrPitch * 16;   // output transpose image, NC/2 = K rows
  // up to one 8-drone wavefront per SIMD the two-sided kernel wins (measured crossover 8-10 k drones on 256 CUs)
  // (order 9 with an even segment count <= 10: the two-sided column-split throughput kernel already wins from one
  // 8-drone wave per CU on -- 4096 x 10: 9.2 against 11.2 us, 8192 x 10: 11.0 against 16.3 -- the straight-line
  // latency kernel below that: 1024 x 10: 7.1 against 7.4 us; tools/order_sizes.py)
  // (order 7, tools/order_sizes.py with PROBE_ORDER=7, 10 segments: 4096 drones 7.4 us against 8.6 (small-batch
  //  kernel) and 11.0 (solve_kernel_reg); 8192: 8.9 / 12.8 / 12.9; 16 384: 13.4 / - / 16.2; 32 768: 21.6 / - / 22.1;
  //  65 536: 42.0 / - / 43.4 (eager launches); 2^20: 0.70 ms against 0.634 -- beyond 256 drones per CU the 16-drone
  //  waves of solve_kernel_reg are ahead.  That holds for its two-waves-per-SIMD instance only: with 11..20 segments
  //  solve_kernel_reg<4, 20> is one wave per SIMD and the column-split kernel stays ahead at every size -- 262 144 x 20:
  //  335 against 370 us, 2^20 x 20: 1.357 against 1.407 ms (59 against 57 % of the HBM peak), 2^20 x 14: 0.941 against
  //  1.020 ms (profiles/r04_order7_long.txt))
  const bool twin_ok = (M >= 4 && M <= (K == 5 ? kTwinMaxSeg : kTwinMaxSeg7) && !ctx->no_twin &&
                        (ctx->twin_max_drones > 0 ? N <= ctx->twin_max_drones
                                                   : (K == 5 || N <= ctx->n_cu * 256 || M > kRegMaxSeg)));
  const int twist_max = ctx->twist_max_drones > 0 ? ctx->twist_max_drones
                                                  : ctx->n_cu * (twin_ok ? 1 : 4) * kTwistDrones;
  if (M >= 2 && M <= (K == 4 ? kTwistMaxSeg : kTwistMaxSeg9) && N <= twist_max && !ctx->no_twist) {
    // small batch: at most one wavefront per SIMD -- halve the dependent chain instead
    const int nt8 = (N + kTwistDrones - 1) / kTwistDrones;
    const int nR = (M - 2) - (M - 2) / 2;
    // inputs + z stash (64 lanes x NU per knot; used by the long-path instances only)
    const size_t lds_bytes = ((size_t)kTwistDrones * (M + 1) * 5 + (size_t)64 * (K - 1) * nR) * sizeof(double);
#define MSNAP_TWIST_EXACT(MM)                                                                          \
  case MM:                                                                                             \
    note_kernel(ctx, "msnap::solve_kernel_twist<%d, %d, %d>", K, (MM - 2) - (MM - 2) / 2, MM);         \
    hipLaunchKernelGGL((solve_kernel_twist<K, (MM - 2) - (MM - 2) / 2, MM>), dim3(nt8), dim3(kWave),   \
                       lds_bytes, ctx->stream, wp, t, shared, N, coef, dur, status, nt8);             \
    break;
    if constexpr (K == 4) {
      switch (M) {   // 2 <= M <= kTwistMaxSeg
        MSNAP_TWIST_EXACT(2) MSNAP_TWIST_EXACT(3) MSNAP_TWIST_EXACT(4) MSNAP_TWIST_EXACT(5)
        MSNAP_TWIST_EXACT(6) MSNAP_TWIST_EXACT(7) MSNAP_TWIST_EXACT(8) MSNAP_TWIST_EXACT(9)
        MSNAP_TWIST_EXACT(10) MSNAP_TWIST_EXACT(11) MSNAP_TWIST_EXACT(12) MSNAP_TWIST_EXACT(13)
        MSNAP_TWIST_EXACT(14) MSNAP_TWIST_EXACT(15) MSNAP_TWIST_EXACT(16) MSNAP_TWIST_EXACT(17)
        MSNAP_TWIST_EXACT(18) MSNAP_TWIST_EXACT(19) MSNAP_TWIST_EXACT(20) MSNAP_TWIST_EXACT(21)
        MSNAP_TWIST_EXACT(22) MSNAP_TWIST_EXACT(23) MSNAP_TWIST_EXACT(24)
        default: return MSNAP_EINVAL;   // unreachable: the range is checked above
      }
    } else {
      switch (M) {   // 2 <= M <= kTwistMaxSeg9
        MSNAP_TWIST_EXACT(2) MSNAP_TWIST_EXACT(3) MSNAP_TWIST_EXACT(4) MSNAP_TWIST_EXACT(5)
        MSNAP_TWIST_EXACT(6) MSNAP_TWIST_EXACT(7) MSNAP_TWIST_EXACT(8) MSNAP_TWIST_EXACT(9)
        MSNAP_TWIST_EXACT(10) MSNAP_TWIST_EXACT(11) MSNAP_TWIST_EXACT(12)
        default: return MSNAP_EINVAL;
      }
    }
#undef MSNAP_TWIST_EXACT
    MSNAP_HIP(ctx, hipGetLastError());
    return MSNAP_OK;
  }
  if (twin_ok) {
    // large batch, even segment count: two-sided column-split kernel
    const int nt8 = (N + kTwinDrones - 1) / kTwinDrones;
    int grid = ctx->n_cu * 8 * 8;
    if (ctx->solve_grid_waves > 0) grid = ctx->solve_grid_waves;
    if (grid > nt8) grid = nt8;
#define MSNAP_TWIN(MM)                                                                                            \
  case MM:                                                                                                        \
    note_kernel(ctx, "msnap::solve_kernel_twin<%d, %d>", K, MM);                                                  \
    hipLaunchKernelGGL((solve_kernel_twin<K, MM>), dim3(grid), dim3(kWave), twin_lds_bytes(K, MM), ctx->stream,   \
                       wp, t, shared, N, coef, dur, status, nt8);                                                 \
    break;
    if constexpr (K == 5) {
      switch (M) {
        MSNAP_TWIN(4) MSNAP_TWIN(5) MSNAP_TWIN(6) MSNAP_TWIN(7) MSNAP_TWIN(8) MSNAP_TWIN(9) MSNAP_TWIN(10)
        MSNAP_TWIN(11) MSNAP_TWIN(12) MSNAP_TWIN(13) MSNAP_TWIN(14) MSNAP_TWIN(15) MSNAP_TWIN(16) MSNAP_TWIN(17)
        MSNAP_TWIN(18) MSNAP_TWIN(19) MSNAP_TWIN(20)
        default: return MSNAP_EINVAL;   // unreachable: the range is checked above
      }
    } else {
      switch (M) {
        MSNAP_TWIN(4) MSNAP_TWIN(5) MSNAP_TWIN(6) MSNAP_TWIN(7) MSNAP_TWIN(8) MSNAP_TWIN(9) MSNAP_TWIN(10)
        MSNAP_TWIN(11) MSNAP_TWIN(12) MSNAP_TWIN(13) MSNAP_TWIN(14) MSNAP_TWIN(15) MSNAP_TWIN(16) MSNAP_TWIN(17)
        MSNAP_TWIN(18) MSNAP_TWIN(19) MSNAP_TWIN(20)
        default: return MSNAP_EINVAL;
      }
    }
#undef MSNAP_TWIN
    MSNAP_HIP(ctx, hipGetLastError());
    return MSNAP_OK;
  }
  if (M >= 2 && M <= kRegMaxSeg2) {   // (one segment: the rolled kernel below)
    const size_t nu = K - 1;
    const size_t in_bytes = solve_input_words(M) * sizeof(double);
    const size_t lds_bytes = (in_bytes > tr_bytes ? in_bytes : tr_bytes) +
                             16 * nu * nu * (size_t)(M > 2 ? M - 2 : 0) * sizeof(double);
    // persistent waves (all resident at once) so that tile k+1's inputs can be prefetched during tile k
    const bool two_per_simd = (K <= 4 && M <= kRegMaxSeg);
    // Waves beyond the resident set (8 or 4 per CU) are started by the hardware as others retire, which
    // staggers the tiles' load / compute / store phases across the chip: 8x the resident set everywhere
    // (2^19..2^20 drones, order 7 M <= 10: 0.681 -> 0.647 ms, 4 tiles per wave still leave the cross-tile
    // prefetch its work; order 9 M > 10: 1.96 -> 1.55 ms; 65 536 drones, tools/reg_grid_probe.py: order 9
    // M <= 10 73.5 -> 71 us, order 7 M > 10 106 -> 101 us; the saturated launches of those two do not care).
    int grid = ctx->n_cu * (two_per_simd ? 8 : 4) * 8;
    if (ctx->solve_grid_waves > 0) grid = ctx->solve_grid_waves;   // msnap_set_option: tests walk several tiles per wave
    if (grid > ntiles) grid = ntiles;
    note_kernel(ctx, "msnap::solve_kernel_reg<%d, %d>", K, M <= kRegMaxSeg ? kRegMaxSeg : kRegMaxSeg2);
    if (M <= kRegMaxSeg)
      hipLaunchKernelGGL((solve_kernel_reg<K, kRegMaxSeg>), dim3(grid), dim3(kWave), lds_bytes, ctx->stream,
                         wp, t, shared, N, M, coef, dur, status, ntiles);
    else
      hipLaunchKernelGGL((solve_kernel_reg<K, kRegMaxSeg2>), dim3(grid), dim3(kWave), lds_bytes, ctx->stream,
                         wp, t, shared, N, M, coef, dur, status, ntiles);
    MSNAP_HIP(ctx, hipGetLastError());
    return MSNAP_OK;
  }
  const size_t words = solve_scratch_words(K, M);
  const size_t lds_bytes = tr_bytes + (words + solve_input_words(M)) * sizeof(double);
  const size_t bytes = words * sizeof(double);
  note_kernel(ctx, "msnap::solve_kernel<%d, %s>", K, lds_bytes <= kMaxLdsBytes ? "false" : "true");
  if (lds_bytes <= kMaxLdsBytes) {
    hipLaunchKernelGGL((solve_kernel<K, false>), dim3(ntiles), dim3(kWave), lds_bytes, ctx->stream, wp, t,
                       shared, N, M, coef, dur, status, (double *)nullptr, ntiles);
  } else {
    // too many segments for LDS: same recurrence on a global scratch slab,
    // persistent grid so the slab stays bounded
    int grid = ctx->n_cu * 8;
    if (grid > ntiles) grid = ntiles;
    int rc = ensure(ctx, ctx->scratch, (size_t)grid * bytes);
    if (rc) return rc;
    hipLaunchKernelGGL((solve_kernel<K, true>), dim3(grid), dim3(kWave), tr_bytes, ctx->stream, wp, t,
                       shared, N, M, coef, dur, status, (double *)ctx->scratch.p, ntiles);
  }
  MSNAP_HIP(ctx, hipGetLastError());
  return MSNAP_OK;
}

bool solve_uses_global_scratch(const msnap_ctx *ctx, int n_seg) {
  if (n_seg <= kRegMaxSeg2) return false;
  const size_t tr_bytes = (size_t)ctx->khalf * kTrPitch * 16;
  return tr_bytes + (solve_scratch_words(ctx->khalf, n_seg) + solve_input_words(n_seg)) * sizeof(double) >
         kMaxLdsBytes;
}

#ifdef MSNAP_TOOLS_TIMELINE
extern "C" int msnap_debug_read_timeline(unsigned long long *out, int n_words) {
  if (hipDeviceSynchronize() != hipSuccess) return MSNAP_EHIP;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_timeline), (size_t)n_words * 8) == hipSuccess ? MSNAP_OK : MSNAP_EHIP;
}
#endif

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
