/* msnap_oracle.c -- plain-C restatement of the reference algorithm.
 * TEST INFRASTRUCTURE ONLY: used by tests/ as a second checker and by
 * bench.py's cpu_baseline leg ("port").  Never linked into libmsnap.so.
 *
 * Follows src/optimizations/calculatingTrajectories.py of the reference:
 *   - deriv_row():   Polynomial.pol_coeffs_at_t / .derivative
 *                    (src/optimizations/uav_trajectory.py:25-36)
 *   - assemble():    the dense 8M x 8M matrix A and vector b, row for row
 *                    (calculatingTrajectories.py:48-131)
 *   - lu_solve():    np.linalg.solve = LAPACK dgesv: LU with partial (row)
 *                    pivoting, then forward / back substitution (:137).
 *                    NumPy/LAPACK are not vendored in the reference; dgesv's
 *                    published algorithm (dgetrf + dgetrs, unblocked) is
 *                    restated here.
 *   - one solve per axis, x,y,z,yaw (calculate_trajectory4D, :200-213) when
 *     faithful != 0; one factorisation with 4 right-hand sides otherwise
 *     (same numbers up to rounding; "B1 optimised CPU" of BASELINE.md).
 * Pinned against the reference's outputs in tests/test_oracle_golden.py.
 *
 * Also here, for inputs too large for the NumPy restatement's Python loops
 * (tests/golden/make_formation_golden.py, the full-size -m gpu tests):
 *   - msnap_oracle_sample():  PiecewisePolynomial.eval on t = s*dt
 *                    (src/optimizations/uav_trajectory.py:154-169, Horner of :17-22)
 *   - msnap_oracle_formation_collide(), msnap_oracle_mesh_sweep(): the two collision
 *     passes.  The reference has neither (parity unpinned); these follow
 *     the definitions of oracle/msnap_oracle.py (formation_collide, mesh_sweep,
 *     point_triangle_dist2) operation for operation and are checked against them
 *     in tests/test_oracle_golden.py.
 * Build with -ffp-contract=off: every multiply and add rounds separately, as in NumPy.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static void deriv_row(int j, double t, int ncoef, double *row) {
  for (int k = 0; k < ncoef; ++k) row[k] = 0.0;
  for (int k = j; k < ncoef; ++k) {
    double f = 1.0;
    for (int q = k - j + 1; q <= k; ++q) f *= (double)q;
    row[k] = f * pow(t, (double)(k - j)); /* pow(0,0) == 1 like Python's 0**0 */
  }
}

/* A: n x n row-major (n = ncoef*M), b: n x nrhs row-major; values: [m][stride] */
static void assemble(int M, int ncoef, const double *times, const double *values, int vstride,
                     int nrhs, double *A, double *b) {
  const int k = ncoef / 2, n = ncoef * M, m = M + 1;
  double row[16];
  memset(A, 0, sizeof(double) * (size_t)n * n);
  memset(b, 0, sizeof(double) * (size_t)n * nrhs);
  double prev_t = 0.0;
  for (int i = 0; i < m; ++i) {
    const double t = times[i] - prev_t; /* :59 */
    if (i == 0 || i == M) {            /* :65 */
      for (int j = 0; j < k; ++j) {
        deriv_row(j, t, ncoef, row);
        if (i == 0)
          memcpy(A + (size_t)j * n, row, sizeof(double) * ncoef); /* :73 */
        else
          memcpy(A + (size_t)(n - k + j) * n + (size_t)ncoef * (M - 1), row, sizeof(double) * ncoef);
      }
      for (int r = 0; r < nrhs; ++r) {
        if (i == 0) b[0 * nrhs + r] = values[(size_t)0 * vstride + r];          /* :82-85 */
        else b[(size_t)(n - k) * nrhs + r] = values[(size_t)i * vstride + r];   /* :87 */
      }
      if (M == 0) break;
    } else {
      const int base = k + (i - 1) * ncoef; /* :112 */
      for (int j = 1; j < ncoef - 1; ++j) {
        deriv_row(j, t, ncoef, row);
        memcpy(A + (size_t)(base + j - 1) * n + (size_t)ncoef * (i - 1), row, sizeof(double) * ncoef);
        deriv_row(j, 0.0, ncoef, row);
        for (int c = 0; c < ncoef; ++c) A[(size_t)(base + j - 1) * n + (size_t)ncoef * i + c] = -row[c];
      }
      deriv_row(0, t, ncoef, row);
      memcpy(A + (size_t)(base + ncoef - 2) * n + (size_t)ncoef * (i - 1), row, sizeof(double) * ncoef);
      deriv_row(0, 0.0, ncoef, row);
      memcpy(A + (size_t)(base + ncoef - 1) * n + (size_t)ncoef * i, row, sizeof(double) * ncoef);
      for (int r = 0; r < nrhs; ++r) {
        b[(size_t)(base + ncoef - 2) * nrhs + r] = values[(size_t)i * vstride + r];
        b[(size_t)(base + ncoef - 1) * nrhs + r] = values[(size_t)i * vstride + r];
      }
    }
    prev_t = times[i]; /* :131 */
  }
}

/* dgesv: in-place LU with partial pivoting, then solve for nrhs columns.
 * returns 0, or j+1 if U(j,j) == 0 (LAPACK info > 0 -> numpy LinAlgError). */
static int lu_solve(int n, int nrhs, double *A, double *b) {
  for (int j = 0; j < n; ++j) {
    int p = j;
    double best = fabs(A[(size_t)j * n + j]);
    for (int r = j + 1; r < n; ++r) {
      const double v = fabs(A[(size_t)r * n + j]);
      if (v > best) { best = v; p = r; }
    }
    if (best == 0.0) return j + 1;
    if (p != j) {
      for (int c = 0; c < n; ++c) {
        const double tmp = A[(size_t)j * n + c];
        A[(size_t)j * n + c] = A[(size_t)p * n + c];
        A[(size_t)p * n + c] = tmp;
      }
      for (int c = 0; c < nrhs; ++c) {
        const double tmp = b[(size_t)j * nrhs + c];
        b[(size_t)j * nrhs + c] = b[(size_t)p * nrhs + c];
        b[(size_t)p * nrhs + c] = tmp;
      }
    }
    const double inv = 1.0 / A[(size_t)j * n + j];
    for (int r = j + 1; r < n; ++r) {
      const double l = A[(size_t)r * n + j] * inv;
      if (l == 0.0) continue;
      A[(size_t)r * n + j] = l;
      for (int c = j + 1; c < n; ++c) A[(size_t)r * n + c] -= l * A[(size_t)j * n + c];
      for (int c = 0; c < nrhs; ++c) b[(size_t)r * nrhs + c] -= l * b[(size_t)j * nrhs + c];
    }
  }
  for (int j = n - 1; j >= 0; --j) {
    for (int c = 0; c < nrhs; ++c) {
      double v = b[(size_t)j * nrhs + c];
      for (int q = j + 1; q < n; ++q) v -= A[(size_t)j * n + q] * b[(size_t)q * nrhs + c];
      b[(size_t)j * nrhs + c] = v / A[(size_t)j * n + j];
    }
  }
  return 0;
}

/* wp [N][M+1][4], t [N][M+1] or [M+1]; coef [N][M][4][ncoef], dur [N][M];
 * info [N] (0 or singular column + 1).  Returns the number of threads used. */
int msnap_oracle_solve_batch(int n_drones, int n_seg, int ncoef, const double *wp, const double *t,
                             int shared_times, int faithful, int n_threads, double *coef, double *dur,
                             int *info) {
  const int M = n_seg, m = M + 1, n = ncoef * M;
  int used = 1;
#ifdef _OPENMP
  if (n_threads <= 0) n_threads = omp_get_max_threads();
  used = n_threads;
#else
  (void)n_threads;
#endif
#pragma omp parallel num_threads(used)
  {
    double *A = (double *)malloc(sizeof(double) * (size_t)n * n);
    double *b = (double *)malloc(sizeof(double) * (size_t)n * 4);
#pragma omp for schedule(static)
    for (int d = 0; d < n_drones; ++d) {
      const double *w = wp + (size_t)d * m * 4;
      const double *tt = shared_times ? t : t + (size_t)d * m;
      double *c = coef + (size_t)d * M * 4 * ncoef;
      int bad = 0;
      if (faithful) {
        for (int a = 0; a < 4; ++a) {
          assemble(M, ncoef, tt, w + a, 4, 1, A, b);
          const int rc = lu_solve(n, 1, A, b);
          if (rc) bad = rc;
          for (int s = 0; s < M; ++s)
            for (int q = 0; q < ncoef; ++q) c[((size_t)s * 4 + a) * ncoef + q] = b[(size_t)s * ncoef + q];
        }
      } else {
        assemble(M, ncoef, tt, w, 4, 4, A, b);
        bad = lu_solve(n, 4, A, b);
        for (int s = 0; s < M; ++s)
          for (int a = 0; a < 4; ++a)
            for (int q = 0; q < ncoef; ++q)
              c[((size_t)s * 4 + a) * ncoef + q] = b[((size_t)s * ncoef + q) * 4 + a];
      }
      for (int s = 0; s < M; ++s) dur[(size_t)d * M + s] = tt[s + 1] - tt[s];
      info[d] = bad;
    }
    free(A);
    free(b);
  }
  return used;
}

int msnap_oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}


/* PiecewisePolynomial.eval (uav_trajectory.py:154-169): strict '<' lookup over the running sum
 * of durations; past the end the last piece extrapolates at t - sum(dur[:-1]).  Horner with a
 * separate multiply and add (uav_trajectory.py:17-22).
 * coef [N][M][4][ncoef], dur [N][M] -> pos [N][S][naxes], sample s at t = s*dt. */
void msnap_oracle_sample(int n_drones, int n_seg, int ncoef, const double *coef, const double *dur,
                         double dt, int n_samples, int naxes, int n_threads, double *pos) {
#ifdef _OPENMP
  if (n_threads <= 0) n_threads = omp_get_max_threads();
#else
  (void)n_threads;
#endif
#pragma omp parallel for schedule(static) num_threads(n_threads)
  for (int d = 0; d < n_drones; ++d) {
    const double *dr = dur + (size_t)d * n_seg;
    for (int s = 0; s < n_samples; ++s) {
      const double t = (double)s * dt;
      double acc = 0.0;
      int seg = -1;
      for (int i = 0; i < n_seg; ++i) {
        if (t < acc + dr[i]) { seg = i; break; }
        acc = acc + dr[i];
      }
      if (seg < 0) {               /* :161-163 */
        seg = n_seg - 1;
        acc = 0.0;
        for (int i = 0; i < n_seg - 1; ++i) acc = acc + dr[i];
      }
      const double tl = t - acc;
      for (int a = 0; a < naxes; ++a) {
        const double *c = coef + (((size_t)d * n_seg + seg) * 4 + a) * ncoef;
        double x = 0.0;
        for (int q = ncoef - 1; q >= 0; --q) x = x * tl + c[q];
        pos[((size_t)d * n_samples + s) * naxes + a] = x;
      }
    }
  }
}

/* pos [N][S][3]: min over other drones j and samples of |p_i - p_j|; partner = lowest j attaining it;
 * hit = min_dist < 2 r.  Squared distance as include/msnap.h defines it: fma(dz, dz, fma(dy, dy, dx*dx))
 * (libm's fma is correctly rounded whether or not the host has the instruction).  NaN samples never win a
 * minimum. */
void msnap_oracle_formation_collide(int n, int n_samples, const double *pos, double radius, int n_threads,
                                    double *min_dist, int *partner, int *hit) {
#ifdef _OPENMP
  if (n_threads <= 0) n_threads = omp_get_max_threads();
#else
  (void)n_threads;
#endif
#pragma omp parallel for schedule(dynamic, 16) num_threads(n_threads)
  for (int i = 0; i < n; ++i) {
    const double *pi = pos + (size_t)i * n_samples * 3;
    double best = INFINITY;
    int bj = -1;
    for (int j = 0; j < n; ++j) {
      if (j == i) continue;
      const double *pj = pos + (size_t)j * n_samples * 3;
      double m = INFINITY;
      for (int s = 0; s < n_samples; ++s) {
        const double dx = pj[3 * s + 0] - pi[3 * s + 0];
        const double dy = pj[3 * s + 1] - pi[3 * s + 1];
        const double dz = pj[3 * s + 2] - pi[3 * s + 2];
        const double d2 = fma(dz, dz, fma(dy, dy, dx * dx));
        if (d2 < m) m = d2;
      }
      if (m < best) { best = m; bj = j; }
    }
    min_dist[i] = sqrt(best);
    partner[i] = bj;
    hit[i] = sqrt(best) < 2.0 * radius;
  }
}

/* closest-point regions (Ericson 5.1.5), as oracle/msnap_oracle.py point_triangle_dist2 */
static double pt_tri_d2(const double *p, const double *t) {
  const double ax = t[0], ay = t[1], az = t[2], bx = t[3], by = t[4], bz = t[5], cx = t[6], cy = t[7], cz = t[8];
  const double abx = bx - ax, aby = by - ay, abz = bz - az;
  const double acx = cx - ax, acy = cy - ay, acz = cz - az;
  const double apx = p[0] - ax, apy = p[1] - ay, apz = p[2] - az;
  const double d1 = abx * apx + aby * apy + abz * apz;
  const double d2 = acx * apx + acy * apy + acz * apz;
  double qx, qy, qz;
  if (d1 <= 0.0 && d2 <= 0.0) { qx = ax; qy = ay; qz = az; }
  else {
    const double bpx = p[0] - bx, bpy = p[1] - by, bpz = p[2] - bz;
    const double d3 = abx * bpx + aby * bpy + abz * bpz;
    const double d4 = acx * bpx + acy * bpy + acz * bpz;
    if (d3 >= 0.0 && d4 <= d3) { qx = bx; qy = by; qz = bz; }
    else {
      const double vc = d1 * d4 - d3 * d2;
      if (vc <= 0.0 && d1 >= 0.0 && d3 <= 0.0) {
        const double v = d1 / (d1 - d3);
        qx = ax + v * abx; qy = ay + v * aby; qz = az + v * abz;
      } else {
        const double cpx = p[0] - cx, cpy = p[1] - cy, cpz = p[2] - cz;
        const double d5 = abx * cpx + aby * cpy + abz * cpz;
        const double d6 = acx * cpx + acy * cpy + acz * cpz;
        if (d6 >= 0.0 && d5 <= d6) { qx = cx; qy = cy; qz = cz; }
        else {
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
              qx = ax + abx * v + acx * w; qy = ay + aby * v + acy * w; qz = az + abz * v + acz * w;
            }
          }
        }
      }
    }
  }
  const double ex = p[0] - qx, ey = p[1] - qy, ez = p[2] - qz;
  return ex * ex + ey * ey + ez * ez;
}

/* pos [N][S][3], tris [T][3][3]: min over samples and triangles of the point-triangle distance;
 * hit = min_dist < r */
void msnap_oracle_mesh_sweep(int n, int n_samples, const double *pos, int n_tris, const double *tris,
                             double radius, int n_threads, double *min_dist, int *hit) {
#ifdef _OPENMP
  if (n_threads <= 0) n_threads = omp_get_max_threads();
#else
  (void)n_threads;
#endif
#pragma omp parallel for schedule(static) num_threads(n_threads)
  for (int i = 0; i < n; ++i) {
    double best = INFINITY;
    for (int s = 0; s < n_samples; ++s)
      for (int t = 0; t < n_tris; ++t) {
        const double v = pt_tri_d2(pos + ((size_t)i * n_samples + s) * 3, tris + (size_t)t * 9);
        if (v < best) best = v;
      }
    min_dist[i] = sqrt(best);
    hit[i] = sqrt(best) < radius;
  }
}
