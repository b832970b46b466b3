/* Plain-C consumer of include/msnap.h: proves the boundary is a C-ABI (no C++ or
 * Python types), solves BASELINE.json configs[0] (4 waypoints, t = 0,1,3,4) on the GPU
 * and compares with the reference's known-answer vector.
 * Built and run by tests/test_aux_gpu.py::test_c_abi_from_plain_c. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "msnap.h"

int main(void) {
  const double wp[4][4] = {{0, 0, 0, 0}, {2, 2.2, 0.3, 0}, {4, 8, 0.8, 0}, {1, 2, 0.4, 0.5}};
  const double t[4] = {0, 1, 3, 4};
  /* x axis, segment 1 and yaw axis, segment 2 of the reference output */
  const double x1[8] = {2.0, 5.640237077579474, 3.6053781455735336, -2.1836605189543383, -1.9836819799625636,
                        0.6011424658810961, 0.2332205334418886, -0.06762862464760394};
  const double yaw2[8] = {0.0, 1.101463296964554, -0.06436634858876582, -0.9704678440823047,
                          -0.003731077074223242, 0.4557143271358343, 0.19468791443672137, -0.21330026879181638};
  double coef[3][4][8], dur[3];
  int32_t status[1];
  msnap_ctx *ctx = NULL;
  int rc = msnap_create(&ctx, 0, 7, 16);
  if (rc != MSNAP_OK) {
    fprintf(stderr, "msnap_create: %s\n", msnap_strerror(rc));
    return 2;
  }
  rc = msnap_solve_batch(ctx, 1, 3, &wp[0][0], t, 0, &coef[0][0][0], dur, status);
  if (rc != MSNAP_OK || status[0] != MSNAP_ST_OK) {
    fprintf(stderr, "msnap_solve_batch: %s (%s), status %d\n", msnap_strerror(rc), msnap_last_hip_error(ctx), status[0]);
    return 3;
  }
  double err = 0.0;
  for (int k = 0; k < 8; ++k) {
    err = fmax(err, fabs(coef[1][0][k] - x1[k]));
    err = fmax(err, fabs(coef[2][3][k] - yaw2[k]));
  }
  /* the float32 row the reference writes to Pol_matrix_*.csv */
  float row[3][33];
  rc = msnap_pack_pol_matrix(ctx, 1, 3, &coef[0][0][0], dur, &row[0][0]);
  if (rc != MSNAP_OK) return 4;
  /* the same solve on a prepared shared grid: the segment count is the caller's statement of what its buffers
   * hold -- another count than the grid's is refused (MSNAP_ESEGMENTS) and nothing is written */
  double gcoef[3][4][8], gdur[3];
  int32_t gstatus[1] = {77};
  if (msnap_grid_segments(ctx) != 0) return 7;
  if (msnap_solve_grid(ctx, 1, 3, &wp[0][0], &gcoef[0][0][0], gdur, gstatus) != MSNAP_ENOGRID) return 8;
  if (msnap_grid_prepare(ctx, 3, t) != MSNAP_OK || msnap_grid_segments(ctx) != 3) return 9;
  gcoef[0][0][0] = -12345.0;
  if (msnap_solve_grid(ctx, 1, 4, &wp[0][0], &gcoef[0][0][0], gdur, gstatus) != MSNAP_ESEGMENTS) return 10;
  if (msnap_solve_grid(ctx, 1, 2, &wp[0][0], &gcoef[0][0][0], gdur, gstatus) != MSNAP_ESEGMENTS) return 11;
  if (gcoef[0][0][0] != -12345.0 || gstatus[0] != 77) return 12;
  if (msnap_solve_grid(ctx, 1, 3, &wp[0][0], &gcoef[0][0][0], gdur, gstatus) != MSNAP_OK || gstatus[0] != MSNAP_ST_OK) return 13;
  for (int s = 0; s < 3; ++s)
    for (int a = 0; a < 4; ++a)
      for (int k = 0; k < 8; ++k) err = fmax(err, fabs(gcoef[s][a][k] - coef[s][a][k]));
  size_t released = 1;
  if (msnap_release_graph_buffers(ctx, &released) != MSNAP_OK || released != 0) return 14;
  msnap_destroy(ctx);
  printf("version %d  dur %.1f %.1f %.1f  max abs err %.3e  row[1][1]=%.7f\n", msnap_version(), dur[0], dur[1], dur[2], err,
         row[1][1]);
  if (!(dur[0] == 1.0 && dur[1] == 2.0 && dur[2] == 1.0)) return 5;
  if (row[1][0] != 2.0f || row[1][1] != 2.0f) return 6;
  return err < 1e-9 ? 0 : 1;
}
