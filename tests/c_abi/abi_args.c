/* Every entry point of include/msnap.h driven through its argument checks and error paths WITHOUT a GPU, for the
 * address/undefined-behaviour sanitizer build of the host side (tests/test_sanitizers.py).  With no device
 * msnap_create must fail with MSNAP_ENODEVICE (there is no CPU fallback) and every call on a null context must
 * come back with MSNAP_EINVAL instead of touching it. */
#include <stdio.h>
#include <string.h>

#include "msnap.h"

static int fails = 0;
#define EXPECT(cond)                                                   \
  do {                                                                 \
    if (!(cond)) {                                                     \
      fprintf(stderr, "abi_args: line %d: %s\n", __LINE__, #cond);    \
      ++fails;                                                         \
    }                                                                  \
  } while (0)

int main(void) {
  double d[64] = {0};
  float f[64] = {0};
  int32_t i[16] = {0};
  unsigned char bytes[256] = {0};
  msnap_ctx *ctx = (msnap_ctx *)0x1;

  EXPECT(msnap_version() >= 100);
  for (int code = 1; code >= -12; --code) EXPECT(msnap_strerror(code) != NULL && strlen(msnap_strerror(code)) > 0);
  EXPECT(strcmp(msnap_strerror(MSNAP_OK), "ok") == 0);
  EXPECT(strcmp(msnap_last_hip_error(NULL), "") == 0);
  EXPECT(strcmp(msnap_last_kernel(NULL), "") == 0);

  EXPECT(msnap_create(NULL, 0, 7, 10) == MSNAP_EINVAL);
  EXPECT(msnap_create(&ctx, 0, 8, 10) == MSNAP_EORDER && ctx == NULL);
  EXPECT(msnap_create(&ctx, 0, 7, 0) == MSNAP_ESEGMENTS && ctx == NULL);
  int rc = msnap_create(&ctx, 0, 7, 10);
  if (rc == MSNAP_OK) {          /* a GPU is present after all: release it, the rest still holds */
    msnap_destroy(ctx);
    ctx = NULL;
  } else {
    EXPECT(rc == MSNAP_ENODEVICE && ctx == NULL);
  }
  EXPECT(msnap_create(&ctx, 1 << 20, 9, 10) == MSNAP_ENODEVICE && ctx == NULL);
  msnap_destroy(NULL);

  long v = 0;
  EXPECT(msnap_set_stream(NULL, NULL) == MSNAP_EINVAL);
  EXPECT(msnap_use_own_stream(NULL) == MSNAP_EINVAL);
  EXPECT(msnap_get_stream(NULL) == NULL);
  EXPECT(msnap_sync(NULL) == MSNAP_EINVAL);
  EXPECT(msnap_set_option(NULL, "no_twist", 1) == MSNAP_EINVAL);
  EXPECT(msnap_get_option(NULL, "no_twist", &v) == MSNAP_EINVAL);
  EXPECT(msnap_timer_start(NULL) == MSNAP_EINVAL);
  EXPECT(msnap_timer_stop(NULL, f) == MSNAP_EINVAL);
  void *hp = NULL;
  EXPECT(msnap_host_alloc(NULL, 16) == MSNAP_EINVAL);
  EXPECT(msnap_host_free(NULL) == MSNAP_OK || msnap_host_free(NULL) == MSNAP_EINVAL);
  (void)hp;

  EXPECT(msnap_solve_batch(NULL, 1, 3, d, d, 0, d, d, i) == MSNAP_EINVAL);
  EXPECT(msnap_solve_batch_device(NULL, 1, 3, d, d, 0, d, d, i) == MSNAP_EINVAL);
  EXPECT(msnap_grid_prepare(NULL, 3, d) == MSNAP_EINVAL);
  EXPECT(msnap_grid_prepare_device(NULL, 3, d) == MSNAP_EINVAL);
  EXPECT(msnap_grid_segments(NULL) == MSNAP_EINVAL);
  EXPECT(msnap_solve_grid(NULL, 1, 3, d, d, d, i) == MSNAP_EINVAL);
  EXPECT(msnap_solve_grid_device(NULL, 1, 3, d, d, d, i) == MSNAP_EINVAL);
  EXPECT(msnap_release_graph_buffers(NULL, NULL) == MSNAP_EINVAL);
  EXPECT(msnap_pack_pol_matrix(NULL, 1, 1, d, d, f) == MSNAP_EINVAL);
  EXPECT(msnap_pack_pol_matrix_device(NULL, 1, 1, d, d, f) == MSNAP_EINVAL);
  EXPECT(msnap_formation_transform(NULL, 1, 1, d, d, d) == MSNAP_EINVAL);
  EXPECT(msnap_formation_transform_device(NULL, 1, 1, d, d, d) == MSNAP_EINVAL);
  EXPECT(msnap_sample(NULL, 1, 1, d, d, 0.1, 2, 3, d) == MSNAP_EINVAL);
  EXPECT(msnap_sample_device(NULL, 1, 1, d, d, 0.1, 2, 3, d) == MSNAP_EINVAL);
  EXPECT(msnap_eval_flat(NULL, 1, 1, d, d, 1, d, d) == MSNAP_EINVAL);
  EXPECT(msnap_eval_flat_device(NULL, 1, 1, d, d, 1, d, d) == MSNAP_EINVAL);
  EXPECT(msnap_snap_cost(NULL, 1, 1, d, d, d) == MSNAP_EINVAL);
  EXPECT(msnap_snap_cost_device(NULL, 1, 1, d, d, d) == MSNAP_EINVAL);
  EXPECT(msnap_formation_collide(NULL, 1, 0, 1, 2, d, d, 0.1, d, i, i) == MSNAP_EINVAL);
  EXPECT(msnap_formation_collide_device(NULL, 1, 0, 1, 2, d, d, 0.1, d, i, i) == MSNAP_EINVAL);
  EXPECT(msnap_collide_rows_t_doubles(0, 5) == 0 && msnap_collide_rows_t_doubles(1, 2) == 128 * 2 * 3);
  EXPECT(msnap_collide_rows_t_doubles(4096, 91) == (size_t)4096 * 91 * 3 && msnap_collide_rows_t_doubles(129, 1) == 256 * 3);
  EXPECT(msnap_sample_collide_device(NULL, 1, 1, d, d, 0.1, 2, d, d) == MSNAP_EINVAL);
  EXPECT(msnap_solve_grid_sample_device(NULL, 1, 1, d, 0.1, 2, d, d, NULL, d, d) == MSNAP_EINVAL);
  EXPECT(msnap_formation_collide_t_device(NULL, 1, 0, 1, 2, d, d, d, 0.1, d, i, i) == MSNAP_EINVAL);
  EXPECT(msnap_formation_part_bytes(0) == 0 && msnap_formation_part_bytes(-3) == 0);
  EXPECT(msnap_formation_part_bytes(1) == 16 && msnap_formation_part_bytes(2) == 24 && msnap_formation_part_bytes(4096) == 49152);
  EXPECT(msnap_formation_collide_takes_broad_phase(NULL, 4096, 0, 4096, 91) == 0);
  {
    int pays = 7;
    EXPECT(msnap_formation_whole_pass_pays(NULL, 4096, 8, &pays) == MSNAP_EINVAL);
  }
  EXPECT(msnap_formation_collide_part(NULL, 2, 2, d, 0, 1, bytes) == MSNAP_EINVAL);
  EXPECT(msnap_formation_collide_part_device(NULL, 2, 2, d, 0, 1, bytes) == MSNAP_EINVAL);
  EXPECT(msnap_formation_collide_finish(NULL, 2, 1, bytes, 0, 2, 0.1, d, i, i) == MSNAP_EINVAL);
  EXPECT(msnap_formation_collide_finish_device(NULL, 2, 1, bytes, 0, 2, 0.1, d, i, i) == MSNAP_EINVAL);
  EXPECT(msnap_mesh_sweep(NULL, 1, 2, d, 1, d, 0.1, d, i) == MSNAP_EINVAL);
  EXPECT(msnap_mesh_sweep_device(NULL, 1, 2, d, 1, d, 0.1, d, i) == MSNAP_EINVAL);
  EXPECT(msnap_mesh_validity(NULL, 1, d, 1, d, 1, d, i) == MSNAP_EINVAL);
  EXPECT(msnap_mesh_validity_device(NULL, 1, d, 1, d, 1, d, i) == MSNAP_EINVAL);

  printf("abi_args: %d failure(s)\n", fails);
  return fails ? 1 : 0;
}
