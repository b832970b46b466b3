// Sanitizer build of the C-ABI's HOST side only (tests/test_sanitizers.py): csrc/msnap_api.hip compiled for the host
// with -fsanitize=address,undefined and linked against these stand-ins for the kernel launchers.  Without a GPU no
// context can be created, so none of them is ever reached: what runs is the argument checking, the option table,
// the error strings and the create/destroy error paths.  Test infrastructure, never part of libmsnap.so.
#include <cstdlib>

#include "msnap_internal.h"

namespace msnap {

static int unreachable() { abort(); }

int launch_solve(msnap_ctx *, int, int, const double *, const double *, int, double *, double *, int32_t *) { return unreachable(); }
int launch_pack(msnap_ctx *, int, int, const double *, const double *, float *) { return unreachable(); }
int launch_formation_transform(msnap_ctx *, int, int, const double *, const double *, double *) { return unreachable(); }
int launch_sample(msnap_ctx *, int, int, const double *, const double *, double, int, int, double *, double *, bool) { return unreachable(); }
int launch_eval_flat(msnap_ctx *, int, int, const double *, const double *, int, const double *, double *) { return unreachable(); }
int launch_snap_cost(msnap_ctx *, int, int, const double *, const double *, double *) { return unreachable(); }
int launch_formation_collide(msnap_ctx *, int, int, int, int, const double *, const double *, double, double *, int32_t *,
                             int32_t *, const double *) { return unreachable(); }
int launch_formation_collide_part(msnap_ctx *, int, int, const double *, int, int, double *, int32_t *) { return unreachable(); }
bool formation_collide_takes_broad_phase(const msnap_ctx *, int, int, int, int) { return false; }
bool collide_counts_by_groups(const msnap_ctx *, int, int, int) { return false; }
int launch_formation_collide_finish(msnap_ctx *, int, int, const void *, size_t, int, int, double, double *, int32_t *,
                                    int32_t *) { return unreachable(); }
int launch_mesh_sweep(msnap_ctx *, int, int, const double *, int, const double *, double, double *, int32_t *) { return unreachable(); }
int launch_mesh_validity(msnap_ctx *, int, const double *, int, const double *, int, const double *, int32_t *) { return unreachable(); }
int solve_kernel_setup(msnap_ctx *) { return unreachable(); }
bool solve_uses_global_scratch(const msnap_ctx *, int) { return unreachable() != 0; }
int launch_grid_prepare(msnap_ctx *, int, const double *, int) { return unreachable(); }
int launch_solve_grid(msnap_ctx *, int, const double *, double *, double *, int32_t *) { return unreachable(); }
int launch_grid_sample(msnap_ctx *, int, const double *, double, int, double *, double *, int32_t *, double *, double *, bool) { return unreachable(); }
bool grid_gemm_supported(const msnap_ctx *, int) { return unreachable() != 0; }

}  // namespace msnap
