// Internal declarations shared by the translation units of libmsnap.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/msnap.h"

#define MSNAP_VERSION_NUM 200  /* 0.2.0: n_seg on msnap_solve_grid[_device], msnap_grid_segments, graph-buffer retention */
/* int32 words of the pairwise pass's broad-phase hand-over block that msnap_get_option reads back */
#define MSNAP_COLLIDE_META_SHARES 128
#define MSNAP_COLLIDE_META_GROUPS 132

namespace msnap {

constexpr int kWave = 64;
constexpr int kAxes = 4;                // x, y, z, yaw: one lane each
constexpr int kDronesPerWave = 16;      // 16 drones x 4 axis-lanes = one wavefront
constexpr size_t kMaxLdsBytes = 160 * 1024;

// doubles one 16-drone tile of the K1 solve keeps between its forward and backward
// sweep (1/T, G_i blocks, z_i vectors) -- LDS, or a global slab for very long paths
inline size_t solve_scratch_words(int khalf, int n_seg) {
  const size_t nu = (size_t)khalf - 1;
  const size_t M = (size_t)n_seg;
  const size_t knots = M > 0 ? M - 1 : 0;
  return 16 * M + 16 * nu * nu * knots + 64 * nu * knots;
}
// doubles of the LDS input stage (raw copy of the tile's waypoints and times)
inline size_t solve_input_words(int n_seg) { return (size_t)16 * (n_seg + 1) * 5; }

// s_waitcnt vmcnt(N): returns once all but the N most recent vector-memory operations of the wave have
// completed (loads and stores share the counter on gfx950 and retire in order).  For hand-issued asm
// prefetch loads followed by AT LEAST N stores; the immediate is a compile-time constant (a run-time
// choice between immediates compiles into a flag dispatch with a static path around every wait, which
// tools/check_prefetch_isa.py could not tell from a missing wait).
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "6-bit vmcnt field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// A workgroup barrier that orders LDS only.  __syncthreads() is a release / acquire fence pair around s_barrier, and the
// release waits for every global store the wave has in flight (s_waitcnt vmcnt(0)): a kernel that has just issued
// stores would sit out their round trip at the next barrier although nothing in the workgroup reads them back.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// growable device buffer
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  bool in_graph = false;   // a stream capture has used this block: a graph may hold pointers into it (ensure())
};
// a block a graph may still replay on, replaced by a larger one: freed by msnap_release_graph_buffers / msnap_destroy
struct RetiredBuf {
  void *p;
  size_t cap;
  RetiredBuf *next;
};

}  // namespace msnap

struct msnap_ctx {
  int device = 0;
  int order = 7;
  int khalf = 4;           // (order+1)/2
  int max_segments = 0;
  int n_cu = 256;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  msnap::DevBuf scratch;   // global-memory scratch for n_seg too large for LDS
  msnap::DevBuf stage[8];  // device staging for the host-pointer entry points
  // chunked host-pointer solves: two streams alternate H2D -> kernel -> D2H over chunks of drones,
  // each with its own staging set (wp, t, coef, dur, status)
  hipStream_t pipe_stream[2] = {nullptr, nullptr};
  hipEvent_t pipe_start = nullptr;
  msnap::DevBuf pipe[2][5];
  size_t pipe_chunk_bytes = 64u << 20;   // output bytes per chunk (MSNAP_PIPE_CHUNK_MB overrides)
  // small host-pointer solves: one page-locked bounce buffer, one upload and one download per call
  void *bounce = nullptr;
  size_t bounce_cap = 0;
  // shared-time-grid operator (K2): built by msnap_grid_prepare
  msnap::DevBuf grid_t, grid_wp, grid_op, grid_dur, grid_status, grid_frag;
  int grid_seg = 0;
  int grid_ready = 0;
  // launch-geometry options (msnap_set_option; the MSNAP_* environment variables of msnap.h only
  // seed them in msnap_create -- nothing on a launch path reads the environment)
  int no_twist = 0;             // "no_twist": keep small batches on the one-sided kernels (A/B timing)
  int gemm_stream_waves_per_cu = 0;   // "gemm_stream_waves_per_cu": wave target of the streaming GEMM's column slicing (0: default)
  int no_grid_sample = 0;      // "no_grid_sample": msnap_solve_grid_sample runs the two kernels (A/B timing)
  int no_twin = 0;             // "no_twin": batches keep solve_kernel_reg where solve_kernel_twin would run (A/B timing)
  int twin_max_drones = 0;      // "twin_max_drones": largest batch that takes solve_kernel_twin (0: default per order)
  int twist_max_drones = 0;     // "twist_max_drones": batches up to this size take the small-batch kernel (0: default)
  int solve_grid_waves = 0;     // "solve_grid_waves": cap on solve_kernel_reg's persistent grid (0: default)
  int gemm_grid_waves = 0;      // "gemm_grid_waves": cap on the shared-grid GEMM's persistent grid (0: default)
  void *mesh_tests = nullptr;   // "mesh_count_tests": device counter of the point-triangle tests evaluated (not culled)
  int collide_waves_per_cu = 0; // "collide_waves_per_cu": shares per CU of the pairwise pass (0: one column block per share)
  int collide_sample_parts = 0; // "collide_sample_parts": waves per share of the pairwise pass (0: chosen per launch)
  int collide_no_sym = 0;       // "collide_no_sym": 1 = the rows of msnap_formation_collide are NOT the slice of the columns at row_offset: one-sided evaluation
  int collide_no_cull = 0;      // "collide_no_cull": 1 = whole-swarm passes without the exact broad phase (A/B, dense swarms)
  int collide_cull_mode = 0;         // "collide_cull_mode": evaluator behind the broad phase: 0 chosen per pass, 1 surviving shares, 2 surviving group pairs
  int collide_cull_min_drones = 0;   // "collide_cull_min_drones": smallest whole swarm that takes the broad phase (0: default 3072)
  int collide_last_cull = 0;    // "collide_last_cull" (read): 1 if the last msnap_formation_collide took the broad-phase path
  int collide_last_shares = 0;  // "collide_last_shares" (read): 8-column x 128-row shares of that pass before the broad phase
  const int32_t *collide_meta = nullptr;   // device: its survivor counts ("collide_last_survivors" / "_group_pairs", read: synchronise)
  int collide_last_by_groups = 0;   // "collide_last_by_groups" (read): that pass's evaluator walked the surviving group pairs
  int collide_last_n = 0;           // its swarm size
  const void *blist_clean = nullptr;   // the group evaluator's reverse lists at this address (for blist_clean_n groups) are all-zero
  int blist_clean_n = 0;
  // one 64-bit word in page-locked host memory the broad phase's evaluator writes its survivor counts to: the next
  // pass's choice of evaluator reads it without synchronising (csrc/msnap_aux.hip::cull_hint_pack)
  unsigned long long *cull_hint = nullptr;
  // what msnap_sample_collide_device left for the pairwise pass, and where (the last few buffers it wrote): 1 the
  // transposed row image, 2 the per-drone boxes and sort keys of a whole-swarm pass behind the broad phase.  A buffer
  // the pass is handed that is not on record -- or whose record does not fit the pass -- is ignored, never misread.
  struct Handover {
    const void *ptr = nullptr;
    int n = 0, s = 0, form = 0;
  } handover[8];
  int handover_next = 0;
  int collide_last_sym = 0;     // "collide_last_sym" (read): 1 if the last msnap_formation_collide evaluated its own-range pairs once
  int mesh_waves_per_cu = 0;    // "mesh_waves_per_cu": wavefronts per CU the mesh sweep's grid is capped at (0: one workgroup per drone)
  int own_stream_priority = 0;  // "own_stream_priority": 0 default, 1 lowest, 2 highest (re-creates own_stream)
  msnap::RetiredBuf *retired = nullptr;   // blocks kept alive for graphs captured before they were outgrown
  char hip_err[256] = {0};
  char last_kernel[96] = {0};   // msnap_last_kernel: the solve kernel instance the last solve entry point launched
};

namespace msnap {

int record_hip_error(msnap_ctx *ctx, hipError_t e, const char *what);
int ensure(msnap_ctx *ctx, DevBuf &b, size_t bytes);
bool stream_is_capturing(const msnap_ctx *ctx);
// form (1 row image, 2 boxes and keys; 0: not on record for n drones x n_samples) of a sampler hand-over buffer
int handover_form(const msnap_ctx *ctx, const void *ptr, int n, int n_samples);
// what the last broad-phase pass evaluated (device-side choice of collide_eval_kernel, restated on its counts)
bool collide_counts_by_groups(const msnap_ctx *ctx, int n_drones, int shares_surviving, int group_pairs_surviving);

// records the kernel instance a solve launcher chose (msnap_last_kernel; bench.py labels its rooflines with it)
void note_kernel(msnap_ctx *ctx, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define MSNAP_HIP(ctx, call)                                         \
  do {                                                               \
    hipError_t e__ = (call);                                         \
    if (e__ != hipSuccess) return msnap::record_hip_error(ctx, e__, #call); \
  } while (0)

// kernel launchers (device pointers, asynchronous on ctx->stream)
int launch_solve(msnap_ctx *ctx, int n_drones, int n_seg, const double *wp, const double *t,
                 int shared_times, double *coef, double *dur, int32_t *status);
int launch_pack(msnap_ctx *ctx, int n_drones, int n_seg, const double *coef, const double *dur,
                float *out);
int launch_formation_transform(msnap_ctx *ctx, int n_poses, int n_offsets, const double *rb_pose,
                               const double *offsets, double *out);
int launch_sample(msnap_ctx *ctx, int n_drones, int n_seg, const double *coef, const double *dur,
                  double dt, int n_samples, int n_axes, double *pos, double *pos_t, bool keys_form);
int launch_eval_flat(msnap_ctx *ctx, int n_drones, int n_seg, const double *coef, const double *dur,
                     int n_samples, const double *ts, double *out);
int launch_snap_cost(msnap_ctx *ctx, int n_drones, int n_seg, const double *coef, const double *dur, double *cost);
int launch_formation_collide(msnap_ctx *ctx, int n_rows, int row_offset, int n_cols, int n_samples,
                             const double *pos_rows, const double *pos_cols, double radius,
                             double *min_dist, int32_t *partner, int32_t *hit, const double *rows_t);
bool formation_collide_takes_broad_phase(const msnap_ctx *ctx, int n_rows, int row_offset, int n_cols, int n_samples);
int launch_formation_collide_part(msnap_ctx *ctx, int n_drones, int n_samples, const double *pos_all, int part,
                                  int n_parts, double *out_d2, int32_t *out_j);
int launch_formation_collide_finish(msnap_ctx *ctx, int n_drones, int n_parts, const void *parts, size_t part_stride,
                                    int row_offset, int n_rows, double radius, double *min_dist, int32_t *partner,
                                    int32_t *hit);
int launch_mesh_sweep(msnap_ctx *ctx, int n_drones, int n_samples, const double *pos, int n_tris,
                      const double *tris, double radius, double *min_dist, int32_t *hit);
int launch_mesh_validity(msnap_ctx *ctx, int n_states, const double *states, int n_rtris, const double *rtris,
                         int n_etris, const double *etris, int32_t *valid);
int solve_kernel_setup(msnap_ctx *ctx);
// true when launch_solve would use the context's single global scratch slab (very long paths):
// such launches must not overlap each other
bool solve_uses_global_scratch(const msnap_ctx *ctx, int n_seg);
int launch_grid_prepare(msnap_ctx *ctx, int n_seg, const double *t, int t_on_device);
int launch_solve_grid(msnap_ctx *ctx, int n_drones, const double *wp, double *coef, double *dur,
                      int32_t *status);
bool grid_gemm_supported(const msnap_ctx *ctx, int n_seg);
// k-step pitch of the packed operator fragments ([column tile][pitch][64]); 0: no GEMM operator for this grid
int grid_frag_ks_pitch(const msnap_ctx *ctx, int n_seg);
int launch_grid_sample(msnap_ctx *ctx, int n_drones, const double *wp, double dt, int n_samples, double *coef,
                       double *dur, int32_t *status, double *pos, double *pos_t, bool keys_form);

}  // namespace msnap
