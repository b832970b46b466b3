/* msnap.h -- C-ABI of the MI355X-native minimum-snap trajectory hot path.
 *
 * The reference (mjmyt/drone_path_planning_python) has no FFI: its de-facto
 * boundary for this path is the Python call
 *     calculate_trajectory4D(waypoints)      src/optimizations/__init__.py:2,
 *                                            src/optimizations/calculatingTrajectories.py:200-213
 * invoked from scripts/drones_pols_generator.py:58.  Every entry point below
 * names the reference interface it replaces.  Plain pointers and sizes only;
 * no torch / numpy types cross this boundary.
 *
 * Conventions
 *   - all arrays row-major, caller-owned, fp64 unless noted;
 *   - entry points WITHOUT a suffix take HOST pointers and are synchronous (device
 *     staging owned by the context; the two solve entry points cut large batches
 *     into chunks that overlap upload, kernel and download -- at full PCIe rate when
 *     the host buffers come from msnap_host_alloc, correct with any host memory);
 *   - entry points ending in _device take DEVICE pointers, are asynchronous on
 *     the context's stream (msnap_sync / stream order to observe results);
 *   - order = polynomial degree, 7 (minimum snap, the reference) or 9
 *     (minimum crackle, BASELINE.json configs[4]; no reference exists);
 *     ncoef = order + 1; coefficients are in ASCENDING powers (c_k t^k), the
 *     reference's Polynomial.p layout (src/optimizations/uav_trajectory.py:17-22);
 *   - functions return 0 or a negative msnap_error; they never throw or abort.
 *     Per-drone problems are reported in status[] (the reference raises
 *     numpy.linalg.LinAlgError / AssertionError out of the ROS callback instead).
 *   - a context is not re-entrant: serialise calls per context (the Python
 *     wrapper holds a lock; rospy runs callback1/callback2 on separate threads,
 *     scripts/drones_pols_generator.py:102-103).  Contexts are independent.
 */
#ifndef MSNAP_H
#define MSNAP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct msnap_ctx msnap_ctx;

enum msnap_error {
  MSNAP_OK = 0,
  MSNAP_EINVAL = -1,      /* null pointer / negative size / bad flag          */
  MSNAP_EHIP = -2,        /* a HIP runtime call failed (msnap_last_hip_error) */
  MSNAP_EORDER = -3,      /* order is not 7 or 9                              */
  MSNAP_ESEGMENTS = -4,   /* n_seg < 1 or n_seg > max_segments of the context */
  MSNAP_ENOMEM = -5,      /* host or device allocation failed                 */
  MSNAP_ENODEVICE = -6,   /* no gfx950 device / device_id out of range        */
  MSNAP_ENOGRID = -7,     /* msnap_solve_grid without a prepared grid          */
  MSNAP_ECAPTURE = -8     /* a scratch buffer of the context would have to grow while its stream is being
                             captured into a graph (growing synchronises and frees), or a query that has to
                             synchronise the stream was made during a capture: run the same call once outside the
                             capture first -- the buffers then have their size */
};

/* Stream capture (hipGraph) and the context's scratch buffers.  A call captured into a graph records raw pointers
 * into the context's internal buffers.  The library therefore remembers every buffer a capture has used: when a
 * LATER call (eager, or another capture) needs such a buffer larger, the old block is not freed -- it is retired and
 * stays valid until msnap_destroy or msnap_release_graph_buffers -- so a graph instantiated earlier keeps replaying
 * on memory that is still its own and still gives the result of the pass it captured.  A replay does not see options
 * set after the capture; a graph that captured msnap_solve_grid_device is tied to the grid prepared at that time --
 * preparing another grid on the context invalidates it (its memory stays valid, its results do not).  msnap_release_graph_buffers frees the retired
 * blocks: call it once every graph that captured calls of this context has been destroyed; it returns the number of
 * bytes released through *bytes (may be NULL). */
int msnap_release_graph_buffers(msnap_ctx *ctx, size_t *bytes);

enum msnap_status {       /* per-drone, written to status[]                   */
  MSNAP_ST_OK = 0,
  MSNAP_ST_SINGULAR = 1,  /* a pivot of the block LDL^T was <= 0 or not finite */
  MSNAP_ST_TIMES = 2,     /* times not strictly increasing (or t[1] <= 2 t[0]) */
  MSNAP_ST_NONFINITE = 3  /* NaN / Inf in the waypoints or times               */
};

int msnap_version(void);                       /* 10000*major + 100*minor + patch */
const char *msnap_strerror(int code);
const char *msnap_last_hip_error(const msnap_ctx *ctx);
/* Name (as a profiler prints it, e.g. "msnap::solve_kernel_twin<5, 10>") of the kernel instance the most recent
 * solve entry point of this context launched -- msnap_solve_batch[_device] or msnap_solve_grid[_device];
 * "" before the first one.  The choice depends on order, segment count, batch size and the options above. */
const char *msnap_last_kernel(const msnap_ctx *ctx);

/* One HIP stream + pinned/device scratch per context. */
int msnap_create(msnap_ctx **out, int device_id, int order, int max_segments);
void msnap_destroy(msnap_ctx *ctx);
/* Borrow an external hipStream_t (e.g. torch's current stream; NULL is the HIP
 * null stream).  msnap_use_own_stream goes back to the context's own stream. */
int msnap_set_stream(msnap_ctx *ctx, void *hip_stream);
int msnap_use_own_stream(msnap_ctx *ctx);
void *msnap_get_stream(msnap_ctx *ctx);
int msnap_sync(msnap_ctx *ctx);
/* Page-locked host memory for the host-pointer entry points (the reference's arrays
 * are ordinary NumPy allocations, src/optimizations/calculatingTrajectories.py:137-144;
 * pinned ones let the copy engines read and write them directly). */
int msnap_host_alloc(void **ptr, size_t bytes);
int msnap_host_free(void *ptr);
/* Options of a context (launch geometry: tests and tuning tools, every default is chosen per
 * launch from the device's CU count; stream priority: callers that overlap two contexts).  Unknown
 * names return MSNAP_EINVAL.
 *   "solve_grid_waves"     cap on the persistent grid of the large-batch solve kernel (0 = default);
 *                          a small cap makes every wave walk several tiles (the regime of a
 *                          saturating batch) on a batch the oracle checks in seconds
 *   "gemm_grid_waves"      the same for the shared-grid GEMM
 *   "gemm_stream_waves_per_cu"  wavefronts per CU up to which the streaming GEMM (16 and more segments) slices its
 *                          column tiles over more waves (0 = default: 16)
 *   "no_grid_sample"       1: msnap_solve_grid_sample_device runs the two kernels where it would fuse (A/B timing)
 *   "twist_max_drones"     largest batch that takes the small-batch two-sided kernel (0 = default)
 *   "no_twist"             1: small batches stay on the one-sided kernels
 *   "twin_max_drones"      largest batch that takes the two-sided column-split throughput kernel (0 = default:
 *                          order 7 up to 128 drones per CU, order 9 any size)
 *   "no_twin"             1: order-9 batches stay on the one-sided throughput kernel where the two-sided
 *                          column-split one would run (A/B timing)
 *   "collide_waves_per_cu" shares per CU of the pairwise pass (0 = one 8-column x 128-row block per share)
 *   "collide_sample_parts" waves per share of the pairwise pass, each a range of the sample chunks
 *                          (0 = chosen per launch: more than one only when the launch is small)
 *   "collide_no_sym"       1: the rows handed to msnap_formation_collide are not the slice of its columns at
 *                          row_offset -- every pair is evaluated one-sidedly (read-only companion
 *                          "collide_last_sym": 1 if the last pass evaluated its own-range pairs once)
 *   "collide_cull_min_drones"  smallest whole swarm that takes the exact broad phase (0 = default 3072; at least
 *                          256: tests lower it to check the path against the oracle on small swarms)
 *   "collide_no_cull"      1: whole-swarm passes (row_offset 0, n_rows == n_cols, 3072..16384 drones) skip the exact
 *                          broad phase -- spatial sort, per-drone bounds, box test per 8-column share -- and
 *                          evaluate every pair; results are identical either way (a dense swarm, where nothing
 *                          can be culled, saves the sort: about a sixth of the pass at 4096 drones).  Read-only
 *                          companions: "collide_last_cull" (1 if the last pass took the broad phase),
 *                          "collide_last_shares" (its 128 x 8 shares before the test), "collide_last_survivors"
 *                          (the shares that pass it) and "collide_last_group_pairs" (the 8 x 8 group pairs that
 *                          pass it; both synchronise the stream).  msnap_set_option refuses the read-only names
 *                          ("collide_last_*") with MSNAP_EINVAL
 *   "collide_cull_mode"    what the broad phase evaluates: 1 the surviving 128 x 8 shares (then the merge of their
 *                          entries), 2 the surviving 8 x 8 group pairs (then the per-group fold of their candidates;
 *                          swarms up to 8192 drones: every group pair has a list slot), 0 (default) chosen per pass
 *                          on the HOST -- the two launch sequences and their buffers differ -- from the survivor
 *                          counts the context's previous pass over a swarm of this size left in page-locked memory
 *                          (read without synchronising; a pass without such counts takes the shares).  Results are
 *                          identical either way
 *   "mesh_waves_per_cu"    wavefronts per CU msnap_mesh_sweep's grid is capped at (0 = one workgroup per drone: a large
 *                          sweep then holds every wave slot of the chip for its whole run).  A sweep on a second
 *                          stream beside other kernels leaves them room with 8..12 (it runs longer itself:
 *                          4096 drones x 96 samples x 68 triangles 33 -> 40..45 us, the pipeline around it 106 -> 102)
 *   "mesh_count_tests"     1: count the point-triangle tests msnap_mesh_sweep evaluates (the ones
 *                          its bounding-box cull does not skip); msnap_get_option returns the count
 *                          since the option was last set (and synchronises the stream); 0: off
 *   "pipe_chunk_mb"        output megabytes per chunk of the chunked host-pointer solves
 *   "own_stream_priority"  0 default, 1 the lowest, 2 the highest priority the device offers: re-creates
 *                          the context's own stream (after draining it).  A pass that should only fill the
 *                          gaps of another context's work -- the mesh sweep beside the pairwise pass --
 *                          runs on a lowest-priority stream: its workgroups are dispatched when the
 *                          other queue has none waiting.  Setting it destroys and re-creates the stream:
 *                          set it BEFORE msnap_get_stream() hands the handle to anybody (a wrapper around
 *                          the old handle -- e.g. torch.cuda.ExternalStream -- would dangle)
 * msnap_create seeds them once from the environment variables MSNAP_SOLVE_GRID_WAVES,
 * MSNAP_GEMM_GRID_WAVES, MSNAP_GEMM_STREAM_WAVES_PER_CU, MSNAP_NO_GRID_SAMPLE, MSNAP_MESH_WAVES_PER_CU, MSNAP_TWIST_MAX_DRONES, MSNAP_NO_TWIST, MSNAP_NO_TWIN, MSNAP_TWIN_MAX_DRONES, MSNAP_COLLIDE_WAVES_PER_CU,
 * MSNAP_COLLIDE_SAMPLE_PARTS, MSNAP_COLLIDE_NO_CULL, MSNAP_COLLIDE_CULL_MIN_DRONES,
 * MSNAP_COLLIDE_CULL_MODE and MSNAP_PIPE_CHUNK_MB; nothing on a launch path reads the environment. */
int msnap_set_option(msnap_ctx *ctx, const char *name, long value);
int msnap_get_option(const msnap_ctx *ctx, const char *name, long *value);
/* hipEvent timing on the context's stream (bench.py roofline leg). */
int msnap_timer_start(msnap_ctx *ctx);
int msnap_timer_stop(msnap_ctx *ctx, float *elapsed_ms);   /* synchronises */

/* ---- a1/a2: calculate_trajectory1D / calculate_trajectory4D, batched ----------
 * replaces src/optimizations/calculatingTrajectories.py:37-197 (per axis) and
 * :200-213 (4 axes) for n_drones independent trajectories.
 *   wp     [n_drones][n_seg+1][4]   x, y, z, yaw per waypoint
 *   t      [n_drones][n_seg+1] absolute times, or [n_seg+1] if shared_times != 0
 *   coef   [n_drones][n_seg][4][ncoef]  out, ascending powers
 *   dur    [n_drones][n_seg]            out, T_i = t[i+1]-t[i]  (time_points, :59-61)
 *   status [n_drones]                   out, msnap_status
 * Drones with status != 0 get NaN coefficients.
 */
int msnap_solve_batch(msnap_ctx *ctx, int n_drones, int n_seg, const double *wp,
                      const double *t, int shared_times, double *coef, double *dur,
                      int32_t *status);
int msnap_solve_batch_device(msnap_ctx *ctx, int n_drones, int n_seg, const double *wp,
                             const double *t, int shared_times, double *coef,
                             double *dur, int32_t *status);

/* ---- a1/a2 on a SHARED time grid: one fp64 MFMA GEMM -----------------------------
 * The reference's own usage: path_to_pol gives every drone the uniform grid
 * t_i = i*10/n (scripts/drones_pols_generator.py:44-46,56), so the matrix of
 * calculate_trajectory1D (src/optimizations/calculatingTrajectories.py:48-131) is
 * shared and the coefficients are linear in the waypoints.
 *   msnap_grid_prepare  builds the (n_seg+1) x (n_seg*ncoef) operator for t on
 *                       the GPU (the solve kernel on unit waypoint vectors);
 *   msnap_solve_grid    applies it to n_drones waypoint sets: same outputs as
 *                       msnap_solve_batch(.., t, shared_times = 1, ..).
 * The operator stays valid until the next msnap_grid_prepare on this context.
 * n_seg of msnap_solve_grid[_device] is the segment count the CALLER sized wp / coef / dur for (as in
 * msnap_solve_batch): it must equal the prepared grid's, else MSNAP_ESEGMENTS and nothing is written -- the reference
 * call sizes its output from its input (calculatingTrajectories.py:45-49), a C caller's buffers cannot be inspected.
 * msnap_grid_segments: segments of the grid this context holds (0: none prepared, negative: error).
 */
int msnap_grid_prepare(msnap_ctx *ctx, int n_seg, const double *t /* host [n_seg+1] */);
int msnap_grid_prepare_device(msnap_ctx *ctx, int n_seg, const double *t /* device */);
int msnap_grid_segments(const msnap_ctx *ctx);
int msnap_solve_grid(msnap_ctx *ctx, int n_drones, int n_seg, const double *wp, double *coef, double *dur,
                     int32_t *status);
int msnap_solve_grid_device(msnap_ctx *ctx, int n_drones, int n_seg, const double *wp, double *coef,
                            double *dur, int32_t *status);

/* ---- a7: the float32 [T | x | y | z | yaw] matrix of path_to_pol -----------------
 * replaces scripts/drones_pols_generator.py:63-77.
 *   out [n_drones][n_seg][1 + 4*ncoef] float32
 */
int msnap_pack_pol_matrix(msnap_ctx *ctx, int n_drones, int n_seg, const double *coef,
                          const double *dur, float *out);
int msnap_pack_pol_matrix_device(msnap_ctx *ctx, int n_drones, int n_seg,
                                 const double *coef, const double *dur, float *out);

/* ---- a8: transform(path) of the formation node, K offsets ------------------------
 * replaces scripts/drones_traj_generator.py:56-89 (K = 2 hard-coded there).
 *   rb_pose [n_poses][7]   x y z qx qy qz qw of the rigid body
 *   offsets [n_offsets][3] body-frame drone positions (identity orientation)
 *   out     [n_offsets][n_poses][7]   p' = R(q) p_k + t,  q' = quaternion of R(q) (KDL GetQuaternion)
 */
int msnap_formation_transform(msnap_ctx *ctx, int n_poses, int n_offsets,
                              const double *rb_pose, const double *offsets, double *out);
int msnap_formation_transform_device(msnap_ctx *ctx, int n_poses, int n_offsets,
                                     const double *rb_pose, const double *offsets,
                                     double *out);

/* ---- a5: PiecewisePolynomial.eval on a uniform grid -------------------------------
 * replaces src/optimizations/uav_trajectory.py:154-169 (strict '<' lookup, the
 * last piece extrapolates) sampled as np.arange(0, .., dt) (scripts/path_vis.py:28,
 * src/trajectory_visualising/visualization.py:53).
 *   pos [n_drones][n_samples][n_axes]  (n_axes = 3: x,y,z; 4: + yaw), sample s at t = s*dt
 */
int msnap_sample(msnap_ctx *ctx, int n_drones, int n_seg, const double *coef,
                 const double *dur, double dt, int n_samples, int n_axes, double *pos);
int msnap_sample_device(msnap_ctx *ctx, int n_drones, int n_seg, const double *coef,
                        const double *dur, double dt, int n_samples, int n_axes,
                        double *pos);

/* ---- Trajectory.eval / Polynomial4D.eval: differential-flatness outputs -------------
 * replaces src/optimizations/uav_trajectory.py:64-85 (pos, vel, acc, omega, yaw from the
 * x,y,z,yaw polynomials) with the piece lookup of Trajectory.eval, :119-127 ('<=').
 *   ts  [n_samples] sample times shared by all drones
 *   out [n_drones][n_samples][13] = pos[3] vel[3] acc[3] omega[3] yaw ; NaN outside [0, duration]
 */
int msnap_eval_flat(msnap_ctx *ctx, int n_drones, int n_seg, const double *coef, const double *dur,
                    int n_samples, const double *ts, double *out);
int msnap_eval_flat_device(msnap_ctx *ctx, int n_drones, int n_seg, const double *coef,
                           const double *dur, int n_samples, const double *ts, double *out);

/* ---- snap cost J = sum_seg int (p^(k))^2 dt per drone and axis (k = 4 at order 7) ------
 * The objective whose KKT system the reference's collocation rows encode
 * (src/optimizations/calculatingTrajectories.py:13-33 states the conditions; the
 * reference never evaluates J).   cost [n_drones][4]
 */
int msnap_snap_cost(msnap_ctx *ctx, int n_drones, int n_seg, const double *coef, const double *dur,
                    double *cost);
int msnap_snap_cost_device(msnap_ctx *ctx, int n_drones, int n_seg, const double *coef,
                           const double *dur, double *cost);

/* ---- drone-vs-drone formation pass (new capability; no reference) -------
 * rows: the n_rows drones this caller owns (a shard), starting at global index
 * row_offset; cols: all n_cols drones (after the all-gather).  Spheres of `radius`.
 *   pos_rows [n_rows][n_samples][3], pos_cols [n_cols][n_samples][3]
 *   min_dist [n_rows]  min over other drones j != global row and samples of |p_i-p_j|, the squared
 *                      distance taken as fma(dz, dz, fma(dy, dy, dx*dx)) (differences rounded once,
 *                      two fused multiply-adds: one rounding less than the plain sum of squares
 *                      and 7 instead of 9 vector operations per pair and sample)
 *   partner  [n_rows]  lowest global j attaining it (-1 if none)
 *   hit      [n_rows]  min_dist < 2*radius
 * pos_rows must be the rows [row_offset, row_offset + n_rows) of pos_cols (the same samples):
 * pairs inside that range are evaluated once and credited to both drones.  The host-pointer entry
 * compares the two arrays and falls back to the one-sided evaluation when they differ; a device-pointer
 * caller whose rows are some other set of drones sets the option "collide_no_sym" first.
 * n_cols == 0 gives (+inf, -1, 0).
 * Non-finite samples never win a minimum (IEEE minNum): a drone whose samples are NaN -- the
 * output of a solve with status != 0 -- reports (+inf, -1, 0) and is invisible to the other
 * drones.  Check status[] of the solve before trusting a "no hit" (the Python pipeline,
 * swarm.formation_pass, refuses such drones).  The same holds for msnap_mesh_sweep.
 */
int msnap_formation_collide(msnap_ctx *ctx, int n_rows, int row_offset, int n_cols,
                            int n_samples, const double *pos_rows, const double *pos_cols,
                            double radius, double *min_dist, int32_t *partner,
                            int32_t *hit);
int msnap_formation_collide_device(msnap_ctx *ctx, int n_rows, int row_offset, int n_cols,
                                   int n_samples, const double *pos_rows,
                                   const double *pos_cols, double radius,
                                   double *min_dist, int32_t *partner, int32_t *hit);

/* The same pass for a caller whose positions come from this library's sampler: msnap_sample_collide_device writes,
 * next to the positions, what the pass over these drones would otherwise compute in a launch of its own -- into
 * pos_rows_t, msnap_collide_rows_t_doubles(n_rows, n_samples) doubles, whose content is the pair's private matter:
 *   - the rows' TRANSPOSED image [n_samples][3][P], P = n_rows rounded up to whole 128-row blocks (the pass reads its
 *     rows from it, 512 contiguous bytes per wave and load, and skips its own transposition pass), or
 *   - where the pass over the n_rows drones as a whole swarm runs behind the exact broad phase: every drone's path
 *     box and sort key (the sampler has the samples in LDS; the pass then starts with its sort, without a key launch).
 * The context remembers which of the two it last wrote and where; a buffer that is not that hand-over (or no longer
 * fits the options in force) is ignored, never misread.  Device pointers only: the pair exists to keep the formation
 * pipeline (sampler -> pairwise pass) on the GPU without the intermediate launch. */
size_t msnap_collide_rows_t_doubles(int n_rows, int n_samples);
/* 1 if msnap_formation_collide_t_device with these arguments reads the sampler's hand-over, 0 if not (paths shorter
 * than 6 samples take plain loops: the caller can then sample with msnap_sample and save the second output). */
int msnap_formation_collide_reads_rows_t(const msnap_ctx *ctx, int n_rows, int row_offset, int n_cols, int n_samples);
int msnap_sample_collide_device(msnap_ctx *ctx, int n_drones, int n_seg, const double *coef,
                                const double *dur, double dt, int n_samples, double *pos, double *pos_t);
/* The shared-grid solve and the sampler as ONE launch, the reference's drones_pols_generator -> drones_traj_generator
 * step for a swarm on a common grid (scripts/drones_pols_generator.py:44-77 -> scripts/drones_traj_generator.py:28-46):
 * the same outputs, bit for bit, as msnap_solve_grid_device followed by msnap_sample_collide_device (pos_t != NULL) or
 * msnap_sample_device with 3 axes (pos_t == NULL) -- the coefficients stay in LDS between the fp64 MFMA product and
 * the Horner loops, which saves the dependent launch and the read-back.  Shapes outside the fused kernel's range
 * (more than 11 segments, where it stops paying; samples beyond its LDS image) run as those two launches; the option "no_grid_sample"
 * forces that (A/B timing).  n_samples == 0: the solve alone. */
int msnap_solve_grid_sample_device(msnap_ctx *ctx, int n_drones, int n_seg, const double *wp, double dt, int n_samples,
                                   double *coef, double *dur, int32_t *status, double *pos /* [n_drones][n_samples][3] */,
                                   double *pos_t /* NULL or msnap_collide_rows_t_doubles(n_drones, n_samples) doubles */);
int msnap_formation_collide_t_device(msnap_ctx *ctx, int n_rows, int row_offset, int n_cols,
                                     int n_samples, const double *pos_rows_t, const double *pos_rows,
                                     const double *pos_cols, double radius, double *min_dist,
                                     int32_t *partner, int32_t *hit);

/* ---- the same pass split over the ranks of a job: every unordered pair on exactly ONE rank --------
 * BASELINE.json north_star: "the drone batch shards across the 8 GPUs ... with an RCCL all-gather for the
 * inter-drone formation collision pass".  After the all-gather every rank holds pos_all [n_drones][n_samples][3].
 * The swarm's unordered pairs form one triangular line of (128-row block, column) units -- the line a single
 * msnap_formation_collide launch walks; part `part` of `n_parts` evaluates the part-th of n_parts equal contiguous
 * ranges of it, each pair once, and credits both drones.  part_out (msnap_formation_part_bytes(n_drones) bytes,
 * 8-byte aligned) receives, for EVERY drone of the swarm, the squared minimum over the pairs this part met
 * (double [n_drones]) followed by the partner (int32 [n_drones]; +inf / -1 where it met none).  The ranks exchange
 * their part_out blocks (one more all-gather of n_parts x 12 B x n_drones) and msnap_formation_collide_finish folds
 * them for the rows [row_offset, row_offset + n_rows) a rank owns: minimum over the parts (lowest partner wins a
 * tie), distance, hit -- bit for bit what one msnap_formation_collide over the whole swarm returns.
 *   parts [n_parts] blocks of msnap_formation_part_bytes(n_drones) bytes, part p at offset p * that
 */
size_t msnap_formation_part_bytes(int n_drones);
/* Who evaluates which pairs on several ranks is a choice between the parts above and "every rank runs the pass over
 * the whole gathered swarm behind the exact broad phase and keeps its rows" (one collective instead of two; pays when
 * the broad phase leaves few pairs).  Both inputs of that choice are the library's own and are exported here, so that
 * a host does not re-type launch thresholds or cost-model constants:
 *   msnap_formation_collide_takes_broad_phase  1 if msnap_formation_collide[_device] with these arguments would run
 *       behind the broad phase (size limits, "collide_no_cull", "collide_no_sym", "collide_cull_min_drones"), else 0;
 *   msnap_formation_whole_pass_pays  after a whole-swarm pass of n_drones on this context: *pays = 1 if, by the
 *       counts that pass left (pairs it evaluated: 8 x 8 per surviving group pair or 128 x 8 per surviving share,
 *       whichever list its evaluator walked) and the evaluator's cost model, the whole pass on every one of n_ranks
 *       ranks is quicker than a rank's 1 / n_ranks of all pairs plus the second collective; 0 if not, or if the last
 *       pass did not take the broad phase.  Synchronises the stream (MSNAP_ECAPTURE during a capture).
 * Read-only options of the same pass: "collide_last_by_groups" (1: its evaluator walked the group pairs) and
 * "collide_last_pairs_evaluated" (drone pairs it evaluated); both synchronise. */
int msnap_formation_collide_takes_broad_phase(const msnap_ctx *ctx, int n_rows, int row_offset, int n_cols,
                                              int n_samples);
int msnap_formation_whole_pass_pays(msnap_ctx *ctx, int n_drones, int n_ranks, int *pays);
int msnap_formation_collide_part(msnap_ctx *ctx, int n_drones, int n_samples, const double *pos_all,
                                 int part, int n_parts, void *part_out);
int msnap_formation_collide_part_device(msnap_ctx *ctx, int n_drones, int n_samples,
                                        const double *pos_all, int part, int n_parts, void *part_out);
int msnap_formation_collide_finish(msnap_ctx *ctx, int n_drones, int n_parts, const void *parts,
                                   int row_offset, int n_rows, double radius, double *min_dist,
                                   int32_t *partner, int32_t *hit);
int msnap_formation_collide_finish_device(msnap_ctx *ctx, int n_drones, int n_parts, const void *parts,
                                          int row_offset, int n_rows, double radius, double *min_dist,
                                          int32_t *partner, int32_t *hit);

/* ---- drone-vs-mesh sweep against resources/stl obstacles (new capability) ----------
 *   tris [n_tris][3][3] fp64 vertices (binary STL float32 widened by the caller)
 *   min_dist [n_drones] min over samples and triangles of the point-triangle distance
 *   hit      [n_drones] min_dist < radius
 */
int msnap_mesh_sweep(msnap_ctx *ctx, int n_drones, int n_samples, const double *pos,
                     int n_tris, const double *tris, double radius, double *min_dist,
                     int32_t *hit);
int msnap_mesh_sweep_device(msnap_ctx *ctx, int n_drones, int n_samples, const double *pos,
                            int n_tris, const double *tris, double radius,
                            double *min_dist, int32_t *hit);

/* ---- rigid-body state validity, batched (the planner's OMPL validity callback) -------
 * replaces isStateValid, src/RigidBodyPlanners/RB_planning_sep_coll_check.py:208-226
 * (robot mesh at (x,y,z) with quaternion_from_euler(0,0,yaw), fcl.collide against the
 * environment mesh, src/RigidBodyPlanners/fcl_checker.py:93-100) for many states at once.
 *   states [n_states][4]  x, y, z, yaw
 *   rtris  [n_rtris][3][3] robot mesh (body frame), etris [n_etris][3][3] environment mesh
 *   valid  [n_states]  1 = no robot triangle intersects an environment triangle
 */
int msnap_mesh_validity(msnap_ctx *ctx, int n_states, const double *states, int n_rtris,
                        const double *rtris, int n_etris, const double *etris, int32_t *valid);
int msnap_mesh_validity_device(msnap_ctx *ctx, int n_states, const double *states, int n_rtris,
                               const double *rtris, int n_etris, const double *etris,
                               int32_t *valid);

#ifdef __cplusplus
}
#endif
#endif /* MSNAP_H */
