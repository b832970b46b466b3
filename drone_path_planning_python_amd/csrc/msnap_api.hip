// C-ABI of libmsnap.so (include/msnap.h): context management, host-pointer
// wrappers, stream / timer plumbing.  All compute happens in the HIP kernels
// of msnap_solve.hip / msnap_aux.hip; there is no CPU fallback.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "msnap_internal.h"

namespace msnap {

int record_hip_error(msnap_ctx *ctx, hipError_t e, const char *what) {
  if (ctx) snprintf(ctx->hip_err, sizeof(ctx->hip_err), "%s: %s", what, hipGetErrorString(e));
  return MSNAP_EHIP;
}

int ensure(msnap_ctx *ctx, DevBuf &b, size_t bytes) {
  if (bytes <= b.cap) return MSNAP_OK;
  if (b.p) {
    MSNAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    MSNAP_HIP(ctx, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
  }
  size_t cap = bytes + bytes / 4 + 256;
  hipError_t e = hipMalloc(&b.p, cap);
  if (e != hipSuccess) {
    b.p = nullptr;
    record_hip_error(ctx, e, "hipMalloc");
    return MSNAP_ENOMEM;
  }
  b.cap = cap;
  return MSNAP_OK;
}

static int check_seg(const msnap_ctx *ctx, int n_seg) {
  if (n_seg < 1 || n_seg > ctx->max_segments) return MSNAP_ESEGMENTS;
  return MSNAP_OK;
}

}  // namespace msnap

using namespace msnap;

extern "C" {

int msnap_version(void) { return MSNAP_VERSION_NUM; }

const char *msnap_strerror(int code) {
  switch (code) {
    case MSNAP_OK: return "ok";
    case MSNAP_EINVAL: return "invalid argument";
    case MSNAP_EHIP: return "HIP runtime error (see msnap_last_hip_error)";
    case MSNAP_EORDER: return "unsupported polynomial order (7 or 9)";
    case MSNAP_ESEGMENTS: return "segment count out of range for this context";
    case MSNAP_ENOMEM: return "out of memory";
    case MSNAP_ENODEVICE: return "no usable gfx950 device";
    case MSNAP_ENOGRID: return "no time grid prepared on this context (msnap_grid_prepare)";
    default: return "unknown msnap error";
  }
}

const char *msnap_last_hip_error(const msnap_ctx *ctx) { return ctx ? ctx->hip_err : ""; }

int msnap_create(msnap_ctx **out, int device_id, int order, int max_segments) {
  if (!out) return MSNAP_EINVAL;
  *out = nullptr;
  if (order != 7 && order != 9) return MSNAP_EORDER;
  if (max_segments < 1) return MSNAP_ESEGMENTS;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return MSNAP_ENODEVICE;
  if (device_id < 0 || device_id >= ndev) return MSNAP_ENODEVICE;
  msnap_ctx *ctx = new (std::nothrow) msnap_ctx();
  if (!ctx) return MSNAP_ENOMEM;
  ctx->device = device_id;
  ctx->order = order;
  ctx->khalf = (order + 1) / 2;
  ctx->max_segments = max_segments;
  {
    const char *e = getenv("MSNAP_NO_TWIST");
    ctx->no_twist = (e && e[0] == '1') ? 1 : 0;
  }
  int rc = MSNAP_OK;
  do {
    if (hipSetDevice(device_id) != hipSuccess) { rc = MSNAP_ENODEVICE; break; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) ctx->n_cu = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) { rc = MSNAP_EHIP; break; }
    ctx->stream = ctx->own_stream;
    if (hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) { rc = MSNAP_EHIP; break; }
    rc = solve_kernel_setup(ctx);
  } while (0);
  if (rc != MSNAP_OK) {
    msnap_destroy(ctx);
    return rc;
  }
  *out = ctx;
  return MSNAP_OK;
}

void msnap_destroy(msnap_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
  if (ctx->scratch.p) (void)hipFree(ctx->scratch.p);
  for (msnap::DevBuf *b : {&ctx->grid_t, &ctx->grid_wp, &ctx->grid_op, &ctx->grid_dur, &ctx->grid_status, &ctx->grid_frag})
    if (b->p) (void)hipFree(b->p);
  for (auto &b : ctx->stage)
    if (b.p) (void)hipFree(b.p);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
}

int msnap_set_stream(msnap_ctx *ctx, void *hip_stream) {
  if (!ctx) return MSNAP_EINVAL;
  ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
  return MSNAP_OK;
}

int msnap_use_own_stream(msnap_ctx *ctx) {
  if (!ctx) return MSNAP_EINVAL;
  ctx->stream = ctx->own_stream;
  return MSNAP_OK;
}

void *msnap_get_stream(msnap_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int msnap_sync(msnap_ctx *ctx) {
  if (!ctx) return MSNAP_EINVAL;
  MSNAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MSNAP_OK;
}

int msnap_timer_start(msnap_ctx *ctx) {
  if (!ctx) return MSNAP_EINVAL;
  MSNAP_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  return MSNAP_OK;
}

int msnap_timer_stop(msnap_ctx *ctx, float *elapsed_ms) {
  if (!ctx || !elapsed_ms) return MSNAP_EINVAL;
  MSNAP_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  MSNAP_HIP(ctx, hipEventSynchronize(ctx->ev1));
  MSNAP_HIP(ctx, hipEventElapsedTime(elapsed_ms, ctx->ev0, ctx->ev1));
  return MSNAP_OK;
}

// ------------------------------------------------------------------ solve
int msnap_solve_batch_device(msnap_ctx *ctx, int n_drones, int n_seg, const double *wp,
                             const double *t, int shared_times, double *coef, double *dur,
                             int32_t *status) {
  if (!ctx || n_drones < 0) return MSNAP_EINVAL;
  int rc = check_seg(ctx, n_seg);
  if (rc) return rc;
  if (n_drones == 0) return MSNAP_OK;
  if (!wp || !t || !coef || !dur || !status) return MSNAP_EINVAL;
  MSNAP_HIP(ctx, hipSetDevice(ctx->device));
  return launch_solve(ctx, n_drones, n_seg, wp, t, shared_times ? 1 : 0, coef, dur, status);
}

int msnap_solve_batch(msnap_ctx *ctx, int n_drones, int n_seg, const double *wp, const double *t,
                      int shared_times, double *coef, double *dur, int32_t *status) {
  if (!ctx || n_drones < 0) return MSNAP_EINVAL;
  int rc = check_seg(ctx, n_seg);
  if (rc) return rc;
  if (n_drones == 0) return MSNAP_OK;
  if (!wp || !t || !coef || !dur || !status) return MSNAP_EINVAL;
  MSNAP_HIP(ctx, hipSetDevice(ctx->device));
  const size_t N = n_drones, m = (size_t)n_seg + 1, nc = ctx->order + 1;
  const size_t b_wp = N * m * 4 * 8, b_t = (shared_times ? 1 : N) * m * 8;
  const size_t b_coef = N * n_seg * 4 * nc * 8, b_dur = N * n_seg * 8, b_st = N * 4;
  if ((rc = ensure(ctx, ctx->stage[0], b_wp))) return rc;
  if ((rc = ensure(ctx, ctx->stage[1], b_t))) return rc;
  if ((rc = ensure(ctx, ctx->stage[2], b_coef))) return rc;
  if ((rc = ensure(ctx, ctx->stage[3], b_dur))) return rc;
  if ((rc = ensure(ctx, ctx->stage[4], b_st))) return rc;
  MSNAP_HIP(ctx, hipMemcpyAsync(ctx->stage[0].p, wp, b_wp, hipMemcpyHostToDevice, ctx->stream));
  MSNAP_HIP(ctx, hipMemcpyAsync(ctx->stage[1].p, t, b_t, hipMemcpyHostToDevice, ctx->stream));
  rc = launch_solve(ctx, n_drones, n_seg, (const double *)ctx->stage[0].p, (const double *)ctx->stage[1].p,
                    shared_times ? 1 : 0, (double *)ctx->stage[2].p, (double *)ctx->stage[3].p,
                    (int32_t *)ctx->stage[4].p);
  if (rc) return rc;
  MSNAP_HIP(ctx, hipMemcpyAsync(coef, ctx->stage[2].p, b_coef, hipMemcpyDeviceToHost, ctx->stream));
  MSNAP_HIP(ctx, hipMemcpyAsync(dur, ctx->stage[3].p, b_dur, hipMemcpyDeviceToHost, ctx->stream));
  MSNAP_HIP(ctx, hipMemcpyAsync(status, ctx->stage[4].p, b_st, hipMemcpyDeviceToHost, ctx->stream));
  MSNAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MSNAP_OK;
}

// ------------------------------------------------------------------ shared grid (K2)
int msnap_grid_prepare_device(msnap_ctx *ctx, int n_seg, const double *t) {
  if (!ctx || !t) return MSNAP_EINVAL;
  int rc = check_seg(ctx, n_seg);
  if (rc) return rc;
  MSNAP_HIP(ctx, hipSetDevice(ctx->device));
  return launch_grid_prepare(ctx, n_seg, t, 1);
}

int msnap_grid_prepare(msnap_ctx *ctx, int n_seg, const double *t) {
  if (!ctx || !t) return MSNAP_EINVAL;
  int rc = check_seg(ctx, n_seg);
  if (rc) return rc;
  MSNAP_HIP(ctx, hipSetDevice(ctx->device));
  rc = launch_grid_prepare(ctx, n_seg, t, 0);
  if (rc) return rc;
  MSNAP_HIP(ctx, hipStreamSynchronize(ctx->stream));   // t is pageable host memory
  return MSNAP_OK;
}

int msnap_solve_grid_device(msnap_ctx *ctx, int n_drones, const double *wp, double *coef, double *dur,
                            int32_t *status) {
  if (!ctx || n_drones < 0) return MSNAP_EINVAL;
  if (!ctx->grid_ready) return MSNAP_ENOGRID;
  if (n_drones == 0) return MSNAP_OK;
  if (!wp || !coef || !dur || !status) return MSNAP_EINVAL;
  MSNAP_HIP(ctx, hipSetDevice(ctx->device));
  return launch_solve_grid(ctx, n_drones, wp, coef, dur, status);
}

int msnap_solve_grid(msnap_ctx *ctx, int n_drones, const double *wp, double *coef, double *dur,
                     int32_t *status) {
  if (!ctx || n_drones < 0) return MSNAP_EINVAL;
  if (!ctx->grid_ready) return MSNAP_ENOGRID;
  if (n_drones == 0) return MSNAP_OK;
  if (!wp || !coef || !dur || !status) return MSNAP_EINVAL;
  MSNAP_HIP(ctx, hipSetDevice(ctx->device));
  const int n_seg = ctx->grid_seg;
  const size_t N = n_drones, m = (size_t)n_seg + 1, nc = ctx->order + 1;
  const size_t b_wp = N * m * 4 * 8;
  const size_t b_coef = N * n_seg * 4 * nc * 8, b_dur = N * n_seg * 8, b_st = N * 4;
  int rc;
  if ((rc = ensure(ctx, ctx->stage[0], b_wp))) return rc;
  if ((rc = ensure(ctx, ctx->stage[2], b_coef))) return rc;
  if ((rc = ensure(ctx, ctx->stage[3], b_dur))) return rc;
  if ((rc = ensure(ctx, ctx->stage[4], b_st))) return rc;
  MSNAP_HIP(ctx, hipMemcpyAsync(ctx->stage[0].p, wp, b_wp, hipMemcpyHostToDevice, ctx->stream));
  rc = launch_solve_grid(ctx, n_drones, (const double *)ctx->stage[0].p, (double *)ctx->stage[2].p,
                         (double *)ctx->stage[3].p, (int32_t *)ctx->stage[4].p);
  if (rc) return rc;
  MSNAP_HIP(ctx, hipMemcpyAsync(coef, ctx->stage[2].p, b_coef, hipMemcpyDeviceToHost, ctx->stream));
  MSNAP_HIP(ctx, hipMemcpyAsync(dur, ctx->stage[3].p, b_dur, hipMemcpyDeviceToHost, ctx->stream));
  MSNAP_HIP(ctx, hipMemcpyAsync(status, ctx->stage[4].p, b_st, hipMemcpyDeviceToHost, ctx->stream));
  MSNAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MSNAP_OK;
}

// ------------------------------------------------------------------ pack
int msnap_pack_pol_matrix_device(msnap_ctx *ctx, int n_drones, int n_seg, const double *coef,
                                 const double *dur, float *out) {
  if (!ctx || n_drones < 0) return MSNAP_EINVAL;
  int rc = check_seg(ctx, n_seg);
  if (rc) return rc;
  if (n_drones == 0) return MSNAP_OK;
  if (!coef || !dur || !out) return MSNAP_EINVAL;
  MSNAP_HIP(ctx, hipSetDevice(ctx->device));
  return launch_pack(ctx, n_drones, n_seg, coef, dur, out);
}

int msnap_pack_pol_matrix(msnap_ctx *ctx, int n_drones, int n_seg, const double *coef,
                          const double *dur, float *out) {
  if (!ctx || n_drones < 0) return MSNAP_EINVAL;
  int rc = check_seg(ctx, n_seg);
  if (rc) return rc;
  if (n_drones == 0) return MSNAP_OK;
  if (!coef || !dur || !out) return MSNAP_EINVAL;
  MSNAP_HIP(ctx, hipSetDevice(ctx->device));
  const size_t N = n_drones, nc = ctx->order + 1;
  const size_t b_coef = N * n_seg * 4 * nc * 8, b_dur = N * n_seg * 8, b_out = N * n_seg * (1 + 4 * nc) * 4;
  if ((rc = ensure(ctx, ctx->stage[2], b_coef))) return rc;
  if ((rc = ensure(ctx, ctx->stage[3], b_dur))) return rc;
  if ((rc = ensure(ctx, ctx->stage[5], b_out))) return rc;
  MSNAP_HIP(ctx, hipMemcpyAsync(ctx->stage[2].p, coef, b_coef, hipMemcpyHostToDevice, ctx->stream));
  MSNAP_HIP(ctx, hipMemcpyAsync(ctx->stage[3].p, dur, b_dur, hipMemcpyHostToDevice, ctx->stream));
  rc = launch_pack(ctx, n_drones, n_seg, (const double *)ctx->stage[2].p, (const double *)ctx->stage[3].p,
                   (float *)ctx->stage[5].p);
  if (rc) return rc;
  MSNAP_HIP(ctx, hipMemcpyAsync(out, ctx->stage[5].p, b_out, hipMemcpyDeviceToHost, ctx->stream));
  MSNAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MSNAP_OK;
}

// ------------------------------------------------------------------ formation transform
int msnap_formation_transform_device(msnap_ctx *ctx, int n_poses, int n_offsets, const double *rb_pose,
                                     const double *offsets, double *out) {
  if (!ctx || n_poses < 0 || n_offsets < 0) return MSNAP_EINVAL;
  if (n_poses == 0 || n_offsets == 0) return MSNAP_OK;
  if (!rb_pose || !offsets || !out) return MSNAP_EINVAL;
  MSNAP_HIP(ctx, hipSetDevice(ctx->device));
  return launch_formation_transform(ctx, n_poses, n_offsets, rb_pose, offsets, out);
}

int msnap_formation_transform(msnap_ctx *ctx, int n_poses, int n_offsets, const double *rb_pose,
                              const double *offsets, double *out) {
  if (!ctx || n_poses < 0 || n_offsets < 0) return MSNAP_EINVAL;
  if (n_poses == 0 || n_offsets == 0) return MSNAP_OK;
  if (!rb_pose || !offsets || !out) return MSNAP_EINVAL;
  MSNAP_HIP(ctx, hipSetDevice(ctx->device));
  const size_t b_in = (size_t)n_poses * 7 * 8, b_off = (size_t)n_offsets * 3 * 8;
  const size_t b_out = (size_t)n_offsets * n_poses * 7 * 8;
  int rc;
  if ((rc = ensure(ctx, ctx->stage[0], b_in))) return rc;
  if ((rc = ensure(ctx, ctx->stage[1], b_off))) return rc;
  if ((rc = ensure(ctx, ctx->stage[2], b_out))) return rc;
  MSNAP_HIP(ctx, hipMemcpyAsync(ctx->stage[0].p, rb_pose, b_in, hipMemcpyHostToDevice, ctx->stream));
  MSNAP_HIP(ctx, hipMemcpyAsync(ctx->stage[1].p, offsets, b_off, hipMemcpyHostToDevice, ctx->stream));
  rc = launch_formation_transform(ctx, n_poses, n_offsets, (const double *)ctx->stage[0].p,
                                  (const double *)ctx->stage[1].p, (double *)ctx->stage[2].p);
  if (rc) return rc;
  MSNAP_HIP(ctx, hipMemcpyAsync(out, ctx->stage[2].p, b_out, hipMemcpyDeviceToHost, ctx->stream));
  MSNAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MSNAP_OK;
}

// ------------------------------------------------------------------ sampler
static int sample_args_ok(const msnap_ctx *ctx, int n_drones, int n_samples, int n_axes, double dt) {
  if (!ctx || n_drones < 0 || n_samples < 0) return 0;
  if (n_axes != 3 && n_axes != 4) return 0;
  if (!(dt >= 0.0)) return 0;
  return 1;
}

int msnap_sample_device(msnap_ctx *ctx, int n_drones, int n_seg, const double *coef, const double *dur,
                        double dt, int n_samples, int n_axes, double *pos) {
  if (!sample_args_ok(ctx, n_drones, n_samples, n_axes, dt)) return MSNAP_EINVAL;
  int rc = check_seg(ctx, n_seg);
  if (rc) return rc;
  if (n_drones == 0 || n_samples == 0) return MSNAP_OK;
  if (!coef || !dur || !pos) return MSNAP_EINVAL;
  MSNAP_HIP(ctx, hipSetDevice(ctx->device));
  return launch_sample(ctx, n_drones, n_seg, coef, dur, dt, n_samples, n_axes, pos);
}

int msnap_sample(msnap_ctx *ctx, int n_drones, int n_seg, const double *coef, const double *dur, double dt,
                 int n_samples, int n_axes, double *pos) {
  if (!sample_args_ok(ctx, n_drones, n_samples, n_axes, dt)) return MSNAP_EINVAL;
  int rc = check_seg(ctx, n_seg);
  if (rc) return rc;
  if (n_drones == 0 || n_samples == 0) return MSNAP_OK;
  if (!coef || !dur || !pos) return MSNAP_EINVAL;
  MSNAP_HIP(ctx, hipSetDevice(ctx->device));
  const size_t N = n_drones, nc = ctx->order + 1;
  const size_t b_coef = N * n_seg * 4 * nc * 8, b_dur = N * n_seg * 8;
  const size_t b_pos = N * (size_t)n_samples * n_axes * 8;
  if ((rc = ensure(ctx, ctx->stage[2], b_coef))) return rc;
  if ((rc = ensure(ctx, ctx->stage[3], b_dur))) return rc;
  if ((rc = ensure(ctx, ctx->stage[6], b_pos))) return rc;
  MSNAP_HIP(ctx, hipMemcpyAsync(ctx->stage[2].p, coef, b_coef, hipMemcpyHostToDevice, ctx->stream));
  MSNAP_HIP(ctx, hipMemcpyAsync(ctx->stage[3].p, dur, b_dur, hipMemcpyHostToDevice, ctx->stream));
  rc = launch_sample(ctx, n_drones, n_seg, (const double *)ctx->stage[2].p, (const double *)ctx->stage[3].p,
                     dt, n_samples, n_axes, (double *)ctx->stage[6].p);
  if (rc) return rc;
  MSNAP_HIP(ctx, hipMemcpyAsync(pos, ctx->stage[6].p, b_pos, hipMemcpyDeviceToHost, ctx->stream));
  MSNAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MSNAP_OK;
}

// ------------------------------------------------------------------ flatness evaluator
int msnap_eval_flat_device(msnap_ctx *ctx, int n_drones, int n_seg, const double *coef, const double *dur,
                           int n_samples, const double *ts, double *out) {
  if (!ctx || n_drones < 0 || n_samples < 0) return MSNAP_EINVAL;
  int rc = check_seg(ctx, n_seg);
  if (rc) return rc;
  if (n_drones == 0 || n_samples == 0) return MSNAP_OK;
  if (!coef || !dur || !ts || !out) return MSNAP_EINVAL;
  MSNAP_HIP(ctx, hipSetDevice(ctx->device));
  return launch_eval_flat(ctx, n_drones, n_seg, coef, dur, n_samples, ts, out);
}

int msnap_eval_flat(msnap_ctx *ctx, int n_drones, int n_seg, const double *coef, const double *dur,
                    int n_samples, const double *ts, double *out) {
  if (!ctx || n_drones < 0 || n_samples < 0) return MSNAP_EINVAL;
  int rc = check_seg(ctx, n_seg);
  if (rc) return rc;
  if (n_drones == 0 || n_samples == 0) return MSNAP_OK;
  if (!coef || !dur || !ts || !out) return MSNAP_EINVAL;
  MSNAP_HIP(ctx, hipSetDevice(ctx->device));
  const size_t N = n_drones, nc = ctx->order + 1;
  const size_t b_coef = N * n_seg * 4 * nc * 8, b_dur = N * n_seg * 8, b_ts = (size_t)n_samples * 8;
  const size_t b_out = N * (size_t)n_samples * 13 * 8;
  if ((rc = ensure(ctx, ctx->stage[2], b_coef))) return rc;
  if ((rc = ensure(ctx, ctx->stage[3], b_dur))) return rc;
  if ((rc = ensure(ctx, ctx->stage[1], b_ts))) return rc;
  if ((rc = ensure(ctx, ctx->stage[6], b_out))) return rc;
  MSNAP_HIP(ctx, hipMemcpyAsync(ctx->stage[2].p, coef, b_coef, hipMemcpyHostToDevice, ctx->stream));
  MSNAP_HIP(ctx, hipMemcpyAsync(ctx->stage[3].p, dur, b_dur, hipMemcpyHostToDevice, ctx->stream));
  MSNAP_HIP(ctx, hipMemcpyAsync(ctx->stage[1].p, ts, b_ts, hipMemcpyHostToDevice, ctx->stream));
  rc = launch_eval_flat(ctx, n_drones, n_seg, (const double *)ctx->stage[2].p, (const double *)ctx->stage[3].p,
                        n_samples, (const double *)ctx->stage[1].p, (double *)ctx->stage[6].p);
  if (rc) return rc;
  MSNAP_HIP(ctx, hipMemcpyAsync(out, ctx->stage[6].p, b_out, hipMemcpyDeviceToHost, ctx->stream));
  MSNAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MSNAP_OK;
}

// ------------------------------------------------------------------ snap cost
int msnap_snap_cost_device(msnap_ctx *ctx, int n_drones, int n_seg, const double *coef, const double *dur,
                           double *cost) {
  if (!ctx || n_drones < 0) return MSNAP_EINVAL;
  int rc = check_seg(ctx, n_seg);
  if (rc) return rc;
  if (n_drones == 0) return MSNAP_OK;
  if (!coef || !dur || !cost) return MSNAP_EINVAL;
  MSNAP_HIP(ctx, hipSetDevice(ctx->device));
  return launch_snap_cost(ctx, n_drones, n_seg, coef, dur, cost);
}

int msnap_snap_cost(msnap_ctx *ctx, int n_drones, int n_seg, const double *coef, const double *dur, double *cost) {
  if (!ctx || n_drones < 0) return MSNAP_EINVAL;
  int rc = check_seg(ctx, n_seg);
  if (rc) return rc;
  if (n_drones == 0) return MSNAP_OK;
  if (!coef || !dur || !cost) return MSNAP_EINVAL;
  MSNAP_HIP(ctx, hipSetDevice(ctx->device));
  const size_t N = n_drones, nc = ctx->order + 1;
  const size_t b_coef = N * n_seg * 4 * nc * 8, b_dur = N * n_seg * 8, b_cost = N * 4 * 8;
  if ((rc = ensure(ctx, ctx->stage[2], b_coef))) return rc;
  if ((rc = ensure(ctx, ctx->stage[3], b_dur))) return rc;
  if ((rc = ensure(ctx, ctx->stage[6], b_cost))) return rc;
  MSNAP_HIP(ctx, hipMemcpyAsync(ctx->stage[2].p, coef, b_coef, hipMemcpyHostToDevice, ctx->stream));
  MSNAP_HIP(ctx, hipMemcpyAsync(ctx->stage[3].p, dur, b_dur, hipMemcpyHostToDevice, ctx->stream));
  rc = launch_snap_cost(ctx, n_drones, n_seg, (const double *)ctx->stage[2].p, (const double *)ctx->stage[3].p,
                        (double *)ctx->stage[6].p);
  if (rc) return rc;
  MSNAP_HIP(ctx, hipMemcpyAsync(cost, ctx->stage[6].p, b_cost, hipMemcpyDeviceToHost, ctx->stream));
  MSNAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MSNAP_OK;
}

// ------------------------------------------------------------------ formation collide
int msnap_formation_collide_device(msnap_ctx *ctx, int n_rows, int row_offset, int n_cols, int n_samples,
                                   const double *pos_rows, const double *pos_cols, double radius,
                                   double *min_dist, int32_t *partner, int32_t *hit) {
  if (!ctx || n_rows < 0 || n_cols < 0 || n_samples < 1 || row_offset < 0 || !(radius >= 0.0))
    return MSNAP_EINVAL;
  if (n_rows == 0) return MSNAP_OK;
  if (!pos_rows || (n_cols > 0 && !pos_cols) || !min_dist || !partner || !hit) return MSNAP_EINVAL;
  MSNAP_HIP(ctx, hipSetDevice(ctx->device));
  return launch_formation_collide(ctx, n_rows, row_offset, n_cols, n_samples, pos_rows, pos_cols, radius,
                                  min_dist, partner, hit);
}

int msnap_formation_collide(msnap_ctx *ctx, int n_rows, int row_offset, int n_cols, int n_samples,
                            const double *pos_rows, const double *pos_cols, double radius, double *min_dist,
                            int32_t *partner, int32_t *hit) {
  if (!ctx || n_rows < 0 || n_cols < 0 || n_samples < 1 || row_offset < 0 || !(radius >= 0.0))
    return MSNAP_EINVAL;
  if (n_rows == 0) return MSNAP_OK;
  if (!pos_rows || (n_cols > 0 && !pos_cols) || !min_dist || !partner || !hit) return MSNAP_EINVAL;
  MSNAP_HIP(ctx, hipSetDevice(ctx->device));
  const size_t b_rows = (size_t)n_rows * n_samples * 3 * 8, b_cols = (size_t)n_cols * n_samples * 3 * 8;
  int rc;
  if ((rc = ensure(ctx, ctx->stage[0], b_rows))) return rc;
  if ((rc = ensure(ctx, ctx->stage[1], b_cols + 8))) return rc;
  if ((rc = ensure(ctx, ctx->stage[2], (size_t)n_rows * 8))) return rc;
  if ((rc = ensure(ctx, ctx->stage[3], (size_t)n_rows * 4))) return rc;
  if ((rc = ensure(ctx, ctx->stage[4], (size_t)n_rows * 4))) return rc;
  MSNAP_HIP(ctx, hipMemcpyAsync(ctx->stage[0].p, pos_rows, b_rows, hipMemcpyHostToDevice, ctx->stream));
  if (b_cols)
    MSNAP_HIP(ctx, hipMemcpyAsync(ctx->stage[1].p, pos_cols, b_cols, hipMemcpyHostToDevice, ctx->stream));
  rc = launch_formation_collide(ctx, n_rows, row_offset, n_cols, n_samples, (const double *)ctx->stage[0].p,
                                (const double *)ctx->stage[1].p, radius, (double *)ctx->stage[2].p,
                                (int32_t *)ctx->stage[3].p, (int32_t *)ctx->stage[4].p);
  if (rc) return rc;
  MSNAP_HIP(ctx, hipMemcpyAsync(min_dist, ctx->stage[2].p, (size_t)n_rows * 8, hipMemcpyDeviceToHost, ctx->stream));
  MSNAP_HIP(ctx, hipMemcpyAsync(partner, ctx->stage[3].p, (size_t)n_rows * 4, hipMemcpyDeviceToHost, ctx->stream));
  MSNAP_HIP(ctx, hipMemcpyAsync(hit, ctx->stage[4].p, (size_t)n_rows * 4, hipMemcpyDeviceToHost, ctx->stream));
  MSNAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MSNAP_OK;
}

// ------------------------------------------------------------------ mesh sweep
int msnap_mesh_sweep_device(msnap_ctx *ctx, int n_drones, int n_samples, const double *pos, int n_tris,
                            const double *tris, double radius, double *min_dist, int32_t *hit) {
  if (!ctx || n_drones < 0 || n_samples < 1 || n_tris < 0 || !(radius >= 0.0)) return MSNAP_EINVAL;
  if (n_drones == 0) return MSNAP_OK;
  if (!pos || (n_tris > 0 && !tris) || !min_dist || !hit) return MSNAP_EINVAL;
  MSNAP_HIP(ctx, hipSetDevice(ctx->device));
  return launch_mesh_sweep(ctx, n_drones, n_samples, pos, n_tris, tris, radius, min_dist, hit);
}

int msnap_mesh_sweep(msnap_ctx *ctx, int n_drones, int n_samples, const double *pos, int n_tris,
                     const double *tris, double radius, double *min_dist, int32_t *hit) {
  if (!ctx || n_drones < 0 || n_samples < 1 || n_tris < 0 || !(radius >= 0.0)) return MSNAP_EINVAL;
  if (n_drones == 0) return MSNAP_OK;
  if (!pos || (n_tris > 0 && !tris) || !min_dist || !hit) return MSNAP_EINVAL;
  MSNAP_HIP(ctx, hipSetDevice(ctx->device));
  const size_t b_pos = (size_t)n_drones * n_samples * 3 * 8, b_tri = (size_t)n_tris * 9 * 8;
  int rc;
  if ((rc = ensure(ctx, ctx->stage[0], b_pos))) return rc;
  if ((rc = ensure(ctx, ctx->stage[1], b_tri + 8))) return rc;
  if ((rc = ensure(ctx, ctx->stage[2], (size_t)n_drones * 8))) return rc;
  if ((rc = ensure(ctx, ctx->stage[4], (size_t)n_drones * 4))) return rc;
  MSNAP_HIP(ctx, hipMemcpyAsync(ctx->stage[0].p, pos, b_pos, hipMemcpyHostToDevice, ctx->stream));
  if (b_tri) MSNAP_HIP(ctx, hipMemcpyAsync(ctx->stage[1].p, tris, b_tri, hipMemcpyHostToDevice, ctx->stream));
  rc = launch_mesh_sweep(ctx, n_drones, n_samples, (const double *)ctx->stage[0].p, n_tris,
                         (const double *)ctx->stage[1].p, radius, (double *)ctx->stage[2].p,
                         (int32_t *)ctx->stage[4].p);
  if (rc) return rc;
  MSNAP_HIP(ctx, hipMemcpyAsync(min_dist, ctx->stage[2].p, (size_t)n_drones * 8, hipMemcpyDeviceToHost, ctx->stream));
  MSNAP_HIP(ctx, hipMemcpyAsync(hit, ctx->stage[4].p, (size_t)n_drones * 4, hipMemcpyDeviceToHost, ctx->stream));
  MSNAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MSNAP_OK;
}

// ------------------------------------------------------------------ mesh-vs-mesh validity
int msnap_mesh_validity_device(msnap_ctx *ctx, int n_states, const double *states, int n_rtris, const double *rtris,
                               int n_etris, const double *etris, int32_t *valid) {
  if (!ctx || n_states < 0 || n_rtris < 0 || n_etris < 0) return MSNAP_EINVAL;
  if (n_states == 0) return MSNAP_OK;
  if (!states || !valid || (n_rtris > 0 && !rtris) || (n_etris > 0 && !etris)) return MSNAP_EINVAL;
  MSNAP_HIP(ctx, hipSetDevice(ctx->device));
  return launch_mesh_validity(ctx, n_states, states, n_rtris, rtris, n_etris, etris, valid);
}

int msnap_mesh_validity(msnap_ctx *ctx, int n_states, const double *states, int n_rtris, const double *rtris,
                        int n_etris, const double *etris, int32_t *valid) {
  if (!ctx || n_states < 0 || n_rtris < 0 || n_etris < 0) return MSNAP_EINVAL;
  if (n_states == 0) return MSNAP_OK;
  if (!states || !valid || (n_rtris > 0 && !rtris) || (n_etris > 0 && !etris)) return MSNAP_EINVAL;
  MSNAP_HIP(ctx, hipSetDevice(ctx->device));
  const size_t b_st = (size_t)n_states * 4 * 8, b_r = (size_t)n_rtris * 9 * 8, b_e = (size_t)n_etris * 9 * 8;
  int rc;
  if ((rc = ensure(ctx, ctx->stage[0], b_st))) return rc;
  if ((rc = ensure(ctx, ctx->stage[1], b_r + 8))) return rc;
  if ((rc = ensure(ctx, ctx->stage[2], b_e + 8))) return rc;
  if ((rc = ensure(ctx, ctx->stage[4], (size_t)n_states * 4))) return rc;
  MSNAP_HIP(ctx, hipMemcpyAsync(ctx->stage[0].p, states, b_st, hipMemcpyHostToDevice, ctx->stream));
  if (b_r) MSNAP_HIP(ctx, hipMemcpyAsync(ctx->stage[1].p, rtris, b_r, hipMemcpyHostToDevice, ctx->stream));
  if (b_e) MSNAP_HIP(ctx, hipMemcpyAsync(ctx->stage[2].p, etris, b_e, hipMemcpyHostToDevice, ctx->stream));
  rc = launch_mesh_validity(ctx, n_states, (const double *)ctx->stage[0].p, n_rtris, (const double *)ctx->stage[1].p,
                            n_etris, (const double *)ctx->stage[2].p, (int32_t *)ctx->stage[4].p);
  if (rc) return rc;
  MSNAP_HIP(ctx, hipMemcpyAsync(valid, ctx->stage[4].p, (size_t)n_states * 4, hipMemcpyDeviceToHost, ctx->stream));
  MSNAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MSNAP_OK;
}

}  // extern "C"
