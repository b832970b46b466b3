// C-ABI of libmsnap.so (include/msnap.h): context management, host-pointer
// wrappers, stream / timer plumbing.  All compute happens in the HIP kernels
// of msnap_solve.hip / msnap_aux.hip; there is no CPU fallback.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include <cstdarg>
#include <cstdio>

#include "msnap_internal.h"

namespace msnap {

int record_hip_error(msnap_ctx *ctx, hipError_t e, const char *what) {
  if (ctx) snprintf(ctx->hip_err, sizeof(ctx->hip_err), "%s: %s", what, hipGetErrorString(e));
  return MSNAP_EHIP;
}

int handover_form(const msnap_ctx *ctx, const void *ptr, int n, int n_samples) {
  if (!ptr) return 0;
  for (const auto &h : ctx->handover)
    if (h.ptr == ptr) return (h.n == n && h.s == n_samples) ? h.form : 0;
  return 0;
}

bool stream_is_capturing(const msnap_ctx *ctx) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  return ctx->stream && hipStreamIsCapturing(ctx->stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
}

int ensure(msnap_ctx *ctx, DevBuf &b, size_t bytes) {
  const bool capturing = stream_is_capturing(ctx);
  if (bytes <= b.cap) {
    // the launch about to be captured records pointers into this block: from now on a graph may replay on it
    if (capturing) b.in_graph = true;
    return MSNAP_OK;
  }
  // growing synchronises the stream and frees the old block: neither is legal while the stream is being
  // captured into a graph, and a capture must not silently record a launch on a buffer that is about to go
  if (capturing) return MSNAP_ECAPTURE;
  if (b.p && b.in_graph) {
    // a graph captured earlier may replay on the old block at any time: it is retired, not freed (msnap.h,
    // "Stream capture"); work queued on it eagerly stays valid for the same reason
    RetiredBuf *r = new (std::nothrow) RetiredBuf{b.p, b.cap, ctx->retired};
    if (!r) return MSNAP_ENOMEM;
    ctx->retired = r;
    b.p = nullptr;
    b.cap = 0;
    b.in_graph = false;
  } else if (b.p) {
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

struct OptionName {
  const char *name, *env;
};
static const OptionName kOptions[] = {
    {"solve_grid_waves", "MSNAP_SOLVE_GRID_WAVES"}, {"gemm_grid_waves", "MSNAP_GEMM_GRID_WAVES"},
    {"twist_max_drones", "MSNAP_TWIST_MAX_DRONES"}, {"no_twist", "MSNAP_NO_TWIST"},
    {"collide_waves_per_cu", "MSNAP_COLLIDE_WAVES_PER_CU"}, {"pipe_chunk_mb", "MSNAP_PIPE_CHUNK_MB"},
    {"collide_sample_parts", "MSNAP_COLLIDE_SAMPLE_PARTS"}, {"no_twin", "MSNAP_NO_TWIN"}, {"no_grid_sample", "MSNAP_NO_GRID_SAMPLE"}, {"gemm_stream_waves_per_cu", "MSNAP_GEMM_STREAM_WAVES_PER_CU"}, {"mesh_waves_per_cu", "MSNAP_MESH_WAVES_PER_CU"}, {"collide_no_cull", "MSNAP_COLLIDE_NO_CULL"}, {"collide_cull_min_drones", "MSNAP_COLLIDE_CULL_MIN_DRONES"}, {"collide_cull_mode", "MSNAP_COLLIDE_CULL_MODE"}, {"twin_max_drones", "MSNAP_TWIN_MAX_DRONES"},
};

void note_kernel(msnap_ctx *ctx, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(ctx->last_kernel, sizeof(ctx->last_kernel), fmt, ap);
  va_end(ap);
}

static bool option_is_read_only(const char *name) { return !strncmp(name, "collide_last_", 13); }

static int *option_slot(msnap_ctx *ctx, const char *name) {
  if (!strcmp(name, "solve_grid_waves")) return &ctx->solve_grid_waves;
  if (!strcmp(name, "gemm_grid_waves")) return &ctx->gemm_grid_waves;
  if (!strcmp(name, "twist_max_drones")) return &ctx->twist_max_drones;
  if (!strcmp(name, "no_twist")) return &ctx->no_twist;
  if (!strcmp(name, "no_twin")) return &ctx->no_twin;
  if (!strcmp(name, "no_grid_sample")) return &ctx->no_grid_sample;
  if (!strcmp(name, "gemm_stream_waves_per_cu")) return &ctx->gemm_stream_waves_per_cu;
  if (!strcmp(name, "mesh_waves_per_cu")) return &ctx->mesh_waves_per_cu;
  if (!strcmp(name, "twin_max_drones")) return &ctx->twin_max_drones;
  if (!strcmp(name, "collide_waves_per_cu")) return &ctx->collide_waves_per_cu;
  if (!strcmp(name, "collide_sample_parts")) return &ctx->collide_sample_parts;
  if (!strcmp(name, "collide_no_sym")) return &ctx->collide_no_sym;
  if (!strcmp(name, "collide_no_cull")) return &ctx->collide_no_cull;
  if (!strcmp(name, "collide_cull_min_drones")) return &ctx->collide_cull_min_drones;
  if (!strcmp(name, "collide_cull_mode")) return &ctx->collide_cull_mode;
  if (!strcmp(name, "collide_last_cull")) return &ctx->collide_last_cull;
  if (!strcmp(name, "collide_last_shares")) return &ctx->collide_last_shares;
  if (!strcmp(name, "collide_last_sym")) return &ctx->collide_last_sym;   // (read: what the last pass did)
  if (!strcmp(name, "own_stream_priority")) return &ctx->own_stream_priority;   // (read side; set has its own branch)
  return nullptr;
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
    case MSNAP_ESEGMENTS: return "segment count out of range for this context, or not the prepared grid's";
    case MSNAP_ENOMEM: return "out of memory";
    case MSNAP_ENODEVICE: return "no usable gfx950 device";
    case MSNAP_ENOGRID: return "no time grid prepared on this context (msnap_grid_prepare)";
    case MSNAP_ECAPTURE: return "a scratch buffer would have to grow during stream capture: run the call once outside the capture first";
    default: return "unknown msnap error";
  }
}

const char *msnap_last_hip_error(const msnap_ctx *ctx) { return ctx ? ctx->hip_err : ""; }

const char *msnap_last_kernel(const msnap_ctx *ctx) { return ctx ? ctx->last_kernel : ""; }

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
  // the only place the environment is read: seeds of the msnap_set_option knobs
  for (const OptionName &o : kOptions) {
    const char *e = getenv(o.env);
    if (e && atol(e) > 0) (void)msnap_set_option(ctx, o.name, atol(e));
  }
  int rc = MSNAP_OK;
  do {
    if (hipSetDevice(device_id) != hipSuccess) { rc = MSNAP_ENODEVICE; break; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) ctx->n_cu = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) { rc = MSNAP_EHIP; break; }
    ctx->stream = ctx->own_stream;
    if (hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) { rc = MSNAP_EHIP; break; }
    if (hipHostMalloc((void **)&ctx->cull_hint, 64, hipHostMallocDefault) != hipSuccess) {
      ctx->cull_hint = nullptr;      // (the pairwise pass then never has a hint and keeps its default evaluator)
      (void)hipGetLastError();
    } else {
      *ctx->cull_hint = 0;
    }
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
  (void)msnap_release_graph_buffers(ctx, nullptr);
  if (ctx->scratch.p) (void)hipFree(ctx->scratch.p);
  if (ctx->mesh_tests) (void)hipFree(ctx->mesh_tests);
  for (msnap::DevBuf *b : {&ctx->grid_t, &ctx->grid_wp, &ctx->grid_op, &ctx->grid_dur, &ctx->grid_status, &ctx->grid_frag})
    if (b->p) (void)hipFree(b->p);
  for (auto &b : ctx->stage)
    if (b.p) (void)hipFree(b.p);
  for (int k = 0; k < 2; ++k) {
    if (ctx->pipe_stream[k]) (void)hipStreamSynchronize(ctx->pipe_stream[k]);
    for (auto &b : ctx->pipe[k])
      if (b.p) (void)hipFree(b.p);
    if (ctx->pipe_stream[k]) (void)hipStreamDestroy(ctx->pipe_stream[k]);
  }
  if (ctx->pipe_start) (void)hipEventDestroy(ctx->pipe_start);
  if (ctx->bounce) (void)hipHostFree(ctx->bounce);
  if (ctx->cull_hint) (void)hipHostFree(ctx->cull_hint);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
}

int msnap_release_graph_buffers(msnap_ctx *ctx, size_t *bytes) {
  if (!ctx) return MSNAP_EINVAL;
  size_t freed = 0;
  int rc = MSNAP_OK;
  (void)hipSetDevice(ctx->device);
  while (ctx->retired) {
    RetiredBuf *r = ctx->retired;
    ctx->retired = r->next;
    if (hipFree(r->p) != hipSuccess) rc = MSNAP_EHIP;      // (hipFree waits for the device's outstanding work)
    freed += r->cap;
    delete r;
  }
  if (bytes) *bytes = freed;
  return rc;
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

int msnap_set_option(msnap_ctx *ctx, const char *name, long value) {
  if (!ctx || !name || value < 0 || value > (1L << 30)) return MSNAP_EINVAL;
  if (!strcmp(name, "pipe_chunk_mb")) {
    ctx->pipe_chunk_bytes = (size_t)(value > 0 ? value : 64) << 20;
    return MSNAP_OK;
  }
  if (!strcmp(name, "own_stream_priority")) {
    // 0 = default, 1 = the lowest priority the device offers, 2 = the highest: the context's own stream is
    // re-created (work queued on it is drained first; an external stream set by msnap_set_stream stays)
    if (value > 2) return MSNAP_EINVAL;
    MSNAP_HIP(ctx, hipSetDevice(ctx->device));
    int least = 0, greatest = 0;
    MSNAP_HIP(ctx, hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t fresh = nullptr;
    MSNAP_HIP(ctx, hipStreamCreateWithPriority(&fresh, hipStreamNonBlocking,
                                               value == 1 ? least : value == 2 ? greatest : 0));
    const bool current = ctx->stream == ctx->own_stream;
    if (ctx->own_stream) {
      (void)hipStreamSynchronize(ctx->own_stream);
      (void)hipStreamDestroy(ctx->own_stream);
    }
    ctx->own_stream = fresh;
    if (current) ctx->stream = fresh;
    ctx->own_stream_priority = (int)value;
    return MSNAP_OK;
  }
  if (!strcmp(name, "mesh_count_tests")) {
    MSNAP_HIP(ctx, hipSetDevice(ctx->device));
    MSNAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (value && !ctx->mesh_tests) MSNAP_HIP(ctx, hipMalloc(&ctx->mesh_tests, 8));
    if (ctx->mesh_tests) MSNAP_HIP(ctx, hipMemset(ctx->mesh_tests, 0, 8));
    if (!value && ctx->mesh_tests) {
      (void)hipFree(ctx->mesh_tests);
      ctx->mesh_tests = nullptr;
    }
    return MSNAP_OK;
  }
  if (option_is_read_only(name)) return MSNAP_EINVAL;      // "collide_last_*": what the last pass did
  int *slot = option_slot(ctx, name);
  if (!slot) return MSNAP_EINVAL;
  *slot = (int)value;
  return MSNAP_OK;
}

int msnap_get_option(const msnap_ctx *ctx, const char *name, long *value) {
  if (!ctx || !name || !value) return MSNAP_EINVAL;
  if (!strcmp(name, "pipe_chunk_mb")) {
    *value = (long)(ctx->pipe_chunk_bytes >> 20);
    return MSNAP_OK;
  }
  if (!strcmp(name, "mesh_count_tests")) {   // the count since it was last set (synchronises the stream)
    unsigned long long n = 0;
    if (ctx->mesh_tests) {
      if (stream_is_capturing(ctx)) return MSNAP_ECAPTURE;
      if (hipSetDevice(ctx->device) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess ||
          hipMemcpy(&n, ctx->mesh_tests, 8, hipMemcpyDeviceToHost) != hipSuccess)
        return MSNAP_EHIP;
    }
    *value = (long)n;
    return MSNAP_OK;
  }
  if (!strcmp(name, "collide_last_group_pairs") || !strcmp(name, "collide_last_survivors") ||
      !strcmp(name, "collide_last_by_groups") || !strcmp(name, "collide_last_pairs_evaluated")) {
    // counts the last broad-phase pass left on the device (synchronises the stream; not during a capture)
    int32_t shares = ctx->collide_last_shares, groups = 0;
    const bool culled = ctx->collide_last_cull && ctx->collide_meta;
    if (culled) {
      if (stream_is_capturing(ctx)) return MSNAP_ECAPTURE;
      if (hipSetDevice(ctx->device) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess ||
          hipMemcpy(&shares, ctx->collide_meta + MSNAP_COLLIDE_META_SHARES, sizeof shares, hipMemcpyDeviceToHost) != hipSuccess ||
          hipMemcpy(&groups, ctx->collide_meta + MSNAP_COLLIDE_META_GROUPS, sizeof groups, hipMemcpyDeviceToHost) != hipSuccess)
        return MSNAP_EHIP;
    }
    // (a large swarm's group evaluator ran only if its survivors fit the list: 1 << 18 slots, csrc/msnap_aux.hip)
    const bool by_groups = culled && (ctx->collide_last_by_groups == 1 || (ctx->collide_last_by_groups == 2 && groups <= (1 << 18)));
    if (!strcmp(name, "collide_last_group_pairs")) *value = groups;
    else if (!strcmp(name, "collide_last_survivors")) *value = shares;
    else if (!strcmp(name, "collide_last_by_groups")) *value = by_groups ? 1 : 0;
    else *value = !culled ? -1 : by_groups ? (long)groups * 64 : (long)shares * 1024;
    return MSNAP_OK;
  }
  const int *slot = option_slot(const_cast<msnap_ctx *>(ctx), name);
  if (!slot) return MSNAP_EINVAL;
  *value = *slot;
  return MSNAP_OK;
}

int msnap_host_alloc(void **ptr, size_t bytes) {
  if (!ptr) return MSNAP_EINVAL;
  *ptr = nullptr;
  if (bytes == 0) return MSNAP_OK;
  hipError_t e = hipHostMalloc(ptr, bytes, hipHostMallocDefault);
  if (e != hipSuccess) {
    *ptr = nullptr;
    (void)hipGetLastError();
    return e == hipErrorNoDevice ? MSNAP_ENODEVICE : MSNAP_ENOMEM;
  }
  return MSNAP_OK;
}

int msnap_host_free(void *ptr) {
  if (!ptr) return MSNAP_OK;
  return hipHostFree(ptr) == hipSuccess ? MSNAP_OK : MSNAP_EHIP;
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
// Host-pointer solve (msnap_solve_batch / msnap_solve_grid): H2D, kernel, D2H.
// Batches whose coefficients exceed one chunk (ctx->pipe_chunk_bytes) are cut into chunks of
// drones that alternate between two streams, each with its own device staging set, so chunk
// c+1's upload and kernel overlap chunk c's download; with page-locked host buffers
// (msnap_host_alloc) both copy engines then run at PCIe rate, with pageable memory the HIP
// runtime serialises the copies and the result is the same as one big copy.  Device staging is
// bounded by two chunks, whatever the batch size.
constexpr size_t kBounceMax = 1u << 20;   // bytes of inputs + outputs that take the single-bounce path

static int solve_host(msnap_ctx *ctx, int n_drones, int n_seg, const double *wp, const double *t, int shared,
                      bool use_grid, double *coef, double *dur, int32_t *status) {
  const size_t m = (size_t)n_seg + 1, nc = ctx->order + 1;
  const size_t pd_wp = m * 4 * 8, pd_t = m * 8;                       // bytes per drone
  const size_t pd_coef = (size_t)n_seg * 4 * nc * 8, pd_dur = (size_t)n_seg * 8, pd_st = 4;
  size_t chunk = ctx->pipe_chunk_bytes / pd_coef;
  chunk -= chunk % kDronesPerWave;
  if (chunk < 4096) chunk = 4096;
  const size_t N = n_drones;
  int rc;
  auto launch = [&](size_t n, const double *dwp, const double *dt, double *dcoef, double *ddur, int32_t *dst) {
    return use_grid ? launch_solve_grid(ctx, (int)n, dwp, dcoef, ddur, dst)
                    : launch_solve(ctx, (int)n, n_seg, dwp, dt, shared, dcoef, ddur, dst);
  };
  // Small batches (the reference's own call is ONE trajectory): everything goes through one
  // page-locked bounce buffer laid out like the device staging block [wp | t | coef | dur | status],
  // so a call is one upload, the kernel, one download and one synchronise instead of five pageable
  // copies (60 -> ~30 us per call).
  auto up256 = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t b_t = t ? (shared ? 1 : N) * pd_t : 0;
  const size_t o_t = up256(N * pd_wp), in_bytes = up256(o_t + b_t);                  // [wp | t]
  const size_t o_coef = in_bytes, o_dur = o_coef + up256(N * pd_coef), o_st = o_dur + up256(N * pd_dur);
  const size_t out_bytes = o_st + up256(N * pd_st) - o_coef;                          // [coef | dur | status]
  if (in_bytes + out_bytes <= kBounceMax && !solve_uses_global_scratch(ctx, n_seg)) {
    if (!ctx->bounce) {
      if (hipHostMalloc(&ctx->bounce, kBounceMax, hipHostMallocDefault) != hipSuccess) {
        ctx->bounce = nullptr;
        (void)hipGetLastError();
      } else {
        ctx->bounce_cap = kBounceMax;
      }
    }
    if (ctx->bounce && (rc = ensure(ctx, ctx->stage[5], kBounceMax)) == MSNAP_OK) {
      char *hb = (char *)ctx->bounce, *db = (char *)ctx->stage[5].p;
      memcpy(hb, wp, N * pd_wp);
      if (t) memcpy(hb + o_t, t, b_t);
      MSNAP_HIP(ctx, hipMemcpyAsync(db, hb, in_bytes, hipMemcpyHostToDevice, ctx->stream));
      rc = launch(N, (const double *)db, (const double *)(db + o_t), (double *)(db + o_coef), (double *)(db + o_dur),
                  (int32_t *)(db + o_st));
      if (rc) return rc;
      MSNAP_HIP(ctx, hipMemcpyAsync(hb + o_coef, db + o_coef, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
      MSNAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
      memcpy(coef, hb + o_coef, N * pd_coef);
      memcpy(dur, hb + o_dur, N * pd_dur);
      memcpy(status, hb + o_st, N * pd_st);
      return MSNAP_OK;
    }
  }
  if (N <= chunk || solve_uses_global_scratch(ctx, n_seg)) {
    // one shot on the context's stream
    if ((rc = ensure(ctx, ctx->stage[0], N * pd_wp))) return rc;
    if (t && (rc = ensure(ctx, ctx->stage[1], (shared ? 1 : N) * pd_t))) return rc;
    if ((rc = ensure(ctx, ctx->stage[2], N * pd_coef))) return rc;
    if ((rc = ensure(ctx, ctx->stage[3], N * pd_dur))) return rc;
    if ((rc = ensure(ctx, ctx->stage[4], N * pd_st))) return rc;
    MSNAP_HIP(ctx, hipMemcpyAsync(ctx->stage[0].p, wp, N * pd_wp, hipMemcpyHostToDevice, ctx->stream));
    if (t)
      MSNAP_HIP(ctx, hipMemcpyAsync(ctx->stage[1].p, t, (shared ? 1 : N) * pd_t, hipMemcpyHostToDevice,
                                    ctx->stream));
    rc = launch(N, (const double *)ctx->stage[0].p, (const double *)ctx->stage[1].p, (double *)ctx->stage[2].p,
                (double *)ctx->stage[3].p, (int32_t *)ctx->stage[4].p);
    if (rc) return rc;
    MSNAP_HIP(ctx, hipMemcpyAsync(coef, ctx->stage[2].p, N * pd_coef, hipMemcpyDeviceToHost, ctx->stream));
    MSNAP_HIP(ctx, hipMemcpyAsync(dur, ctx->stage[3].p, N * pd_dur, hipMemcpyDeviceToHost, ctx->stream));
    MSNAP_HIP(ctx, hipMemcpyAsync(status, ctx->stage[4].p, N * pd_st, hipMemcpyDeviceToHost, ctx->stream));
    MSNAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MSNAP_OK;
  }

  for (int k = 0; k < 2; ++k)
    if (!ctx->pipe_stream[k])
      MSNAP_HIP(ctx, hipStreamCreateWithFlags(&ctx->pipe_stream[k], hipStreamNonBlocking));
  if (!ctx->pipe_start) MSNAP_HIP(ctx, hipEventCreateWithFlags(&ctx->pipe_start, hipEventDisableTiming));
  const bool per_drone_t = t && !shared;
  for (int k = 0; k < 2; ++k) {
    if ((rc = ensure(ctx, ctx->pipe[k][0], chunk * pd_wp))) return rc;
    if (per_drone_t && (rc = ensure(ctx, ctx->pipe[k][1], chunk * pd_t))) return rc;
    if ((rc = ensure(ctx, ctx->pipe[k][2], chunk * pd_coef))) return rc;
    if ((rc = ensure(ctx, ctx->pipe[k][3], chunk * pd_dur))) return rc;
    if ((rc = ensure(ctx, ctx->pipe[k][4], chunk * pd_st))) return rc;
  }
  if (t && shared) {
    if ((rc = ensure(ctx, ctx->stage[1], pd_t))) return rc;
    MSNAP_HIP(ctx, hipMemcpyAsync(ctx->stage[1].p, t, pd_t, hipMemcpyHostToDevice, ctx->stream));
  }
  // the chunk streams start after whatever the context's stream holds (incl. the shared grid upload)
  MSNAP_HIP(ctx, hipEventRecord(ctx->pipe_start, ctx->stream));
  hipStream_t const user_stream = ctx->stream;
  rc = MSNAP_OK;
  hipError_t herr = hipSuccess;
  const char *what = "";
#define PIPE_HIP(call)                                   \
  if (rc == MSNAP_OK && herr == hipSuccess) {            \
    herr = (call);                                       \
    if (herr != hipSuccess) what = #call;                \
  }
  for (int k = 0; k < 2; ++k) PIPE_HIP(hipStreamWaitEvent(ctx->pipe_stream[k], ctx->pipe_start, 0));
  size_t c = 0;
  for (size_t d0 = 0; d0 < N && rc == MSNAP_OK && herr == hipSuccess; d0 += chunk, ++c) {
    const size_t n = (N - d0 < chunk) ? N - d0 : chunk;
    const int k = (int)(c & 1);
    hipStream_t s = ctx->pipe_stream[k];
    DevBuf *b = ctx->pipe[k];
    PIPE_HIP(hipMemcpyAsync(b[0].p, (const char *)wp + d0 * pd_wp, n * pd_wp, hipMemcpyHostToDevice, s));
    if (per_drone_t)
      PIPE_HIP(hipMemcpyAsync(b[1].p, (const char *)t + d0 * pd_t, n * pd_t, hipMemcpyHostToDevice, s));
    if (herr != hipSuccess) break;
    ctx->stream = s;   // the launchers enqueue on ctx->stream
    rc = launch(n, (const double *)b[0].p, per_drone_t ? (const double *)b[1].p : (const double *)ctx->stage[1].p,
                (double *)b[2].p, (double *)b[3].p, (int32_t *)b[4].p);
    ctx->stream = user_stream;
    PIPE_HIP(hipMemcpyAsync((char *)coef + d0 * pd_coef, b[2].p, n * pd_coef, hipMemcpyDeviceToHost, s));
    PIPE_HIP(hipMemcpyAsync((char *)dur + d0 * pd_dur, b[3].p, n * pd_dur, hipMemcpyDeviceToHost, s));
    PIPE_HIP(hipMemcpyAsync((char *)status + d0 * pd_st, b[4].p, n * pd_st, hipMemcpyDeviceToHost, s));
  }
#undef PIPE_HIP
  // always drain both streams, also on the error path: the staging sets must be idle on return
  for (int k = 0; k < 2; ++k) {
    hipError_t e = hipStreamSynchronize(ctx->pipe_stream[k]);
    if (herr == hipSuccess && e != hipSuccess) { herr = e; what = "hipStreamSynchronize(pipe)"; }
  }
  if (rc != MSNAP_OK) return rc;
  if (herr != hipSuccess) return record_hip_error(ctx, herr, what);
  return MSNAP_OK;
}

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
  return solve_host(ctx, n_drones, n_seg, wp, t, shared_times ? 1 : 0, /*use_grid=*/false, coef, dur, status);
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

int msnap_grid_segments(const msnap_ctx *ctx) {
  if (!ctx) return MSNAP_EINVAL;
  return ctx->grid_ready ? ctx->grid_seg : 0;
}

int msnap_solve_grid_device(msnap_ctx *ctx, int n_drones, int n_seg, const double *wp, double *coef, double *dur,
                            int32_t *status) {
  if (!ctx || n_drones < 0) return MSNAP_EINVAL;
  if (!ctx->grid_ready) return MSNAP_ENOGRID;
  if (n_seg != ctx->grid_seg) return MSNAP_ESEGMENTS;      // the caller's buffers are sized for another grid
  if (n_drones == 0) return MSNAP_OK;
  if (!wp || !coef || !dur || !status) return MSNAP_EINVAL;
  MSNAP_HIP(ctx, hipSetDevice(ctx->device));
  return launch_solve_grid(ctx, n_drones, wp, coef, dur, status);
}

int msnap_solve_grid(msnap_ctx *ctx, int n_drones, int n_seg, const double *wp, double *coef, double *dur,
                     int32_t *status) {
  if (!ctx || n_drones < 0) return MSNAP_EINVAL;
  if (!ctx->grid_ready) return MSNAP_ENOGRID;
  if (n_seg != ctx->grid_seg) return MSNAP_ESEGMENTS;
  if (n_drones == 0) return MSNAP_OK;
  if (!wp || !coef || !dur || !status) return MSNAP_EINVAL;
  MSNAP_HIP(ctx, hipSetDevice(ctx->device));
  return solve_host(ctx, n_drones, ctx->grid_seg, wp, nullptr, 1, /*use_grid=*/true, coef, dur, status);
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
  return launch_sample(ctx, n_drones, n_seg, coef, dur, dt, n_samples, n_axes, pos, nullptr, false);
}

size_t msnap_collide_rows_t_doubles(int n_rows, int n_samples) {
  if (n_rows <= 0 || n_samples <= 0) return 0;
  return ((size_t)n_rows + 127) / 128 * 128 * (size_t)n_samples * 3;
}

int msnap_formation_collide_reads_rows_t(const msnap_ctx *ctx, int n_rows, int row_offset, int n_cols, int n_samples) {
  (void)row_offset;
  if (!ctx || n_rows <= 0 || n_cols <= 0 || n_samples < 6) return 0;      // (paths shorter than one sample chunk: plain loops)
  return 1;
}

int msnap_formation_collide_takes_broad_phase(const msnap_ctx *ctx, int n_rows, int row_offset, int n_cols, int n_samples) {
  if (!ctx || n_rows <= 0 || n_cols <= 0 || n_samples < 1 || row_offset < 0) return 0;
  return formation_collide_takes_broad_phase(ctx, n_rows, row_offset, n_cols, n_samples) ? 1 : 0;
}

int msnap_formation_whole_pass_pays(msnap_ctx *ctx, int n_drones, int n_ranks, int *pays) {
  if (!ctx || !pays || n_drones < 0 || n_ranks < 1) return MSNAP_EINVAL;
  *pays = 0;
  long evaluated = -1, shares = 0, groups = 0;
  int rc = msnap_get_option(ctx, "collide_last_pairs_evaluated", &evaluated);
  if (rc) return rc;
  if (evaluated < 0 || n_drones < 2) return MSNAP_OK;      // the last pass evaluated all pairs: the parts divide them
  // (what the NEXT whole passes of this swarm evaluate: the evaluator follows these counts from then on)
  if ((rc = msnap_get_option(ctx, "collide_last_survivors", &shares)) || (rc = msnap_get_option(ctx, "collide_last_group_pairs", &groups)))
    return rc;
  evaluated = collide_counts_by_groups(ctx, ctx->collide_last_n, (int)shares, (int)groups) ? groups * 64 : shares * 1024;
  // whole pass on every rank: the sort / bound / select launches + the surviving pairs at about 2/3 of the all-pairs
  // kernel's pace; parts: 1 / n_ranks of all pairs + transposition, merge, fold and the second collective -- the fixed
  // costs of the two sides are about equal (measured at 4096 drones), which leaves the arithmetic
  const double all_pairs = 0.5 * (double)n_drones * (double)(n_drones - 1);
  *pays = ((double)evaluated / all_pairs) * 1.5 * (double)n_ranks < 1.0 ? 1 : 0;
  return MSNAP_OK;
}

// What the pass over these drones as a whole swarm will read from the sampler's second output, put on record for that
// pass: behind the broad phase the boxes and sort keys (true: the sampler has the samples in LDS; the pass then needs
// no key launch), otherwise the transposed row image (false).  A buffer seen before keeps its slot.
static bool record_handover(msnap_ctx *ctx, const double *pos_t, int n_drones, int n_samples) {
  const bool keys = formation_collide_takes_broad_phase(ctx, n_drones, 0, n_drones, n_samples);
  msnap_ctx::Handover *rec = nullptr;
  for (auto &h : ctx->handover)
    if (h.ptr == (const void *)pos_t) rec = &h;
  if (!rec) rec = &ctx->handover[ctx->handover_next++ % 8];
  *rec = {pos_t, n_drones, n_samples, keys ? 2 : 1};
  return keys;
}

int msnap_sample_collide_device(msnap_ctx *ctx, int n_drones, int n_seg, const double *coef, const double *dur,
                                double dt, int n_samples, double *pos, double *pos_t) {
  if (!sample_args_ok(ctx, n_drones, n_samples, 3, dt)) return MSNAP_EINVAL;
  int rc = check_seg(ctx, n_seg);
  if (rc) return rc;
  if (n_drones == 0 || n_samples == 0) return MSNAP_OK;
  if (!coef || !dur || !pos || !pos_t) return MSNAP_EINVAL;
  MSNAP_HIP(ctx, hipSetDevice(ctx->device));
  const bool keys = record_handover(ctx, pos_t, n_drones, n_samples);
  return launch_sample(ctx, n_drones, n_seg, coef, dur, dt, n_samples, 3, pos, pos_t, keys);
}

int msnap_solve_grid_sample_device(msnap_ctx *ctx, int n_drones, int n_seg, const double *wp, double dt, int n_samples,
                                   double *coef, double *dur, int32_t *status, double *pos, double *pos_t) {
  if (!sample_args_ok(ctx, n_drones, n_samples, 3, dt)) return MSNAP_EINVAL;
  if (!ctx->grid_ready) return MSNAP_ENOGRID;
  if (n_seg != ctx->grid_seg) return MSNAP_ESEGMENTS;
  if (n_drones == 0) return MSNAP_OK;
  if (!wp || !coef || !dur || !status) return MSNAP_EINVAL;
  MSNAP_HIP(ctx, hipSetDevice(ctx->device));
  if (n_samples == 0) return launch_solve_grid(ctx, n_drones, wp, coef, dur, status);
  if (!pos) return MSNAP_EINVAL;
  const bool keys = pos_t && record_handover(ctx, pos_t, n_drones, n_samples);
  return launch_grid_sample(ctx, n_drones, wp, dt, n_samples, coef, dur, status, pos, pos_t, keys);
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
                     dt, n_samples, n_axes, (double *)ctx->stage[6].p, nullptr, false);
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
                                  min_dist, partner, hit, nullptr);
}

int msnap_formation_collide_t_device(msnap_ctx *ctx, int n_rows, int row_offset, int n_cols, int n_samples,
                                     const double *pos_rows_t, const double *pos_rows, const double *pos_cols,
                                     double radius, double *min_dist, int32_t *partner, int32_t *hit) {
  if (!ctx || n_rows < 0 || n_cols < 0 || n_samples < 1 || row_offset < 0 || !(radius >= 0.0))
    return MSNAP_EINVAL;
  if (n_rows == 0) return MSNAP_OK;
  if (!pos_rows_t || !pos_rows || (n_cols > 0 && !pos_cols) || !min_dist || !partner || !hit) return MSNAP_EINVAL;
  MSNAP_HIP(ctx, hipSetDevice(ctx->device));
  return launch_formation_collide(ctx, n_rows, row_offset, n_cols, n_samples, pos_rows, pos_cols, radius,
                                  min_dist, partner, hit, pos_rows_t);
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
  // The once-per-pair evaluation credits column-side minima to the ROW drones, which is only right when
  // pos_rows is the slice [row_offset, row_offset + n_rows) of pos_cols.  Host arrays can be compared: a
  // caller whose rows are some other set of drones gets the one-sided evaluation instead of wrong partners.
  const int saved_no_sym = ctx->collide_no_sym;
  if ((long long)row_offset + n_rows <= n_cols && pos_rows != pos_cols + (size_t)row_offset * n_samples * 3 &&
      memcmp(pos_rows, pos_cols + (size_t)row_offset * n_samples * 3, b_rows) != 0)
    ctx->collide_no_sym = 1;
  rc = launch_formation_collide(ctx, n_rows, row_offset, n_cols, n_samples, (const double *)ctx->stage[0].p,
                                (const double *)ctx->stage[1].p, radius, (double *)ctx->stage[2].p,
                                (int32_t *)ctx->stage[3].p, (int32_t *)ctx->stage[4].p, nullptr);
  ctx->collide_no_sym = saved_no_sym;
  if (rc) return rc;
  MSNAP_HIP(ctx, hipMemcpyAsync(min_dist, ctx->stage[2].p, (size_t)n_rows * 8, hipMemcpyDeviceToHost, ctx->stream));
  MSNAP_HIP(ctx, hipMemcpyAsync(partner, ctx->stage[3].p, (size_t)n_rows * 4, hipMemcpyDeviceToHost, ctx->stream));
  MSNAP_HIP(ctx, hipMemcpyAsync(hit, ctx->stage[4].p, (size_t)n_rows * 4, hipMemcpyDeviceToHost, ctx->stream));
  MSNAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MSNAP_OK;
}

// ------------------------------------------------------------------ formation pass in parts (one per rank)
size_t msnap_formation_part_bytes(int n_drones) {
  if (n_drones < 0) return 0;
  return ((size_t)n_drones * (sizeof(double) + sizeof(int32_t)) + 7) & ~(size_t)7;
}

int msnap_formation_collide_part_device(msnap_ctx *ctx, int n_drones, int n_samples, const double *pos_all, int part,
                                        int n_parts, void *part_out) {
  if (!ctx || n_drones < 0 || n_samples < 1 || n_parts < 1 || part < 0 || part >= n_parts) return MSNAP_EINVAL;
  if (n_drones == 0) return MSNAP_OK;
  if (!pos_all || !part_out || ((uintptr_t)part_out & 7)) return MSNAP_EINVAL;
  MSNAP_HIP(ctx, hipSetDevice(ctx->device));
  return launch_formation_collide_part(ctx, n_drones, n_samples, pos_all, part, n_parts, (double *)part_out,
                                       (int32_t *)((unsigned char *)part_out + (size_t)n_drones * sizeof(double)));
}

int msnap_formation_collide_finish_device(msnap_ctx *ctx, int n_drones, int n_parts, const void *parts, int row_offset,
                                          int n_rows, double radius, double *min_dist, int32_t *partner, int32_t *hit) {
  if (!ctx || n_drones < 0 || n_parts < 1 || row_offset < 0 || n_rows < 0 || !(radius >= 0.0)) return MSNAP_EINVAL;
  if ((long long)row_offset + n_rows > n_drones) return MSNAP_EINVAL;
  if (n_rows == 0) return MSNAP_OK;
  if (!parts || !min_dist || !partner || !hit || ((uintptr_t)parts & 7)) return MSNAP_EINVAL;
  MSNAP_HIP(ctx, hipSetDevice(ctx->device));
  return launch_formation_collide_finish(ctx, n_drones, n_parts, parts, msnap_formation_part_bytes(n_drones),
                                         row_offset, n_rows, radius, min_dist, partner, hit);
}

int msnap_formation_collide_part(msnap_ctx *ctx, int n_drones, int n_samples, const double *pos_all, int part,
                                 int n_parts, void *part_out) {
  if (!ctx || n_drones < 0 || n_samples < 1 || n_parts < 1 || part < 0 || part >= n_parts) return MSNAP_EINVAL;
  if (n_drones == 0) return MSNAP_OK;
  if (!pos_all || !part_out) return MSNAP_EINVAL;
  MSNAP_HIP(ctx, hipSetDevice(ctx->device));
  const size_t b_pos = (size_t)n_drones * n_samples * 3 * 8, b_out = msnap_formation_part_bytes(n_drones);
  int rc;
  if ((rc = ensure(ctx, ctx->stage[1], b_pos + 8))) return rc;
  if ((rc = ensure(ctx, ctx->stage[2], b_out))) return rc;
  MSNAP_HIP(ctx, hipMemcpyAsync(ctx->stage[1].p, pos_all, b_pos, hipMemcpyHostToDevice, ctx->stream));
  rc = msnap_formation_collide_part_device(ctx, n_drones, n_samples, (const double *)ctx->stage[1].p, part, n_parts,
                                           ctx->stage[2].p);
  if (rc) return rc;
  MSNAP_HIP(ctx, hipMemcpyAsync(part_out, ctx->stage[2].p, b_out, hipMemcpyDeviceToHost, ctx->stream));
  MSNAP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MSNAP_OK;
}

int msnap_formation_collide_finish(msnap_ctx *ctx, int n_drones, int n_parts, const void *parts, int row_offset,
                                   int n_rows, double radius, double *min_dist, int32_t *partner, int32_t *hit) {
  if (!ctx || n_drones < 0 || n_parts < 1 || row_offset < 0 || n_rows < 0 || !(radius >= 0.0)) return MSNAP_EINVAL;
  if ((long long)row_offset + n_rows > n_drones) return MSNAP_EINVAL;
  if (n_rows == 0) return MSNAP_OK;
  if (!parts || !min_dist || !partner || !hit) return MSNAP_EINVAL;
  MSNAP_HIP(ctx, hipSetDevice(ctx->device));
  const size_t b_in = msnap_formation_part_bytes(n_drones) * (size_t)n_parts;
  int rc;
  if ((rc = ensure(ctx, ctx->stage[1], b_in + 8))) return rc;
  if ((rc = ensure(ctx, ctx->stage[2], (size_t)n_rows * 8))) return rc;
  if ((rc = ensure(ctx, ctx->stage[3], (size_t)n_rows * 4))) return rc;
  if ((rc = ensure(ctx, ctx->stage[4], (size_t)n_rows * 4))) return rc;
  MSNAP_HIP(ctx, hipMemcpyAsync(ctx->stage[1].p, parts, b_in, hipMemcpyHostToDevice, ctx->stream));
  rc = msnap_formation_collide_finish_device(ctx, n_drones, n_parts, ctx->stage[1].p, row_offset, n_rows, radius,
                                             (double *)ctx->stage[2].p, (int32_t *)ctx->stage[3].p,
                                             (int32_t *)ctx->stage[4].p);
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
