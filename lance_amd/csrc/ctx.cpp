// ctx.cpp -- context lifecycle, thread-local error channel, scratch arena, memory helpers.
#include "common.h"

namespace lh {
static thread_local char g_err[1024] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace lh

void *lance_hip_ctx::scratch(const char *name, size_t bytes) {
  if (bytes == 0) bytes = 16;
  auto it = slots.find(name);
  if (it != slots.end() && it->second.second >= bytes) return it->second.first;
  if (capturing) {
    lh::set_error("scratch '%s' would have to grow to %zu bytes while a search graph is being captured", name, bytes);
    return nullptr;
  }
  if (it != slots.end()) {
    drop_graphs();   // captured searches hold the old block's address
    // make sure nothing in flight still uses the old block
    (void)hipStreamSynchronize(stream);
    (void)hipFree(it->second.first);
    slots.erase(it);
  }
  size_t cap = bytes + bytes / 4;
  cap = (cap + 255) & ~(size_t)255;
  void *p = nullptr;
  hipError_t e = hipMalloc(&p, cap);
  if (e != hipSuccess) {
    lh::set_error("hipMalloc(%zu) for scratch '%s' failed: %s", cap, name, hipGetErrorString(e));
    return nullptr;
  }
  slots[name] = {p, cap};
  return p;
}

// One-call buffers as large as a column (the long-row flat filter's bf16 plane): exactly `bytes` (no growth headroom), and a failed
// allocation is an answer, not an error -- the caller takes its route without the buffer.
void *lance_hip_ctx::scratch_exact(const char *name, size_t bytes) {
  if (bytes == 0) bytes = 16;
  auto it = slots.find(name);
  if (it != slots.end() && it->second.second >= bytes) return it->second.first;
  if (capturing) return nullptr;
  if (it != slots.end()) {
    drop_graphs();
    (void)hipStreamSynchronize(stream);
    (void)hipFree(it->second.first);
    slots.erase(it);
  }
  const size_t cap = (bytes + 255) & ~(size_t)255;
  void *p = nullptr;
  if (hipMalloc(&p, cap) != hipSuccess) {
    (void)hipGetLastError();      // clear the sticky error: the caller goes on without the buffer
    return nullptr;
  }
  slots[name] = {p, cap};
  return p;
}

void lance_hip_ctx::scratch_release(const char *name) {
  auto it = slots.find(name);
  if (it == slots.end()) return;
  (void)hipFree(it->second.first);
  slots.erase(it);
}

void lance_hip_ctx::drop_graphs() {
  graph_seen_once.clear();      // a key seen once must size the (reallocated) arena again before it is captured
  graph_seen_next = 0;
  if (graphs.empty()) return;
  (void)hipStreamSynchronize(stream);
  for (auto &kv : graphs)
    if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
  graphs.clear();
}

void *lance_hip_ctx::host_staging(size_t bytes) {
  if (bytes <= pinned_bytes) return pinned;
  if (pinned) (void)hipHostFree(pinned);
  pinned = nullptr;
  pinned_bytes = 0;
  size_t cap = bytes < (1 << 20) ? (1 << 20) : bytes + bytes / 4;
  if (hipHostMalloc(&pinned, cap, hipHostMallocDefault) != hipSuccess) {
    lh::set_error("hipHostMalloc(%zu) failed", cap);
    pinned = nullptr;
    return nullptr;
  }
  pinned_bytes = cap;
  return pinned;
}

uint32_t *lance_hip_ctx::host_flag_word() {
  if (host_flags) return host_flags;
  void *p = nullptr;
  if (hipHostMalloc(&p, 64, hipHostMallocDefault) != hipSuccess) {
    lh::set_error("hipHostMalloc(64) for the host status words failed");
    return nullptr;
  }
  host_flags = static_cast<uint32_t *>(p);
  host_flags[0] = 0;
  return host_flags;
}

void lance_hip_ctx::time_begin(const char *kernel) {
  hipEvent_t a, b;
  if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
  (void)hipEventRecord(a, stream);
  timers[kernel].pending.push_back({a, b});
}
void lance_hip_ctx::time_end(const char *kernel) {
  auto &t = timers[kernel];
  if (t.pending.empty()) return;
  (void)hipEventRecord(t.pending.back().second, stream);
}

extern "C" {

const char *lance_hip_last_error(void) { return lh::g_err; }
const char *lance_hip_version(void) { return "lance_hip 0.1 (gfx950)"; }

int lance_hip_ctx_create(int device_id, void *stream, lance_hip_ctx **out) {
  LH_REQUIRE(out != nullptr, "ctx_create: out is NULL");
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0) {
    lh::set_error("no HIP device available (%s)", e == hipSuccess ? "count=0" : hipGetErrorString(e));
    return LANCE_HIP_ERUNTIME;
  }
  LH_REQUIRE(device_id >= 0 && device_id < count, "ctx_create: device %d out of range (0..%d)", device_id, count - 1);
  LH_CHECK_HIP(hipSetDevice(device_id));
  auto *c = new lance_hip_ctx();
  c->device = device_id;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) c->num_cus = prop.multiProcessorCount;
  if (stream) {
    c->stream = reinterpret_cast<hipStream_t>(stream);
  } else {
    hipError_t se = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (se != hipSuccess) {
      delete c;
      lh::set_error("hipStreamCreate failed: %s", hipGetErrorString(se));
      return LANCE_HIP_ERUNTIME;
    }
    c->owns_stream = true;
  }
  *out = c;
  return LANCE_HIP_OK;
}

void lance_hip_ctx_destroy(lance_hip_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  ctx->drop_graphs();
  for (auto &kv : ctx->slots) (void)hipFree(kv.second.first);
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  if (ctx->host_flags) (void)hipHostFree(ctx->host_flags);
  for (auto &kv : ctx->timers)
    for (auto &ev : kv.second.pending) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

int lance_hip_synchronize(lance_hip_ctx *ctx) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(ctx, "ctx is NULL");
  LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return LANCE_HIP_OK;
}

int lance_hip_malloc(lance_hip_ctx *ctx, size_t bytes, void **out) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(ctx && out, "malloc: NULL argument");
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  hipError_t e = hipMalloc(out, bytes ? bytes : 16);
  if (e != hipSuccess) {
    lh::set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return LANCE_HIP_ENOMEM;
  }
  return LANCE_HIP_OK;
}
int lance_hip_free(lance_hip_ctx *ctx, void *ptr) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(ctx, "ctx is NULL");
  if (ptr) LH_CHECK_HIP(hipFree(ptr));
  return LANCE_HIP_OK;
}
int lance_hip_memcpy_h2d(lance_hip_ctx *ctx, void *dst, const void *src_host, size_t bytes) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(ctx && (bytes == 0 || (dst && src_host)), "memcpy_h2d: NULL argument");
  if (bytes == 0) return LANCE_HIP_OK;
  LH_CHECK_HIP(hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
  LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return LANCE_HIP_OK;
}
int lance_hip_memcpy_d2h(lance_hip_ctx *ctx, void *dst_host, const void *src, size_t bytes) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(ctx && (bytes == 0 || (dst_host && src)), "memcpy_d2h: NULL argument");
  if (bytes == 0) return LANCE_HIP_OK;
  LH_CHECK_HIP(hipMemcpyAsync(dst_host, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return LANCE_HIP_OK;
}

int lance_hip_timing_enable(lance_hip_ctx *ctx, int on) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(ctx, "ctx is NULL");
  ctx->timing = on != 0;
  return LANCE_HIP_OK;
}

int lance_hip_timing_query(lance_hip_ctx *ctx, const char *kernel, double *ms_total, uint64_t *launches) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(ctx && kernel, "timing_query: NULL argument");
  LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  if (strncmp(kernel, "count:", 6) == 0) {      // enqueue counter of a pipeline stage (common.h: stage_counts); not reset by the query
    auto it = ctx->stage_counts.find(kernel + 6);
    if (ms_total) *ms_total = 0.0;
    if (launches) *launches = it == ctx->stage_counts.end() ? 0 : it->second;
    return LANCE_HIP_OK;
  }
  auto &t = ctx->timers[kernel];
  for (auto &ev : t.pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) {
      t.ms += ms;
      t.launches += 1;
    }
    (void)hipEventDestroy(ev.first);
    (void)hipEventDestroy(ev.second);
  }
  t.pending.clear();
  if (ms_total) *ms_total = t.ms;
  if (launches) *launches = t.launches;
  t.ms = 0.0;
  t.launches = 0;
  return LANCE_HIP_OK;
}

}  // extern "C"
