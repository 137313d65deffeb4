// common.h -- context, error channel, scratch arena, launch helpers (host side).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <initializer_list>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/lance_hip.h"

namespace lh {

void set_error(const char *fmt, ...);

#define LH_CHECK_HIP(expr)                                                              \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      lh::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return LANCE_HIP_ERUNTIME;                                                        \
    }                                                                                   \
  } while (0)

#define LH_REQUIRE(cond, ...)      \
  do {                             \
    if (!(cond)) {                 \
      lh::set_error(__VA_ARGS__);  \
      return LANCE_HIP_EINVAL;     \
    }                              \
  } while (0)

#define LH_TRY(expr)              \
  do {                            \
    int _r = (expr);              \
    if (_r != LANCE_HIP_OK) return _r; \
  } while (0)

// a captured search (search.hip: ivfpq_search_enqueue): the ~25 launches of one batch replayed as one hipGraphLaunch
struct GraphEntry {
  hipGraphExec_t exec = nullptr;
  bool failed = false;               // a capture of this key failed before: it stays on the plain path
  uint32_t *flags = nullptr;         // what the captured call handed back through flags_out
  const uint32_t *replay = nullptr;  // ... and left in last_replay_counter
  uint64_t last_use = 0;             // ctx->graph_tick of the last capture / replay (eviction: least recently used)
  std::vector<const char *> paths;   // the named pipeline stages the captured call went through (ScopedTimer names): a replay counts them
};

struct KernelTimer {
  double ms = 0.0;
  uint64_t launches = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

}  // namespace lh

// Grow-only device scratch: named slots so that steady-state calls never hipMalloc.
struct lance_hip_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool owns_stream = false;
  int num_cus = 256;
  std::map<std::string, std::pair<void *, size_t>> slots;
  void *pinned = nullptr;
  size_t pinned_bytes = 0;
  bool timing = false;
  const uint32_t *last_replay_counter = nullptr;  // device word written by the exact kernel of the last search
  std::map<std::string, lh::KernelTimer> timers;
  // Every extern "C" entry point holds this for its duration (lh::CtxLock): two host threads may share a context -- their
  // calls serialise, the work of each is ordered on the context's stream -- and threads that want to overlap use a context
  // each (the reference calls this path from many rayon / tokio threads at once, v2.rs:232-306).  Recursive: entry points
  // call each other.
  std::recursive_mutex mu;
  // captured searches, keyed by the packed arguments of the call.  Every node holds scratch-arena pointers, so the cache is
  // dropped whenever a slot is reallocated; growth DURING a capture is refused (the uncaptured first call has sized the arena).
  std::map<std::string, lh::GraphEntry> graphs;
  // keys seen exactly once (their first call ran uncaptured and sized the scratch arena): a bounded FIFO, so that callers that never
  // repeat a call -- fresh output buffers every time -- cost a string each and never touch the graphs above
  std::vector<std::string> graph_seen_once;
  size_t graph_seen_next = 0;
  uint64_t graph_tick = 0;
  bool capturing = false;
  std::vector<const char *> *capture_paths = nullptr;   // while capturing: the stage names of the call being captured
  void drop_graphs();
  // How often each named pipeline stage was ENQUEUED on this context -- plain launches, captures and graph replays alike (HIP-event
  // timing, by contrast, forces the plain path).  Read through lance_hip_timing_query("count:<stage>"): tests assert which kernels
  // served a call that went through a replayed graph.
  std::map<std::string, uint64_t> stage_counts;
  void count_stage(const char *name) {
    ++stage_counts[name];
    if (capture_paths) capture_paths->push_back(name);
  }

  // returns nullptr on failure (error set)
  void *scratch(const char *name, size_t bytes);
  void *scratch_exact(const char *name, size_t bytes);   // no headroom; nullptr (no error set) when the device cannot give it
  void scratch_release(const char *name);      // gives a slot back (large one-call buffers); the caller has synchronised the stream
  void *host_staging(size_t bytes);
  uint32_t *host_flags = nullptr;      // 64 pinned bytes: status words kernels leave for the host (read after a stream wait: no copy kernel)
  uint32_t *host_flag_word();          // nullptr on failure (error set)
  template <typename T>
  T *scratch_t(const char *name, size_t count) {
    return reinterpret_cast<T *>(scratch(name, count * sizeof(T)));
  }
  void time_begin(const char *kernel);
  void time_end(const char *kernel);
};

namespace lh {
struct ScopedTimer {
  lance_hip_ctx *c; const char *k;
  ScopedTimer(lance_hip_ctx *c_, const char *k_) : c(c_), k(k_) { c->count_stage(k); if (c->timing) c->time_begin(k); }
  ~ScopedTimer() { if (c->timing) c->time_end(k); }
};
// hipMemsetAsync as a plain kernel launch (dtype.hip).  Same cost as the runtime's own fill kernel, but a KERNEL node when a
// search is captured into a HIP graph: replays of graphs holding memset nodes returned empty results on ROCm 7.2 (gpurun r04c,
// tests/test_zz_gpu_graph.py) while the first launch of the same graph was right.
hipError_t memset_async(void *ptr, int value, size_t bytes, hipStream_t stream);
struct FillSpec { void *ptr; int value; size_t bytes; };
hipError_t memset_multi(hipStream_t stream, std::initializer_list<FillSpec> fills);      // up to six word-aligned fills in one launch (dtype.hip)
struct CtxLock {
  lance_hip_ctx *c;
  explicit CtxLock(lance_hip_ctx *c_) : c(c_) { if (c) c->mu.lock(); }
  ~CtxLock() { if (c) c->mu.unlock(); }
  CtxLock(const CtxLock &) = delete;
  CtxLock &operator=(const CtxLock &) = delete;
};
inline uint64_t cdiv(uint64_t a, uint64_t b) { return (a + b - 1) / b; }
}  // namespace lh
