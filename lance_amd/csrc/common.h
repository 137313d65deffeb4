// common.h -- context, error channel, scratch arena, launch helpers (host side).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
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
  uint32_t seen = 0;                 // calls with this key so far (the first one runs uncaptured: it sizes the scratch arena)
  uint32_t *flags = nullptr;         // what the captured call handed back through flags_out
  const uint32_t *replay = nullptr;  // ... and left in last_replay_counter
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
  bool capturing = false;
  void drop_graphs();

  // returns nullptr on failure (error set)
  void *scratch(const char *name, size_t bytes);
  void *host_staging(size_t bytes);
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
  ScopedTimer(lance_hip_ctx *c_, const char *k_) : c(c_), k(k_) { if (c->timing) c->time_begin(k); }
  ~ScopedTimer() { if (c->timing) c->time_end(k); }
};
// hipMemsetAsync as a plain kernel launch (dtype.hip).  Same cost as the runtime's own fill kernel, but a KERNEL node when a
// search is captured into a HIP graph: replays of graphs holding memset nodes returned empty results on ROCm 7.2 (gpurun r04c,
// tests/test_zz_gpu_graph.py) while the first launch of the same graph was right.
hipError_t memset_async(void *ptr, int value, size_t bytes, hipStream_t stream);
struct CtxLock {
  lance_hip_ctx *c;
  explicit CtxLock(lance_hip_ctx *c_) : c(c_) { if (c) c->mu.lock(); }
  ~CtxLock() { if (c) c->mu.unlock(); }
  CtxLock(const CtxLock &) = delete;
  CtxLock &operator=(const CtxLock &) = delete;
};
inline uint64_t cdiv(uint64_t a, uint64_t b) { return (a + b - 1) / b; }
}  // namespace lh
