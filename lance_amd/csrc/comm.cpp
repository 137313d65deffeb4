// comm.cpp -- the collectives of the row-sharded index build behind the C ABI.
//
// A Python host orders its RCCL calls through torch.distributed (lance_amd/dist.py).  A host without torch -- the Rust side of the
// reference -- needs the same loop behind plain C: a communicator handle (created here from an ncclUniqueId the host ships to its
// ranks, or adopted from an ncclComm_t the host already owns) and `lance_hip_kmeans_train_sharded`, which runs the whole Lloyd
// loop on the context's stream: local E-step + partial sums (kmeans.hip) -> ncclAllReduce SUM of the fused f32 buffer
// [k*d sums | k counts] and of the f64 per-cluster losses, ncclAllReduce MAX of the radii -> the update kernel (centroids, loss,
// balance factor, convergence, shared-seed split, next bias).  The host looks at the state every 8 iterations.
//   KMeans::train_kmeans   rust/lance-index/src/vector/kmeans.rs:610-719 (one exchange per iteration replaces the rayon reduction)
// librccl is resolved at first use: liblance_hip.so itself has no load-time dependency on it.  Inside a process that already maps an
// RCCL (torch bundles its own, soname librccl.so.1) THAT library must answer -- an adopted ncclComm_t is only meaningful to the
// library that made it -- so the lookup goes: symbols already visible (RTLD_DEFAULT), then the soname / file name with RTLD_NOLOAD
// (finds a library mapped RTLD_LOCAL, as Python's loader does), and only then a fresh load by name.
// A host that brings its own transport (MPI, gloo, a test harness running two ranks on one GPU) passes an all-reduce callback instead
// (lance_hip_comm_from_callback): the loop below is then the same code with the three exchanges routed through it.
#include <dlfcn.h>

#include <mutex>

#include "common.h"
#include "kernels.h"

namespace {

typedef struct { char internal[128]; } rccl_unique_id;      // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void *rccl_comm_t;
enum { RCCL_SUM = 0, RCCL_MAX = 2 };                          // ncclRedOp_t
enum { RCCL_F32 = 7, RCCL_F64 = 8 };                          // ncclDataType_t

struct RcclApi {
  void *lib = nullptr;
  bool resident = false;      // the process already maps an RCCL (torch's): its symbols are resolved through RTLD_DEFAULT
  int (*GetUniqueId)(rccl_unique_id *) = nullptr;
  int (*CommInitRank)(rccl_comm_t *, int, rccl_unique_id, int) = nullptr;
  int (*CommDestroy)(rccl_comm_t) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  bool ok = false;
};

RcclApi &rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    // An RCCL the process already exposes must answer (an adopted ncclComm_t belongs to THAT library, whatever its file is called or
    // however it was linked in).  RTLD_DEFAULT is a null pointer on glibc: the hit is kept in its own flag, not in `lib`.
    api.resident = dlsym(RTLD_DEFAULT, "ncclAllReduce") && dlsym(RTLD_DEFAULT, "ncclCommInitRank");
    if (!api.resident) {
      for (const char *name : {"librccl.so.1", "librccl.so"}) {
        api.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
        if (api.lib) break;
      }
      if (!api.lib)
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
          api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
          if (api.lib) break;
        }
      if (!api.lib) return;
    }
    void *from = api.resident ? RTLD_DEFAULT : api.lib;
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(from, "ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(from, "ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(from, "ncclCommDestroy"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(from, "ncclAllReduce"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(from, "ncclGetErrorString"));
    api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce;
  });
  return api;
}

int rccl_fail(const char *what, int rc) {
  RcclApi &a = rccl();
  lh::set_error("%s failed: %s (RCCL result %d)", what, a.GetErrorString ? a.GetErrorString(rc) : "?", rc);
  return LANCE_HIP_ERUNTIME;
}

}  // namespace

struct lance_hip_comm {
  rccl_comm_t comm = nullptr;
  int nranks = 1, rank = 0;
  bool owned = false;
  lance_hip_allreduce_fn fn = nullptr;      // host transport (lance_hip_comm_from_callback) instead of RCCL
  void *user = nullptr;
};

namespace {
// one exchange of the Lloyd loop: in place, on the context's stream (RCCL) or through the host's callback
int comm_allreduce(lance_hip_ctx *ctx, lance_hip_comm *comm, void *buf, size_t count, int dtype, int op) {
  if (comm->fn) {
    const int rc = comm->fn(comm->user, buf, (uint64_t)count, dtype, op, ctx->stream);
    if (rc != 0) { lh::set_error("all-reduce callback failed with %d", rc); return LANCE_HIP_ERUNTIME; }
    return LANCE_HIP_OK;
  }
  const int rc = rccl().AllReduce(buf, buf, count, dtype == LANCE_HIP_COMM_F64 ? RCCL_F64 : RCCL_F32, op == LANCE_HIP_COMM_MAX ? RCCL_MAX : RCCL_SUM,
                                  comm->comm, ctx->stream);
  if (rc != 0) return rccl_fail("ncclAllReduce", rc);
  return LANCE_HIP_OK;
}
}  // namespace

extern "C" {

int lance_hip_comm_unique_id(char *id_out_host) {
  LH_REQUIRE(id_out_host, "comm_unique_id: NULL argument");
  RcclApi &a = rccl();
  LH_REQUIRE(a.ok, "RCCL is not available (librccl.so could not be loaded)");
  rccl_unique_id id;
  const int rc = a.GetUniqueId(&id);
  if (rc != 0) return rccl_fail("ncclGetUniqueId", rc);
  memcpy(id_out_host, id.internal, sizeof(id.internal));
  return LANCE_HIP_OK;
}

int lance_hip_comm_create(lance_hip_ctx *ctx, const char *id_host, int nranks, int rank, lance_hip_comm **out) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(ctx && id_host && out, "comm_create: NULL argument");
  LH_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "comm_create: rank %d of %d", rank, nranks);
  RcclApi &a = rccl();
  LH_REQUIRE(a.ok, "RCCL is not available (librccl.so could not be loaded)");
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  rccl_unique_id id;
  memcpy(id.internal, id_host, sizeof(id.internal));
  rccl_comm_t c = nullptr;
  const int rc = a.CommInitRank(&c, nranks, id, rank);
  if (rc != 0) return rccl_fail("ncclCommInitRank", rc);
  auto *h = new lance_hip_comm();
  h->comm = c; h->nranks = nranks; h->rank = rank; h->owned = true;
  *out = h;
  return LANCE_HIP_OK;
}

int lance_hip_comm_adopt(void *nccl_comm, int nranks, int rank, lance_hip_comm **out) {
  LH_REQUIRE(nccl_comm && out, "comm_adopt: NULL argument");
  LH_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "comm_adopt: rank %d of %d", rank, nranks);
  LH_REQUIRE(rccl().ok, "RCCL is not available (librccl.so could not be loaded)");
  auto *h = new lance_hip_comm();
  h->comm = nccl_comm; h->nranks = nranks; h->rank = rank; h->owned = false;
  *out = h;
  return LANCE_HIP_OK;
}

int lance_hip_comm_from_callback(lance_hip_allreduce_fn fn, void *user, int nranks, int rank, lance_hip_comm **out) {
  LH_REQUIRE(fn && out, "comm_from_callback: NULL argument");
  LH_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "comm_from_callback: rank %d of %d", rank, nranks);
  auto *h = new lance_hip_comm();
  h->fn = fn; h->user = user; h->nranks = nranks; h->rank = rank;
  *out = h;
  return LANCE_HIP_OK;
}

void lance_hip_comm_destroy(lance_hip_comm *comm) {
  if (!comm) return;
  if (comm->owned && comm->comm && rccl().ok) (void)rccl().CommDestroy(comm->comm);
  delete comm;
}

int lance_hip_kmeans_train_sharded_x(lance_hip_ctx *ctx, lance_hip_comm *comm, int dtype, int metric, const void *x_local, uint64_t n_local, uint32_t d,
                                     uint32_t k, uint64_t n_total, uint32_t max_iters, double tol, float balance_factor, uint64_t seed,
                                     float *centroids, double *loss_out_host, uint32_t *iters_out_host) {
  lh::CtxLock _ctx_lock(ctx);
  // Argument errors that every rank sees alike may return at once ...
  LH_REQUIRE(ctx && centroids && (n_local == 0 || x_local), "kmeans_train_sharded: NULL argument");
  LH_REQUIRE(d > 0 && k > 0 && n_total >= k && n_total >= n_local, "kmeans_train_sharded: need n_total >= k and n_total >= n_local");
  LH_REQUIRE(metric == LANCE_HIP_L2 || metric == LANCE_HIP_DOT || metric == LANCE_HIP_COSINE, "kmeans_train_sharded: bad metric %d", metric);
  LH_REQUIRE(k <= 4096, "kmeans_train_sharded: k=%u > 4096 is not a flat Lloyd problem (the reference trains k > 256 hierarchically: "
             "lance_hip_kmeans_split is the unit of work of the multi-GPU hierarchical trainer)", k);
  LH_REQUIRE(!comm || comm->fn || rccl().ok, "RCCL is not available (librccl.so could not be loaded)");
  LH_TRY(lh::check_dtype(dtype, "kmeans_train_sharded"));
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  // ... everything that can fail on ONE rank only -- its row count, its scratch, the f32 copy of an f16 / int8 shard, its first E-step --
  // is folded into one status word that the ranks exchange BEFORE the first all-reduce of the loop: a rank that returned early would
  // leave its peers waiting in that all-reduce for ever.  The status slot itself is the one allocation that cannot be.
  float *status = ctx->scratch_t<float>("shard.status", 4);
  if (!status) return LANCE_HIP_ENOMEM;
  int pre = LANCE_HIP_OK;
  void *state = nullptr;
  float *bias = nullptr, *buf = nullptr, *radius = nullptr;
  double *losses = nullptr;
  const float *xf = nullptr;
  const float bf_scaled = balance_factor / (float)n_total;      // train_kmeans :1344: params.balance_factor /= data.len()
  if (n_local >= (1ull << 32)) {
    lh::set_error("kmeans_train_sharded: %llu rows on one rank (limit 2^32 - 1)", (unsigned long long)n_local);
    pre = LANCE_HIP_EINVAL;
  }
  if (pre == LANCE_HIP_OK) {
    state = ctx->scratch("shard.state", 256);
    bias = ctx->scratch_t<float>("shard.bias", k);
    buf = ctx->scratch_t<float>("shard.buf", (size_t)k * d + k);
    losses = ctx->scratch_t<double>("shard.losses", k);
    radius = ctx->scratch_t<float>("shard.radius", k);
    if (!state || !bias || !buf || !losses || !radius) pre = LANCE_HIP_ENOMEM;
  }
  // f16 / int8 shards: widened once, here (exact; the flat sharded loop accumulates in f32 -- the M-step of an f16 column on ONE GPU
  // rounds like half::f16, one more reason the two agree to round-off only)
  if (pre == LANCE_HIP_OK) pre = lh::as_f32(ctx, dtype, x_local, (size_t)n_local * d, "shard.x", &xf);
  if (pre == LANCE_HIP_OK) pre = lance_hip_kmeans_shard_begin(ctx, k, bf_scaled, seed, state, bias);
  double loss = 0.0;
  uint32_t iters = 0;
  int active = 1;
  for (uint32_t it = 1; it <= max_iters; ++it) {
    const int erc = pre != LANCE_HIP_OK ? pre : lance_hip_kmeans_shard_estep(ctx, metric, xf, n_local, d, centroids, k, bias, state, buf, losses, radius);
    if (comm && it == 1) {
      char saved[1024];
      snprintf(saved, sizeof(saved), "%s", lance_hip_last_error());
      const float mine = erc != LANCE_HIP_OK ? 1.0f : 0.0f;
      float any = 1.0f;
      LH_CHECK_HIP(hipMemcpyAsync(status, &mine, 4, hipMemcpyHostToDevice, ctx->stream));
      LH_TRY(comm_allreduce(ctx, comm, status, 1, LANCE_HIP_COMM_F32, LANCE_HIP_COMM_MAX));
      LH_CHECK_HIP(hipMemcpyAsync(&any, status, 4, hipMemcpyDeviceToHost, ctx->stream));
      LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
      if (erc != LANCE_HIP_OK) { lh::set_error("%s", saved); return erc; }
      if (any != 0.0f) { lh::set_error("kmeans_train_sharded: set-up or the first E-step failed on another rank (this rank stops with it)"); return LANCE_HIP_ERUNTIME; }
    } else if (erc != LANCE_HIP_OK) {
      return erc;
    }
    if (comm) {
      LH_TRY(comm_allreduce(ctx, comm, buf, (size_t)k * d + k, LANCE_HIP_COMM_F32, LANCE_HIP_COMM_SUM));
      LH_TRY(comm_allreduce(ctx, comm, losses, k, LANCE_HIP_COMM_F64, LANCE_HIP_COMM_SUM));
      LH_TRY(comm_allreduce(ctx, comm, radius, k, LANCE_HIP_COMM_F32, LANCE_HIP_COMM_MAX));
    }
    LH_TRY(lance_hip_kmeans_shard_update(ctx, state, buf, losses, radius, centroids, bias, k, d, n_total, bf_scaled, tol, it));
    if (it % 8 == 0 || it == max_iters) {
      LH_TRY(lance_hip_kmeans_shard_end(ctx, state, &loss, &iters, &active));
      if (!active) break;
    }
  }
  if (iters == 0) LH_TRY(lance_hip_kmeans_shard_end(ctx, state, &loss, &iters, &active));
  if (loss_out_host) *loss_out_host = loss;
  if (iters_out_host) *iters_out_host = iters;
  return LANCE_HIP_OK;
}

int lance_hip_kmeans_train_sharded(lance_hip_ctx *ctx, lance_hip_comm *comm, int metric, const float *x_local, uint64_t n_local, uint32_t d,
                                   uint32_t k, uint64_t n_total, uint32_t max_iters, double tol, float balance_factor, uint64_t seed,
                                   float *centroids, double *loss_out_host, uint32_t *iters_out_host) {
  return lance_hip_kmeans_train_sharded_x(ctx, comm, LANCE_HIP_F32, metric, x_local, n_local, d, k, n_total, max_iters, tol, balance_factor, seed, centroids,
                                          loss_out_host, iters_out_host);
}

}  // extern "C"
