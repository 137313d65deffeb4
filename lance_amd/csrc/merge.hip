// merge.hip -- (dist, rowid) merge of per-shard candidate lists on the device: the final step of a list-sharded search
// (every rank all-gathers the shards' local top-k*refine and merges them identically).
//
//   SortExec([_distance asc, _rowid asc]).with_fetch(k)          rust/lance/src/dataset/scanner.rs:3440-3468
//   refine: take the k*refine best by PQ distance, re-rank by the exact distance, fetch k      scanner.rs:2884-2904
#include "common.h"
#include "exact.cuh"
#include "kernels.h"
#include "search_common.cuh"

namespace lh {

// one workgroup per query; P = next power of two >= C (<= 4096)
__global__ __launch_bounds__(256) void merge_topk_kernel(const int64_t *__restrict__ ids, const float *__restrict__ dists,
                                                         const float *__restrict__ exact, int C, int P, int keff, int k,
                                                         int64_t *__restrict__ out_ids, float *__restrict__ out_dists) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint64_t *rid = reinterpret_cast<uint64_t *>(smem);
  uint32_t *key = reinterpret_cast<uint32_t *>(rid + P);
  uint32_t *pos = key + P;
  const int q = blockIdx.x;
  const int64_t *qi = ids + (int64_t)q * C;
  const float *qd = dists + (int64_t)q * C;
  for (int i = threadIdx.x; i < P; i += 256) {
    uint32_t kk = 0xFFFFFFFFu;
    uint64_t r = ~0ull;
    if (i < C && qi[i] >= 0) { kk = order_key(qd[i]); r = (uint64_t)qi[i]; }
    key[i] = kk; rid[i] = r; pos[i] = (uint32_t)i;
  }
  __syncthreads();
  bitonic_sort_kr<256>(key, rid, pos, P);
  if (exact) {
    // keep the keff best by PQ distance, then order THOSE by (exact distance, row id)
    const float *qe = exact + (int64_t)q * C;
    for (int i = threadIdx.x; i < P; i += 256) {
      uint32_t kk = 0xFFFFFFFFu;
      if (i < keff && rid[i] != ~0ull) kk = order_key(qe[pos[i]]);
      else rid[i] = ~0ull;
      key[i] = kk;
    }
    __syncthreads();
    bitonic_sort_kr<256>(key, rid, pos, P);
  }
  for (int i = threadIdx.x; i < k; i += 256) {
    const bool have = i < P && rid[i] != ~0ull;
    out_ids[(int64_t)q * k + i] = have ? (int64_t)rid[i] : -1;
    out_dists[(int64_t)q * k + i] = have ? key_to_float(key[i]) : INFINITY;
  }
}

}  // namespace lh

using namespace lh;

extern "C" int lance_hip_merge_topk(lance_hip_ctx *ctx, const int64_t *ids, const float *dists, const float *exact_dists, uint32_t nq,
                                    uint32_t c, uint32_t keff, uint32_t k, int64_t *out_ids, float *out_dists) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(ctx && (nq == 0 || (ids && dists && out_ids && out_dists)), "merge_topk: NULL argument");
  LH_REQUIRE(c > 0 && c <= 4096, "merge_topk: %u candidates per query not supported (1..4096)", c);
  LH_REQUIRE(k > 0, "merge_topk: k must be > 0");
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  if (nq == 0) return LANCE_HIP_OK;
  const int P = next_pow2(std::max<int>((int)c, 64));
  hipLaunchKernelGGL(merge_topk_kernel, dim3(nq), dim3(256), (size_t)P * 16, ctx->stream, ids, dists, exact_dists, (int)c, P, (int)keff, (int)k,
                     out_ids, out_dists);
  LH_CHECK_HIP(hipGetLastError());
  LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return LANCE_HIP_OK;
}
