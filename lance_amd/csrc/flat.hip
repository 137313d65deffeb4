// flat.hip -- exhaustive KNN over raw vectors (BASELINE config 1; ground truth; un-indexed data).
//
//   KNNVectorDistanceExec / compute_distance   lance/src/io/exec/knn.rs:218-246, lance-index flat.rs:95-148
//   l2_distance_arrow_batch                    lance-linalg l2.rs:245-266 (-> l2_scalar order)
//   SortExec(dist asc, rowid asc).fetch(k)     lance/src/dataset/scanner.rs:3386-3406
//
// Lanes own QUERIES (query vector in VGPRs), database rows stream through LDS tiles that
// every lane reads with wave-uniform addresses (broadcast).  Each lane keeps its running
// top-k as a sorted (key, row id) list in global scratch and its current worst entry in
// registers, so the common case is one compare per (query, row); insertions are rare
// (~k ln(n/k) per query).  Row ranges are split across workgroups; a merge kernel takes
// the final (dist, rowid) order -- a total order, so the result equals the reference's
// SortExec exactly, ties included.
#include <algorithm>

#include "common.h"
#include "exact.cuh"
#include "kernels.h"

#pragma clang fp contract(off)

namespace lh {

struct FlatArgs {
  const float *x;
  const uint64_t *row_ids;
  int64_t n;
  int d;
  const float *q;
  int nq, k, nsplit;
  int64_t rows_per_split;
  uint32_t *lkeys;  // [nsplit][nq][k]
  uint64_t *lrids;
};

__device__ __forceinline__ void flat_insert(uint32_t *lk, uint64_t *lr, int k, int &cnt, uint32_t key, uint64_t rid,
                                            uint32_t &wkey, uint64_t &wrid) {
  // list sorted ascending by (key, rid); drop the last element when full
  int pos = cnt < k ? cnt : k - 1;
  while (pos > 0) {
    const uint32_t pk = lk[pos - 1];
    const uint64_t pr = lr[pos - 1];
    if (pk < key || (pk == key && pr < rid)) break;
    lk[pos] = pk; lr[pos] = pr;
    --pos;
  }
  lk[pos] = key; lr[pos] = rid;
  if (cnt < k) ++cnt;
  if (cnt == k) { wkey = lk[k - 1]; wrid = lr[k - 1]; }
}

template <int D, int METRIC, int TR>
__global__ __launch_bounds__(256) void flat_scan_kernel(FlatArgs p) {
  __shared__ __attribute__((aligned(16))) float tile[TR * D];
  __shared__ uint64_t trid[TR];
  const int qi = blockIdx.x * 256 + threadIdx.x;
  const int sp = blockIdx.y;
  const bool valid = qi < p.nq;
  RegVec<D> a;
#pragma unroll
  for (int i = 0; i < RegVec<D>::Q; ++i) a.q[i] = f4{0.f, 0.f, 0.f, 0.f};
  if (valid) {
    const float *src = p.q + (int64_t)qi * D;
#pragma unroll
    for (int i = 0; i < RegVec<D>::Q * 4; ++i) a.q[i >> 2][i & 3] = i < D ? src[i] : 0.0f;
  }
  uint32_t *lk = p.lkeys + ((int64_t)sp * p.nq + (valid ? qi : 0)) * p.k;
  uint64_t *lr = p.lrids + ((int64_t)sp * p.nq + (valid ? qi : 0)) * p.k;
  int cnt = 0;
  uint32_t wkey = 0xFFFFFFFFu;
  uint64_t wrid = ~0ull;
  const int64_t r0 = (int64_t)sp * p.rows_per_split;
  const int64_t r1 = min(p.n, r0 + p.rows_per_split);
  constexpr bool NEG = METRIC != METRIC_DOT;
  for (int64_t t0 = r0; t0 < r1; t0 += TR) {
    const int tr = (int)min<int64_t>(TR, r1 - t0);
    __syncthreads();
    for (int i = threadIdx.x; i < tr * D; i += 256) {
      const float v = p.x[t0 * D + i];
      tile[i] = NEG ? -v : v;
    }
    for (int i = threadIdx.x; i < tr; i += 256) trid[i] = p.row_ids ? p.row_ids[t0 + i] : (uint64_t)(t0 + i);
    __syncthreads();
    if (valid) {
      for (int r = 0; r < tr; ++r) {
        const float v = finish_metric<METRIC>(dist_exact<D, METRIC, NEG>(a, &tile[r * D]));
        const uint32_t key = order_key(v);
        if (key < wkey || cnt < p.k || (key == wkey && trid[r] < wrid)) flat_insert(lk, lr, p.k, cnt, key, trid[r], wkey, wrid);
      }
    }
  }
  if (valid)
    for (int i = cnt; i < p.k; ++i) { lk[i] = 0xFFFFFFFFu; lr[i] = ~0ull; }
}

// generic dimension: query read through global/L1 per row (correctness fallback)
template <int METRIC>
__global__ __launch_bounds__(256) void flat_scan_generic_kernel(FlatArgs p, int tr_max) {
  extern __shared__ __attribute__((aligned(16))) char fsm[];
  uint64_t *trid = reinterpret_cast<uint64_t *>(fsm);
  float *tile = reinterpret_cast<float *>(trid + tr_max);
  const int qi = blockIdx.x * 256 + threadIdx.x;
  const int sp = blockIdx.y;
  const bool valid = qi < p.nq;
  const float *qv = p.q + (int64_t)(valid ? qi : 0) * p.d;
  uint32_t *lk = p.lkeys + ((int64_t)sp * p.nq + (valid ? qi : 0)) * p.k;
  uint64_t *lr = p.lrids + ((int64_t)sp * p.nq + (valid ? qi : 0)) * p.k;
  int cnt = 0;
  uint32_t wkey = 0xFFFFFFFFu;
  uint64_t wrid = ~0ull;
  const int64_t r0 = (int64_t)sp * p.rows_per_split;
  const int64_t r1 = min(p.n, r0 + p.rows_per_split);
  float qnorm = 0.0f;
  if constexpr (METRIC == METRIC_COSINE) qnorm = norm_l2_rt(qv, p.d);
  for (int64_t t0 = r0; t0 < r1; t0 += tr_max) {
    const int tr = (int)min<int64_t>(tr_max, r1 - t0);
    __syncthreads();
    for (int i = threadIdx.x; i < tr * p.d; i += 256) tile[i] = p.x[t0 * p.d + i];
    for (int i = threadIdx.x; i < tr; i += 256) trid[i] = p.row_ids ? p.row_ids[t0 + i] : (uint64_t)(t0 + i);
    __syncthreads();
    if (valid) {
      for (int r = 0; r < tr; ++r) {
        float v;
        if constexpr (METRIC == METRIC_COSINE) v = cosine_exact_rt(qv, qnorm, &tile[r * p.d], p.d);
        else v = finish_metric<METRIC>(dist_exact_rt<METRIC>(qv, &tile[r * p.d], p.d));
        const uint32_t key = order_key(v);
        if (key < wkey || cnt < p.k || (key == wkey && trid[r] < wrid)) flat_insert(lk, lr, p.k, cnt, key, trid[r], wkey, wrid);
      }
    }
  }
  if (valid)
    for (int i = cnt; i < p.k; ++i) { lk[i] = 0xFFFFFFFFu; lr[i] = ~0ull; }
}

__global__ __launch_bounds__(256) void flat_merge_kernel(const uint32_t *__restrict__ lkeys, const uint64_t *__restrict__ lrids,
                                                         int nq, int k, int nsplit, int P, uint64_t *__restrict__ out_ids,
                                                         float *__restrict__ out_dists) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint64_t *rid = reinterpret_cast<uint64_t *>(smem);
  uint32_t *key = reinterpret_cast<uint32_t *>(rid + P);
  const int q = blockIdx.x;
  for (int i = threadIdx.x; i < P; i += 256) {
    uint32_t kk = 0xFFFFFFFFu;
    uint64_t r = ~0ull;
    if (i < nsplit * k) {
      const int sp = i / k, j = i % k;
      kk = lkeys[((int64_t)sp * nq + q) * k + j];
      r = lrids[((int64_t)sp * nq + q) * k + j];
    }
    key[i] = kk; rid[i] = r;
  }
  __syncthreads();
  for (int k2 = 2; k2 <= P; k2 <<= 1) {
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < P / 2; i += 256) {
        const int ix = 2 * j * (i / j) + (i % j);
        const int px = ix + j;
        const bool up = (ix & k2) == 0;
        const uint32_t kx = key[ix], ky = key[px];
        const uint64_t rx = rid[ix], ry = rid[px];
        const bool gt = kx > ky || (kx == ky && rx > ry);
        if (gt == up) { key[ix] = ky; key[px] = kx; rid[ix] = ry; rid[px] = rx; }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < k; i += 256) {
    const bool ok = rid[i] != ~0ull;
    out_ids[(int64_t)q * k + i] = rid[i];
    out_dists[(int64_t)q * k + i] = ok ? key_to_float(key[i]) : INFINITY;
  }
}

template <int D>
static void launch_flat_fixed(lance_hip_ctx *ctx, const FlatArgs &a, int metric, dim3 grid) {
  constexpr int TR = (8192 / D) > 256 ? 256 : (8192 / D);
  if (metric == METRIC_DOT)
    hipLaunchKernelGGL((flat_scan_kernel<D, METRIC_DOT, TR>), grid, dim3(256), 0, ctx->stream, a);
  else
    hipLaunchKernelGGL((flat_scan_kernel<D, METRIC_L2, TR>), grid, dim3(256), 0, ctx->stream, a);
}

}  // namespace lh

using namespace lh;

extern "C" int lance_hip_flat_topk(lance_hip_ctx *ctx, int dtype, int metric, const void *x, const uint64_t *row_ids,
                                   uint64_t n, uint32_t d, const void *q, uint32_t nq, uint32_t k, uint64_t *ids,
                                   float *dists) {
  LH_REQUIRE(ctx && (n == 0 || x) && (nq == 0 || (q && ids && dists)), "flat_topk: NULL argument");
  LH_TRY(check_dtype(dtype, "flat_topk"));
  LH_REQUIRE(!(dtype == LANCE_HIP_F16 && metric != LANCE_HIP_L2), "flat_topk: f16 supports L2 only in this version");
  LH_REQUIRE(metric == LANCE_HIP_L2 || metric == LANCE_HIP_DOT || metric == LANCE_HIP_COSINE, "flat_topk: bad metric %d", metric);
  LH_REQUIRE(k > 0 && k <= 1024, "flat_topk: k=%u not supported (1..1024)", k);
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  if (nq == 0) return LANCE_HIP_OK;
  const int qblocks = (int)cdiv(nq, 256);
  int nsplit = (int)cdiv(2ull * ctx->num_cus, qblocks);
  nsplit = std::max(1, std::min<int>(nsplit, (int)(2048 / k)));
  nsplit = (int)std::min<uint64_t>(nsplit, std::max<uint64_t>(1, cdiv(n, 256)));
  FlatArgs a;
  const float *xf, *qf;
  LH_TRY(as_f32(ctx, dtype, x, (size_t)n * d, "f16.x", &xf));
  LH_TRY(as_f32(ctx, dtype, q, (size_t)nq * d, "f16.q", &qf));
  a.x = xf; a.row_ids = row_ids; a.n = (int64_t)n; a.d = (int)d;
  a.q = qf; a.nq = (int)nq; a.k = (int)k; a.nsplit = nsplit;
  a.rows_per_split = (int64_t)cdiv(n > 0 ? n : 1, nsplit);
  a.lkeys = ctx->scratch_t<uint32_t>("flat.lkeys", (size_t)nsplit * nq * k);
  a.lrids = ctx->scratch_t<uint64_t>("flat.lrids", (size_t)nsplit * nq * k);
  if (!a.lkeys || !a.lrids) return LANCE_HIP_ENOMEM;
  const dim3 grid(qblocks, nsplit);
  {
    ScopedTimer t(ctx, "flat_scan");
    switch (metric == LANCE_HIP_COSINE ? 0u : d) {
      case 8: launch_flat_fixed<8>(ctx, a, metric, grid); break;
      case 16: launch_flat_fixed<16>(ctx, a, metric, grid); break;
      case 32: launch_flat_fixed<32>(ctx, a, metric, grid); break;
      case 64: launch_flat_fixed<64>(ctx, a, metric, grid); break;
      case 96: launch_flat_fixed<96>(ctx, a, metric, grid); break;
      case 128: launch_flat_fixed<128>(ctx, a, metric, grid); break;
      default: {
        int tr = (int)(8192 / d);
        tr = std::max(1, std::min(tr, 256));
        const size_t lds = (size_t)tr * 8 + (size_t)tr * d * 4;
        LH_REQUIRE(lds <= 160 * 1024, "flat_topk: dimension %u too large", d);
        if (metric == LANCE_HIP_COSINE)
          hipLaunchKernelGGL((flat_scan_generic_kernel<METRIC_COSINE>), grid, dim3(256), lds, ctx->stream, a, tr);
        else if (metric == LANCE_HIP_DOT)
          hipLaunchKernelGGL((flat_scan_generic_kernel<METRIC_DOT>), grid, dim3(256), lds, ctx->stream, a, tr);
        else
          hipLaunchKernelGGL((flat_scan_generic_kernel<METRIC_L2>), grid, dim3(256), lds, ctx->stream, a, tr);
      }
    }
  }
  int P = 64;
  while (P < nsplit * (int)k) P <<= 1;
  hipLaunchKernelGGL(flat_merge_kernel, dim3(nq), dim3(256), (size_t)P * 12, ctx->stream, a.lkeys, a.lrids, (int)nq, (int)k, nsplit, P,
                     ids, dists);
  LH_CHECK_HIP(hipGetLastError());
  LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return LANCE_HIP_OK;
}
