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
#include <cstdlib>

#include "common.h"
#include "exact.cuh"
#include "index.h"
#include "kernels.h"
#include "search_common.cuh"

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
// H32: f16 column (operands arrive widened) -- dot products / norms in the 32-lane order (dot.rs:91-102, norm_l2.rs:60-85) and
// the scalar cosine of the Cosine trait default (cosine.rs:171-179)
template <int METRIC, bool H32 = false>
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
  if constexpr (METRIC == METRIC_COSINE) qnorm = H32 ? norm_l2_rt<float, 32>(qv, p.d) : norm_l2_rt(qv, p.d);
  for (int64_t t0 = r0; t0 < r1; t0 += tr_max) {
    const int tr = (int)min<int64_t>(tr_max, r1 - t0);
    __syncthreads();
    for (int i = threadIdx.x; i < tr * p.d; i += 256) tile[i] = p.x[t0 * p.d + i];
    for (int i = threadIdx.x; i < tr; i += 256) trid[i] = p.row_ids ? p.row_ids[t0 + i] : (uint64_t)(t0 + i);
    __syncthreads();
    if (valid) {
      for (int r = 0; r < tr; ++r) {
        float v;
        if constexpr (METRIC == METRIC_COSINE) v = H32 ? cosine_scalar32_rt(qv, qnorm, &tile[r * p.d], p.d) : cosine_exact_rt(qv, qnorm, &tile[r * p.d], p.d);
        else v = finish_metric<METRIC>(dist_exact_rt<METRIC, float, (H32 && METRIC == METRIC_DOT) ? 32 : 16>(qv, &tile[r * p.d], p.d));
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


// ---------------------------------------------------------------------------------------------------------
// v2 (fixed D, k <= 128, L2 / dot): lanes own DATABASE ROWS (row vector in VGPRs, the pairwise.hip shape,
// which streams rows at ~4 TB/s and runs the distance loop at ~70% of the non-FMA VALU rate), queries are
// staged in LDS tiles together with a per-query threshold pair T = (key, rowid).  A row whose (key, rowid)
// is <= T is appended to that query's candidate pool (wave-aggregated atomic).  The rows are visited in a
// few EPOCHS of geometrically growing size; after each epoch a per-query select kernel sorts the pool by
// (key, rowid), keeps the best k and tightens T to the k-th pair.  T is always the k-th smallest pair of a
// SUBSET of the rows, hence never below the true k-th pair: no row of the exact answer is ever filtered,
// and the final (key, rowid) sort reproduces SortExec's total order, ties included.  A pool overflow
// (possible only for adversarial row orders) is detected and repaired by re-running the scan with the
// tightened T, which strictly decreases each round.

template <int D, int METRIC, int QT, int BS, typename TX>
__global__ __launch_bounds__(BS) void flat_filter_kernel(FlatPool p) {
  __shared__ __attribute__((aligned(16))) float tile[QT * D];
  __shared__ uint32_t tk[QT];
  __shared__ uint64_t tr[QT];
  const int64_t row = p.r0 + (int64_t)blockIdx.x * BS + threadIdx.x;
  const bool valid = row < p.r1;
  RegVec<D> a;
#pragma unroll
  for (int i = 0; i < RegVec<D>::Q; ++i) a.q[i] = f4{0.f, 0.f, 0.f, 0.f};
  uint64_t rid = ~0ull;
  if (valid) {
    // the row in the column's own element type, widened per element (l2.rs:128-159 / :253-260): no f32 copy of the column
    const TX *src = static_cast<const TX *>(p.x_native) + row * D;
    if constexpr (D % 4 == 0) {
#pragma unroll
      for (int i = 0; i < D / 4; ++i) a.q[i] = load4(src + 4 * i);
    } else {
#pragma unroll
      for (int i = 0; i < D; ++i) a.q[i >> 2][i & 3] = ld_elem(src, i);
    }
    rid = p.row_ids ? p.row_ids[row] : (uint64_t)row;
  }
  const int lane = threadIdx.x & 63;
  constexpr bool NEG = METRIC != METRIC_DOT;
  for (int q0 = blockIdx.z * QT; q0 < p.nq; q0 += QT * gridDim.z) {
    const int qt = min(QT, p.nq - q0);
    __syncthreads();
    for (int i = threadIdx.x * 4; i < qt * D; i += BS * 4) {
      const f4 v = *reinterpret_cast<const f4 *>(&p.q[(int64_t)q0 * D + i]);
      *reinterpret_cast<f4 *>(&tile[i]) = NEG ? -v : v;
    }
    for (int i = threadIdx.x; i < qt; i += BS) { tk[i] = p.tkey[q0 + i]; tr[i] = p.trid[q0 + i]; }
    __syncthreads();
    for (int c = 0; c < qt; ++c) {
      const float v = finish_metric<METRIC>(dist_exact<D, METRIC, NEG>(a, &tile[c * D]));
      const uint32_t key = order_key(v);
      const uint32_t t = tk[c];
      bool pass = valid && key <= t;
      if (pass && key == t) pass = rid <= tr[c];
      const uint64_t m = __ballot(pass);
      if (m) {
        const int leader = __ffsll((long long)m) - 1;
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&p.cnt[q0 + c], (uint32_t)__popcll(m));
        base = __shfl(base, leader);
        if (pass) {
          const uint32_t pos = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
          if (pos < (uint32_t)p.cap) {
            p.pkeys[(int64_t)(q0 + c) * p.cap + pos] = key;
            p.prids[(int64_t)(q0 + c) * p.cap + pos] = rid;
          }
        }
      }
    }
  }
}

// per-query: sort the pool by (key, rowid), keep the best k, tighten T; on the last epoch emit the answer
__global__ __launch_bounds__(256) void flat_select_kernel(FlatPool p, int last, uint64_t *__restrict__ out_ids,
                                                          float *__restrict__ out_dists) {
  extern __shared__ __attribute__((aligned(16))) char ssm[];
  const int q = blockIdx.x;
  const uint32_t total = p.cnt[q];
  const int c = (int)min(total, (uint32_t)p.cap);
  if (total > (uint32_t)p.cap && threadIdx.x == 0) atomicOr(p.overflow, 1u);
  int P = 64;
  while (P < c) P <<= 1;
  uint64_t *rid = reinterpret_cast<uint64_t *>(ssm);
  uint32_t *key = reinterpret_cast<uint32_t *>(rid + P);
  uint32_t *pk = p.pkeys + (int64_t)q * p.cap;
  uint64_t *pr = p.prids + (int64_t)q * p.cap;
  for (int i = threadIdx.x; i < P; i += 256) {
    key[i] = i < c ? pk[i] : 0xFFFFFFFFu;
    rid[i] = i < c ? pr[i] : ~0ull;
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
  const int keep = min(c, p.k);
  for (int i = threadIdx.x; i < keep; i += 256) { pk[i] = key[i]; pr[i] = rid[i]; }
  if (threadIdx.x == 0) {
    p.cnt[q] = (uint32_t)keep;
    if (c >= p.k) { p.tkey[q] = key[p.k - 1]; p.trid[q] = rid[p.k - 1]; }
  }
  if (last) {
    for (int i = threadIdx.x; i < p.k; i += 256) {
      const bool ok = i < keep;
      out_ids[(int64_t)q * p.k + i] = ok ? rid[i] : ~0ull;
      out_dists[(int64_t)q * p.k + i] = ok ? key_to_float(key[i]) : INFINITY;
    }
  }
}

__global__ void flat_pool_reset_kernel(FlatPool p, int reset_t) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < p.nq) {
    p.cnt[i] = 0;
    if (reset_t) { p.tkey[i] = 0xFFFFFFFFu; p.trid[i] = ~0ull; }
  }
  if (i == 0) *p.overflow = 0;
}

// cosine_fast's y-side norm for every row, d % 16 == 0: 16 FMA lane accumulators -> f32x8 tree (cosine.rs:143-175).
// One 16-lane group per row (coalesced 64-byte reads), the tree via shuffles.  out = sqrt(y_norm).
// xb != NULL: the same pass also writes the rows' bf16 plane (round-to-nearest-even) for the long-row matrix-core filter
// (flat_mfma_wide.hip) -- the f32 column is read once, not twice.
__device__ __forceinline__ uint16_t cr_bf16_rne(float x) {
  const uint32_t u = __float_as_uint(x);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40u);
  return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__global__ __launch_bounds__(256) void cosine_rownorm_kernel(const float *__restrict__ x, int64_t n, int d, float *__restrict__ out,
                                                             uint16_t *__restrict__ xb) {
  const int lane = threadIdx.x & 63, i = lane & 15;
  const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
  float a = 0.0f;
  if (row < n) {
    const float *src = x + row * d;
    if (xb) {
      uint16_t *dst = xb + row * d;
      for (int c = 0; c < d; c += 16) { const float v = src[c + i]; a = __fmaf_rn(v, v, a); dst[c + i] = cr_bf16_rne(v); }
    } else {
      for (int c = 0; c < d; c += 16) { const float v = src[c + i]; a = __fmaf_rn(v, v, a); }
    }
  }
  const float t = a + __shfl(a, lane + 8);        // i < 8: t_i = a_i + a_{i+8}
  const float s = t + __shfl(t, lane + 4);        // i < 4: s_i = t_i + t_{i+4}
  const float u = s + __shfl(s, lane + 2);        // i < 2: u0 = s0 + s2, u1 = s1 + s3
  float y = u + __shfl(u, lane + 1);              // i = 0: (s0+s2) + (s1+s3)
  y = y + 0.0f;                                   // + reduce_sum(yn8 = 0)
  y = y + 0.0f;                                   // + norm_l2(empty tail)^2
  if (row < n && i == 0) out[row] = sqrtf(y);
}

__global__ __launch_bounds__(64) void query_norm_kernel(const float *__restrict__ q, int nq, int d, float *__restrict__ out) {
  const int qi = blockIdx.x * 64 + threadIdx.x;
  if (qi < nq) out[qi] = norm_l2_rt(q + (int64_t)qi * d, d);
}

template <int D>
static void launch_flat_filter(lance_hip_ctx *ctx, const FlatPool &a, int metric) {
  constexpr int QT = D <= 32 ? 256 : 64;
  constexpr int BS = 256;
  const int64_t rows = a.r1 - a.r0;
  const int rblocks = (int)cdiv((uint64_t)rows, BS);
  const int qtiles = (int)cdiv((uint64_t)a.nq, QT);
  int z = (int)cdiv(4ull * ctx->num_cus, (uint64_t)rblocks);
  z = std::max(1, std::min(z, qtiles));
  const dim3 grid(rblocks, 1, z);
  auto go = [&](auto tag) {
    using TX = decltype(tag);
    if (metric == METRIC_DOT)
      hipLaunchKernelGGL((flat_filter_kernel<D, METRIC_DOT, QT, BS, TX>), grid, dim3(BS), 0, ctx->stream, a);
    else
      hipLaunchKernelGGL((flat_filter_kernel<D, METRIC_L2, QT, BS, TX>), grid, dim3(BS), 0, ctx->stream, a);
  };
  if (a.x_dtype == LANCE_HIP_F16) go(__half());
  else if (a.x_dtype == LANCE_HIP_I8) go(int8_t());
  else go(float());
}

constexpr int FLAT_CAP = 4096;      // pool entries per query
constexpr int FLAT_QCHUNK = 2048;   // queries per pass over the rows (pool = 2048 x 4096 x 12 B = 100 MB)

static bool flat_fixed_dim(uint32_t d) { return d == 8 || d == 16 || d == 32 || d == 64 || d == 96 || d == 128; }

static bool flat_v2_supported(int metric, uint32_t d, uint32_t k) {
  if (k > 128) return false;
  if (metric == LANCE_HIP_COSINE) return d % 16 == 0 && d >= 32;     // d = 8 / 16 take cosine_once (cosine.rs:233-238)
  return true;
}

// x: the rows as f32, or NULL when flat_reads_native() said the kernels take the column (x_native, x_dtype) as it is
static int flat_topk_v2(lance_hip_ctx *ctx, int metric, const float *x, const void *x_native, int x_dtype, const uint64_t *row_ids, int64_t n,
                        int d, const float *q, int nq, int k, uint64_t *ids, float *dists) {
  const int qch = std::min(nq, FLAT_QCHUNK);
  FlatPool a;
  a.x = x; a.x_native = x ? static_cast<const void *>(x) : x_native; a.x_dtype = x ? LANCE_HIP_F32 : x_dtype;
  a.row_ids = row_ids; a.k = k; a.cap = FLAT_CAP;
  a.tkey = ctx->scratch_t<uint32_t>("flat2.tkey", qch);
  a.trid = ctx->scratch_t<uint64_t>("flat2.trid", qch);
  a.cnt = ctx->scratch_t<uint32_t>("flat2.cnt", qch);
  a.pkeys = ctx->scratch_t<uint32_t>("flat2.pkeys", (size_t)qch * FLAT_CAP);
  a.prids = ctx->scratch_t<uint64_t>("flat2.prids", (size_t)qch * FLAT_CAP);
  a.overflow = ctx->scratch_t<uint32_t>("flat2.ovf", 1);
  if (!a.tkey || !a.trid || !a.cnt || !a.pkeys || !a.prids || !a.overflow) return LANCE_HIP_ENOMEM;
  const int64_t growth = std::max<int64_t>(2, FLAT_CAP / (16 * (int64_t)k));
  const bool fixed = metric != LANCE_HIP_COSINE && flat_fixed_dim((uint32_t)d) &&
                     (((reinterpret_cast<uintptr_t>(a.x_native) | reinterpret_cast<uintptr_t>(q)) & 15) == 0);
  LH_REQUIRE(x || fixed, "flat_topk: internal: native rows need the fixed-dimension kernels");
  // long rows (no fixed-dimension kernel): query batches run the epochs after the first on the matrix cores (flat_mfma_wide.hip);
  // the rows' bf16 plane and norms are made once per call (cosine: by the row-norm pass below, which reads the column anyway)
  bool wide_mfma = !fixed && flat_mfma_wide_supported(metric, d, std::min(nq, qch), n, x, q);
  const uint16_t *wxb = nullptr;
  const float *wxn2 = nullptr;
  // The plane is as large as half the column (exact size, no arena headroom).  If the device cannot give it the scan runs on the exact
  // kernel, as before the matrix-core route existed (the call used to fail with ENOMEM on columns near device capacity);
  // a plane above 4 GiB goes back on EVERY way out of this function, error paths included (up to 4 GiB it stays: a repeated call does
  // not pay the allocation).
  struct PlaneGuard {
    lance_hip_ctx *c; bool armed;
    ~PlaneGuard() { if (armed) { (void)hipStreamSynchronize(c->stream); c->scratch_release("fw.xb"); } }
  } plane_guard{ctx, false};
  const bool big_plane = (uint64_t)n * (uint64_t)d * 2 > (4ull << 30);
  if (metric == LANCE_HIP_COSINE) {
    float *sy = ctx->scratch_t<float>("flat2.row_sy", (size_t)std::max<int64_t>(n, 1));
    float *qn = ctx->scratch_t<float>("flat2.q_norm", (size_t)nq);
    uint16_t *xb = wide_mfma ? static_cast<uint16_t *>(ctx->scratch_exact("fw.xb", (size_t)std::max<int64_t>(n, 1) * d * 2)) : nullptr;
    if (wide_mfma && !xb) wide_mfma = false;
    plane_guard.armed = wide_mfma && big_plane;
    if (!sy || !qn) return LANCE_HIP_ENOMEM;
    if (n > 0) hipLaunchKernelGGL(cosine_rownorm_kernel, dim3((unsigned)cdiv((uint64_t)n * 16, 256)), dim3(256), 0, ctx->stream, x, n, d, sy, xb);
    hipLaunchKernelGGL(query_norm_kernel, dim3(cdiv(nq, 64)), dim3(64), 0, ctx->stream, q, nq, d, qn);
    a.row_sy = sy;
    wxb = xb; wxn2 = sy;      // (the cosine filter reads row_sy, not |x|^2)
  } else if (wide_mfma) {
    const int prc = flat_mfma_wide_prepare_rows(ctx, x, n, d, &wxb, &wxn2);
    if (prc == LH_NOT_TAKEN) wide_mfma = false;
    else if (prc != LANCE_HIP_OK) return prc;
    plane_guard.armed = wide_mfma && big_plane;
  }
  const float *qn_all = metric == LANCE_HIP_COSINE ? ctx->scratch_t<float>("flat2.q_norm", (size_t)nq) : nullptr;
  auto filter = [&](const FlatPool &e) {
    if (!fixed) { launch_wide_filter(ctx, e, d, metric); return; }
    switch (d) {
      case 8: launch_flat_filter<8>(ctx, e, metric); break;
      case 16: launch_flat_filter<16>(ctx, e, metric); break;
      case 32: launch_flat_filter<32>(ctx, e, metric); break;
      case 64: launch_flat_filter<64>(ctx, e, metric); break;
      case 96: launch_flat_filter<96>(ctx, e, metric); break;
      default: launch_flat_filter<128>(ctx, e, metric); break;
    }
  };
  const size_t sel_lds = (size_t)FLAT_CAP * 12;
  for (int qc0 = 0; qc0 < nq; qc0 += qch) {
    a.q = q + (int64_t)qc0 * d;
    a.q_norm = qn_all ? qn_all + qc0 : nullptr;
    a.nq = std::min(qch, nq - qc0);
    uint64_t *oid = ids + (int64_t)qc0 * k;
    float *od = dists + (int64_t)qc0 * k;
    hipLaunchKernelGGL(flat_pool_reset_kernel, dim3(cdiv(a.nq, 256)), dim3(256), 0, ctx->stream, a, 1);
    // query batches: after the first epoch (which gives every query a threshold) the epochs run on the matrix cores
    const bool use_mfma = fixed && flat_mfma_supported(metric, d, a.nq, a.x_native, a.q);
    const uint16_t *qhi = nullptr, *qlo = nullptr;
    const float *qn2 = nullptr;
    if (use_mfma) LH_TRY(flat_mfma_prepare(ctx, a.q, a.nq, d, &qhi, &qlo, &qn2));
    const bool use_wide = wide_mfma && flat_mfma_wide_supported(metric, d, a.nq, n, x, a.q);      // (a short last chunk takes the exact kernel)
    const uint16_t *wqb = nullptr;
    const float *wqn2 = nullptr;
    if (use_wide) LH_TRY(flat_mfma_wide_prepare_queries(ctx, a.q, a.nq, d, &wqb, &wqn2));
    int64_t seen = 0;
    if (n == 0) {
      a.r0 = a.r1 = 0;
      hipLaunchKernelGGL(flat_select_kernel, dim3(a.nq), dim3(256), sel_lds, ctx->stream, a, 1, oid, od);
    }
    while (seen < n) {
      const int64_t want = seen == 0 ? FLAT_CAP / 2 : seen * (growth - 1);
      a.r0 = seen;
      a.r1 = std::min<int64_t>(n, seen + std::max<int64_t>(want, 1));
      if (n - a.r1 < (a.r1 - a.r0) / 4) a.r1 = n;            // do not leave a small tail epoch
      {
        ScopedTimer t(ctx, "flat_scan");
        if (use_mfma && seen > 0 && seen >= k) LH_TRY(launch_flat_filter_mfma(ctx, a, d, metric, qhi, qlo, qn2));
        else if (use_wide && seen > 0 && seen >= k) LH_TRY(launch_flat_filter_mfma_wide(ctx, a, d, metric, wxb, wxn2, wqb, wqn2));
        else filter(a);
      }
      seen = a.r1;
      hipLaunchKernelGGL(flat_select_kernel, dim3(a.nq), dim3(256), sel_lds, ctx->stream, a, seen == n ? 1 : 0, oid, od);
    }
    // overflow repair: rows <= T (T only ever tightens) are re-collected from scratch until every pool fits
    for (int round = 0; round < 64; ++round) {
      uint32_t ovf = 0;
      LH_CHECK_HIP(hipMemcpyAsync(&ovf, a.overflow, 4, hipMemcpyDeviceToHost, ctx->stream));
      LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
      if (!ovf) break;
      LH_REQUIRE(round < 63, "flat_topk: candidate pool did not converge");
      hipLaunchKernelGGL(flat_pool_reset_kernel, dim3(cdiv(a.nq, 256)), dim3(256), 0, ctx->stream, a, 0);
      a.r0 = 0; a.r1 = n;
      filter(a);
      hipLaunchKernelGGL(flat_select_kernel, dim3(a.nq), dim3(256), sel_lds, ctx->stream, a, 1, oid, od);
    }
  }
  LH_CHECK_HIP(hipGetLastError());
  return LANCE_HIP_OK;      // (plane_guard gives a > 4 GiB plane back after waiting for the stream)
}

template <int D>
static void launch_flat_fixed(lance_hip_ctx *ctx, const FlatArgs &a, int metric, dim3 grid) {
  constexpr int TR = (8192 / D) > 256 ? 256 : (8192 / D);
  if (metric == METRIC_DOT)
    hipLaunchKernelGGL((flat_scan_kernel<D, METRIC_DOT, TR>), grid, dim3(256), 0, ctx->stream, a);
  else
    hipLaunchKernelGGL((flat_scan_kernel<D, METRIC_L2, TR>), grid, dim3(256), 0, ctx->stream, a);
}


// ---------------------------------------------------------------------------------------------------------
// IVF_FLAT (IvfSubIndex = FlatIndex over FlatFloatStorage): find_partitions, then an exact scan of the raw
// vectors of every probed partition and the (dist, rowid) merge.
//   FlatIndex::search            lance-index/src/vector/flat/index.rs:82-177
//   FlatDistanceCal::distance_all  flat/storage.rs:345-402  (distance_type.func()(query, vector): a1 / a2 order)
//   SortExec merge               lance/src/dataset/scanner.rs:3440-3468
// One workgroup per (query, probe): the query sits in LDS (negated for L2), lanes own rows of the partition.
// Pass 0 (bound) walks each query's nearest partition and takes the k-th smallest of the 256 lane minima as the
// query's threshold; pass 1 walks all probed partitions and appends rows under the threshold to the query's
// pool; flat_select_kernel sorts the pool by (dist, rowid).  Same pool / overflow-repair machinery as flat v2.
struct IvfFlatArgs {
  const float *vec;              // [n][d] partition-ordered
  const uint64_t *row_ids;       // [n]
  const uint32_t *part_offsets;  // [nlist+1]
  const uint32_t *probes;        // [nq][nprobes]
  const float *q;                // [nq][d]
  int nprobes, d, bound;
  uint32_t *ppart;               // [nq][cap] partition of every pool entry (boundary-tie check)
  const int2_host *items;        // partition-major pass: (partition, first row) per 256-row block
  const uint32_t *pair_starts;   // [nlist+1] (query, probe) pairs grouped by partition
  const uint32_t *pair_idx;      // [nq*nprobes] pair index (query = idx / pdiv), grouped
  int pdiv;                      // nprobes for the main pass, 1 for the bound pass (pairs = queries)
  uint32_t *lanemin;             // bound pass: [nq][256] running minimum key per (query, lane)
  uint32_t *flags;               // [nq] 1 = the survivors of a boundary tie depend on the reference heap: replay
  const uint32_t *allow = nullptr;   // prefilter: one bit per storage position (flat/index.rs:129-165), NULL = none
  FlatPool pool;
};

// k-th smallest (rank kk) of one u32 per lane of a 256-lane workgroup
__device__ __forceinline__ uint32_t kth_smallest_256(uint32_t v, int kk, uint32_t *sorted, uint32_t *slot) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k2 = 2; k2 <= 64; k2 <<= 1) {
#pragma unroll
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      const uint32_t o = __shfl_xor(v, j, 64);
      const bool up = (lane & k2) == 0, lower = (lane & j) == 0;
      v = (lower == up) ? min(v, o) : max(v, o);
    }
  }
  sorted[threadIdx.x] = v;
  __syncthreads();
  int rank = lane;
  for (int w = 0; w < 4; ++w) {
    if (w == wave) continue;
    const uint32_t *run = sorted + w * 64;
    int lo = 0, hi = 64;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      const bool before = w < wave ? run[mid] <= v : run[mid] < v;
      if (before) lo = mid + 1; else hi = mid;
    }
    rank += lo;
  }
  if (rank == kk) *slot = v;
  __syncthreads();
  return *slot;
}

// distance of one stored row as FlatDistanceCal sees it (flat/storage.rs:345-402: distance_type.func()(query, vector)).  Cosine:
// f32::cosine = norm_l2(query) once, then cosine_fast per row (cosine.rs:36-39,143-175) -- the rows of a cosine IVF_FLAT index
// are stored normalised (IvfTransformer::new_flat, ivf.rs:147-160) and the query arrives normalised (knn.rs:498), but the
// distance function is still the full cosine.
// H32: the rows are widened f16 values -- FlatDistanceCal over a Float16 column calls f16's dot / cosine (32-lane dot_scalar and
// norm_l2_impl, the scalar cosine; dot.rs:91-102, norm_l2.rs:60-85, cosine.rs:171-179); L2 stays the 16-lane l2_scalar.
template <int METRIC, bool H32 = false>
__device__ __forceinline__ float ivfflat_dist_rt(const float *__restrict__ q, float qnorm, const float *__restrict__ row, int d) {
  if constexpr (METRIC == METRIC_COSINE) return H32 ? cosine_scalar32_rt(q, qnorm, row, d) : cosine_exact_rt(q, qnorm, row, d);
  else return finish_metric<METRIC>(dist_exact_rt<METRIC, float, (H32 && METRIC == METRIC_DOT) ? 32 : 16>(q, row, d));
}

template <int D, int METRIC, bool H32 = false>   // D = 0: run-time dimension (H32 only there)
__global__ __launch_bounds__(256) void ivfflat_kernel(IvfFlatArgs a) {
  static_assert(!H32 || D == 0, "the f16 orders run on the run-time-dimension kernel");
  extern __shared__ __attribute__((aligned(16))) float qt[];   // [d padded to 4]
  __shared__ uint32_t sorted[256];
  __shared__ uint32_t slot;
  const int qi = a.bound ? blockIdx.x : blockIdx.x / a.nprobes;
  const int pr = a.bound ? 0 : blockIdx.x % a.nprobes;
  const uint32_t part = a.probes[(int64_t)qi * a.nprobes + pr];
  const uint32_t r0 = a.part_offsets[part], r1 = a.part_offsets[part + 1];
  constexpr bool NEG = METRIC != METRIC_DOT && D != 0;
  for (int i = threadIdx.x; i < a.d; i += 256) {
    const float v = a.q[(int64_t)qi * a.d + i];
    qt[i] = NEG ? -v : v;
  }
  const uint32_t tk = a.bound ? 0xFFFFFFFFu : a.pool.tkey[qi];
  const uint64_t tr = ~0ull;   // every row tied with the bound stays in the pool: the tie check below needs all of them
  __syncthreads();
  float qnorm = 0.0f;
  if constexpr (METRIC == METRIC_COSINE) qnorm = H32 ? norm_l2_rt<float, 32>(qt, a.d) : norm_l2_rt(qt, a.d);
  uint32_t mn = 0xFFFFFFFFu;
  for (uint32_t base = r0; base < r1; base += 256) {
    const uint32_t row = base + threadIdx.x;
    if (row < r1 && row_allowed(a.allow, row)) {
      float v;
      if constexpr (D != 0) {
        RegVec<D> rv;
        const float *src = a.vec + (int64_t)row * D;
#pragma unroll
        for (int i = 0; i < D / 4; ++i) rv.q[i] = *reinterpret_cast<const f4 *>(src + 4 * i);
        v = finish_metric<METRIC>(dist_exact<D, METRIC, NEG>(rv, qt));
      } else {
        v = ivfflat_dist_rt<METRIC, H32>(qt, qnorm, a.vec + (int64_t)row * a.d, a.d);
      }
      const uint32_t key = order_key(v);
      if (a.bound) {
        mn = min(mn, key);
      } else {
        const uint64_t rid = a.row_ids[row];
        if (key < tk || (key == tk && rid <= tr)) {
          const uint32_t pos = atomicAdd(&a.pool.cnt[qi], 1u);
          if (pos < (uint32_t)a.pool.cap) {
            a.pool.pkeys[(int64_t)qi * a.pool.cap + pos] = key;
            a.pool.prids[(int64_t)qi * a.pool.cap + pos] = rid;
            a.ppart[(int64_t)qi * a.pool.cap + pos] = part;
          }
        }
      }
    }
  }
  if (a.bound) {
    if (threadIdx.x == 0) slot = 0xFFFFFFFFu;
    __syncthreads();
    const uint32_t t = kth_smallest_256(mn, a.pool.k - 1, sorted, &slot);
    if (threadIdx.x == 0) a.pool.tkey[qi] = t;     // 0xFFFFFFFF when the partition has fewer than k rows
  }
}


// per-query: sort the pool by (key, rowid), emit the best k, tighten T (repair rounds), and detect the one case
// the (dist, rowid) order cannot decide: more than k rows of ONE partition at or below the k-th distance with a tie
// at the boundary -- there FlatIndex's BinaryHeap (flat/index.rs:113-123) decides which tied rows survive.
__global__ __launch_bounds__(256) void ivfflat_select_kernel(IvfFlatArgs a, uint64_t *__restrict__ out_ids, float *__restrict__ out_dists) {
  extern __shared__ __attribute__((aligned(16))) char ssm[];
  const FlatPool &p = a.pool;
  const int q = blockIdx.x;
  const uint32_t total = p.cnt[q];
  const int c = (int)min(total, (uint32_t)p.cap);
  if (total > (uint32_t)p.cap && threadIdx.x == 0) atomicOr(p.overflow, 1u);
  int P = 64;
  while (P < c) P <<= 1;
  uint64_t *rid = reinterpret_cast<uint64_t *>(ssm);
  uint32_t *key = reinterpret_cast<uint32_t *>(rid + P);
  uint32_t *prt = key + P;
  __shared__ int s_amb;
  if (threadIdx.x == 0) s_amb = 0;
  for (int i = threadIdx.x; i < P; i += 256) {
    key[i] = i < c ? p.pkeys[(int64_t)q * p.cap + i] : 0xFFFFFFFFu;
    rid[i] = i < c ? p.prids[(int64_t)q * p.cap + i] : ~0ull;
    prt[i] = i < c ? a.ppart[(int64_t)q * p.cap + i] : 0u;
  }
  __syncthreads();
  bitonic_sort_kr<256>(key, rid, prt, P);
  if (threadIdx.x == 0 && c >= p.k) { p.tkey[q] = key[p.k - 1]; p.trid[q] = ~0ull; }
  if (c > p.k && key[p.k] == key[p.k - 1]) {
    const uint32_t tf = key[p.k - 1];
    int lo = p.k, hi = c;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (key[mid] <= tf) lo = mid + 1; else hi = mid; }
    const int L = lo;
    for (int i = threadIdx.x; i < L; i += 256) {
      int same = 0;
      for (int j = 0; j < L; ++j) same += prt[j] == prt[i] ? 1 : 0;
      if (same > p.k) s_amb = 1;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) a.flags[q] = (uint32_t)s_amb;
  const int keep = min(c, p.k);
  for (int i = threadIdx.x; i < p.k; i += 256) {
    const bool ok = i < keep;
    out_ids[(int64_t)q * p.k + i] = ok ? rid[i] : ~0ull;
    out_dists[(int64_t)q * p.k + i] = ok ? key_to_float(key[i]) : INFINITY;
  }
}

// Exact replay of a flagged query: every probed partition through a max-heap of k with std BinaryHeap semantics
// (push while len < k, else replace the root only if root.dist > dist), rows in scan order; distances by all 64
// lanes, a ballot drops rows that cannot enter, lane 0 replays the rest; partition heaps are merged by (dist, rowid).
template <int METRIC, bool H32 = false>
__global__ __launch_bounds__(64) void ivfflat_exact_kernel(IvfFlatArgs a, uint64_t *__restrict__ out_ids, float *__restrict__ out_dists) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int qi = blockIdx.x;
  if (!a.flags[qi]) return;
  const int lane = threadIdx.x, k = a.pool.k;
  const int dpad = (a.d + 3) & ~3;
  float *qv = reinterpret_cast<float *>(smem);
  uint64_t *trid = reinterpret_cast<uint64_t *>(qv + dpad + (dpad & 1));
  uint32_t *tkey = reinterpret_cast<uint32_t *>(trid + k);
  uint32_t *hk = tkey + k;
  uint32_t *hp = hk + k + 1;
  uint32_t *skey = hp + k + 1;
  __shared__ int s_hlen, s_tcnt;
  if (lane == 0) { s_hlen = 0; s_tcnt = 0; }
  for (int t = lane; t < a.d; t += 64) qv[t] = a.q[(int64_t)qi * a.d + t];
  __syncthreads();
  float qnorm = 0.0f;
  if constexpr (METRIC == METRIC_COSINE) qnorm = H32 ? norm_l2_rt<float, 32>(qv, a.d) : norm_l2_rt(qv, a.d);
  for (int pi = 0; pi < a.nprobes; ++pi) {
    const uint32_t part = a.probes[(int64_t)qi * a.nprobes + pi];
    const uint32_t off = a.part_offsets[part];
    const int np = (int)(a.part_offsets[part + 1] - off);
    if (np == 0) continue;
    if (lane == 0) s_hlen = 0;
    __syncthreads();
    for (int base = 0; base < np; base += 64) {
      const int row = base + lane;
      uint32_t key = 0xFFFFFFFFu;
      bool cand = false;
      if (row < np && row_allowed(a.allow, off + (uint32_t)row)) {
        key = order_key(ivfflat_dist_rt<METRIC, H32>(qv, qnorm, a.vec + (int64_t)(off + row) * a.d, a.d));
        cand = s_hlen < k || key < hk[0];
      }
      const uint64_t mask = __ballot(cand);
      skey[lane] = key;
      __syncthreads();
      if (lane == 0 && mask) {
        int hl = s_hlen;
        uint64_t mm = mask;
        while (mm) {
          const int b = __ffsll((long long)mm) - 1;
          mm &= mm - 1;
          const uint32_t kk = skey[b];
          if (hl < k) {
            heap_push(hk, hp, hl, kk, off + (uint32_t)(base + b));
          } else if (hk[0] > kk) {
            heap_pop(hk, hp, hl);
            heap_push(hk, hp, hl, kk, off + (uint32_t)(base + b));
          }
        }
        s_hlen = hl;
      }
      __syncthreads();
    }
    if (lane == 0) {
      int tc = s_tcnt;
      for (int i = 0; i < s_hlen; ++i) {
        const uint32_t kk = hk[i];
        const uint64_t rr = a.row_ids[hp[i]];
        if (tc == k) {
          const uint32_t wk = tkey[tc - 1];
          const uint64_t wr = trid[tc - 1];
          if (!(kk < wk || (kk == wk && rr < wr))) continue;
        }
        int pos = tc < k ? tc : k - 1;
        while (pos > 0) {
          const uint32_t pk = tkey[pos - 1];
          const uint64_t pr = trid[pos - 1];
          if (pk < kk || (pk == kk && pr < rr)) break;
          tkey[pos] = pk; trid[pos] = pr;
          --pos;
        }
        tkey[pos] = kk; trid[pos] = rr;
        if (tc < k) ++tc;
      }
      s_tcnt = tc;
    }
    __syncthreads();
  }
  const int got = s_tcnt;
  for (int i = lane; i < k; i += 64) {
    out_ids[(int64_t)qi * k + i] = i < got ? trid[i] : ~0ull;
    out_dists[(int64_t)qi * k + i] = i < got ? key_to_float(tkey[i]) : INFINITY;
  }
}

__global__ __launch_bounds__(256) void gather_vectors_kernel(const float *__restrict__ x, const uint64_t *__restrict__ row_ids,
                                                             const uint32_t *__restrict__ perm, int64_t n_out, int d,
                                                             float *__restrict__ out, uint64_t *__restrict__ rid_out) {
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= n_out * d) return;
  const int64_t s = g / d;
  const uint32_t r = perm[s];
  out[g] = x[(int64_t)r * d + (g - s * d)];
  if (g == s * d) rid_out[s] = row_ids ? row_ids[r] : (uint64_t)r;
}


// Partition-major main pass (fixed D): one workgroup per 256-row block of a partition.  Lanes own the rows (vector
// in VGPRs, loaded once); the queries that probe this partition -- found by grouping the (query, probe) pairs by
// partition with the stable counting sort -- stream through LDS tiles with their thresholds.  Rows are read once
// per block instead of once per (query, probe) pair: the query-major kernel moved 40 GB through L2 per 2000-query
// batch at C2, this one moves the 512 MB of vectors once.
template <int D, int METRIC, int QT, bool BOUND>
__global__ __launch_bounds__(256) void ivfflat_pm_kernel(IvfFlatArgs a) {
  __shared__ __attribute__((aligned(16))) float tile[QT * D];
  __shared__ uint32_t tk[QT];
  __shared__ int tq[QT];
  const int part = a.items[blockIdx.x].x;
  const uint32_t row = (uint32_t)a.items[blockIdx.x].y + threadIdx.x;
  const bool valid = row < a.part_offsets[part + 1] && row_allowed(a.allow, row);
  const uint32_t qs = a.pair_starts[part], qe = a.pair_starts[part + 1];
  if (qs == qe) return;
  RegVec<D> rv;
#pragma unroll
  for (int i = 0; i < RegVec<D>::Q; ++i) rv.q[i] = f4{0.f, 0.f, 0.f, 0.f};
  uint64_t rid = ~0ull;
  if (valid) {
    const float *src = a.vec + (int64_t)row * D;
#pragma unroll
    for (int i = 0; i < D / 4; ++i) rv.q[i] = *reinterpret_cast<const f4 *>(src + 4 * i);
    rid = a.row_ids[row];
  }
  const int lane = threadIdx.x & 63;
  constexpr bool NEG = METRIC != METRIC_DOT;
  for (uint32_t j0 = qs; j0 < qe; j0 += QT) {
    const int qt = (int)min((uint32_t)QT, qe - j0);
    __syncthreads();
    for (int j = threadIdx.x; j < qt; j += 256) {
      const int qi = (int)(a.pair_idx[j0 + j] / (uint32_t)a.pdiv);
      tq[j] = qi;
      tk[j] = BOUND ? 0xFFFFFFFFu : a.pool.tkey[qi];
    }
    __syncthreads();
    for (int i = threadIdx.x * 4; i < qt * D; i += 256 * 4) {
      const int j = i / D, e = i - j * D;
      const f4 v = *reinterpret_cast<const f4 *>(&a.q[(int64_t)tq[j] * D + e]);
      *reinterpret_cast<f4 *>(&tile[i]) = NEG ? -v : v;
    }
    __syncthreads();
    for (int c = 0; c < qt; ++c) {
      const float v = finish_metric<METRIC>(dist_exact<D, METRIC, NEG>(rv, &tile[c * D]));
      const uint32_t key = order_key(v);
      if constexpr (BOUND) {
        // lane l of every block of this partition folds its row into lanemin[q][l]: 256 minima over distinct rows,
        // whose k-th smallest (ivfflat_kth_kernel) is the query's bound
        if (valid) atomicMin(&a.lanemin[(int64_t)tq[c] * 256 + threadIdx.x], key);
        continue;
      }
      const bool pass = valid && key <= tk[c];       // ties with the bound are kept (the boundary-tie check needs them)
      const uint64_t m = __ballot(pass);
      if (m) {
        const int qi = tq[c];
        const int leader = __ffsll((long long)m) - 1;
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&a.pool.cnt[qi], (uint32_t)__popcll(m));
        base = __shfl(base, leader);
        if (pass) {
          const uint32_t pos = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
          if (pos < (uint32_t)a.pool.cap) {
            a.pool.pkeys[(int64_t)qi * a.pool.cap + pos] = key;
            a.pool.prids[(int64_t)qi * a.pool.cap + pos] = rid;
            a.ppart[(int64_t)qi * a.pool.cap + pos] = (uint32_t)part;
          }
        }
      }
    }
  }
}

template <int METRIC, bool BOUND>
static bool launch_ivfflat_pm(lance_hip_ctx *ctx, const IvfFlatArgs &a, unsigned n_items) {
  switch (a.d) {
    case 8: hipLaunchKernelGGL((ivfflat_pm_kernel<8, METRIC, 256, BOUND>), dim3(n_items), dim3(256), 0, ctx->stream, a); return true;
    case 16: hipLaunchKernelGGL((ivfflat_pm_kernel<16, METRIC, 256, BOUND>), dim3(n_items), dim3(256), 0, ctx->stream, a); return true;
    case 32: hipLaunchKernelGGL((ivfflat_pm_kernel<32, METRIC, 256, BOUND>), dim3(n_items), dim3(256), 0, ctx->stream, a); return true;
    case 64: hipLaunchKernelGGL((ivfflat_pm_kernel<64, METRIC, 64, BOUND>), dim3(n_items), dim3(256), 0, ctx->stream, a); return true;
    case 96: hipLaunchKernelGGL((ivfflat_pm_kernel<96, METRIC, 64, BOUND>), dim3(n_items), dim3(256), 0, ctx->stream, a); return true;
    case 128: hipLaunchKernelGGL((ivfflat_pm_kernel<128, METRIC, 64, BOUND>), dim3(n_items), dim3(256), 0, ctx->stream, a); return true;
    default: return false;
  }
}

__global__ __launch_bounds__(256) void ivfflat_first_probe_kernel(const uint32_t *__restrict__ probes, int nq, int nprobes, uint32_t *__restrict__ keys,
                                                                  uint32_t *__restrict__ lanemin) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < nq) keys[i] = probes[(int64_t)i * nprobes];
  for (int64_t j = i; j < (int64_t)nq * 256; j += (int64_t)gridDim.x * 256) lanemin[j] = 0xFFFFFFFFu;
}

__global__ __launch_bounds__(256) void ivfflat_kth_kernel(const uint32_t *__restrict__ lanemin, int k, uint32_t *__restrict__ tkey) {
  __shared__ uint32_t sorted[256];
  __shared__ uint32_t slot;
  if (threadIdx.x == 0) slot = 0xFFFFFFFFu;
  __syncthreads();
  const uint32_t t = kth_smallest_256(lanemin[(int64_t)blockIdx.x * 256 + threadIdx.x], k - 1, sorted, &slot);
  if (threadIdx.x == 0) tkey[blockIdx.x] = t;
}

template <int METRIC>
static void launch_ivfflat(lance_hip_ctx *ctx, const IvfFlatArgs &a, unsigned grid, bool fixed, bool h32 = false) {
  const size_t lds = (size_t)((a.d + 3) & ~3) * 4;
  if constexpr (METRIC != METRIC_L2) if (h32) {   // f16 column under dot / cosine: the 32-lane orders, run-time dimension
    hipLaunchKernelGGL((ivfflat_kernel<0, METRIC, true>), dim3(grid), dim3(256), lds, ctx->stream, a);
    return;
  }
  if constexpr (METRIC != METRIC_COSINE) if (fixed) {   // cosine: run-time dimension kernel only (cosine_fast has its own lane layout)
    switch (a.d) {
      case 8: hipLaunchKernelGGL((ivfflat_kernel<8, METRIC>), dim3(grid), dim3(256), lds, ctx->stream, a); return;
      case 16: hipLaunchKernelGGL((ivfflat_kernel<16, METRIC>), dim3(grid), dim3(256), lds, ctx->stream, a); return;
      case 32: hipLaunchKernelGGL((ivfflat_kernel<32, METRIC>), dim3(grid), dim3(256), lds, ctx->stream, a); return;
      case 64: hipLaunchKernelGGL((ivfflat_kernel<64, METRIC>), dim3(grid), dim3(256), lds, ctx->stream, a); return;
      case 96: hipLaunchKernelGGL((ivfflat_kernel<96, METRIC>), dim3(grid), dim3(256), lds, ctx->stream, a); return;
      case 128: hipLaunchKernelGGL((ivfflat_kernel<128, METRIC>), dim3(grid), dim3(256), lds, ctx->stream, a); return;
      default: break;
    }
  }
  hipLaunchKernelGGL((ivfflat_kernel<0, METRIC>), dim3(grid), dim3(256), lds, ctx->stream, a);
}

}  // namespace lh

using namespace lh;

extern "C" int lance_hip_ivfflat_create(lance_hip_ctx *ctx, int dtype, int metric, uint32_t d, const void *centroids, uint32_t nlist,
                                        const void *x, const uint32_t *part_ids, const uint64_t *row_ids, uint64_t n,
                                        lance_hip_index **out) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(ctx && centroids && out && (n == 0 || (x && part_ids)), "ivfflat_create: NULL argument");
  LH_TRY(check_dtype(dtype, "ivfflat_create"));
  LH_REQUIRE(metric == LANCE_HIP_L2 || metric == LANCE_HIP_DOT || metric == LANCE_HIP_COSINE, "ivfflat_create: bad metric %d", metric);
  LH_REQUIRE(!(dtype == LANCE_HIP_I8 && metric == LANCE_HIP_COSINE), "ivfflat_create: int8 cosine is not supported (no normalised int8 rows)");
  LH_REQUIRE(nlist > 0 && nlist <= 65536 && d > 0, "ivfflat_create: nlist=%u / d=%u not supported", nlist, d);
  LH_REQUIRE(n < (1ull << 32), "ivfflat_create: n too large for this version");
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  auto *ix = new lance_hip_index();
  ix->device = ctx->device; ix->metric = metric; ix->dtype = dtype; ix->d = d; ix->nlist = nlist; ix->m = 0;
  auto fail = [&](int r) { delete ix; return r; };
  if (hipMalloc(reinterpret_cast<void **>(&ix->centroids), (size_t)nlist * d * 4) != hipSuccess) return fail(LANCE_HIP_ENOMEM);
  if (hipMalloc(reinterpret_cast<void **>(&ix->part_offsets), (size_t)(nlist + 1) * 4) != hipSuccess) return fail(LANCE_HIP_ENOMEM);
  int r = widen_into(ctx, model_dtype(dtype), centroids, (size_t)nlist * d, ix->centroids);
  if (r != LANCE_HIP_OK) return fail(r);
  uint32_t *perm = ctx->scratch_t<uint32_t>("index.perm", (size_t)(n ? n : 1));
  if (!perm) return fail(LANCE_HIP_ENOMEM);
  r = stable_group(ctx, part_ids, (int64_t)n, (int64_t)n, (int)nlist, 1, ix->part_offsets, perm, (int64_t)n, nullptr);
  if (r != LANCE_HIP_OK) return fail(r);
  ix->part_offsets_h.resize(nlist + 1);
  if (hipMemcpyAsync(ix->part_offsets_h.data(), ix->part_offsets, (size_t)(nlist + 1) * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess) { set_error("ivfflat_create: HIP failure"); return fail(LANCE_HIP_ERUNTIME); }
  ix->n = ix->part_offsets_h[nlist];       // rows with part id NONE (non-finite vectors) are dropped
  for (uint32_t p = 0; p < nlist; ++p) ix->max_part = std::max(ix->max_part, ix->part_offsets_h[p + 1] - ix->part_offsets_h[p]);
  if (hipMalloc(reinterpret_cast<void **>(&ix->vectors), std::max<size_t>((size_t)ix->n * d * 4, 16)) != hipSuccess) return fail(LANCE_HIP_ENOMEM);
  if (hipMalloc(reinterpret_cast<void **>(&ix->row_ids), std::max<size_t>((size_t)ix->n * 8, 16)) != hipSuccess) return fail(LANCE_HIP_ENOMEM);
  const float *xf;
  r = as_f32(ctx, dtype, x, (size_t)n * d, "f16.x", &xf);
  if (r != LANCE_HIP_OK) return fail(r);
  if (ix->n > 0) {
    hipLaunchKernelGGL(gather_vectors_kernel, dim3((unsigned)cdiv((uint64_t)ix->n * d, 256)), dim3(256), 0, ctx->stream, xf, row_ids, perm,
                       (int64_t)ix->n, (int)d, ix->vectors, ix->row_ids);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
      set_error("ivfflat_create: gather kernel failed");
      return fail(LANCE_HIP_ERUNTIME);
    }
  }
  {  // 256-row blocks of every partition for the partition-major pass
    std::vector<int2_host> items;
    for (uint32_t p = 0; p < nlist; ++p)
      for (uint32_t r0 = ix->part_offsets_h[p]; r0 < ix->part_offsets_h[p + 1]; r0 += 256) items.push_back(int2_host{(int)p, (int)r0});
    ix->n_flat_items = (uint32_t)items.size();
    if (hipMalloc(reinterpret_cast<void **>(&ix->flat_items), std::max<size_t>(items.size() * sizeof(int2_host), 16)) != hipSuccess) return fail(LANCE_HIP_ENOMEM);
    if (!items.empty() && hipMemcpy(ix->flat_items, items.data(), items.size() * sizeof(int2_host), hipMemcpyHostToDevice) != hipSuccess) {
      set_error("ivfflat_create: HIP failure");
      return fail(LANCE_HIP_ERUNTIME);
    }
  }
  *out = ix;
  return LANCE_HIP_OK;
}

// More rows tie at a query's bound than its pool holds (thousands of duplicate vectors, or distances that are all NaN): the
// threshold cannot separate them, so the repair rounds never fit.  Those queries -- and only those -- are handed to the
// heap-emulating exact kernel, which needs no pool.
__global__ __launch_bounds__(256) void ivfflat_flag_overfull_kernel(FlatPool p, int nq, uint32_t *__restrict__ flags) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q < nq) flags[q] = p.cnt[q] > (uint32_t)p.cap ? 1u : 0u;
}

__global__ __launch_bounds__(256) void ivfflat_flag_all_kernel(int nq, uint32_t *__restrict__ flags) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q < nq) flags[q] = 1u;
}

static int ivfflat_search_impl(lance_hip_ctx *ctx, const lance_hip_index *idx, const void *q, uint32_t nq, uint32_t k, uint32_t nprobes,
                               const uint32_t *allow, uint64_t *ids, float *dists) {
  LH_REQUIRE(ctx && idx && (nq == 0 || (q && ids && dists)), "ivfflat_search: NULL argument");
  LH_REQUIRE(idx->m == 0 && idx->vectors, "ivfflat_search: not an IVF_FLAT index");
  // FlatIndex::search takes any k (flat/index.rs:82-177).  The threshold machinery of the fast kernels selects among one value per
  // lane (k <= 128); beyond that every query goes through the heap-emulating exact kernel, whose heap lives in LDS (k <= 4096)
  LH_REQUIRE(k > 0 && k <= 4096, "ivfflat_search: k=%u not supported (1..4096)", k);
  const bool big_k = k > 128;
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  if (nq == 0) return LANCE_HIP_OK;
  if (nprobes > idx->nlist) nprobes = idx->nlist;
  LH_REQUIRE(nprobes > 0, "ivfflat_search: nprobes must be > 0");
  const int d = (int)idx->d;
  const float *qf;
  LH_TRY(as_f32(ctx, idx->dtype, q, (size_t)nq * d, "f16.q", &qf));
  uint32_t *probes = ctx->scratch_t<uint32_t>("ivfflat.probes", (size_t)nq * nprobes);
  float *pd = ctx->scratch_t<float>("ivfflat.pdists", (size_t)nq * nprobes);
  if (!probes || !pd) return LANCE_HIP_ENOMEM;
  const bool cosine = idx->metric == LANCE_HIP_COSINE;
  if (cosine) {
    // knn.rs:498 normalises the key of a cosine query; the coarse quantiser of a cosine index works in L2 on normalised
    // vectors (ivf/v2.rs:455-465), the partition scan uses the cosine distance itself (flat/storage.rs:345-402)
    float *qn = ctx->scratch_t<float>("ivfflat.qn", (size_t)nq * d);
    if (!qn) return LANCE_HIP_ENOMEM;
    LH_TRY(launch_normalize(ctx, qf, (int64_t)nq, d, qn, idx->dtype == LANCE_HIP_F16));   // an f16 key is normalised in f16 arithmetic
    qf = qn;
  }
  // f16 column under dot / cosine: the distance functions of half::f16 (32-lane dot / norm, scalar cosine) on the widened values
  const bool h32 = idx->dtype == LANCE_HIP_F16 && idx->metric != LANCE_HIP_L2;
  LH_TRY(find_partitions_f32(ctx, cosine ? LANCE_HIP_L2 : idx->metric, qf, nq, idx->d, idx->centroids, idx->nlist, nprobes, probes, pd,
                             idx->dtype == LANCE_HIP_F16 && idx->metric == LANCE_HIP_DOT && d > 16));
  const int qch = (int)std::min<uint32_t>(nq, FLAT_QCHUNK);
  IvfFlatArgs a;
  a.vec = idx->vectors; a.row_ids = idx->row_ids; a.part_offsets = idx->part_offsets; a.nprobes = (int)nprobes; a.d = d;
  a.allow = allow;
  FlatPool &pl = a.pool;
  constexpr int IVFFLAT_CAP = FLAT_CAP / 2;       // 2048 pool entries per query: (key, rowid, partition) sorted in 32 KiB of LDS
  pl.x = nullptr; pl.row_ids = nullptr; pl.r0 = pl.r1 = 0; pl.k = (int)k; pl.cap = IVFFLAT_CAP;
  a.ppart = ctx->scratch_t<uint32_t>("ivfflat.ppart", (size_t)qch * IVFFLAT_CAP);
  a.flags = ctx->scratch_t<uint32_t>("ivfflat.flags", (size_t)qch);
  if (!a.ppart || !a.flags) return LANCE_HIP_ENOMEM;
  pl.tkey = ctx->scratch_t<uint32_t>("flat2.tkey", qch);
  pl.trid = ctx->scratch_t<uint64_t>("flat2.trid", qch);
  pl.cnt = ctx->scratch_t<uint32_t>("flat2.cnt", qch);
  pl.pkeys = ctx->scratch_t<uint32_t>("flat2.pkeys", (size_t)qch * FLAT_CAP);
  pl.prids = ctx->scratch_t<uint64_t>("flat2.prids", (size_t)qch * FLAT_CAP);
  pl.overflow = ctx->scratch_t<uint32_t>("flat2.ovf", 1);
  if (!pl.tkey || !pl.trid || !pl.cnt || !pl.pkeys || !pl.prids || !pl.overflow) return LANCE_HIP_ENOMEM;
  const bool fixed = !cosine && !h32 && flat_fixed_dim(idx->d);
  const size_t sel_lds = (size_t)IVFFLAT_CAP * 16;
  const size_t ex_lds = (size_t)(((d + 3) & ~3) + 2) * 4 + (size_t)k * 12 + (size_t)(k + 1) * 8 + 64 * 4 + 64;
  LH_REQUIRE(ex_lds <= 160 * 1024, "ivfflat_search: k=%u with %d-dimensional vectors does not fit the exact kernel's LDS", k, d);
  auto finish = [&](int nqc, uint64_t *oid, float *od, bool select = true) {
    if (select) hipLaunchKernelGGL(ivfflat_select_kernel, dim3(nqc), dim3(256), sel_lds, ctx->stream, a, oid, od);
    ScopedTimer t(ctx, "ivfflat_exact");
    if (h32 && cosine) hipLaunchKernelGGL((ivfflat_exact_kernel<METRIC_COSINE, true>), dim3(nqc), dim3(64), ex_lds, ctx->stream, a, oid, od);
    else if (h32) hipLaunchKernelGGL((ivfflat_exact_kernel<METRIC_DOT, true>), dim3(nqc), dim3(64), ex_lds, ctx->stream, a, oid, od);
    else if (idx->metric == LANCE_HIP_DOT) hipLaunchKernelGGL((ivfflat_exact_kernel<METRIC_DOT>), dim3(nqc), dim3(64), ex_lds, ctx->stream, a, oid, od);
    else if (cosine) hipLaunchKernelGGL((ivfflat_exact_kernel<METRIC_COSINE>), dim3(nqc), dim3(64), ex_lds, ctx->stream, a, oid, od);
    else hipLaunchKernelGGL((ivfflat_exact_kernel<METRIC_L2>), dim3(nqc), dim3(64), ex_lds, ctx->stream, a, oid, od);
  };
  const bool pm = fixed && idx->n_flat_items > 0 &&     // partition-major for the fixed dimensions, query-major kernel otherwise
                  ((reinterpret_cast<uintptr_t>(qf) & 15) == 0);
  uint32_t *pair_starts = nullptr, *pair_idx = nullptr;
  if (pm) {
    pair_starts = ctx->scratch_t<uint32_t>("ivfflat.pair_starts", (size_t)idx->nlist + 1);
    pair_idx = ctx->scratch_t<uint32_t>("ivfflat.pair_idx", (size_t)qch * nprobes);
    if (!pair_starts || !pair_idx) return LANCE_HIP_ENOMEM;
  }
  uint32_t *pair_starts0 = nullptr, *pair_idx0 = nullptr, *keys0 = nullptr, *lanemin = nullptr;
  if (pm) {
    pair_starts0 = ctx->scratch_t<uint32_t>("ivfflat.pair_starts0", (size_t)idx->nlist + 1);
    pair_idx0 = ctx->scratch_t<uint32_t>("ivfflat.pair_idx0", (size_t)qch);
    keys0 = ctx->scratch_t<uint32_t>("ivfflat.keys0", (size_t)qch);
    lanemin = ctx->scratch_t<uint32_t>("ivfflat.lanemin", (size_t)qch * 256);
    if (!pair_starts0 || !pair_idx0 || !keys0 || !lanemin) return LANCE_HIP_ENOMEM;
  }
  a.items = idx->flat_items; a.pair_starts = pair_starts; a.pair_idx = pair_idx; a.pdiv = (int)nprobes; a.lanemin = lanemin;
  bool grouped = false;
  auto scan = [&](int bound, int nqc) {
    a.bound = bound;
    const unsigned grid = (unsigned)(bound ? nqc : nqc * (int)nprobes);
    ScopedTimer t(ctx, bound ? "ivfflat_bound" : "ivfflat_scan");
    if (bound && pm) {
      // bound pass, partition-major: group the queries by their nearest partition, fold every row of that partition
      // into 256 per-query lane minima, take the k-th smallest
      hipLaunchKernelGGL(ivfflat_first_probe_kernel, dim3(cdiv(nqc, 256)), dim3(256), 0, ctx->stream, a.probes, nqc, (int)nprobes, keys0, lanemin);
      (void)stable_group(ctx, keys0, (int64_t)nqc, (int64_t)nqc, (int)idx->nlist, 1, pair_starts0, pair_idx0, (int64_t)nqc, nullptr);
      IvfFlatArgs b = a;
      b.pair_starts = pair_starts0; b.pair_idx = pair_idx0; b.pdiv = 1;
      if (idx->metric == LANCE_HIP_DOT) launch_ivfflat_pm<METRIC_DOT, true>(ctx, b, idx->n_flat_items);
      else launch_ivfflat_pm<METRIC_L2, true>(ctx, b, idx->n_flat_items);
      hipLaunchKernelGGL(ivfflat_kth_kernel, dim3(nqc), dim3(256), 0, ctx->stream, lanemin, (int)k, pl.tkey);
      return;
    }
    if (!bound && pm) {
      if (!grouped) {
        (void)stable_group(ctx, a.probes, (int64_t)nqc * nprobes, (int64_t)nqc * nprobes, (int)idx->nlist, 1, pair_starts, pair_idx,
                           (int64_t)nqc * nprobes, nullptr);
        grouped = true;
      }
      if (idx->metric == LANCE_HIP_DOT) launch_ivfflat_pm<METRIC_DOT, false>(ctx, a, idx->n_flat_items);
      else launch_ivfflat_pm<METRIC_L2, false>(ctx, a, idx->n_flat_items);
      return;
    }
    if (idx->metric == LANCE_HIP_DOT) launch_ivfflat<METRIC_DOT>(ctx, a, grid, fixed, h32);
    else if (cosine) launch_ivfflat<METRIC_COSINE>(ctx, a, grid, false, h32);
    else launch_ivfflat<METRIC_L2>(ctx, a, grid, fixed);
  };
  for (uint32_t qc0 = 0; qc0 < nq; qc0 += (uint32_t)qch) {
    const int nqc = (int)std::min<uint32_t>((uint32_t)qch, nq - qc0);
    a.q = qf + (int64_t)qc0 * d;
    a.probes = probes + (int64_t)qc0 * nprobes;
    pl.q = a.q; pl.nq = nqc;
    uint64_t *oid = ids + (int64_t)qc0 * k;
    float *od = dists + (int64_t)qc0 * k;
    hipLaunchKernelGGL(flat_pool_reset_kernel, dim3(cdiv(nqc, 256)), dim3(256), 0, ctx->stream, pl, 1);
    grouped = false;
    if (big_k) {     // every query flagged: the exact kernel answers all of them
      hipLaunchKernelGGL(ivfflat_flag_all_kernel, dim3(cdiv(nqc, 256)), dim3(256), 0, ctx->stream, nqc, a.flags);
      finish(nqc, oid, od, false);
      continue;
    }
    scan(1, nqc);
    scan(0, nqc);
    finish(nqc, oid, od);
    for (int round = 0; round < 64; ++round) {
      uint32_t ovf = 0;
      LH_CHECK_HIP(hipMemcpyAsync(&ovf, pl.overflow, 4, hipMemcpyDeviceToHost, ctx->stream));
      LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
      if (!ovf) break;
      if (round >= 6) {
        // the bound stopped tightening: the overfull queries are replayed exactly (every other query already has its answer)
        hipLaunchKernelGGL(ivfflat_flag_overfull_kernel, dim3(cdiv(nqc, 256)), dim3(256), 0, ctx->stream, pl, nqc, a.flags);
        finish(nqc, oid, od, false);
        break;
      }
      hipLaunchKernelGGL(flat_pool_reset_kernel, dim3(cdiv(nqc, 256)), dim3(256), 0, ctx->stream, pl, 0);
      scan(0, nqc);
      finish(nqc, oid, od);
    }
  }
  LH_CHECK_HIP(hipGetLastError());
  LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return LANCE_HIP_OK;
}

extern "C" int lance_hip_ivfflat_search(lance_hip_ctx *ctx, const lance_hip_index *idx, const void *q, uint32_t nq, uint32_t k,
                                        uint32_t nprobes, uint64_t *ids, float *dists) {
  lh::CtxLock _ctx_lock(ctx);
  return ivfflat_search_impl(ctx, idx, q, nq, k, nprobes, nullptr, ids, dists);
}

extern "C" int lance_hip_ivfflat_search_filtered(lance_hip_ctx *ctx, const lance_hip_index *idx, const void *q, uint32_t nq, uint32_t k,
                                                 uint32_t nprobes, const uint8_t *allow_by_rowid, uint64_t n_allow, uint64_t *ids,
                                                 float *dists) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(ctx && idx && (allow_by_rowid || n_allow == 0), "ivfflat_search_filtered: NULL argument");
  LH_REQUIRE(idx->m == 0 && idx->vectors, "ivfflat_search_filtered: not an IVF_FLAT index");
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  const uint32_t *bits = nullptr;
  LH_TRY(build_allow_bits(ctx, idx->row_ids, idx->n, allow_by_rowid, n_allow, &bits));
  return ivfflat_search_impl(ctx, idx, q, nq, k, nprobes, bits, ids, dists);
}

extern "C" int lance_hip_flat_topk(lance_hip_ctx *ctx, int dtype, int metric, const void *x, const uint64_t *row_ids,
                                   uint64_t n, uint32_t d, const void *q, uint32_t nq, uint32_t k, uint64_t *ids,
                                   float *dists) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(ctx && (n == 0 || x) && (nq == 0 || (q && ids && dists)), "flat_topk: NULL argument");
  LH_TRY(check_dtype(dtype, "flat_topk"));
  LH_REQUIRE(metric == LANCE_HIP_L2 || metric == LANCE_HIP_DOT || metric == LANCE_HIP_COSINE, "flat_topk: bad metric %d", metric);
  LH_REQUIRE(k > 0 && k <= 1024, "flat_topk: k=%u not supported (1..1024)", k);
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  if (nq == 0) return LANCE_HIP_OK;
  // f16 columns under dot / cosine: half::f16's own distance functions (32-lane dot_scalar / norm_l2_impl, scalar cosine;
  // dot.rs:91-102, norm_l2.rs:60-85, cosine.rs:171-179) -- the run-time-dimension kernel below carries those orders
  const bool h32 = dtype == LANCE_HIP_F16 && metric != LANCE_HIP_L2;
  if (flat_small_supported(metric, dtype, d, nq, k, n)) {      // one to four queries: a single streaming pass (flat_small.hip)
    const float *qs2;
    LH_TRY(as_f32(ctx, dtype, q, (size_t)nq * d, "f16.q", &qs2));
    bool done = false;
    LH_TRY(flat_topk_small(ctx, metric, dtype, x, row_ids, n, d, qs2, nq, k, ids, dists, &done));
    if (done) return LANCE_HIP_OK;
  }
  if (!h32 && flat_v2_supported(metric, d, k) && !getenv("LANCE_HIP_FLAT_V1")) {
    const float *xf2 = nullptr, *qf2;
    LH_TRY(as_f32(ctx, dtype, q, (size_t)nq * d, "f16.q", &qf2));
    // f16 / int8 columns: the fixed-dimension L2 / dot kernels (exact filter and MFMA filter) read the rows as they are;
    // cosine and the any-dimension kernels still take an f32 copy
    const size_t esz = dtype == LANCE_HIP_F16 ? 2 : 1;
    const bool native = dtype != LANCE_HIP_F32 && metric != LANCE_HIP_COSINE && flat_fixed_dim(d) && ((size_t)d * esz) % 16 == 0 &&
                        ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(qf2)) & 15) == 0 && !getenv("LANCE_HIP_NO_NATIVE_FLAT");
    if (!native) LH_TRY(as_f32(ctx, dtype, x, (size_t)n * d, "f16.x", &xf2));
    return flat_topk_v2(ctx, metric, xf2, x, dtype, row_ids, (int64_t)n, (int)d, qf2, (int)nq, (int)k, ids, dists);
  }
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
    switch ((metric == LANCE_HIP_COSINE || h32) ? 0u : d) {
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
        if (h32 && metric == LANCE_HIP_COSINE)
          hipLaunchKernelGGL((flat_scan_generic_kernel<METRIC_COSINE, true>), grid, dim3(256), lds, ctx->stream, a, tr);
        else if (h32)
          hipLaunchKernelGGL((flat_scan_generic_kernel<METRIC_DOT, true>), grid, dim3(256), lds, ctx->stream, a, tr);
        else if (metric == LANCE_HIP_COSINE)
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
