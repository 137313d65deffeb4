// kmeans.hip -- Lloyd k-means exactly as lance-index trains it, batched over independent
// problems (1 for IVF, M for the PQ sub-quantisers).
//
//   KMeans::train_kmeans            rust/lance-index/src/vector/kmeans.rs:610-719
//   compute_membership_and_loss     :250-281      (E-step -> pairwise.hip assign kernel)
//   compute_cluster_sizes           :210-232
//   compute_balance_loss            :234-237
//   KMeansAlgoFloat::to_kmeans      :371-446      (M-step: per-centroid sums in ROW ORDER)
//   split_clusters                  :174-207
//   kmeans_random_init              :149-170
//   PQBuildParams::build_from_fsl   pq/builder.rs:89-157
//
// Determinism: the M-step first groups row indices by cluster id with a stable sort
// (group.hip) and then lets one lane per (centroid, dimension) add the members in ascending
// row order -- the same f32 addition chain a reference rayon thread executes -- so given
// the same initial centroids the trained centroids are bit-identical to the CPU result.
// Per-cluster loss is accumulated in f64 in the same order (kmeans.rs:274-277).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <thread>
#include <vector>

#include <hip/hip_fp16.h>

#include "common.h"
#include "exact.cuh"
#include "f16.h"
#include "kernels.h"
#include "rng.h"

#pragma clang fp contract(off)

namespace lh {

// one lane per (centroid, dim): sequential row-order sum, then *= 1/count (kmeans.rs:388-418)
__device__ __forceinline__ void kmeans_accumulate_body(int bx, const float *__restrict__ x, int64_t ldx, int x_batch_off,
                                                       int d, int k, const uint32_t *__restrict__ sorted_rows,
                                                       int64_t rows_stride, const uint32_t *__restrict__ starts,
                                                       float *__restrict__ cent, int64_t cent_batch_stride,
                                                       const uint8_t *__restrict__ active, int scale, int f16) {
  const int b = blockIdx.y;
  if (active && !active[b]) return;
  const int64_t g = (int64_t)bx * 256 + threadIdx.x;
  if (g >= (int64_t)k * d) return;
  const int c = (int)(g / d), dim = (int)(g % d);
  const uint32_t *st = starts + (int64_t)b * (k + 1);
  const uint32_t *rows = sorted_rows + (int64_t)b * rows_stride;
  const float *xb = x + (int64_t)b * x_batch_off + dim;
  const uint32_t s = st[c], e = st[c + 1];
  float acc = 0.0f;
  uint32_t i = s;
  if (f16) {
    // T = half::f16: `*c += *v` and `*v *= norm` are f16 operations (f32 op, round to binary16)
    for (; i < e; ++i) acc = __half2float(__float2half_rn(acc + xb[(int64_t)rows[i] * ldx]));
  } else {
    for (; i + 32 <= e; i += 32) {      // 32 gathers in flight per lane; the adds stay in row order
      float v[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) v[u] = xb[(int64_t)rows[i + u] * ldx];
#pragma unroll
      for (int u = 0; u < 32; ++u) acc += v[u];
    }
    for (; i + 8 <= e; i += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = xb[(int64_t)rows[i + u] * ldx];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; i < e; ++i) acc += xb[(int64_t)rows[i] * ldx];
  }
  const uint32_t cnt = e - s;
  if (scale && cnt > 0) {
    if (f16) {
      const float cnt_h = __half2float(__float2half_rn((float)cnt));        // T::from_usize(cnt)
      const float norm = __half2float(__float2half_rn(1.0f / cnt_h));        // T::one() / cnt
      acc = __half2float(__float2half_rn(acc * norm));
    } else {
      const float norm = 1.0f / (float)cnt;
      acc *= norm;
    }
  }
  cent[(int64_t)b * cent_batch_stride + (int64_t)c * d + dim] = acc;
}
__global__ __launch_bounds__(256) void kmeans_accumulate_kernel(const float *__restrict__ x, int64_t ldx, int x_batch_off,
                                                                int d, int k, const uint32_t *__restrict__ sorted_rows,
                                                                int64_t rows_stride, const uint32_t *__restrict__ starts,
                                                                float *__restrict__ cent, int64_t cent_batch_stride,
                                                                const uint8_t *__restrict__ active, int scale, int f16) {
  kmeans_accumulate_body((int)blockIdx.x, x, ldx, x_batch_off, d, k, sorted_rows, rows_stride, starts, cent, cent_batch_stride, active, scale, f16);
}

// one WAVE per centroid: the 64 lanes fetch 64 member distances at a time (two dependent loads, but 64 wide),
// lane 0 then adds them in row order -- the f64 chain of kmeans.rs:274-277 is kept, only its operands are
// fetched in parallel.  (One lane per centroid spent 70 us per iteration on 2 x 32 serial memory round trips.)
__device__ __forceinline__ void kmeans_stats_body(int bx, float (*buf)[2][64], const float *__restrict__ dists, int64_t dist_stride, int k,
                                                  const uint32_t *__restrict__ sorted_rows, int64_t rows_stride,
                                                  const uint32_t *__restrict__ starts, double *__restrict__ losses,
                                                  float *__restrict__ radius, uint32_t *__restrict__ last_row,
                                                  const uint8_t *__restrict__ active) {
  const int b = blockIdx.y;
  if (active && !active[b]) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = bx * 4 + wave;
  if (c >= k) return;
  const uint32_t *st = starts + (int64_t)b * (k + 1);
  const uint32_t *rows = sorted_rows + (int64_t)b * rows_stride;
  const float *db = dists + (int64_t)b * dist_stride;
  double loss = 0.0;
  float rad = 0.0f;
  const uint32_t s = st[c], e = st[c + 1];
  int pb = 0;
  float nxt = (s + lane < e) ? db[rows[s + lane]] : 0.0f;
  for (uint32_t i = s; i < e; i += 64) {
    buf[wave][pb][lane] = nxt;
    const uint32_t j = i + 64 + lane;
    nxt = j < e ? db[rows[j]] : 0.0f;           // next chunk in flight while lane 0 adds this one
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane == 0) {
      const int cnt = (int)min(64u, e - i);
      const float *v = buf[wave][pb];
      int t = 0;
      for (; t + 4 <= cnt; t += 4) {
        const f4 q4 = *reinterpret_cast<const f4 *>(v + t);
        rad = fmaxf(rad, q4.x); loss += (double)q4.x;   // f32::max; f64 chain in row order
        rad = fmaxf(rad, q4.y); loss += (double)q4.y;
        rad = fmaxf(rad, q4.z); loss += (double)q4.z;
        rad = fmaxf(rad, q4.w); loss += (double)q4.w;
      }
      for (; t < cnt; ++t) { rad = fmaxf(rad, v[t]); loss += (double)v[t]; }
    }
    pb ^= 1;
  }
  if (lane == 0) {
    losses[(int64_t)b * k + c] = loss;
    radius[(int64_t)b * k + c] = rad;
    last_row[(int64_t)b * k + c] = e > s ? rows[e - 1] : 0xFFFFFFFFu;
  }
}
__global__ __launch_bounds__(256) void kmeans_stats_kernel(const float *__restrict__ dists, int64_t dist_stride, int k,
                                                           const uint32_t *__restrict__ sorted_rows, int64_t rows_stride,
                                                           const uint32_t *__restrict__ starts, double *__restrict__ losses,
                                                           float *__restrict__ radius, uint32_t *__restrict__ last_row,
                                                           const uint8_t *__restrict__ active) {
  __shared__ __attribute__((aligned(16))) float buf[4][2][64];
  kmeans_stats_body((int)blockIdx.x, buf, dists, dist_stride, k, sorted_rows, rows_stride, starts, losses, radius, last_row, active);
}

// The trainer's M-step in ONE launch: blocks [0, acc_blocks) are kmeans_accumulate_kernel's, the rest kmeans_stats_kernel's (they read the
// same member lists and are independent of each other; a launch less per Lloyd iteration: the build is bound by its kernel boundaries).
struct KmMstepArgs {
  const float *x; int64_t ldx; int x_batch_off, d, k;
  const uint32_t *sorted_rows; int64_t rows_stride; const uint32_t *starts;
  float *cent; int64_t cent_batch_stride; const uint8_t *active; int scale, f16;
  const float *dists; int64_t dist_stride; double *losses; float *radius; uint32_t *last_row;
  int acc_blocks;
};
__global__ __launch_bounds__(256) void kmeans_mstep_kernel(KmMstepArgs a) {
  __shared__ __attribute__((aligned(16))) float buf[4][2][64];
  if ((int)blockIdx.x < a.acc_blocks)
    kmeans_accumulate_body((int)blockIdx.x, a.x, a.ldx, a.x_batch_off, a.d, a.k, a.sorted_rows, a.rows_stride, a.starts, a.cent, a.cent_batch_stride, a.active,
                           a.scale, a.f16);
  else
    kmeans_stats_body((int)blockIdx.x - a.acc_blocks, buf, a.dists, a.dist_stride, a.k, a.sorted_rows, a.rows_stride, a.starts, a.losses, a.radius,
                      a.last_row, a.active);
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float *__restrict__ x, int64_t ldx, int x_batch_off, int d,
                                                          int k, const uint64_t *__restrict__ idx,
                                                          float *__restrict__ out, int64_t out_batch_stride) {
  const int b = blockIdx.y;
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= (int64_t)k * d) return;
  const int c = (int)(g / d), dim = (int)(g % d);
  out[(int64_t)b * out_batch_stride + g] = x[(int64_t)idx[(int64_t)b * k + c] * ldx + (int64_t)b * x_batch_off + dim];
}

// ---- per-iteration control on the device -------------------------------------------------------------------------
// Everything train_kmeans does between two E-steps besides the sums -- cluster sizes and the largest cluster
// (compute_cluster_sizes :210-232), the balance factor update and loss (:234-237, :680-704), the empty-cluster split
// (split_clusters :174-207, same RNG stream as the oracle), the convergence test (:704) and the next iteration's bias --
// runs in one small workgroup per problem, so the Lloyd loop needs no host round trip per iteration.  The scalar chains
// (f64 loss sum in cluster order, the running-max rule, the split's rejection loop) are executed by lane 0 in the
// reference's order; only order-free work (sizes, sum of squares, bias) is spread over the lanes.
struct KmState {
  double loss, last_loss;
  float adjusted, bf_used;
  uint32_t iters, pad;
  Rng rng;
};

struct KmCtl {
  KmState *state;               // [B]
  uint8_t *active;              // [B]
  const uint32_t *starts;       // [B][k+1]
  const double *losses;         // [B][k]
  const float *radius;          // [B][k]
  const uint32_t *last_row;     // [B][k]
  float *bias;                  // [B][k]
  float *cent;                  // [B][k][d]
  uint32_t *sizes;              // [B][k] cluster sizes carried to the next iteration (after a split)
  int64_t n;
  int k, d, f16, use_bias;
  float balance_factor_scaled;
  double tol;
};

__global__ __launch_bounds__(64) void kmeans_state_init_kernel(KmCtl c, const uint64_t *__restrict__ seeds) {
  const int b = blockIdx.x;
  if (threadIdx.x == 0) {
    KmState &s = c.state[b];
    s.loss = DBL_MAX; s.last_loss = DBL_MAX; s.adjusted = FLT_MAX; s.iters = 0; s.pad = 0;
    s.bf_used = FLT_MAX < c.balance_factor_scaled ? FLT_MAX : c.balance_factor_scaled;   // f32::min(adjusted, balance_factor)
    s.rng.seed(seeds[b] ^ 0x5bd1e995ULL);
    c.active[b] = 1;
  }
  for (int i = threadIdx.x; i < c.k; i += 64) { c.sizes[(int64_t)b * c.k + i] = 0; c.bias[(int64_t)b * c.k + i] = 0.0f; }
}

__device__ __forceinline__ float km_round(float v, int f16) { return f16 ? __half2float(__float2half_rn(v)) : v; }

__global__ __launch_bounds__(256) void kmeans_control_kernel(KmCtl c) {
  extern __shared__ __attribute__((aligned(8))) char km_smem[];
  // per-cluster inputs staged in LDS by all lanes, so that lane 0's sequential chains read LDS, not one global word at a time
  double *lbs = reinterpret_cast<double *>(km_smem);            // [k] per-cluster losses
  uint32_t *sz = reinterpret_cast<uint32_t *>(lbs + c.k);       // [k] cluster sizes
  uint32_t *lasts = sz + c.k;                                   // [k] last member row
  __shared__ unsigned long long s_sq;
  __shared__ int s_empty;
  const int b = blockIdx.x, k = c.k;
  if (!c.active[b]) return;
  const uint32_t *st = c.starts + (int64_t)b * (k + 1);
  if (threadIdx.x == 0) { s_sq = 0ull; s_empty = 0; }
  __syncthreads();
  unsigned long long sq = 0ull;
  int empty = 0;
  for (int i = threadIdx.x; i < k; i += 256) {
    const uint32_t v = st[i + 1] - st[i];
    sz[i] = v;
    lbs[i] = c.losses[(int64_t)b * k + i];
    lasts[i] = c.last_row[(int64_t)b * k + i];
    sq += (unsigned long long)v * v;
    empty |= v == 0;
  }
  atomicAdd(&s_sq, sq);
  if (empty) s_empty = 1;
  __syncthreads();
  // compute_cluster_sizes: the running-max rule picks, among the clusters of maximal size, the one whose last member comes
  // first in row order (and cluster 0 when every cluster is empty).  That is an order-free selection -- largest size, then
  // smallest last row (distinct for non-empty clusters), then smallest id -- so all lanes take part: a single lane walking k
  // LDS words with a dependent compare each was two thirds of this kernel's 30 us (98 launches per index build).
  __shared__ uint32_t s_bsz[4], s_blast[4], s_bid[4];
  {
    uint32_t bsz = 0, blast = 0xFFFFFFFFu, bid = 0xFFFFFFFFu;
    for (int i = threadIdx.x; i < k; i += 256) {
      const uint32_t v = sz[i], lr = lasts[i];
      if (v > bsz || (v == bsz && ((v > 0 && lr < blast) || (v == 0 && (uint32_t)i < bid)))) { bsz = v; blast = lr; bid = (uint32_t)i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint32_t osz = __shfl_xor(bsz, o, 64), olast = __shfl_xor(blast, o, 64), oid = __shfl_xor(bid, o, 64);
      if (osz > bsz || (osz == bsz && ((osz > 0 && olast < blast) || (osz == 0 && oid < bid)))) { bsz = osz; blast = olast; bid = oid; }
    }
    if ((threadIdx.x & 63) == 0) { s_bsz[threadIdx.x >> 6] = bsz; s_blast[threadIdx.x >> 6] = blast; s_bid[threadIdx.x >> 6] = bid; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    KmState &s = c.state[b];
    const double *lb = lbs;
    const float *rb = c.radius + (int64_t)b * k;
    s.iters += 1u;      // (counted on the device: the launch takes no per-iteration argument, so a block of iterations can be a HIP graph)
    uint32_t bsz = s_bsz[0], blast = s_blast[0], bid = s_bid[0];
    for (int w = 1; w < 4; ++w) {
      const uint32_t osz = s_bsz[w], olast = s_blast[w], oid = s_bid[w];
      if (osz > bsz || (osz == bsz && ((osz > 0 && olast < blast) || (osz == 0 && oid < bid)))) { bsz = osz; blast = olast; bid = oid; }
    }
    const int max_id = bid == 0xFFFFFFFFu ? 0 : (int)bid;
    s.adjusted = (rb[max_id] - (float)lb[max_id] / (float)sz[max_id]) / (float)c.n;
    const float size_loss = (float)(uint64_t)s_sq;
    const float balance_loss = s.bf_used * (size_loss - (float)((uint64_t)c.n * (uint64_t)c.n) / (float)k);
    double lsum = 0.0;     // f64 chain in cluster order (the reference's order): unrolled so that the LDS loads run ahead of the adds
#pragma unroll 8
    for (int i = 0; i < k; ++i) lsum = lsum + lb[i];
    s.last_loss = lsum + (double)balance_loss;
    if (s_empty) {
      // split_clusters: an empty cluster takes half of a size-weighted random one, both perturbed by +-1/1024 on
      // alternating dimensions.  The reference's rejection loop never terminates when no cluster has >= 2 members (all
      // p <= 0, e.g. every distance NaN after an f16 M-step overflow): stop splitting instead of hanging.
      float *cent = c.cent + (int64_t)b * k * c.d;
      const float eps = 1.0f / 1024.0f;
      for (int i = 0; i < k; ++i) {
        if (sz[i] != 0) continue;
        bool splittable = false;
        for (int t = 0; t < k; ++t) if (sz[t] >= 2) { splittable = true; break; }
        if (!splittable) break;
        int j = 0;
        for (;;) {
          const float p = ((float)sz[j] - 1.0f) / (float)(c.n - k);
          if (s.rng.next_f32() < p) break;
          j += 1;
          j %= k;
        }
        sz[i] = sz[j] / 2;
        sz[j] -= sz[i];
        for (int t = 0; t < c.d; ++t) {
          const float cj = cent[(int64_t)j * c.d + t];
          if (t % 2 == 0) {
            cent[(int64_t)i * c.d + t] = km_round(cj * (1.0f + eps), c.f16);
            cent[(int64_t)j * c.d + t] = km_round(cj * (1.0f - eps), c.f16);
          } else {
            cent[(int64_t)i * c.d + t] = km_round(cj * (1.0f - eps), c.f16);
            cent[(int64_t)j * c.d + t] = km_round(cj * (1.0f + eps), c.f16);
          }
        }
      }
    }
    if (fabs(s.loss - s.last_loss) < c.tol * s.last_loss) {
      c.active[b] = 0;
    } else {
      s.loss = s.last_loss;
    }
    s.bf_used = s.adjusted < c.balance_factor_scaled ? s.adjusted : c.balance_factor_scaled;   // next iteration's factor
  }
  __syncthreads();
  const float bf = c.state[b].bf_used;
  for (int i = threadIdx.x; i < k; i += 256) {
    c.sizes[(int64_t)b * k + i] = sz[i];
    if (c.use_bias) c.bias[(int64_t)b * k + i] = bf * (float)sz[i];
  }
}

// Batched trainer.  x: [n][ldx] floats; problem b uses columns [b*x_batch_off, +d).
// cent: [B][k][d] (in/out: holds the initial centroids when have_init).
int kmeans_train_batched(lance_hip_ctx *ctx, int metric, const float *x, int64_t n, int64_t ldx, int x_batch_off, int d,
                         int k, int B, uint32_t max_iters, double tol, float balance_factor_scaled, bool have_init,
                         const uint64_t *seeds, float *cent, double *loss_out, uint32_t *iters_out, bool f16_arith) {
  LH_REQUIRE(n >= k, "KMeans: training does not have sufficient data points: n(%lld) is smaller than k(%d)", (long long)n, k);
  LH_REQUIRE(k <= 4096, "kmeans_train: k=%d > 4096 not supported in this version", k);
  LH_REQUIRE(metric == METRIC_L2 || metric == METRIC_DOT, "kmeans_train: metric must be L2 or Dot");
  // kmeans.rs:623-627
  if (n >= (int64_t)k * 512) n = (int64_t)k * 512;

  uint64_t *seeds_d = ctx->scratch_t<uint64_t>("kmeans.seeds", (size_t)B);
  if (!seeds_d) return LANCE_HIP_ENOMEM;
  LH_CHECK_HIP(hipMemcpyAsync(seeds_d, seeds, (size_t)B * 8, hipMemcpyHostToDevice, ctx->stream));
  if (!have_init) {
    // kmeans_random_init: reservoir choose_multiple on the host, gather on the device
    std::vector<uint64_t> idx((size_t)B * k);
    // The reservoir walk is sequential in n per problem (one RNG draw per row: 0.4 ms at n = 65,536) -- but the B problems are
    // independent streams: one host thread each.  (Round 5: the sixteen PQ sub-quantisers spent 6.4 of train_pq's 14.5 ms here, on
    // one host core, with the GPU idle.)
    if (B > 1) {
      const int nthreads = std::max(1, std::min<int>(B, (int)std::thread::hardware_concurrency() ? (int)std::thread::hardware_concurrency() : 4));
      std::vector<std::thread> pool;
      pool.reserve((size_t)nthreads);
      for (int t = 0; t < nthreads; ++t)
        pool.emplace_back([&, t]() {
          for (int b = t; b < B; b += nthreads) kmeans_init_indices((uint64_t)n, (uint32_t)k, seeds[b], idx.data() + (size_t)b * k);
        });
      for (auto &th : pool) th.join();
    } else {
      kmeans_init_indices((uint64_t)n, (uint32_t)k, seeds[0], idx.data());
    }
    uint64_t *didx = ctx->scratch_t<uint64_t>("kmeans.initidx", idx.size());
    if (!didx) return LANCE_HIP_ENOMEM;
    LH_CHECK_HIP(hipMemcpyAsync(didx, idx.data(), idx.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)cdiv((uint64_t)k * d, 256), B), dim3(256), 0, ctx->stream, x,
                       ldx, x_batch_off, d, k, didx, cent, (int64_t)k * d);
    LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));  // idx goes out of scope
  }

  uint32_t *ids = ctx->scratch_t<uint32_t>("kmeans.ids", (size_t)B * n);
  float *dists = ctx->scratch_t<float>("kmeans.dists", (size_t)B * n);
  uint32_t *starts = ctx->scratch_t<uint32_t>("kmeans.starts", (size_t)B * (k + 1));
  uint32_t *sorted_rows = ctx->scratch_t<uint32_t>("kmeans.sorted", (size_t)B * n);
  float *bias = ctx->scratch_t<float>("kmeans.bias", (size_t)B * k);
  uint8_t *active_d = ctx->scratch_t<uint8_t>("kmeans.active", (size_t)B);
  uint32_t *sizes_d = ctx->scratch_t<uint32_t>("kmeans.sizes", (size_t)B * k);
  KmState *state_d = ctx->scratch_t<KmState>("kmeans.state", (size_t)B);
  // stats block: [losses f64 B*k][radius f32 B*k][last_row u32 B*k]
  const size_t stats_bytes = (size_t)B * k * (8 + 4 + 4);
  char *stats_d = reinterpret_cast<char *>(ctx->scratch("kmeans.stats", stats_bytes));
  if (!ids || !dists || !starts || !sorted_rows || !bias || !active_d || !sizes_d || !state_d || !stats_d) return LANCE_HIP_ENOMEM;
  double *losses_d = reinterpret_cast<double *>(stats_d);
  float *radius_d = reinterpret_cast<float *>(stats_d + (size_t)B * k * 8);
  uint32_t *last_d = reinterpret_cast<uint32_t *>(stats_d + (size_t)B * k * 12);
  const bool use_bias = balance_factor_scaled != 0.0f;

  KmCtl ctl;
  ctl.state = state_d; ctl.active = active_d; ctl.starts = starts; ctl.losses = losses_d; ctl.radius = radius_d; ctl.last_row = last_d;
  ctl.bias = bias; ctl.cent = cent; ctl.sizes = sizes_d; ctl.n = n; ctl.k = k; ctl.d = d; ctl.f16 = f16_arith ? 1 : 0;
  ctl.use_bias = use_bias ? 1 : 0; ctl.balance_factor_scaled = balance_factor_scaled; ctl.tol = tol;
  hipLaunchKernelGGL(kmeans_state_init_kernel, dim3(B), dim3(64), 0, ctx->stream, ctl, seeds_d);

  // The iterations are enqueued back to back; converged problems turn themselves off on the device (`active`).  The host
  // looks at the flags only every few iterations to stop enqueueing once every problem has converged.
  static const uint32_t check_every = getenv("LANCE_HIP_KMEANS_CHECK") ? (uint32_t)std::max(1, atoi(getenv("LANCE_HIP_KMEANS_CHECK"))) : 8;
  std::vector<uint8_t> active(B, 1);
  auto enqueue_iteration = [&]() -> int {
    PairwiseArgs pa;
    pa.x = x; pa.n = n; pa.ldx = ldx; pa.x_batch_off = x_batch_off;
    pa.cent = cent; pa.k = k; pa.cent_batch_stride = (int64_t)k * d;
    pa.bias = use_bias ? bias : nullptr; pa.bias_batch_stride = k;
    pa.ids = ids; pa.dists = dists; pa.out_batch_stride = n;
    pa.active = active_d;
    pa.lanes32 = f16_arith && metric == LANCE_HIP_DOT && d > 16;   // KMeansAlgoFloat<Float16Type> + dot: dot_scalar::<f16, f32, 32>
    LH_TRY(launch_assign(ctx, pa, d, metric, B));
    LH_TRY(stable_group(ctx, ids, n, n, k, B, starts, sorted_rows, n, active_d));
    {
      ScopedTimer t(ctx, "kmeans_mstep");
      KmMstepArgs ma;
      ma.x = x; ma.ldx = ldx; ma.x_batch_off = x_batch_off; ma.d = d; ma.k = k; ma.sorted_rows = sorted_rows; ma.rows_stride = n; ma.starts = starts;
      ma.cent = cent; ma.cent_batch_stride = (int64_t)k * d; ma.active = active_d; ma.scale = 1; ma.f16 = f16_arith ? 1 : 0;
      ma.dists = dists; ma.dist_stride = n; ma.losses = losses_d; ma.radius = radius_d; ma.last_row = last_d;
      ma.acc_blocks = (int)cdiv((uint64_t)k * d, 256);
      hipLaunchKernelGGL(kmeans_mstep_kernel, dim3((unsigned)(ma.acc_blocks + (int)cdiv(k, 4)), B), dim3(256), 0, ctx->stream, ma);
      hipLaunchKernelGGL(kmeans_control_kernel, dim3(B), dim3(256), (size_t)k * 16, ctx->stream, ctl);
    }
    LH_CHECK_HIP(hipGetLastError());
    return LANCE_HIP_OK;
  };
  auto all_converged = [&]() -> int {      // 1: every problem has turned itself off; 0: not yet; < 0: error
    if (hipMemcpyAsync(active.data(), active_d, B, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
      set_error("kmeans_train: reading the convergence flags failed");
      return -1;
    }
    for (int b = 0; b < B; ++b) if (active[b]) return 0;
    return 1;
  };
  // An iteration is ~10 small launches (assign: prep + MFMA sweep + re-check, four grouping kernels, stats, accumulate, control) whose
  // arguments never change: the build was launch-latency bound (profiles/r03: train_pq 14 ms = 50 iterations x ~280 us of which ~160 us
  // are kernels).  The first block of `check_every` iterations runs on the plain path (it sizes the scratch arena); the second is captured
  // into a HIP graph and every further block is one hipGraphLaunch.  OPT-IN (LANCE_HIP_KMEANS_GRAPH=1): measured no gain (24.4 vs 25.0 ms at
  // C2, gpurun r04p -- the cost is the kernel boundaries, not the launches) while every training call, and every sub-problem of a
  // hierarchical one, paid a capture + instantiate.
  static const bool graph_on = getenv("LANCE_HIP_KMEANS_GRAPH") && getenv("LANCE_HIP_KMEANS_GRAPH")[0] == '1';
  hipGraphExec_t gexec = nullptr;
  bool graph_failed = !graph_on || ctx->timing || ctx->capturing;
  uint32_t it = 0;
  int rc_loop = LANCE_HIP_OK;
  while (it < max_iters) {
    const uint32_t blk = std::min<uint32_t>(check_every, max_iters - it);
    bool done_block = false;
    if (blk == check_every && it >= check_every && !graph_failed) {
      if (!gexec) {
        hipGraph_t g = nullptr;
        if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
          ctx->capturing = true;
          int rc = LANCE_HIP_OK;
          for (uint32_t t = 0; t < blk && rc == LANCE_HIP_OK; ++t) rc = enqueue_iteration();
          ctx->capturing = false;
          const hipError_t ee = hipStreamEndCapture(ctx->stream, &g);
          if (!(rc == LANCE_HIP_OK && ee == hipSuccess && g && hipGraphInstantiate(&gexec, g, nullptr, nullptr, 0) == hipSuccess && gexec)) {
            gexec = nullptr; graph_failed = true;      // nothing was executed: this block runs on the plain path below
          }
          if (g) (void)hipGraphDestroy(g);
          (void)hipGetLastError();
        } else {
          (void)hipGetLastError();
          graph_failed = true;
        }
      }
      if (gexec) {
        if (hipGraphLaunch(gexec, ctx->stream) != hipSuccess) { set_error("kmeans_train: hipGraphLaunch failed"); rc_loop = LANCE_HIP_ERUNTIME; break; }
        done_block = true;
      }
    }
    if (!done_block)
      for (uint32_t t = 0; t < blk; ++t) { rc_loop = enqueue_iteration(); if (rc_loop != LANCE_HIP_OK) break; }
    if (rc_loop != LANCE_HIP_OK) break;
    it += blk;
    if (it < max_iters) {
      const int c = all_converged();
      if (c < 0) { rc_loop = LANCE_HIP_ERUNTIME; break; }
      if (c) break;
    }
  }
  if (gexec) { (void)hipStreamSynchronize(ctx->stream); (void)hipGraphExecDestroy(gexec); }
  if (rc_loop != LANCE_HIP_OK) return rc_loop;
  std::vector<KmState> st_h(B);
  LH_CHECK_HIP(hipMemcpyAsync(st_h.data(), state_d, (size_t)B * sizeof(KmState), hipMemcpyDeviceToHost, ctx->stream));
  LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  for (int b = 0; b < B; ++b) {
    if (loss_out) loss_out[b] = st_h[b].last_loss;
    if (iters_out) iters_out[b] = st_h[b].iters;
  }
  return LANCE_HIP_OK;
}

__global__ __launch_bounds__(256) void counts_to_float_kernel(const uint32_t *__restrict__ starts, int k, float *__restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < k) out[c] = (float)(starts[c + 1] - starts[c]);
}

__global__ __launch_bounds__(256) void finalize_centroids_kernel(const float *__restrict__ buf, int k, int d, float *__restrict__ cent) {
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= (int64_t)k * d) return;
  const float cnt = buf[(int64_t)k * d + g / d];
  float v = buf[g];
  if (cnt > 0.0f) {
    const float norm = 1.0f / cnt;
    v *= norm;
  }
  cent[g] = v;
}

__global__ __launch_bounds__(256) void gather_rows_u32_kernel(const float *__restrict__ x, int d, const uint32_t *__restrict__ idx,
                                                              int64_t cnt, float *__restrict__ out) {
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= cnt * d) return;
  const int64_t r = g / d;
  out[g] = x[(int64_t)idx[r] * d + (g - r * d)];
}

// train_hierarchical_kmeans (kmeans.rs:746-1003): host-driven; every k-means / assignment runs on
// the device.  The cluster heap follows Rust std BinaryHeap push/pop with Ord = (not finalized, size).
namespace {
struct HCluster {
  size_t id;
  std::vector<uint32_t> idx;
  std::vector<float> centroid;
  bool finalized;
};
inline bool hc_le(const HCluster &a, const HCluster &b) {
  const int ka = a.finalized ? 0 : 1, kb = b.finalized ? 0 : 1;
  if (ka != kb) return ka < kb;
  return a.idx.size() <= b.idx.size();
}
struct HHeap {
  std::vector<HCluster> d;
  void sift_up(size_t start, size_t pos) {
    HCluster e = std::move(d[pos]);
    while (pos > start) {
      const size_t parent = (pos - 1) / 2;
      if (hc_le(e, d[parent])) break;
      d[pos] = std::move(d[parent]);
      pos = parent;
    }
    d[pos] = std::move(e);
  }
  void push(HCluster c) { d.push_back(std::move(c)); sift_up(0, d.size() - 1); }
  HCluster pop() {
    HCluster item = std::move(d.back());
    d.pop_back();
    if (!d.empty()) {
      std::swap(item, d[0]);
      const size_t end = d.size();
      size_t pos = 0, child = 1;
      HCluster e = std::move(d[0]);
      while (end >= 2 && child <= end - 2) {
        if (hc_le(d[child], d[child + 1])) child += 1;
        d[pos] = std::move(d[child]);
        pos = child;
        child = 2 * pos + 1;
      }
      if (child == end - 1) { d[pos] = std::move(d[child]); pos = child; }
      d[pos] = std::move(e);
      sift_up(0, pos);
    }
    return item;
  }
};
}  // namespace

static int hier_assign_host(lance_hip_ctx *ctx, int metric, const float *x, int64_t n, int d, const float *cent, int k,
                            std::vector<uint32_t> &mem, bool f16_arith) {
  uint32_t *ids = ctx->scratch_t<uint32_t>("hier.ids", (size_t)n);
  if (!ids) return LANCE_HIP_ENOMEM;
  PairwiseArgs pa;
  pa.x = x; pa.n = n; pa.ldx = d; pa.cent = cent; pa.k = k; pa.ids = ids; pa.out_batch_stride = n;
  pa.lanes32 = f16_arith && metric == LANCE_HIP_DOT && d > 16;
  LH_TRY(launch_assign(ctx, pa, d, metric, 1));
  mem.resize((size_t)n);
  LH_CHECK_HIP(hipMemcpyAsync(mem.data(), ids, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
  LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return LANCE_HIP_OK;
}

// One split of train_hierarchical_kmeans (kmeans.rs:866-905): k-means with `cluster_k` centroids over the rows `idx` of x, then the
// membership of those rows.  cdev: [cluster_k][d] device centroids; mem: host membership (LANCE_HIP_NONE: no centroid).
static int hier_split(lance_hip_ctx *ctx, int metric, const float *x, int d, const uint32_t *idx_host, size_t cluster_size, int cluster_k,
                      uint32_t max_iters, double tol, float bf_scaled, uint64_t seed, bool f16_arith, float *cdev, std::vector<uint32_t> &mem) {
  uint32_t *didx = ctx->scratch_t<uint32_t>("hier.idx", cluster_size);
  float *sub = ctx->scratch_t<float>("hier.sub", cluster_size * d);
  if (!didx || !sub) return LANCE_HIP_ENOMEM;
  LH_CHECK_HIP(hipMemcpyAsync(didx, idx_host, cluster_size * 4, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(gather_rows_u32_kernel, dim3((unsigned)cdiv(cluster_size * d, 256)), dim3(256), 0, ctx->stream, x, d, didx,
                     (int64_t)cluster_size, sub);
  LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  uint64_t sd = seed;
  LH_TRY(kmeans_train_batched(ctx, metric, sub, (int64_t)cluster_size, d, 0, d, cluster_k, 1, max_iters, tol, bf_scaled, false, &sd, cdev, nullptr,
                              nullptr, f16_arith));
  return hier_assign_host(ctx, metric, sub, (int64_t)cluster_size, d, cdev, cluster_k, mem, f16_arith);
}

int kmeans_train_hierarchical(lance_hip_ctx *ctx, int metric, const float *x, int64_t n, int d, int target_k, uint32_t max_iters,
                              double tol, float bf_scaled, int hierarchical_k, uint64_t seed, float *cent_out, uint32_t *n_out,
                              bool f16_arith) {
  uint64_t run = 0;
  const int initial_k = (int)std::min<int64_t>(std::min(hierarchical_k, target_k), n);
  float *cdev = ctx->scratch_t<float>("hier.cent", (size_t)std::max(hierarchical_k, initial_k) * d);
  if (!cdev) return LANCE_HIP_ENOMEM;
  uint64_t sd = seed + run++;
  LH_TRY(kmeans_train_batched(ctx, metric, x, n, d, 0, d, initial_k, 1, max_iters, tol, bf_scaled, false, &sd, cdev, nullptr, nullptr, f16_arith));
  std::vector<uint32_t> mem;
  LH_TRY(hier_assign_host(ctx, metric, x, n, d, cdev, initial_k, mem, f16_arith));
  std::vector<float> c0((size_t)initial_k * d);
  LH_CHECK_HIP(hipMemcpyAsync(c0.data(), cdev, c0.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
  LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  HHeap heap;
  size_t next_id = 0;
  for (int i = 0; i < initial_k; ++i) {
    HCluster c;
    for (int64_t r = 0; r < n; ++r)
      if (mem[r] == (uint32_t)i) c.idx.push_back((uint32_t)r);
    if (c.idx.empty()) continue;
    c.id = next_id++; c.finalized = false;
    c.centroid.assign(c0.begin() + (size_t)i * d, c0.begin() + (size_t)(i + 1) * d);
    heap.push(std::move(c));
  }
  std::vector<float> sc;
  while ((int)heap.d.size() < target_k) {
    if (heap.d.empty()) break;
    HCluster big = heap.pop();
    if (big.finalized || big.idx.size() <= 1) { heap.push(std::move(big)); break; }
    const size_t cluster_size = big.idx.size();
    const size_t remaining_k = (size_t)target_k - heap.d.size();
    size_t cluster_k;
    if (cluster_size <= (size_t)hierarchical_k) {
      cluster_k = std::min<size_t>(std::min<size_t>(2, remaining_k), cluster_size);
    } else {
      cluster_k = std::max<size_t>(std::min<size_t>(std::min<size_t>(cluster_size / hierarchical_k, remaining_k), hierarchical_k), 2);
    }
    sd = seed + run++;
    LH_TRY(hier_split(ctx, metric, x, d, big.idx.data(), cluster_size, (int)cluster_k, max_iters, tol, bf_scaled, sd, f16_arith, cdev, mem));
    bool all_same = true, have_first = false;
    uint32_t first = 0;
    for (size_t r = 0; r < cluster_size; ++r) {
      if (mem[r] == LANCE_HIP_NONE) continue;
      if (have_first) { if (mem[r] != first) all_same = false; } else { first = mem[r]; have_first = true; }
    }
    if (all_same) {
      big.finalized = true;
      heap.push(std::move(big));
      continue;
    }
    sc.resize(cluster_k * d);
    LH_CHECK_HIP(hipMemcpyAsync(sc.data(), cdev, sc.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < cluster_k; ++i) {
      HCluster c;
      for (size_t r = 0; r < cluster_size; ++r)
        if (mem[r] == (uint32_t)i) c.idx.push_back(big.idx[r]);
      if (c.idx.empty()) continue;
      c.id = next_id++; c.finalized = false;
      c.centroid.assign(sc.begin() + i * d, sc.begin() + (i + 1) * d);
      heap.push(std::move(c));
    }
  }
  std::sort(heap.d.begin(), heap.d.end(), [](const HCluster &a, const HCluster &b) { return a.id < b.id; });
  std::vector<float> flat;
  flat.reserve(heap.d.size() * d);
  for (auto &c : heap.d) flat.insert(flat.end(), c.centroid.begin(), c.centroid.end());
  LH_CHECK_HIP(hipMemcpyAsync(cent_out, flat.data(), flat.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  if (n_out) *n_out = (uint32_t)heap.d.size();
  return LANCE_HIP_OK;
}

}  // namespace lh

using namespace lh;

extern "C" {

int lance_hip_assign(lance_hip_ctx *ctx, int dtype, int metric, const void *x, uint64_t n, uint32_t d,
                     const void *centroids, uint32_t k, const float *bias, uint32_t *ids, float *dists) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(ctx && x && centroids && ids, "assign: NULL argument");
  LH_TRY(check_dtype(dtype, "assign"));
  LH_REQUIRE(d > 0 && k > 0, "assign: d and k must be > 0");
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  // f16: l2_scalar<f16,f32,16> / dot_scalar<f16,f32,32> widen every element before the arithmetic (l2.rs:128-159, dot.rs:91-102)
  const float *xf, *cf;
  LH_TRY(as_f32(ctx, dtype, x, (size_t)n * d, "f16.x", &xf));
  LH_TRY(as_f32(ctx, model_dtype(dtype), centroids, (size_t)k * d, "f16.cent", &cf));
  PairwiseArgs pa;
  pa.x = xf; pa.n = (int64_t)n; pa.ldx = d;
  pa.cent = cf; pa.k = (int)k;
  pa.bias = bias; pa.ids = ids; pa.dists = dists; pa.out_batch_stride = (int64_t)n;
  pa.lanes32 = dtype == LANCE_HIP_F16 && metric == LANCE_HIP_DOT && d > 16;
  LH_TRY(launch_assign(ctx, pa, (int)d, metric == LANCE_HIP_COSINE ? LANCE_HIP_L2 : metric, 1));
  LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return LANCE_HIP_OK;
}

// KMeans::new_with_params on f32 containers (f16_arith: the values are f16, see f16.h)
static int kmeans_train_impl(lance_hip_ctx *ctx, int metric, const float *x, uint64_t n, uint32_t d, uint32_t k, uint32_t max_iters,
                             double tol, float balance_factor, uint32_t hierarchical_k, const float *init, uint64_t seed, float *cent,
                             double *loss_out_host, uint32_t *iters_out_host, uint32_t *k_out_host, bool f16_arith) {
  const int km = metric == LANCE_HIP_COSINE ? LANCE_HIP_L2 : metric;
  // train_kmeans :1344: params.balance_factor /= data.len()
  const float bf = balance_factor / (float)n;
  if (k_out_host) *k_out_host = k;
  // new_with_params :1027: hierarchical clustering if k > 256 and hierarchical_k > 1 (initial centroids are ignored there)
  if (k > 256 && hierarchical_k > 1) {
    if (loss_out_host) *loss_out_host = 0.0;  // "Loss is not meaningful for hierarchical clustering" (:1001)
    if (iters_out_host) *iters_out_host = 0;
    return kmeans_train_hierarchical(ctx, km, x, (int64_t)n, (int)d, (int)k, max_iters, tol, bf, (int)hierarchical_k, seed, cent,
                                     k_out_host, f16_arith);
  }
  if (init && init != cent) LH_CHECK_HIP(hipMemcpyAsync(cent, init, (size_t)k * d * 4, hipMemcpyDeviceToDevice, ctx->stream));
  uint64_t seeds[1] = {seed};
  return kmeans_train_batched(ctx, km, x, (int64_t)n, d, 0, (int)d, (int)k, 1, max_iters, tol, bf, init != nullptr, seeds, cent,
                              loss_out_host, iters_out_host, f16_arith);
}

int lance_hip_kmeans_train_ex(lance_hip_ctx *ctx, int dtype, int metric, const void *x, uint64_t n, uint32_t d, uint32_t k,
                              uint32_t max_iters, double tol, float balance_factor, uint32_t hierarchical_k,
                              const void *init_centroids, uint64_t seed, void *centroids_out, double *loss_out_host,
                              uint32_t *iters_out_host, uint32_t *k_out_host) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(ctx && x && centroids_out, "kmeans_train: NULL argument");
  LH_TRY(check_dtype(dtype, "kmeans_train"));
  LH_REQUIRE(d > 0 && k > 0 && n > 0, "kmeans_train: empty problem");
  LH_REQUIRE(n >= k, "KMeans: training does not have sufficient data points: n(%llu) is smaller than k(%u)", (unsigned long long)n, k);
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  if (dtype == LANCE_HIP_F32)
    return kmeans_train_impl(ctx, metric, static_cast<const float *>(x), n, d, k, max_iters, tol, balance_factor, hierarchical_k,
                             static_cast<const float *>(init_centroids), seed, static_cast<float *>(centroids_out), loss_out_host,
                             iters_out_host, k_out_host, false);
  if (dtype == LANCE_HIP_I8) {   // Int8 columns train on the f32 conversion (lance-arrow lib.rs:284-306); the model is f32
    const float *xw;
    LH_TRY(as_f32(ctx, dtype, x, (size_t)n * d, "f16.x", &xw));
    return kmeans_train_impl(ctx, metric, xw, n, d, k, max_iters, tol, balance_factor, hierarchical_k,
                             static_cast<const float *>(init_centroids), seed, static_cast<float *>(centroids_out), loss_out_host,
                             iters_out_host, k_out_host, false);
  }
  // f16: KMeansAlgoFloat<Float16Type> -- widen, train with f16 M-step arithmetic, narrow the model
  const float *xf;
  LH_TRY(as_f32(ctx, dtype, x, (size_t)n * d, "f16.x", &xf));
  float *cw = ctx->scratch_t<float>("f16.kmeans_out", (size_t)k * d);
  float *iw = nullptr;
  if (!cw) return LANCE_HIP_ENOMEM;
  if (init_centroids) {
    iw = ctx->scratch_t<float>("f16.kmeans_init", (size_t)k * d);
    if (!iw) return LANCE_HIP_ENOMEM;
    LH_TRY(widen_into(ctx, model_dtype(dtype), init_centroids, (size_t)k * d, iw));
  }
  uint32_t kout = k;
  LH_TRY(kmeans_train_impl(ctx, metric, xf, n, d, k, max_iters, tol, balance_factor, hierarchical_k, iw, seed, cw, loss_out_host,
                           iters_out_host, &kout, true));
  if (k_out_host) *k_out_host = kout;
  LH_TRY(from_f32(ctx, model_dtype(dtype), cw, centroids_out, (size_t)kout * d));
  LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return LANCE_HIP_OK;
}

int lance_hip_kmeans_train(lance_hip_ctx *ctx, int dtype, int metric, const void *x, uint64_t n, uint32_t d,
                           uint32_t k, uint32_t max_iters, double tol, float balance_factor,
                           const void *init_centroids, uint64_t seed, void *centroids_out,
                           double *loss_out_host, uint32_t *iters_out_host) {
  lh::CtxLock _ctx_lock(ctx);
  // KMeansParams default hierarchical_k = 16 (kmeans.rs:92-103)
  return lance_hip_kmeans_train_ex(ctx, dtype, metric, x, n, d, k, max_iters, tol, balance_factor, 16, init_centroids, seed,
                                   centroids_out, loss_out_host, iters_out_host, nullptr);
}

int lance_hip_kmeans_estep_partial(lance_hip_ctx *ctx, int dtype, int metric, const void *x, uint64_t n, uint32_t d,
                                   const void *centroids, uint32_t k, const float *bias, float *buf, double *losses,
                                   float *radius, double *loss_out_host) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(ctx && x && centroids && buf, "kmeans_estep_partial: NULL argument");
  LH_REQUIRE(dtype == LANCE_HIP_F32, "kmeans_estep_partial: only f32 is implemented in this version");
  LH_REQUIRE(n < (1ull << 32) && k <= 4096, "kmeans_estep_partial: n or k too large for this version");
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  const size_t nn = n ? n : 1;
  uint32_t *ids = ctx->scratch_t<uint32_t>("kmeans.ids", nn);
  float *dists = ctx->scratch_t<float>("kmeans.dists", nn);
  uint32_t *starts = ctx->scratch_t<uint32_t>("kmeans.starts", (size_t)k + 1);
  uint32_t *sorted_rows = ctx->scratch_t<uint32_t>("kmeans.sorted", nn);
  char *stats_d = reinterpret_cast<char *>(ctx->scratch("kmeans.stats", (size_t)k * 16));
  if (!ids || !dists || !starts || !sorted_rows || !stats_d) return LANCE_HIP_ENOMEM;
  double *losses_d = reinterpret_cast<double *>(stats_d);
  float *radius_d = reinterpret_cast<float *>(stats_d + (size_t)k * 8);
  uint32_t *last_d = reinterpret_cast<uint32_t *>(stats_d + (size_t)k * 12);
  PairwiseArgs pa;
  pa.x = static_cast<const float *>(x); pa.n = (int64_t)n; pa.ldx = d;
  pa.cent = static_cast<const float *>(centroids); pa.k = (int)k; pa.bias = bias;
  pa.ids = ids; pa.dists = dists; pa.out_batch_stride = (int64_t)n;
  LH_TRY(launch_assign(ctx, pa, (int)d, metric == LANCE_HIP_COSINE ? LANCE_HIP_L2 : metric, 1));
  LH_TRY(stable_group(ctx, ids, (int64_t)n, (int64_t)n, (int)k, 1, starts, sorted_rows, (int64_t)n, nullptr));
  hipLaunchKernelGGL(kmeans_stats_kernel, dim3((unsigned)cdiv(k, 4), 1), dim3(256), 0, ctx->stream, dists, (int64_t)n, (int)k,
                     sorted_rows, (int64_t)n, starts, losses_d, radius_d, last_d, (const uint8_t *)nullptr);
  hipLaunchKernelGGL(kmeans_accumulate_kernel, dim3((unsigned)cdiv((uint64_t)k * d, 256), 1), dim3(256), 0, ctx->stream,
                     static_cast<const float *>(x), (int64_t)d, 0, (int)d, (int)k, sorted_rows, (int64_t)n, starts, buf, (int64_t)k * d,
                     (const uint8_t *)nullptr, 0, 0);
  hipLaunchKernelGGL(counts_to_float_kernel, dim3((unsigned)cdiv(k, 256)), dim3(256), 0, ctx->stream, starts, (int)k, buf + (size_t)k * d);
  LH_CHECK_HIP(hipGetLastError());
  if (losses) LH_CHECK_HIP(hipMemcpyAsync(losses, losses_d, (size_t)k * 8, hipMemcpyDeviceToDevice, ctx->stream));
  if (radius) LH_CHECK_HIP(hipMemcpyAsync(radius, radius_d, (size_t)k * 4, hipMemcpyDeviceToDevice, ctx->stream));
  if (loss_out_host) {
    std::vector<double> lh_(k);
    LH_CHECK_HIP(hipMemcpyAsync(lh_.data(), losses_d, (size_t)k * 8, hipMemcpyDeviceToHost, ctx->stream));
    LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    double tot = 0.0;
    for (uint32_t c = 0; c < k; ++c) tot = tot + lh_[c];
    *loss_out_host = tot;
  }
  LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return LANCE_HIP_OK;
}

// ---- row-sharded Lloyd iteration without host round trips (multi-GPU) ---------------------------------------
// One iteration = shard_estep (local E-step + local partial sums / losses / radii, enqueued) -> the caller's collectives
// on the same stream (all-reduce SUM of [k*d sums | k counts] and of the f64 losses, MAX of the radii) -> shard_update
// (finalise centroids, loss, balance factor, convergence, empty-cluster split with the shared seed, next bias; enqueued).
// Nothing synchronises with the host until shard_end.  Every rank executes the same update on the same reduced numbers, so
// all ranks hold identical centroids and take identical decisions.
struct KmShardState {
  KmState s;
  uint32_t active, pad;
};

__global__ __launch_bounds__(256) void kmeans_shard_update_kernel(KmShardState *__restrict__ st, const float *__restrict__ buf,
                                                                  const double *__restrict__ losses, const float *__restrict__ radius,
                                                                  float *__restrict__ cent, float *__restrict__ bias, int k, int d, int64_t n_total,
                                                                  float balance_factor_scaled, double tol, uint32_t it) {
  extern __shared__ __attribute__((aligned(8))) char km_smem[];
  double *lbs = reinterpret_cast<double *>(km_smem);
  uint32_t *sz = reinterpret_cast<uint32_t *>(lbs + k);
  __shared__ unsigned long long s_sq;
  __shared__ int s_empty;
  if (!st->active) return;
  if (threadIdx.x == 0) { s_sq = 0ull; s_empty = 0; }
  __syncthreads();
  // centroids = sums / counts (to_kmeans :410-418); counts travel as f32 in the reduced buffer
  for (int64_t g = threadIdx.x; g < (int64_t)k * d; g += 256) {
    const float cnt = buf[(int64_t)k * d + g / d];
    float v = buf[g];
    if (cnt > 0.0f) { const float norm = 1.0f / cnt; v *= norm; }
    cent[g] = v;
  }
  unsigned long long sq = 0ull;
  int empty = 0;
  for (int i = threadIdx.x; i < k; i += 256) {
    const uint32_t v = (uint32_t)buf[(int64_t)k * d + i];
    sz[i] = v; lbs[i] = losses[i];
    sq += (unsigned long long)v * v;
    empty |= v == 0;
  }
  atomicAdd(&s_sq, sq);
  if (empty) s_empty = 1;
  __syncthreads();
  if (threadIdx.x == 0) {
    KmState &s = st->s;
    s.iters += 1u;      // (counted on the device: the launch takes no per-iteration argument, so a block of iterations can be a HIP graph)
    uint64_t max_size = 0;
    int max_id = 0;
    for (int i = 0; i < k; ++i) if (sz[i] > max_size) { max_size = sz[i]; max_id = i; }   // first maximal cluster
    s.adjusted = (radius[max_id] - (float)lbs[max_id] / (float)sz[max_id]) / (float)n_total;
    const float size_loss = (float)(uint64_t)s_sq;
    const float balance_loss = s.bf_used * (size_loss - (float)((uint64_t)n_total * (uint64_t)n_total) / (float)k);
    double lsum = 0.0;
    for (int i = 0; i < k; ++i) lsum = lsum + lbs[i];
    s.last_loss = lsum + (double)balance_loss;
    if (s_empty) {
      const float eps = 1.0f / 1024.0f;
      for (int i = 0; i < k; ++i) {
        if (sz[i] != 0) continue;
        bool splittable = false;
        for (int t = 0; t < k; ++t) if (sz[t] >= 2) { splittable = true; break; }
        if (!splittable) break;
        int j = 0;
        for (;;) {
          const float p = ((float)sz[j] - 1.0f) / (float)(n_total - k);
          if (s.rng.next_f32() < p) break;
          j += 1;
          j %= k;
        }
        sz[i] = sz[j] / 2;
        sz[j] -= sz[i];
        for (int t = 0; t < d; ++t) {
          const float cj = cent[(int64_t)j * d + t];
          cent[(int64_t)i * d + t] = cj * ((t % 2 == 0) ? (1.0f + eps) : (1.0f - eps));
          cent[(int64_t)j * d + t] = cj * ((t % 2 == 0) ? (1.0f - eps) : (1.0f + eps));
        }
      }
    }
    if (fabs(s.loss - s.last_loss) < tol * s.last_loss) st->active = 0;
    else s.loss = s.last_loss;
    s.bf_used = s.adjusted < balance_factor_scaled ? s.adjusted : balance_factor_scaled;
  }
  __syncthreads();
  const float bf = st->s.bf_used;
  for (int i = threadIdx.x; i < k; i += 256) bias[i] = bf * (float)sz[i];
}

__global__ void kmeans_shard_init_kernel(KmShardState *st, float *bias, int k, float balance_factor_scaled, uint64_t seed) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    KmState &s = st->s;
    s.loss = DBL_MAX; s.last_loss = DBL_MAX; s.adjusted = FLT_MAX; s.iters = 0; s.pad = 0;
    s.bf_used = FLT_MAX < balance_factor_scaled ? FLT_MAX : balance_factor_scaled;
    s.rng.seed(seed ^ 0x5bd1e995ULL);
    st->active = 1; st->pad = 0;
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < k; i += gridDim.x * blockDim.x) bias[i] = 0.0f;
}

// a converged run must keep contributing ZEROS? no: every rank converges at the same iteration (same reduced numbers), and
// the caller stops issuing collectives once shard_end reports inactive; estep on an inactive state leaves the buffers as is.
__global__ void kmeans_shard_gate_kernel(const KmShardState *st, uint8_t *active_byte) { *active_byte = st->active ? 1 : 0; }

// kmeans_random_init's row choice (kmeans.rs:149-170 shape; the engine's seeded reservoir, rng.h) -- host only
int lance_hip_kmeans_init_indices(uint64_t n, uint32_t k, uint64_t seed, uint64_t *out_host) {
  LH_REQUIRE(out_host && n >= k, "kmeans_init_indices: need n >= k and an output array");
  kmeans_init_indices(n, k, seed, out_host);
  return LANCE_HIP_OK;
}

int lance_hip_kmeans_shard_begin(lance_hip_ctx *ctx, uint32_t k, float balance_factor_scaled, uint64_t seed, void *state, float *bias) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(ctx && state && bias, "kmeans_shard_begin: NULL argument");
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(kmeans_shard_init_kernel, dim3(4), dim3(256), 0, ctx->stream, static_cast<KmShardState *>(state), bias, (int)k,
                     balance_factor_scaled, seed);
  LH_CHECK_HIP(hipGetLastError());
  return LANCE_HIP_OK;
}

int lance_hip_kmeans_shard_estep(lance_hip_ctx *ctx, int metric, const float *x, uint64_t n, uint32_t d, const float *centroids, uint32_t k,
                                 const float *bias, const void *state, float *buf, double *losses, float *radius);
// the shard in the column's own element type: f16 / int8 rows are widened (exactly) into the context's scratch on every call -- a caller
// that iterates keeps its own f32 copy and uses the f32 entry point (lance_hip_kmeans_train_sharded_x widens once for the whole loop)
int lance_hip_kmeans_shard_estep_x(lance_hip_ctx *ctx, int dtype, int metric, const void *x, uint64_t n, uint32_t d, const float *centroids, uint32_t k,
                                   const float *bias, const void *state, float *buf, double *losses, float *radius) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(ctx && (n == 0 || x), "kmeans_shard_estep: NULL argument");
  LH_TRY(check_dtype(dtype, "kmeans_shard_estep"));
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  const float *xf;
  LH_TRY(as_f32(ctx, dtype, x, (size_t)n * d, "shard.x", &xf));
  return lance_hip_kmeans_shard_estep(ctx, metric, xf, n, d, centroids, k, bias, state, buf, losses, radius);
}

int lance_hip_kmeans_shard_estep(lance_hip_ctx *ctx, int metric, const float *x, uint64_t n, uint32_t d, const float *centroids, uint32_t k,
                                 const float *bias, const void *state, float *buf, double *losses, float *radius) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(ctx && centroids && buf && losses && radius && state && (n == 0 || x), "kmeans_shard_estep: NULL argument");
  LH_REQUIRE(n < (1ull << 32) && k <= 4096, "kmeans_shard_estep: n or k too large for this version");
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  const size_t nn = n ? n : 1;
  uint32_t *ids = ctx->scratch_t<uint32_t>("kmeans.ids", nn);
  float *dists = ctx->scratch_t<float>("kmeans.dists", nn);
  uint32_t *starts = ctx->scratch_t<uint32_t>("kmeans.starts", (size_t)k + 1);
  uint32_t *sorted_rows = ctx->scratch_t<uint32_t>("kmeans.sorted", nn);
  uint32_t *last_d = ctx->scratch_t<uint32_t>("kmeans.last", (size_t)k);
  uint8_t *act = ctx->scratch_t<uint8_t>("kmeans.active", 1);
  if (!ids || !dists || !starts || !sorted_rows || !last_d || !act) return LANCE_HIP_ENOMEM;
  hipLaunchKernelGGL(kmeans_shard_gate_kernel, dim3(1), dim3(1), 0, ctx->stream, static_cast<const KmShardState *>(state), act);
  PairwiseArgs pa;
  pa.x = x; pa.n = (int64_t)n; pa.ldx = d;
  pa.cent = centroids; pa.k = (int)k; pa.bias = bias;
  pa.ids = ids; pa.dists = dists; pa.out_batch_stride = (int64_t)n;
  pa.active = act;
  if (n > 0) LH_TRY(launch_assign(ctx, pa, (int)d, metric == LANCE_HIP_COSINE ? LANCE_HIP_L2 : metric, 1));
  LH_TRY(stable_group(ctx, ids, (int64_t)n, (int64_t)n, (int)k, 1, starts, sorted_rows, (int64_t)n, act));
  hipLaunchKernelGGL(kmeans_stats_kernel, dim3((unsigned)cdiv(k, 4), 1), dim3(256), 0, ctx->stream, dists, (int64_t)n, (int)k, sorted_rows,
                     (int64_t)n, starts, losses, radius, last_d, act);
  hipLaunchKernelGGL(kmeans_accumulate_kernel, dim3((unsigned)cdiv((uint64_t)k * d, 256), 1), dim3(256), 0, ctx->stream, x, (int64_t)d, 0,
                     (int)d, (int)k, sorted_rows, (int64_t)n, starts, buf, (int64_t)k * d, act, 0, 0);
  hipLaunchKernelGGL(counts_to_float_kernel, dim3((unsigned)cdiv(k, 256)), dim3(256), 0, ctx->stream, starts, (int)k, buf + (size_t)k * d);
  LH_CHECK_HIP(hipGetLastError());
  return LANCE_HIP_OK;
}

int lance_hip_kmeans_shard_update(lance_hip_ctx *ctx, void *state, const float *buf, const double *losses, const float *radius,
                                  float *centroids, float *bias, uint32_t k, uint32_t d, uint64_t n_total, float balance_factor_scaled,
                                  double tol, uint32_t it) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(ctx && state && buf && losses && radius && centroids && bias, "kmeans_shard_update: NULL argument");
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(kmeans_shard_update_kernel, dim3(1), dim3(256), (size_t)k * 12 + 8, ctx->stream, static_cast<KmShardState *>(state), buf,
                     losses, radius, centroids, bias, (int)k, (int)d, (int64_t)n_total, balance_factor_scaled, tol, it);
  LH_CHECK_HIP(hipGetLastError());
  return LANCE_HIP_OK;
}

int lance_hip_kmeans_shard_end(lance_hip_ctx *ctx, const void *state, double *loss_host, uint32_t *iters_host, int *active_host) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(ctx && state, "kmeans_shard_end: NULL argument");
  KmShardState h;
  LH_CHECK_HIP(hipMemcpyAsync(&h, state, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  if (loss_host) *loss_host = h.s.last_loss;
  if (iters_host) *iters_host = h.s.iters;
  if (active_host) *active_host = (int)h.active;
  return LANCE_HIP_OK;
}

int lance_hip_kmeans_split(lance_hip_ctx *ctx, int dtype, int metric, const float *x, uint64_t n, uint32_t d, const uint32_t *rows_host,
                           uint64_t n_rows, uint32_t k, uint32_t max_iters, double tol, float balance_factor_scaled, uint64_t seed,
                           float *centroids_out_host, uint32_t *membership_out_host) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(ctx && x && centroids_out_host && membership_out_host && (rows_host || n_rows == n), "kmeans_split: NULL argument");
  LH_REQUIRE(dtype == LANCE_HIP_F32 || dtype == LANCE_HIP_F16, "kmeans_split: f32 and f16 samples (Int8 columns train on their f32 copy)");
  LH_REQUIRE(n_rows > 0 && n_rows <= n && n < (1ull << 32) && k > 0 && k <= n_rows, "kmeans_split: %llu rows of %llu, k = %u", (unsigned long long)n_rows,
             (unsigned long long)n, k);
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  const float *xf = x;      // the sample as f32 (an f16 column's sample widened ONCE by the caller: thousands of splits read it)
  const int km = metric == LANCE_HIP_COSINE ? LANCE_HIP_L2 : metric;
  float *cdev = ctx->scratch_t<float>("hier.cent", (size_t)std::max<uint32_t>(k, 16) * d);
  if (!cdev) return LANCE_HIP_ENOMEM;
  std::vector<uint32_t> all, mem;
  if (!rows_host) { all.resize(n_rows); for (uint64_t i = 0; i < n_rows; ++i) all[i] = (uint32_t)i; rows_host = all.data(); }
  LH_TRY(hier_split(ctx, km, xf, (int)d, rows_host, (size_t)n_rows, (int)k, max_iters, tol, balance_factor_scaled, seed, dtype == LANCE_HIP_F16, cdev, mem));
  LH_CHECK_HIP(hipMemcpyAsync(centroids_out_host, cdev, (size_t)k * d * 4, hipMemcpyDeviceToHost, ctx->stream));
  LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  memcpy(membership_out_host, mem.data(), (size_t)n_rows * 4);
  return LANCE_HIP_OK;
}

int lance_hip_kmeans_finalize(lance_hip_ctx *ctx, int dtype, const float *buf, uint32_t k, uint32_t d, void *centroids_out) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(ctx && buf && centroids_out, "kmeans_finalize: NULL argument");
  LH_REQUIRE(dtype == LANCE_HIP_F32, "kmeans_finalize: only f32 is implemented in this version");
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(finalize_centroids_kernel, dim3((unsigned)cdiv((uint64_t)k * d, 256)), dim3(256), 0, ctx->stream, buf, (int)k,
                     (int)d, static_cast<float *>(centroids_out));
  LH_CHECK_HIP(hipGetLastError());
  LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return LANCE_HIP_OK;
}

}  // extern "C"

namespace lh {
// [rows][d] -> [m][rows][sd]: every sub-quantiser's training matrix contiguous.  The batched Lloyd kernels address problem b's row r as
// x + b * x_batch_off + r * ldx, so the sliced layout is just (ldx = sd, x_batch_off = rows * sd) to them -- but the E-step's row loads
// become 32 rows x 32 contiguous bytes instead of 32 rows 512 bytes apart, each sub-quantiser's 2 MB stays in the L2 of the XCDs that work
// on it (the 32 MB row-major matrix was re-fetched by the workgroups of all sixteen column slices: 50 iterations x 16 x 32 MB through the
// fabric), and the M-step's row gathers hit a compact matrix.  Same values, same row order: the trained codebook is bit-identical.
__global__ __launch_bounds__(256) void pq_slice_transpose_kernel(const f4 *__restrict__ x, int64_t rows, int d4, int sd4, f4 *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * d4) return;
  const int64_t row = i / d4;
  const int c4 = (int)(i - row * d4);
  const int b = c4 / sd4, e4 = c4 - b * sd4;
  out[((int64_t)b * rows + row) * sd4 + e4] = x[i];
}
}  // namespace lh

extern "C" {

int lance_hip_pq_train(lance_hip_ctx *ctx, int dtype, const void *residuals, uint64_t n, uint32_t d, uint32_t m,
                       uint32_t nbits, uint32_t max_iters, uint32_t sample_rate, uint64_t seed, void *codebook_out,
                       uint32_t *iters_out_host) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(ctx && residuals && codebook_out, "pq_train: NULL argument");
  LH_TRY(check_dtype(dtype, "pq_train"));
  LH_REQUIRE(m > 0 && d % m == 0, "num_sub_vectors must divide vector dimension %u, but got %u", d, m);
  LH_REQUIRE(nbits == 8 || nbits == 4, "ProductQuantization: num_bits %u not supported", nbits);
  const uint32_t kc = 1u << nbits;
  LH_REQUIRE(n >= kc, "Not enough rows to train PQ. Requires %u rows but only %llu available", kc, (unsigned long long)n);
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  // train_kmeans :1328-1340: slice to sample_rate * k rows
  uint64_t rows = n;
  if (rows > (uint64_t)sample_rate * kc) rows = (uint64_t)sample_rate * kc;
  std::vector<uint64_t> seeds(m);
  for (uint32_t i = 0; i < m; ++i) seeds[i] = seed + i;
  std::vector<double> loss(m);
  const bool f16 = dtype == LANCE_HIP_F16;
  const float *rf;
  LH_TRY(as_f32(ctx, dtype, residuals, (size_t)rows * d, "f16.x", &rf));
  float *cb = static_cast<float *>(codebook_out);
  if (f16) {
    cb = ctx->scratch_t<float>("f16.codebook_out", (size_t)m * kc * (d / m));
    if (!cb) return LANCE_HIP_ENOMEM;
  }
  // balance factor 0 (KMeansParams::new, pq/builder.rs:113-128); L2 always (builder.rs:455)
  const uint32_t sd = d / m;
  static const bool no_slices = getenv("LANCE_HIP_PQ_NO_SLICES") != nullptr;      // A/B: train on the row-major matrix
  int64_t ldx = d;
  int xoff = (int)sd;
  if (!no_slices && m > 1 && sd % 4 == 0 && (reinterpret_cast<uintptr_t>(rf) & 15) == 0 && (uint64_t)rows * sd < (1ull << 31)) {
    float *xt = ctx->scratch_t<float>("pq.slices", (size_t)rows * d);
    if (!xt) return LANCE_HIP_ENOMEM;
    hipLaunchKernelGGL(lh::pq_slice_transpose_kernel, dim3((unsigned)cdiv(rows * (uint64_t)(d / 4), 256)), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const f4 *>(rf), (int64_t)rows, (int)(d / 4), (int)(sd / 4), reinterpret_cast<f4 *>(xt));
    LH_CHECK_HIP(hipGetLastError());
    rf = xt; ldx = sd; xoff = (int)(rows * sd);
  }
  LH_TRY(kmeans_train_batched(ctx, LANCE_HIP_L2, rf, (int64_t)rows, ldx, xoff, (int)sd, (int)kc, (int)m, max_iters, 1e-4, 0.0f,
                              false, seeds.data(), cb, loss.data(), iters_out_host, f16));
  if (f16) {
    LH_TRY(from_f32(ctx, model_dtype(dtype), cb, codebook_out, (size_t)m * kc * (d / m)));
    LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  }
  return LANCE_HIP_OK;
}

}  // extern "C"
