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
#include <cfloat>
#include <cmath>
#include <vector>

#include "common.h"
#include "exact.cuh"
#include "kernels.h"
#include "rng.h"

#pragma clang fp contract(off)

namespace lh {

// one lane per (centroid, dim): sequential row-order sum, then *= 1/count (kmeans.rs:388-418)
__global__ __launch_bounds__(256) void kmeans_accumulate_kernel(const float *__restrict__ x, int64_t ldx, int x_batch_off,
                                                                int d, int k, const uint32_t *__restrict__ sorted_rows,
                                                                int64_t rows_stride, const uint32_t *__restrict__ starts,
                                                                float *__restrict__ cent, int64_t cent_batch_stride,
                                                                const uint8_t *__restrict__ active, int scale) {
  const int b = blockIdx.y;
  if (active && !active[b]) return;
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= (int64_t)k * d) return;
  const int c = (int)(g / d), dim = (int)(g % d);
  const uint32_t *st = starts + (int64_t)b * (k + 1);
  const uint32_t *rows = sorted_rows + (int64_t)b * rows_stride;
  const float *xb = x + (int64_t)b * x_batch_off + dim;
  const uint32_t s = st[c], e = st[c + 1];
  float acc = 0.0f;
  uint32_t i = s;
  for (; i + 8 <= e; i += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = xb[(int64_t)rows[i + u] * ldx];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  for (; i < e; ++i) acc += xb[(int64_t)rows[i] * ldx];
  const uint32_t cnt = e - s;
  if (scale && cnt > 0) {
    const float norm = 1.0f / (float)cnt;
    acc *= norm;
  }
  cent[(int64_t)b * cent_batch_stride + (int64_t)c * d + dim] = acc;
}

// one lane per centroid: loss (f64, row order), radius, index of the last member row.
__global__ __launch_bounds__(256) void kmeans_stats_kernel(const float *__restrict__ dists, int64_t dist_stride, int k,
                                                           const uint32_t *__restrict__ sorted_rows, int64_t rows_stride,
                                                           const uint32_t *__restrict__ starts, double *__restrict__ losses,
                                                           float *__restrict__ radius, uint32_t *__restrict__ last_row,
                                                           const uint8_t *__restrict__ active) {
  const int b = blockIdx.y;
  if (active && !active[b]) return;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= k) return;
  const uint32_t *st = starts + (int64_t)b * (k + 1);
  const uint32_t *rows = sorted_rows + (int64_t)b * rows_stride;
  const float *db = dists + (int64_t)b * dist_stride;
  double loss = 0.0;
  float rad = 0.0f;
  const uint32_t s = st[c], e = st[c + 1];
  uint32_t i = s;
  // gathers issued 8 at a time; the f64 chain itself stays in row order (kmeans.rs:274-277)
  for (; i + 8 <= e; i += 8) {
    float dv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) dv[u] = db[rows[i + u]];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      rad = fmaxf(rad, dv[u]);  // f32::max
      loss += (double)dv[u];
    }
  }
  for (; i < e; ++i) {
    const float dv = db[rows[i]];
    rad = fmaxf(rad, dv);
    loss += (double)dv;
  }
  losses[(int64_t)b * k + c] = loss;
  radius[(int64_t)b * k + c] = rad;
  last_row[(int64_t)b * k + c] = e > s ? rows[e - 1] : 0xFFFFFFFFu;
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

// split_clusters (kmeans.rs:174-207), host side on the (rare) iteration that has an empty cluster.
static void split_clusters_host(size_t n, std::vector<uint64_t> &cnts, float *centroids, size_t dim, Rng &rng) {
  const size_t k = cnts.size();
  const float eps = 1.0f / 1024.0f;
  for (size_t i = 0; i < k; i++) {
    if (cnts[i] == 0) {
      size_t j = 0;
      for (;;) {
        const float p = ((float)cnts[j] - 1.0f) / (float)(n - k);
        if (rng.next_f32() < p) break;
        j += 1;
        j %= k;
      }
      cnts[i] = cnts[j] / 2;
      cnts[j] -= cnts[i];
      for (size_t t = 0; t < dim; t++) {
        if (t % 2 == 0) {
          centroids[i * dim + t] = centroids[j * dim + t] * (1.0f + eps);
          centroids[j * dim + t] *= 1.0f - eps;
        } else {
          centroids[i * dim + t] = centroids[j * dim + t] * (1.0f - eps);
          centroids[j * dim + t] *= 1.0f + eps;
        }
      }
    }
  }
}

// Batched trainer.  x: [n][ldx] floats; problem b uses columns [b*x_batch_off, +d).
// cent: [B][k][d] (in/out: holds the initial centroids when have_init).
int kmeans_train_batched(lance_hip_ctx *ctx, int metric, const float *x, int64_t n, int64_t ldx, int x_batch_off, int d,
                         int k, int B, uint32_t max_iters, double tol, float balance_factor_scaled, bool have_init,
                         const uint64_t *seeds, float *cent, double *loss_out, uint32_t *iters_out) {
  LH_REQUIRE(n >= k, "KMeans: training does not have sufficient data points: n(%lld) is smaller than k(%d)", (long long)n, k);
  LH_REQUIRE(k <= 4096, "kmeans_train: k=%d > 4096 not supported in this version", k);
  LH_REQUIRE(metric == METRIC_L2 || metric == METRIC_DOT, "kmeans_train: metric must be L2 or Dot");
  // kmeans.rs:623-627
  if (n >= (int64_t)k * 512) n = (int64_t)k * 512;

  std::vector<Rng> split_rng(B);
  for (int b = 0; b < B; ++b) split_rng[b].seed(seeds[b] ^ 0x5bd1e995ULL);

  if (!have_init) {
    // kmeans_random_init: reservoir choose_multiple on the host, gather on the device
    std::vector<uint64_t> idx((size_t)B * k);
    for (int b = 0; b < B; ++b) kmeans_init_indices((uint64_t)n, (uint32_t)k, seeds[b], idx.data() + (size_t)b * k);
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
  // stats block: [losses f64 B*k][radius f32 B*k][last_row u32 B*k][starts copy u32 B*(k+1)]
  const size_t stats_bytes = (size_t)B * k * (8 + 4 + 4);
  char *stats_d = reinterpret_cast<char *>(ctx->scratch("kmeans.stats", stats_bytes));
  if (!ids || !dists || !starts || !sorted_rows || !bias || !active_d || !stats_d) return LANCE_HIP_ENOMEM;
  double *losses_d = reinterpret_cast<double *>(stats_d);
  float *radius_d = reinterpret_cast<float *>(stats_d + (size_t)B * k * 8);
  uint32_t *last_d = reinterpret_cast<uint32_t *>(stats_d + (size_t)B * k * 12);

  std::vector<char> stats_h(stats_bytes);
  std::vector<uint32_t> starts_h((size_t)B * (k + 1));
  std::vector<std::vector<uint64_t>> sizes(B, std::vector<uint64_t>(k, 0));
  std::vector<float> adjusted(B, FLT_MAX);
  std::vector<double> loss(B, DBL_MAX), last_loss(B, DBL_MAX);
  std::vector<uint8_t> active(B, 1);
  std::vector<uint32_t> iters(B, 0);
  std::vector<float> bias_h((size_t)B * k, 0.0f);
  std::vector<float> cent_h;
  const bool use_bias = balance_factor_scaled != 0.0f;
  int n_active = B;

  for (uint32_t it = 1; it <= max_iters && n_active > 0; ++it) {
    std::vector<float> bf_used(B, 0.0f);
    for (int b = 0; b < B; ++b) {
      const float bf = adjusted[b] < balance_factor_scaled ? adjusted[b] : balance_factor_scaled;  // f32::min
      bf_used[b] = bf;
      if (use_bias)
        for (int c = 0; c < k; ++c) bias_h[(size_t)b * k + c] = bf * (float)sizes[b][c];
    }
    LH_CHECK_HIP(hipMemcpyAsync(active_d, active.data(), B, hipMemcpyHostToDevice, ctx->stream));
    if (use_bias) LH_CHECK_HIP(hipMemcpyAsync(bias, bias_h.data(), bias_h.size() * 4, hipMemcpyHostToDevice, ctx->stream));

    PairwiseArgs pa;
    pa.x = x; pa.n = n; pa.ldx = ldx; pa.x_batch_off = x_batch_off;
    pa.cent = cent; pa.k = k; pa.cent_batch_stride = (int64_t)k * d;
    pa.bias = use_bias ? bias : nullptr; pa.bias_batch_stride = k;
    pa.ids = ids; pa.dists = dists; pa.out_batch_stride = n;
    pa.active = active_d;
    LH_TRY(launch_assign(ctx, pa, d, metric, B));
    LH_TRY(stable_group(ctx, ids, n, n, k, B, starts, sorted_rows, n, active_d));
    {
      ScopedTimer t(ctx, "kmeans_mstep");
      hipLaunchKernelGGL(kmeans_stats_kernel, dim3((unsigned)cdiv(k, 256), B), dim3(256), 0, ctx->stream, dists, n, k,
                         sorted_rows, n, starts, losses_d, radius_d, last_d, active_d);
      hipLaunchKernelGGL(kmeans_accumulate_kernel, dim3((unsigned)cdiv((uint64_t)k * d, 256), B), dim3(256), 0, ctx->stream,
                         x, ldx, x_batch_off, d, k, sorted_rows, n, starts, cent, (int64_t)k * d, active_d, 1);
    }
    LH_CHECK_HIP(hipGetLastError());
    LH_CHECK_HIP(hipMemcpyAsync(stats_h.data(), stats_d, stats_bytes, hipMemcpyDeviceToHost, ctx->stream));
    LH_CHECK_HIP(hipMemcpyAsync(starts_h.data(), starts, starts_h.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    const double *losses_h = reinterpret_cast<const double *>(stats_h.data());
    const float *radius_h = reinterpret_cast<const float *>(stats_h.data() + (size_t)B * k * 8);
    const uint32_t *last_h = reinterpret_cast<const uint32_t *>(stats_h.data() + (size_t)B * k * 12);

    for (int b = 0; b < B; ++b) {
      if (!active[b]) continue;
      iters[b] = it;
      const uint32_t *st = &starts_h[(size_t)b * (k + 1)];
      // compute_cluster_sizes: the running-max rule picks, among the clusters of maximal
      // size, the one whose last member comes first in row order.
      uint64_t max_size = 0;
      int max_id = 0;
      uint32_t max_last = 0xFFFFFFFFu;
      for (int c = 0; c < k; ++c) {
        sizes[b][c] = st[c + 1] - st[c];
        const uint32_t lr = last_h[(size_t)b * k + c];
        if (sizes[b][c] > max_size || (sizes[b][c] == max_size && max_size > 0 && lr < max_last)) {
          max_size = sizes[b][c]; max_id = c; max_last = lr;
        }
      }
      const double *lb = losses_h + (size_t)b * k;
      const float *rb = radius_h + (size_t)b * k;
      adjusted[b] = (rb[max_id] - (float)lb[max_id] / (float)sizes[b][max_id]) / (float)n;
      uint64_t size_loss_u = 0;
      for (int c = 0; c < k; ++c) size_loss_u += sizes[b][c] * sizes[b][c];
      const float size_loss = (float)size_loss_u;
      const float balance_loss = bf_used[b] * (size_loss - (float)((uint64_t)n * (uint64_t)n) / (float)k);
      double lsum = 0.0;
      for (int c = 0; c < k; ++c) lsum = lsum + lb[c];
      last_loss[b] = lsum + (double)balance_loss;

      bool any_empty = false;
      for (int c = 0; c < k; ++c) any_empty |= sizes[b][c] == 0;
      if (any_empty) {
        cent_h.resize((size_t)k * d);
        float *cb = cent + (size_t)b * k * d;
        LH_CHECK_HIP(hipMemcpyAsync(cent_h.data(), cb, cent_h.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
        LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        split_clusters_host((size_t)n, sizes[b], cent_h.data(), (size_t)d, split_rng[b]);
        LH_CHECK_HIP(hipMemcpyAsync(cb, cent_h.data(), cent_h.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
      }
      if (std::fabs(loss[b] - last_loss[b]) < tol * last_loss[b]) {
        active[b] = 0;
        --n_active;
      } else {
        loss[b] = last_loss[b];
      }
    }
  }
  for (int b = 0; b < B; ++b) {
    if (loss_out) loss_out[b] = last_loss[b];
    if (iters_out) iters_out[b] = iters[b];
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

}  // namespace lh

using namespace lh;

extern "C" {

int lance_hip_assign(lance_hip_ctx *ctx, int dtype, int metric, const void *x, uint64_t n, uint32_t d,
                     const void *centroids, uint32_t k, const float *bias, uint32_t *ids, float *dists) {
  LH_REQUIRE(ctx && x && centroids && ids, "assign: NULL argument");
  LH_REQUIRE(dtype == LANCE_HIP_F32, "assign: only f32 is implemented in this version");
  if (dtype != LANCE_HIP_F32) return LANCE_HIP_ENOTSUP;
  LH_REQUIRE(d > 0 && k > 0, "assign: d and k must be > 0");
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  PairwiseArgs pa;
  pa.x = static_cast<const float *>(x); pa.n = (int64_t)n; pa.ldx = d;
  pa.cent = static_cast<const float *>(centroids); pa.k = (int)k;
  pa.bias = bias; pa.ids = ids; pa.dists = dists; pa.out_batch_stride = (int64_t)n;
  LH_TRY(launch_assign(ctx, pa, (int)d, metric == LANCE_HIP_COSINE ? LANCE_HIP_L2 : metric, 1));
  LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return LANCE_HIP_OK;
}

int lance_hip_kmeans_train(lance_hip_ctx *ctx, int dtype, int metric, const void *x, uint64_t n, uint32_t d,
                           uint32_t k, uint32_t max_iters, double tol, float balance_factor,
                           const void *init_centroids, uint64_t seed, void *centroids_out,
                           double *loss_out_host, uint32_t *iters_out_host) {
  LH_REQUIRE(ctx && x && centroids_out, "kmeans_train: NULL argument");
  LH_REQUIRE(dtype == LANCE_HIP_F32, "kmeans_train: only f32 is implemented in this version");
  LH_REQUIRE(d > 0 && k > 0 && n > 0, "kmeans_train: empty problem");
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  float *cent = static_cast<float *>(centroids_out);
  if (init_centroids && init_centroids != centroids_out)
    LH_CHECK_HIP(hipMemcpyAsync(cent, init_centroids, (size_t)k * d * 4, hipMemcpyDeviceToDevice, ctx->stream));
  // train_kmeans :1344: params.balance_factor /= data.len()
  const float bf = balance_factor / (float)n;
  uint64_t seeds[1] = {seed};
  return kmeans_train_batched(ctx, metric == LANCE_HIP_COSINE ? LANCE_HIP_L2 : metric, static_cast<const float *>(x),
                              (int64_t)n, d, 0, (int)d, (int)k, 1, max_iters, tol, bf, init_centroids != nullptr, seeds,
                              cent, loss_out_host, iters_out_host);
}

int lance_hip_kmeans_estep_partial(lance_hip_ctx *ctx, int dtype, int metric, const void *x, uint64_t n, uint32_t d,
                                   const void *centroids, uint32_t k, const float *bias, float *buf, double *losses,
                                   float *radius, double *loss_out_host) {
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
  hipLaunchKernelGGL(kmeans_stats_kernel, dim3((unsigned)cdiv(k, 256), 1), dim3(256), 0, ctx->stream, dists, (int64_t)n, (int)k,
                     sorted_rows, (int64_t)n, starts, losses_d, radius_d, last_d, (const uint8_t *)nullptr);
  hipLaunchKernelGGL(kmeans_accumulate_kernel, dim3((unsigned)cdiv((uint64_t)k * d, 256), 1), dim3(256), 0, ctx->stream,
                     static_cast<const float *>(x), (int64_t)d, 0, (int)d, (int)k, sorted_rows, (int64_t)n, starts, buf, (int64_t)k * d,
                     (const uint8_t *)nullptr, 0);
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

int lance_hip_kmeans_finalize(lance_hip_ctx *ctx, int dtype, const float *buf, uint32_t k, uint32_t d, void *centroids_out) {
  LH_REQUIRE(ctx && buf && centroids_out, "kmeans_finalize: NULL argument");
  LH_REQUIRE(dtype == LANCE_HIP_F32, "kmeans_finalize: only f32 is implemented in this version");
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(finalize_centroids_kernel, dim3((unsigned)cdiv((uint64_t)k * d, 256)), dim3(256), 0, ctx->stream, buf, (int)k,
                     (int)d, static_cast<float *>(centroids_out));
  LH_CHECK_HIP(hipGetLastError());
  LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return LANCE_HIP_OK;
}

int lance_hip_pq_train(lance_hip_ctx *ctx, int dtype, const void *residuals, uint64_t n, uint32_t d, uint32_t m,
                       uint32_t nbits, uint32_t max_iters, uint32_t sample_rate, uint64_t seed, void *codebook_out,
                       uint32_t *iters_out_host) {
  LH_REQUIRE(ctx && residuals && codebook_out, "pq_train: NULL argument");
  LH_REQUIRE(dtype == LANCE_HIP_F32, "pq_train: only f32 is implemented in this version");
  LH_REQUIRE(m > 0 && d % m == 0, "num_sub_vectors must divide vector dimension %u, but got %u", d, m);
  LH_REQUIRE(nbits == 8, "pq_train: only num_bits=8 is implemented in this version (got %u)", nbits);
  const uint32_t kc = 1u << nbits;
  LH_REQUIRE(n >= kc, "Not enough rows to train PQ. Requires %u rows but only %llu available", kc, (unsigned long long)n);
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  // train_kmeans :1328-1340: slice to sample_rate * k rows
  uint64_t rows = n;
  if (rows > (uint64_t)sample_rate * kc) rows = (uint64_t)sample_rate * kc;
  std::vector<uint64_t> seeds(m);
  for (uint32_t i = 0; i < m; ++i) seeds[i] = seed + i;
  std::vector<double> loss(m);
  // balance factor 0 (KMeansParams::new, pq/builder.rs:113-128); L2 always (builder.rs:455)
  return kmeans_train_batched(ctx, LANCE_HIP_L2, static_cast<const float *>(residuals), (int64_t)rows, d, (int)(d / m),
                              (int)(d / m), (int)kc, (int)m, max_iters, 1e-4, 0.0f, false, seeds.data(),
                              static_cast<float *>(codebook_out), loss.data(), iters_out_host);
}

}  // extern "C"
