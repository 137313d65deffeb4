// build.hip -- the IvfTransformer chain and per-partition storage on the device.
//
//   NormalizeTransformer / normalize_fsl   lance-linalg kernels.rs:141-146,172-211
//   KeepFiniteVectors / is_finite          lance-index utils.rs:263-286
//   PartitionTransformer                   ivf/transform.rs:75-137  (assign kernel)
//   ResidualTransform / do_compute_residual residual.rs:58-102
//   PQTransformer / transform_impl         pq.rs:116-191            (batched assign kernel)
//   shuffle + per-partition storage        v3/shuffler.rs:105-218, builder.rs:685-846,
//                                          pq/storage.rs:183-290,430-449 (transpose)
// HBM layout of an index (ours, chosen for the GPU scan): rows grouped by partition in
// ascending input order; PQ codes ROW-major [n][m] so one lane fetches a row's m bytes with
// 16-byte loads (the reference's per-partition [m][n_p] transpose serves CPU SIMD gathers;
// import/export convert).
#include <vector>

#include <hip/hip_fp16.h>

#include "common.h"
#include "exact.cuh"
#include <cstdlib>

#include "index.h"
#include "kernels.h"

#pragma clang fp contract(off)

namespace lh {

// a4: l2_norm = sqrt(sequential sum of x^2 in T); out = x / l2_norm (kernels.rs:141-146).  One lane per row.
// F16: T = half::f16 (do_normalize_fsl::<Float16Type>) on f32 containers holding f16 values -- every operation of the `half`
// crate (2.7.1, x86_64) is the f32 operation rounded to binary16: powi(2) -> rh(x*x); `Sum for f16` adds the widened terms
// in f32 and rounds once; sqrt -> rh(sqrtf); x / l2_norm -> rh(x / norm).
template <bool F16>
__global__ __launch_bounds__(256) void normalize_kernel(const float *__restrict__ x, int64_t n, int d, float *__restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  const float *v = x + r * d;
  float acc = 0.0f;
  for (int i = 0; i < d; ++i) {
    float p = v[i] * v[i];
    if (F16) p = __half2float(__float2half_rn(p));
    acc = acc + p;
  }
  if (F16) acc = __half2float(__float2half_rn(acc));
  float norm = sqrtf(acc);
  if (F16) norm = __half2float(__float2half_rn(norm));
  float *o = out + r * d;
  for (int i = 0; i < d; ++i) {
    float qv = v[i] / norm;
    if (F16) qv = __half2float(__float2half_rn(qv));
    o[i] = qv;
  }
}

// The same arithmetic for long rows (C3: d = 1536), coalesced: a workgroup takes 64 rows and walks them in 64-column tiles staged
// through LDS (each row segment is one 256-byte read); lane r of the first wave adds row r's squares in element order -- the
// reference's sequential sum, untouched -- and in a second sweep every lane rescales the tile it loads.  With one lane per row
// reading its own row (normalize_kernel) every load instruction touches 64 cache lines: 44 ms per 1M x 1536 against the
// 18 GB / 4 TB/s = 4.5 ms this layout moves.
template <bool F16>
__global__ __launch_bounds__(256) void normalize_tiled_kernel(const float *__restrict__ x, int64_t n, int d, float *__restrict__ out,
                                                              uint16_t *__restrict__ plane16 = nullptr, int dp = 0, float *__restrict__ n2 = nullptr) {
  __shared__ float tile[64][65];
  __shared__ float norms[64];
  const int64_t row0 = (int64_t)blockIdx.x * 64;
  const int tid = threadIdx.x, col = tid & 63, rsub = tid >> 6;
  float acc = 0.0f;
  for (int c0 = 0; c0 < d; c0 += 64) {
#pragma unroll 4
    for (int j = 0; j < 16; ++j) {
      const int r = rsub + 4 * j;
      float v = 0.0f;
      if (row0 + r < n && c0 + col < d) v = x[(row0 + r) * d + c0 + col];
      tile[r][col] = v;
    }
    __syncthreads();
    if (tid < 64) {
      const int w = min(64, d - c0);
      for (int i = 0; i < w; ++i) {
        float p = tile[tid][i] * tile[tid][i];
        if (F16) p = __half2float(__float2half_rn(p));
        acc = acc + p;
      }
    }
    __syncthreads();
  }
  if (tid < 64) {
    if (F16) acc = __half2float(__float2half_rn(acc));
    float norm = sqrtf(acc);
    if (F16) norm = __half2float(__float2half_rn(norm));
    norms[tid] = norm;
  }
  __syncthreads();
  for (int c0 = 0; c0 < d; c0 += 64) {
#pragma unroll 4
    for (int j = 0; j < 16; ++j) {
      const int r = rsub + 4 * j;
      if (row0 + r < n && c0 + col < d) {
        float qv = x[(row0 + r) * d + c0 + col] / norms[r];
        if (F16) qv = __half2float(__float2half_rn(qv));
        out[(row0 + r) * d + c0 + col] = qv;
        // |qv| <= 1 (+ an ulp): x 2^14 is a normal binary16 down to |qv| = 2^-28
        if (plane16) plane16[(row0 + r) * dp + c0 + col] = __half_as_ushort(__float2half_rn(qv * 16384.0f));
      }
    }
  }
  if (plane16) {
    for (int i = tid; i < 64 * (dp - d); i += 256) {      // the plane's padding columns
      const int r = i / (dp - d), c = d + i % (dp - d);
      if (row0 + r < n) plane16[(row0 + r) * dp + c] = 0;
    }
    // a unit vector's squared norm, as an upper bound for the coarse quantiser's margin; a row without a direction (zero / non-finite) has NaN
    if (tid < 64 && row0 + tid < n) n2[row0 + tid] = (norms[tid] > 0.0f && norms[tid] < INFINITY) ? 1.0001f : __uint_as_float(0x7FC00000u);
  }
}

int launch_normalize_planes(lance_hip_ctx *ctx, const float *x, int64_t n, int d, float *out, bool f16, uint16_t *plane16, int dp, float *n2) {
  if (n <= 0) return LANCE_HIP_OK;
  LH_REQUIRE(d >= 64 && dp >= d && plane16 && n2, "normalize: the plane-writing form needs d >= 64");
  if (f16) hipLaunchKernelGGL(normalize_tiled_kernel<true>, dim3((unsigned)cdiv(n, 64)), dim3(256), 0, ctx->stream, x, n, d, out, plane16, dp, n2);
  else hipLaunchKernelGGL(normalize_tiled_kernel<false>, dim3((unsigned)cdiv(n, 64)), dim3(256), 0, ctx->stream, x, n, d, out, plane16, dp, n2);
  LH_CHECK_HIP(hipGetLastError());
  return LANCE_HIP_OK;
}

int launch_normalize(lance_hip_ctx *ctx, const float *x, int64_t n, int d, float *out, bool f16) {
  if (n <= 0) return LANCE_HIP_OK;
  static const bool no_tiled = getenv("LANCE_HIP_NO_TILED_NORMALIZE") != nullptr;
  if (d >= 64 && !no_tiled) {
    if (f16) hipLaunchKernelGGL(normalize_tiled_kernel<true>, dim3((unsigned)cdiv(n, 64)), dim3(256), 0, ctx->stream, x, n, d, out);
    else hipLaunchKernelGGL(normalize_tiled_kernel<false>, dim3((unsigned)cdiv(n, 64)), dim3(256), 0, ctx->stream, x, n, d, out);
  } else {
    if (f16) hipLaunchKernelGGL(normalize_kernel<true>, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, ctx->stream, x, n, d, out);
    else hipLaunchKernelGGL(normalize_kernel<false>, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, ctx->stream, x, n, d, out);
  }
  LH_CHECK_HIP(hipGetLastError());
  return LANCE_HIP_OK;
}

// KeepFiniteVectors: rows with any non-finite element get part id NONE.
__global__ __launch_bounds__(256) void finite_mask_kernel(const float *__restrict__ x, int64_t n, int d, uint32_t *__restrict__ part_ids) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  const float *v = x + r * d;
  bool ok = true;
  for (int i = 0; i < d; ++i) ok &= isfinite(v[i]);
  if (!ok) part_ids[r] = LANCE_HIP_NONE;
}

// a10: out[r] = x[r] - centroids[part_ids[r]] ; coalesced over the flattened [n*d] range.
__global__ __launch_bounds__(256) void residual_kernel(const float *__restrict__ x, int64_t n, int d,
                                                       const float *__restrict__ cent, const uint32_t *__restrict__ part_ids,
                                                       float *__restrict__ out, int f16) {
  const int64_t total = n * d;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)gridDim.x * 256) {
    const int64_t r = g / d;
    const int t = (int)(g - r * d);
    const uint32_t p = part_ids[r];
    float v = p == LANCE_HIP_NONE ? 0.0f : x[g] - cent[(int64_t)p * d + t];
    if (f16) v = __half2float(__float2half_rn(v));  // `*v - *cent` in half::f16 (residual.rs:96)
    out[g] = v;
  }
}

__global__ __launch_bounds__(256) void gather_codes_kernel(const uint8_t *__restrict__ codes, const uint64_t *__restrict__ row_ids,
                                                           const uint32_t *__restrict__ perm, int64_t n_out, int m,
                                                           uint8_t *__restrict__ codes_out, uint64_t *__restrict__ row_ids_out) {
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= n_out) return;
  const uint32_t r = perm[s];
  row_ids_out[s] = row_ids ? row_ids[r] : (uint64_t)r;
  const uint8_t *src = codes + (int64_t)r * m;
  uint8_t *dst = codes_out + s * m;
  if ((m & 15) == 0 && ((reinterpret_cast<uintptr_t>(codes) | reinterpret_cast<uintptr_t>(codes_out)) & 15) == 0) {
    for (int i = 0; i < m; i += 16) *reinterpret_cast<uint4 *>(dst + i) = *reinterpret_cast<const uint4 *>(src + i);
  } else {
    for (int i = 0; i < m; ++i) dst[i] = src[i];
  }
}

__device__ __forceinline__ uint32_t find_partition(const uint32_t *__restrict__ offs, int nlist, uint32_t slot) {
  // largest p with offs[p] <= slot
  int lo = 0, hi = nlist;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (offs[mid] <= slot) lo = mid; else hi = mid;
  }
  return (uint32_t)lo;
}

// transposed per-partition blocks <-> row-major.  dir 0: transposed -> row-major, 1: reverse
__global__ __launch_bounds__(256) void retile_codes_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int64_t n,
                                                           int m, const uint32_t *__restrict__ offs, int nlist, int dir) {
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= n * m) return;
  const uint32_t slot = (uint32_t)(g / m);
  const int mm = (int)(g % m);
  const uint32_t p = find_partition(offs, nlist, slot);
  const uint32_t off = offs[p], np = offs[p + 1] - off;
  const int64_t t = (int64_t)off * m + (int64_t)mm * np + (slot - off);
  if (dir == 0) out[g] = in[t]; else out[t] = in[g];
}

static int dev_dup(lance_hip_ctx *ctx, const void *src, size_t bytes, void **out) {
  void *p = nullptr;
  hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
  if (e != hipSuccess) {
    set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return LANCE_HIP_ENOMEM;
  }
  if (src && bytes) {
    e = hipMemcpyAsync(p, src, bytes, hipMemcpyDefault, ctx->stream);
    if (e != hipSuccess) {
      (void)hipFree(p);
      set_error("hipMemcpyAsync failed: %s", hipGetErrorString(e));
      return LANCE_HIP_ERUNTIME;
    }
  }
  *out = p;
  return LANCE_HIP_OK;
}

static int check_pq_params(uint32_t d, uint32_t m, uint32_t nbits) {
  LH_REQUIRE(m > 0 && d % m == 0, "num_sub_vectors must divide vector dimension %u, but got %u", d, m);
  LH_REQUIRE(nbits == 8 || nbits == 4, "ProductQuantization: num_bits %u not supported", nbits);   // pq.rs:104-112
  LH_REQUIRE(!(nbits == 4 && m % 2 != 0), "PQ: num_sub_vectors must be divisible by 2 for num_bits=4, but got %u", m);  // pq.rs:134-142
  return LANCE_HIP_OK;
}

int launch_residual(lance_hip_ctx *ctx, const float *x, int64_t n, int d, const float *cent, const uint32_t *part_ids, float *out, bool f16) {
  if (n == 0) return LANCE_HIP_OK;
  const unsigned grid = (unsigned)std::min<uint64_t>(cdiv((uint64_t)n * d, 256), 65536);
  hipLaunchKernelGGL(residual_kernel, dim3(grid), dim3(256), 0, ctx->stream, x, n, d, cent, part_ids, out, f16 ? 1 : 0);
  LH_CHECK_HIP(hipGetLastError());
  return LANCE_HIP_OK;
}

// 4-bit packing, pq.rs:168-172: byte b = (code[2b+1] << 4) | code[2b]
__global__ __launch_bounds__(256) void pack_nibbles_kernel(const uint8_t *__restrict__ in, int64_t n, int m, uint8_t *__restrict__ out) {
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= n * (m / 2)) return;
  const int64_t r = g / (m / 2);
  const int b = (int)(g % (m / 2));
  out[g] = (uint8_t)((in[r * m + 2 * b + 1] << 4) | in[r * m + 2 * b]);
}

int pq_encode_launch(lance_hip_ctx *ctx, int metric, const float *x, int64_t n, int d, const float *codebook, int m,
                     uint8_t *codes, int nbits, bool lanes32) {
  const int kc = 1 << nbits;
  uint8_t *dst = codes;
  if (nbits == 4) {
    dst = ctx->scratch_t<uint8_t>("encode.codes8", (size_t)n * m);
    if (!dst) return LANCE_HIP_ENOMEM;
  }
  PairwiseArgs pa;
  pa.x = x; pa.n = n; pa.ldx = d; pa.x_batch_off = d / m;
  pa.cent = codebook; pa.k = kc; pa.cent_batch_stride = (int64_t)kc * (d / m);
  pa.codes = dst; pa.codes_ld = m;
  pa.lanes32 = lanes32;      // f16 sub-vectors of more than 16 elements under dot: dot_scalar::<f16, f32, 32> (dot.rs:91-102)
  LH_TRY(launch_assign(ctx, pa, d / m, metric, m));
  if (nbits == 4 && n > 0) {
    hipLaunchKernelGGL(pack_nibbles_kernel, dim3((unsigned)cdiv((uint64_t)n * (m / 2), 256)), dim3(256), 0, ctx->stream, dst, n, m, codes);
    LH_CHECK_HIP(hipGetLastError());
  }
  return LANCE_HIP_OK;
}

}  // namespace lh

using namespace lh;

lance_hip_index::~lance_hip_index() {
  (void)hipSetDevice(device);
  if (centroids) (void)hipFree(centroids);
  if (codebook) (void)hipFree(codebook);
  if (cb_mean) (void)hipFree(cb_mean);
  if (pt) {
    if (pt->g) (void)hipFree(pt->g);
    if (pt->cen_t) (void)hipFree(pt->cen_t);
    if (pt->row_beta) (void)hipFree(pt->row_beta);
    if (pt->beta_min) (void)hipFree(pt->beta_min);
    if (pt->beta_abs) (void)hipFree(pt->beta_abs);
    if (pt->beta_mean) (void)hipFree(pt->beta_mean);
    delete pt;
  }
  if (cq) {
    if (cq->cpl) (void)hipFree(cq->cpl);
    if (cq->maxbits) (void)hipFree(cq->maxbits);
    delete cq;
  }
  if (ms) {
    if (ms->cbh) (void)hipFree(ms->cbh);
    if (ms->cbn2) (void)hipFree(ms->cbn2);
    if (ms->row_cn2) (void)hipFree(ms->row_cn2);
    if (ms->cmaxp) (void)hipFree(ms->cmaxp);
    delete ms;
  }
  if (part_offsets) (void)hipFree(part_offsets);
  if (codes) (void)hipFree(codes);
  if (row_ids) (void)hipFree(row_ids);
  if (vectors) (void)hipFree(vectors);
  if (flat_items) (void)hipFree(flat_items);
  if (raw_u8) (void)hipFree(raw_u8);
}

// tail of lance_hip_ivfpq_encode: the optional loss and the synchronisation
static int encode_finish(lance_hip_ctx *ctx, uint64_t n, uint32_t nlist, const float *dists, const uint32_t *part_ids, double *loss_out_host) {
  LH_CHECK_HIP(hipGetLastError());
  if (loss_out_host) {
    // sum of the assignment distances (compute_partitions, kmeans.rs:1276-1290): f64, host side
    std::vector<float> dh(n);
    std::vector<uint32_t> ph(n);
    LH_CHECK_HIP(hipMemcpyAsync(dh.data(), dists, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    LH_CHECK_HIP(hipMemcpyAsync(ph.data(), part_ids, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<double> losses(nlist, 0.0);
    for (uint64_t r = 0; r < n; ++r)
      if (ph[r] != LANCE_HIP_NONE) losses[ph[r]] += (double)dh[r];
    double tot = 0.0;
    for (uint32_t c = 0; c < nlist; ++c) tot = tot + losses[c];
    *loss_out_host = tot;
  }
  LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return LANCE_HIP_OK;
}

extern "C" {

int lance_hip_normalize(lance_hip_ctx *ctx, int dtype, const void *x, uint64_t n, uint32_t d, void *out) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(ctx && x && out, "normalize: NULL argument");
  LH_REQUIRE(dtype == LANCE_HIP_F32 || dtype == LANCE_HIP_F16, "normalize: f32 and f16 columns (normalize_fsl accepts float arrays only, kernels.rs:170-186)");
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  if (n == 0) return LANCE_HIP_OK;
  if (dtype == LANCE_HIP_F16) {   // f16 in, f16 out, f16 arithmetic
    const float *xf;
    LH_TRY(as_f32(ctx, dtype, x, (size_t)n * d, "f16.x", &xf));
    float *of = ctx->scratch_t<float>("f16.normalize_out", (size_t)n * d);
    if (!of) return LANCE_HIP_ENOMEM;
    LH_TRY(launch_normalize(ctx, xf, (int64_t)n, (int)d, of, true));
    LH_TRY(from_f32(ctx, dtype, of, out, (size_t)n * d));
  } else {
    LH_TRY(launch_normalize(ctx, static_cast<const float *>(x), (int64_t)n, (int)d, static_cast<float *>(out), false));
  }
  LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return LANCE_HIP_OK;
}

int lance_hip_residual(lance_hip_ctx *ctx, int dtype, const void *x, uint64_t n, uint32_t d, const void *centroids,
                       const uint32_t *part_ids, void *out) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(ctx && x && centroids && part_ids && out, "residual: NULL argument");
  LH_TRY(check_dtype(dtype, "residual"));
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  if (n == 0) return LANCE_HIP_OK;
  const bool f16 = dtype == LANCE_HIP_F16;
  const float *xf, *cf;
  LH_TRY(as_f32(ctx, dtype, x, (size_t)n * d, "f16.x", &xf));
  // the number of centroids is not part of this call: widen lazily is impossible, so f16 callers pass
  // centroids through lance_hip_ivfpq_encode; here the f16 centroid table is bounded by max(part_ids)+1
  uint32_t kmax = 0;
  if (f16) {
    std::vector<uint32_t> ph(n);
    LH_CHECK_HIP(hipMemcpyAsync(ph.data(), part_ids, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    for (uint64_t r = 0; r < n; ++r)
      if (ph[r] != LANCE_HIP_NONE) kmax = std::max(kmax, ph[r] + 1);
  }
  LH_TRY(as_f32(ctx, model_dtype(dtype), centroids, (size_t)kmax * d, "f16.cent", &cf));
  float *of = static_cast<float *>(out);
  if (f16) {
    of = ctx->scratch_t<float>("f16.residual_out", (size_t)n * d);
    if (!of) return LANCE_HIP_ENOMEM;
  }
  LH_TRY(launch_residual(ctx, xf, (int64_t)n, (int)d, cf, part_ids, of, f16));
  if (f16) LH_TRY(from_f32(ctx, dtype, of, out, (size_t)n * d));
  LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return LANCE_HIP_OK;
}

int lance_hip_pq_encode(lance_hip_ctx *ctx, int dtype, int metric, const void *x, uint64_t n, uint32_t d,
                        const void *codebook, uint32_t m, uint32_t nbits, uint8_t *codes) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(ctx && x && codebook && codes, "pq_encode: NULL argument");
  LH_TRY(check_dtype(dtype, "pq_encode"));
  LH_TRY(check_pq_params(d, m, nbits));
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  // f16 columns: L2 takes l2_scalar::<f16, f32, 16> (as f32 does); dot products of f16 sub-vectors are dot_scalar::<f16, f32, 32>
  // (dot.rs:91-102), which for sub-vectors of up to 16 elements is the same sequence of additions as the 16-lane form
  const float *xf, *cbf;
  LH_TRY(as_f32(ctx, dtype, x, (size_t)n * d, "f16.x", &xf));
  LH_TRY(as_f32(ctx, model_dtype(dtype), codebook, ((size_t)1 << nbits) * d, "f16.codebook", &cbf));
  LH_TRY(pq_encode_launch(ctx, metric == LANCE_HIP_COSINE ? LANCE_HIP_L2 : metric, xf, (int64_t)n, (int)d, cbf, (int)m, codes, (int)nbits,
                          dtype == LANCE_HIP_F16 && metric == LANCE_HIP_DOT && d / m > 16));
  LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return LANCE_HIP_OK;
}

int lance_hip_ivfpq_encode(lance_hip_ctx *ctx, int dtype, int metric, const void *x, uint64_t n, uint32_t d,
                           const void *centroids, uint32_t nlist, const void *codebook, uint32_t m, uint32_t nbits,
                           uint32_t *part_ids, uint8_t *codes, double *loss_out_host) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(ctx && x && centroids && codebook && part_ids && codes, "ivfpq_encode: NULL argument");
  LH_TRY(check_dtype(dtype, "ivfpq_encode"));
  LH_TRY(check_pq_params(d, m, nbits));
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  if (n == 0) { if (loss_out_host) *loss_out_host = 0.0; return LANCE_HIP_OK; }
  const bool f16 = dtype == LANCE_HIP_F16;
  const float *xs = nullptr, *centf, *cbf;
  LH_TRY(as_f32(ctx, model_dtype(dtype), centroids, (size_t)nlist * d, "f16.cent", &centf));
  LH_TRY(as_f32(ctx, model_dtype(dtype), codebook, ((size_t)1 << nbits) * d, "f16.codebook", &cbf));
  centroids = centf; codebook = cbf;
  const int scan_metric = metric == LANCE_HIP_COSINE ? LANCE_HIP_L2 : metric;
  float *dists = ctx->scratch_t<float>("encode.dists", (size_t)n);
  if (!dists) return LANCE_HIP_ENOMEM;
  PairwiseArgs pa;
  pa.n = (int64_t)n; pa.ldx = d;
  pa.cent = centf; pa.k = (int)nlist;
  pa.ids = part_ids; pa.dists = dists; pa.out_batch_stride = (int64_t)n;
  pa.check_finite = true;  // KeepFiniteVectors fused into the assign kernel
  pa.lanes32 = f16 && metric == LANCE_HIP_DOT && d > 16;   // coarse quantiser of an f16 column under dot: dot_scalar::<f16, f32, 32>
  // Native route (L2 / dot): the MFMA assign kernels and the fused residual + encode kernel read the rows in the column's own
  // element type -- no f32 copy of the column, no residual array.  Cosine normalises first and takes the staged route.
  // (round 3: the sub-quantiser argmin of the fused encode runs on the matrix cores when the shape allows -- pq_mfma.hip)
  // Round 6: the shapes of the north-star configurations (d <= 128, sub-dimension 4 / 8, 8-bit codes) take ONE kernel for the whole chain
  // (xform_fused.hip): the rows are read once, in their own element type.  Cosine normalises into an f32 copy first and runs the same
  // kernel on that (L2 on unit vectors, residual rounded to binary16 for f16 columns).
  if (metric != LANCE_HIP_COSINE &&
      xform_fused_supported(dtype, scan_metric, (int)d, (int)m, (int)nbits, (int64_t)n, (int)nlist, x, centf, cbf, codes, pa.lanes32)) {
    LH_TRY(launch_xform_fused(ctx, dtype, scan_metric, x, (int64_t)n, (int)d, centf, (int)nlist, cbf, (int)m, part_ids, dists, codes, f16));
    return encode_finish(ctx, n, nlist, dists, part_ids, loss_out_host);
  }
  const bool mfma_enc = metric != LANCE_HIP_COSINE && pq_mfma_encode_supported(dtype, (int)d, (int)m, (int)nbits, x, centf, cbf, (int64_t)n);
  const bool native = metric != LANCE_HIP_COSINE && (mfma_enc || encode_fused_supported(dtype, (int)d, (int)m, (int)nbits, x, centf, cbf));
  bool assign_native = false;
  if (native) {
    pa.x_native = x; pa.x_dtype = dtype;
    if (dtype == LANCE_HIP_F32) pa.x = static_cast<const float *>(x);
    assign_native = dtype == LANCE_HIP_F32 || assign_reads_native(pa, (int)d, 1);
  }
  if (!assign_native) {
    LH_TRY(as_f32(ctx, dtype, x, (size_t)n * d, "f16.x", &xs));
    pa.x_native = nullptr; pa.x = xs;
  }
  if (metric == LANCE_HIP_COSINE) {
    float *xn = ctx->scratch_t<float>("encode.norm", (size_t)n * d);
    if (!xn) return LANCE_HIP_ENOMEM;
    // long rows: the normalise kernel also leaves the unit rows as a binary16 plane for the K-tiled coarse quantiser (one f16 matrix product
    // instead of three bf16 ones + a separate pass that splits the rows: mfma_assign.hip)
    static const bool no_f16 = getenv("LANCE_HIP_NO_MFMA_F16") != nullptr;
    const int dp16 = ((int)d + 31) / 32 * 32;
    uint16_t *plane16 = nullptr;
    float *plane_n2 = nullptr;
    if (!no_f16 && d > 128 && d <= 4096 && n >= 2048 && nlist >= 64 && (uint64_t)n * dp16 * 2 <= (16ull << 30)) {
      plane16 = static_cast<uint16_t *>(ctx->scratch_exact("encode.plane16", (size_t)n * dp16 * 2));
      plane_n2 = plane16 ? ctx->scratch_t<float>("encode.plane_n2", (size_t)n) : nullptr;
      if (!plane_n2) plane16 = nullptr;
    }
    if (plane16) {
      LH_TRY(launch_normalize_planes(ctx, xs, (int64_t)n, (int)d, xn, f16, plane16, dp16, plane_n2));
      pa.x_plane16 = plane16; pa.x_plane_n2 = plane_n2; pa.x_plane_dp = dp16;
    } else {
      LH_TRY(launch_normalize(ctx, xs, (int64_t)n, (int)d, xn, f16));
    }
    xs = xn; pa.x = xs;
    if (xform_fused_supported(LANCE_HIP_F32, LANCE_HIP_L2, (int)d, (int)m, (int)nbits, (int64_t)n, (int)nlist, xn, centf, cbf, codes, false)) {
      LH_TRY(launch_xform_fused(ctx, LANCE_HIP_F32, LANCE_HIP_L2, xn, (int64_t)n, (int)d, centf, (int)nlist, cbf, (int)m, part_ids, dists, codes, f16));
      return encode_finish(ctx, n, nlist, dists, part_ids, loss_out_host);
    }
  }
  LH_TRY(launch_assign(ctx, pa, (int)d, scan_metric, 1));
  // long rows (d > 128) that exist as f32 -- an f32 column, or the f32 copy the steps above made: residual + encode in one kernel per
  // 128-column block (xform_fused.hip: xf_tail_kernel), no residual array (round 6)
  const float *tail_x = xs ? xs : (dtype == LANCE_HIP_F32 ? static_cast<const float *>(x) : nullptr);
  if (tail_x && xform_tail_supported((int)d, (int)m, (int)nbits, (int64_t)n, tail_x, centf, cbf)) {
    LH_TRY(launch_xform_tail(ctx, tail_x, (int64_t)n, (int)d, centf, part_ids, scan_metric == LANCE_HIP_L2 ? 1 : 0, f16, cbf, (int)m, codes));
    return encode_finish(ctx, n, nlist, dists, part_ids, loss_out_host);
  }
  if (native) {
    if (mfma_enc) LH_TRY(launch_pq_mfma_encode(ctx, dtype, x, (int64_t)n, (int)d, centf, part_ids, scan_metric == LANCE_HIP_L2 ? 1 : 0, cbf, (int)m, codes));
    else LH_TRY(launch_encode_fused(ctx, dtype, x, (int64_t)n, (int)d, centf, part_ids, scan_metric == LANCE_HIP_L2 ? 1 : 0, cbf, (int)m, codes));
  } else {
    const float *enc_in = xs;
    if (scan_metric == LANCE_HIP_L2) {
      float *res = ctx->scratch_t<float>("encode.residual", (size_t)n * d);
      if (!res) return LANCE_HIP_ENOMEM;
      LH_TRY(launch_residual(ctx, xs, (int64_t)n, (int)d, centf, part_ids, res, f16));
      enc_in = res;
    }
    // the quantizer is built with DistanceType::L2 whatever the index metric (lance/src/index/vector/builder.rs:456) and
    // ProductQuantizer::transform encodes with the quantizer's distance type (pq.rs:143,165): L2-nearest codeword, dot too
    LH_TRY(pq_encode_launch(ctx, LANCE_HIP_L2, enc_in, (int64_t)n, (int)d, cbf, (int)m, codes, (int)nbits, false));
  }
  return encode_finish(ctx, n, nlist, dists, part_ids, loss_out_host);
}

static int index_alloc_common(lance_hip_ctx *ctx, int dtype, int metric, uint32_t d, const void *centroids, uint32_t nlist,
                              const void *codebook, uint32_t m, uint32_t nbits, lance_hip_index **out) {
  LH_REQUIRE(ctx && centroids && codebook && out, "index: NULL argument");
  LH_TRY(check_dtype(dtype, "index"));
  // f16 columns under dot: the table entries are dot products of f16 sub-vectors (32-lane dot_scalar, dot.rs:91-102) -- identical
  // to the 16-lane form up to 16 elements (the partition-major kernels' sub-dimensions); longer sub-vectors take the query-major
  // kernels, whose table build switches to the 32-lane order (search.hip: lut_entry_rt)
  LH_REQUIRE(metric == LANCE_HIP_L2 || metric == LANCE_HIP_COSINE || metric == LANCE_HIP_DOT, "index: bad metric %d", metric);
  LH_REQUIRE(nlist > 0 && nlist <= 65536, "index: nlist=%u not supported in this version (1..65536)", nlist);
  LH_TRY(check_pq_params(d, m, nbits));
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  auto *ix = new lance_hip_index();
  ix->device = ctx->device; ix->metric = metric; ix->d = d; ix->nlist = nlist; ix->m = m; ix->nbits = nbits;
  ix->dtype = dtype;
  // the index keeps f32 copies of the model (f16 widens exactly); the scan rounds the residual query to f16 when dtype is f16
  int r = dev_dup(ctx, nullptr, (size_t)nlist * d * 4, reinterpret_cast<void **>(&ix->centroids));
  if (r == LANCE_HIP_OK) r = widen_into(ctx, model_dtype(dtype), centroids, (size_t)nlist * d, ix->centroids);
  const size_t cb_elems = (size_t)m * ((size_t)1 << nbits) * (d / m);
  if (r == LANCE_HIP_OK) r = dev_dup(ctx, nullptr, cb_elems * 4, reinterpret_cast<void **>(&ix->codebook));
  if (r == LANCE_HIP_OK) r = widen_into(ctx, model_dtype(dtype), codebook, cb_elems, ix->codebook);
  if (r == LANCE_HIP_OK && m != 0 && nbits == 8) r = qscan_index_constants(ctx, ix);
  if (r == LANCE_HIP_OK) r = dev_dup(ctx, nullptr, (size_t)(nlist + 1) * 4, reinterpret_cast<void **>(&ix->part_offsets));
  if (r != LANCE_HIP_OK) { delete ix; return r; }
  *out = ix;
  return LANCE_HIP_OK;
}

static int index_finish_offsets(lance_hip_ctx *ctx, lance_hip_index *ix) {
  ix->part_offsets_h.resize(ix->nlist + 1);
  LH_CHECK_HIP(hipMemcpyAsync(ix->part_offsets_h.data(), ix->part_offsets, (size_t)(ix->nlist + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
  uint32_t nonfinite = 0;
  if (ix->cb_mean) LH_CHECK_HIP(hipMemcpyAsync(&nonfinite, ix->cb_mean + ix->d + 1, 4, hipMemcpyDeviceToHost, ctx->stream));
  LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  ix->model_finite = nonfinite == 0;
  ix->max_part = 0;
  for (uint32_t p = 0; p < ix->nlist; ++p)
    ix->max_part = std::max(ix->max_part, ix->part_offsets_h[p + 1] - ix->part_offsets_h[p]);
  return LANCE_HIP_OK;
}

int lance_hip_index_create(lance_hip_ctx *ctx, int dtype, int metric, uint32_t d, const void *centroids, uint32_t nlist,
                           const void *codebook, uint32_t m, uint32_t nbits, const uint32_t *part_ids,
                           const uint8_t *codes, const uint64_t *row_ids, uint64_t n, lance_hip_index **out) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(part_ids && codes, "index_create: NULL argument");
  LH_REQUIRE(n < (1ull << 32), "index_create: n too large for this version");
  lance_hip_index *ix = nullptr;
  LH_TRY(index_alloc_common(ctx, dtype, metric, d, centroids, nlist, codebook, m, nbits, &ix));
  uint32_t *perm = ctx->scratch_t<uint32_t>("index.perm", (size_t)(n ? n : 1));
  int r = perm ? LANCE_HIP_OK : LANCE_HIP_ENOMEM;
  if (r == LANCE_HIP_OK) r = stable_group(ctx, part_ids, (int64_t)n, (int64_t)n, (int)nlist, 1, ix->part_offsets, perm, (int64_t)n, nullptr);
  if (r == LANCE_HIP_OK) r = index_finish_offsets(ctx, ix);
  if (r == LANCE_HIP_OK) {
    ix->n = ix->part_offsets_h[nlist];
    r = dev_dup(ctx, nullptr, (size_t)ix->n * ix->code_bytes(), reinterpret_cast<void **>(&ix->codes));
    if (r == LANCE_HIP_OK) r = dev_dup(ctx, nullptr, (size_t)ix->n * 8, reinterpret_cast<void **>(&ix->row_ids));
  }
  if (r == LANCE_HIP_OK && ix->n > 0) {
    hipLaunchKernelGGL(gather_codes_kernel, dim3((unsigned)cdiv(ix->n, 256)), dim3(256), 0, ctx->stream, codes, row_ids, perm,
                       (int64_t)ix->n, (int)ix->code_bytes(), ix->codes, ix->row_ids);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
      set_error("index_create: gather kernel failed");
      r = LANCE_HIP_ERUNTIME;
    }
  }
  if (r != LANCE_HIP_OK) { delete ix; return r; }
  *out = ix;
  return LANCE_HIP_OK;
}

int lance_hip_index_from_storage(lance_hip_ctx *ctx, int dtype, int metric, uint32_t d, const void *centroids,
                                 uint32_t nlist, const void *codebook, uint32_t m, uint32_t nbits,
                                 const uint32_t *part_offsets_host, const uint8_t *codes, int transposed,
                                 const uint64_t *row_ids, uint64_t n, lance_hip_index **out) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(part_offsets_host && (n == 0 || (codes && row_ids)), "index_from_storage: NULL argument");
  LH_REQUIRE(part_offsets_host[0] == 0 && part_offsets_host[nlist] == n, "index_from_storage: offsets do not cover n rows");
  lance_hip_index *ix = nullptr;
  LH_TRY(index_alloc_common(ctx, dtype, metric, d, centroids, nlist, codebook, m, nbits, &ix));
  int r = LANCE_HIP_OK;
  ix->n = n;
  if (hipMemcpyAsync(ix->part_offsets, part_offsets_host, (size_t)(nlist + 1) * 4, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) r = LANCE_HIP_ERUNTIME;
  if (r == LANCE_HIP_OK) r = index_finish_offsets(ctx, ix);
  const uint32_t mbytes = ix ? ix->code_bytes() : m;
  if (r == LANCE_HIP_OK) r = dev_dup(ctx, transposed ? nullptr : codes, (size_t)n * mbytes, reinterpret_cast<void **>(&ix->codes));
  if (r == LANCE_HIP_OK) r = dev_dup(ctx, row_ids, (size_t)n * 8, reinterpret_cast<void **>(&ix->row_ids));
  if (r == LANCE_HIP_OK && transposed && n > 0) {
    hipLaunchKernelGGL(retile_codes_kernel, dim3((unsigned)cdiv(n * mbytes, 256)), dim3(256), 0, ctx->stream, codes, ix->codes,
                       (int64_t)n, (int)mbytes, ix->part_offsets, (int)nlist, 0);
    if (hipGetLastError() != hipSuccess) r = LANCE_HIP_ERUNTIME;
  }
  if (r == LANCE_HIP_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) r = LANCE_HIP_ERUNTIME;
  if (r != LANCE_HIP_OK) { if (r == LANCE_HIP_ERUNTIME) set_error("index_from_storage: HIP failure"); delete ix; return r; }
  *out = ix;
  return LANCE_HIP_OK;
}

void lance_hip_index_destroy(lance_hip_index *idx) { delete idx; }

int lance_hip_index_set_raw(lance_hip_index *idx, const void *x, uint64_t n_raw) {
  LH_REQUIRE(idx, "index is NULL");
  std::lock_guard<std::mutex> lk(idx->lazy_mu);
  idx->raw = x;   // element type = the index's dtype
  idx->n_raw = n_raw;
  if (idx->raw_u8) {   // the compact refine copy belonged to the previous column (searches of this index must have finished: the caller swaps its data)
    (void)hipSetDevice(idx->device);
    (void)hipFree(idx->raw_u8);
    idx->raw_u8 = nullptr;
  }
  idx->raw_compact_state = 0;
  ++idx->raw_gen;      // a captured search of the previous attachment must not be replayed (same pointer, new contents; the freed u8 copy)
  return LANCE_HIP_OK;
}

int lance_hip_index_info(const lance_hip_index *idx, uint64_t *n_rows, uint32_t *nlist, uint32_t *m, uint32_t *d) {
  LH_REQUIRE(idx, "index is NULL");
  if (n_rows) *n_rows = idx->n;
  if (nlist) *nlist = idx->nlist;
  if (m) *m = idx->m;
  if (d) *d = idx->d;
  return LANCE_HIP_OK;
}

int lance_hip_index_export(lance_hip_ctx *ctx, const lance_hip_index *idx, uint32_t *part_offsets_host,
                           uint8_t *codes_transposed_host, uint64_t *row_ids_host) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(ctx && idx, "index_export: NULL argument");
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  if (part_offsets_host) memcpy(part_offsets_host, idx->part_offsets_h.data(), (size_t)(idx->nlist + 1) * 4);
  if (codes_transposed_host && idx->n > 0) {
    const uint32_t mbytes = idx->code_bytes();
    uint8_t *tmp = ctx->scratch_t<uint8_t>("index.export", (size_t)idx->n * mbytes);
    if (!tmp) return LANCE_HIP_ENOMEM;
    hipLaunchKernelGGL(retile_codes_kernel, dim3((unsigned)cdiv(idx->n * mbytes, 256)), dim3(256), 0, ctx->stream, idx->codes, tmp,
                       (int64_t)idx->n, (int)mbytes, idx->part_offsets, (int)idx->nlist, 1);
    LH_CHECK_HIP(hipGetLastError());
    LH_CHECK_HIP(hipMemcpyAsync(codes_transposed_host, tmp, (size_t)idx->n * mbytes, hipMemcpyDeviceToHost, ctx->stream));
  }
  if (row_ids_host && idx->n > 0)
    LH_CHECK_HIP(hipMemcpyAsync(row_ids_host, idx->row_ids, (size_t)idx->n * 8, hipMemcpyDeviceToHost, ctx->stream));
  LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return LANCE_HIP_OK;
}

}  // extern "C"
