// encode_fused.hip -- residual + PQ encode in ONE kernel, rows read in the column's own element type.
//
//   ResidualTransform / do_compute_residual   residual.rs:58-102      (x - centroid[part], in the element type: f16 rounds)
//   PQTransformer / transform_impl<8>         pq.rs:116-191           (per sub-vector: L2-nearest codeword, unwrap_or(0))
//
// The separate kernels write the residuals as an [n][d] f32 array and read it back (1 KB per 128-d row, 102 GB at the C4
// shape) after widening f16 / int8 columns into another f32 copy.  Here a lane owns a row: it loads the row once (native
// element type, widened exactly in registers), subtracts its partition's centroid, and walks the M sub-quantisers with the
// codebook staged through LDS (negated tile, broadcast ds_read_b128, the exact l2_scalar order of dist_exact) -- the only
// HBM traffic is the row and its M code bytes.
#include <hip/hip_fp16.h>

#include "common.h"
#include "exact.cuh"
#include "kernels.h"

#pragma clang fp contract(off)

namespace lh {

__device__ __forceinline__ f4 ef_load4(const float *p) { return *reinterpret_cast<const f4 *>(p); }
__device__ __forceinline__ f4 ef_load4(const __half *p) {
  const uint2 u = *reinterpret_cast<const uint2 *>(p);
  const __half2 a = *reinterpret_cast<const __half2 *>(&u.x), b = *reinterpret_cast<const __half2 *>(&u.y);
  return f4{__low2float(a), __high2float(a), __low2float(b), __high2float(b)};
}
__device__ __forceinline__ f4 ef_load4(const int8_t *p) {
  const uint32_t u = *reinterpret_cast<const uint32_t *>(p);
  return f4{(float)(int8_t)(u & 255u), (float)(int8_t)((u >> 8) & 255u), (float)(int8_t)((u >> 16) & 255u), (float)(int8_t)(u >> 24)};
}

constexpr int EF_MT_FLOATS = 8192;   // floats of codebook staged per group (32 KiB): MT = 8192 / (256 * SD) sub-quantisers

template <int D, int SD, typename TX>
__global__ __launch_bounds__(256) void encode_fused_kernel(const TX *__restrict__ x, int64_t n, const float *__restrict__ cent,
                                                           const uint32_t *__restrict__ part_ids, int residual, int f16,
                                                           const float *__restrict__ codebook, uint8_t *__restrict__ codes) {
  constexpr int M = D / SD;
  constexpr int MT = EF_MT_FLOATS / (256 * SD) < M ? EF_MT_FLOATS / (256 * SD) : M;
  __shared__ __attribute__((aligned(16))) float tile[MT * 256 * SD];
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool valid = row < n;
  RegVec<D> r;
  const uint32_t part = valid ? part_ids[row] : LANCE_HIP_NONE;
#pragma unroll
  for (int i = 0; i < D / 4; ++i) {
    f4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (valid && (part != LANCE_HIP_NONE || !residual)) {
      v = ef_load4(x + row * D + 4 * i);
      if (residual) {
        const f4 c = *reinterpret_cast<const f4 *>(cent + (int64_t)part * D + 4 * i);
        v = v - c;
        if (f16) {   // `*v - *cent` in half::f16 (residual.rs:96)
          v.x = __half2float(__float2half_rn(v.x)); v.y = __half2float(__float2half_rn(v.y));
          v.z = __half2float(__float2half_rn(v.z)); v.w = __half2float(__float2half_rn(v.w));
        }
      }
    }
    r.q[i] = v;   // rows without a partition encode the zero vector, as the separate residual kernel gives them
  }
  uint32_t packed[(M + 3) / 4];
#pragma unroll
  for (int i = 0; i < (M + 3) / 4; ++i) packed[i] = 0u;
#pragma unroll
  for (int g0 = 0; g0 < M; g0 += MT) {
    __syncthreads();
    // stage -codebook[g0 .. g0+MT) so that x - c is evaluated as x + (-c) (dist_exact BNEG: packed adds, same bits)
    for (int i = threadIdx.x * 4; i < MT * 256 * SD; i += 256 * 4) {
      const f4 v = *reinterpret_cast<const f4 *>(codebook + (int64_t)g0 * 256 * SD + i);
      *reinterpret_cast<f4 *>(&tile[i]) = -v;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int mm = g0 + t;
      if (mm < M) {
        RegVec<SD> a;
#pragma unroll
        for (int u = 0; u < SD / 4; ++u) a.q[u] = r.q[(mm * SD) / 4 + u];
        float minv = INFINITY;
        uint32_t mini = 0u;   // pq.rs:165 unwrap_or(0): an all-NaN sub-vector encodes 0
        const float *tb = tile + t * 256 * SD;
#pragma unroll 4
        for (int c = 0; c < 256; ++c) {
          const float v = dist_exact<SD, METRIC_L2, true>(a, tb + c * SD);
          if (v < minv) { minv = v; mini = (uint32_t)c; }
        }
        packed[mm >> 2] |= mini << (8 * (mm & 3));
      }
    }
  }
  if (valid) {
    uint8_t *dst = codes + row * M;
    if constexpr (M % 16 == 0) {
#pragma unroll
      for (int i = 0; i < M / 16; ++i)
        *reinterpret_cast<uint4 *>(dst + 16 * i) = make_uint4(packed[4 * i], packed[4 * i + 1], packed[4 * i + 2], packed[4 * i + 3]);
    } else {
#pragma unroll
      for (int i = 0; i < (M + 3) / 4; ++i) *reinterpret_cast<uint32_t *>(dst + 4 * i) = packed[i];
    }
  }
}

bool encode_fused_supported(int dtype, int d, int m, int nbits, const void *x, const float *cent, const float *codebook) {
  static const bool off = getenv("LANCE_HIP_NO_FUSED_ENCODE") != nullptr;
  if (off || nbits != 8 || m <= 0 || d % m != 0) return false;
  const int sd = d / m;
  if (!((d == 128 && (sd == 4 || sd == 8 || sd == 16)) || (d == 64 && (sd == 4 || sd == 8)))) return false;
  const size_t es = dtype == LANCE_HIP_F16 ? 2 : (dtype == LANCE_HIP_I8 ? 1 : 4);
  if (reinterpret_cast<uintptr_t>(x) % (4 * es)) return false;
  if ((reinterpret_cast<uintptr_t>(cent) & 15) || (reinterpret_cast<uintptr_t>(codebook) & 15)) return false;
  return true;
}

template <int D, int SD>
static void launch_ef_tx(lance_hip_ctx *ctx, int dtype, const void *x, int64_t n, const float *cent, const uint32_t *part_ids, int residual,
                         const float *codebook, uint8_t *codes) {
  const dim3 grid((unsigned)cdiv(n, 256));
  if (dtype == LANCE_HIP_F16)
    hipLaunchKernelGGL((encode_fused_kernel<D, SD, __half>), grid, dim3(256), 0, ctx->stream, static_cast<const __half *>(x), n, cent, part_ids,
                       residual, 1, codebook, codes);
  else if (dtype == LANCE_HIP_I8)
    hipLaunchKernelGGL((encode_fused_kernel<D, SD, int8_t>), grid, dim3(256), 0, ctx->stream, static_cast<const int8_t *>(x), n, cent, part_ids,
                       residual, 0, codebook, codes);
  else
    hipLaunchKernelGGL((encode_fused_kernel<D, SD, float>), grid, dim3(256), 0, ctx->stream, static_cast<const float *>(x), n, cent, part_ids,
                       residual, 0, codebook, codes);
}

int launch_encode_fused(lance_hip_ctx *ctx, int dtype, const void *x, int64_t n, int d, const float *cent, const uint32_t *part_ids,
                        int residual, const float *codebook, int m, uint8_t *codes) {
  if (n == 0) return LANCE_HIP_OK;
  const int sd = d / m;
  ScopedTimer t(ctx, "encode_fused");
  if (d == 128 && sd == 4) launch_ef_tx<128, 4>(ctx, dtype, x, n, cent, part_ids, residual, codebook, codes);
  else if (d == 128 && sd == 8) launch_ef_tx<128, 8>(ctx, dtype, x, n, cent, part_ids, residual, codebook, codes);
  else if (d == 128 && sd == 16) launch_ef_tx<128, 16>(ctx, dtype, x, n, cent, part_ids, residual, codebook, codes);
  else if (d == 64 && sd == 4) launch_ef_tx<64, 4>(ctx, dtype, x, n, cent, part_ids, residual, codebook, codes);
  else if (d == 64 && sd == 8) launch_ef_tx<64, 8>(ctx, dtype, x, n, cent, part_ids, residual, codebook, codes);
  else { set_error("encode_fused: unsupported shape d=%d m=%d", d, m); return LANCE_HIP_EINVAL; }
  LH_CHECK_HIP(hipGetLastError());
  return LANCE_HIP_OK;
}

}  // namespace lh
