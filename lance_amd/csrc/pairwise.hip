// pairwise.hip -- exact-order distance kernels with one operand per lane.
//
//   assign   : argmin over centroids           (kmeans.rs:317-369, :1350-1369; kernels.rs:79-111)
//   matrix   : all distances row x centroid    (kmeans.rs:1134-1158 before the partial sort)
//
// Shape of every kernel here: each lane owns ONE row vector `a` held entirely in VGPRs
// (compile-time D <= 128); the other operand (centroid / codebook tile) is staged in LDS
// and read with wave-uniform addresses (broadcast ds_read_b128), so the inner loop is
// pure packed-f32 VALU (v_pk_add_f32 / v_pk_mul_f32) fed by LDS broadcasts.  The work is
// VALU-bound: 3 flops per element with no FMA allowed (bit parity with the reference's
// un-fused l2_scalar); f32 MFMA runs at the same rate as the VALU on gfx950 and cannot
// reproduce the (x-y)^2 lane order, so it is deliberately not used here.
#include "common.h"
#include "exact.cuh"
#include <algorithm>
#include <cstdlib>

#include "kernels.h"

#pragma clang fp contract(off)

namespace lh {

template <int D>
__device__ __forceinline__ void load_row(RegVec<D> &a, const float *__restrict__ src, bool aligned16) {
  if constexpr (D % 4 == 0) {
    if (aligned16) {
#pragma unroll
      for (int i = 0; i < D / 4; ++i) a.q[i] = *reinterpret_cast<const f4 *>(src + 4 * i);
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < RegVec<D>::Q * 4; ++i) a.q[i >> 2][i & 3] = i < D ? src[i] : 0.0f;
}

template <int D>
__device__ __forceinline__ void zero_row(RegVec<D> &a) {
#pragma unroll
  for (int i = 0; i < RegVec<D>::Q; ++i) a.q[i] = f4{0.0f, 0.0f, 0.0f, 0.0f};
}

template <int D>
__device__ __forceinline__ bool row_is_finite(const RegVec<D> &a) {
  bool ok = true;
#pragma unroll
  for (int i = 0; i < D; ++i) ok &= isfinite(a.get(i));
  return ok;
}

// MODE 0: argmin  MODE 1: full distance matrix.  BS = workgroup size (rows per workgroup).
// gridDim.z = ksplit: split z handles centroid tiles z, z+ksplit, ... (more workgroups for
// small row counts); partial argmins are merged by argmin_merge_kernel.
template <int D, int METRIC, int CT, int MODE, int BS>
__global__ __launch_bounds__(BS) void pairwise_kernel(PairwiseArgs p) {
  __shared__ __attribute__((aligned(16))) float tile[CT * D];
  // MODE 1, one-wave workgroups (query batches): the distances of a tile are staged [row][centroid] in LDS and
  // written out as 256-byte row segments instead of 64 scattered 4-byte stores per centroid
  constexpr bool STAGE = MODE == 1 && BS == 64 && CT <= 64;
  __shared__ float dstage[STAGE ? 64 * 65 : 1];
  const int b = blockIdx.y;
  if (p.active && !p.active[b]) return;
  const float *xb = p.x + (int64_t)b * p.x_batch_off;
  const float *cb = p.cent + (int64_t)b * p.cent_batch_stride;
  const float *biasb = p.bias ? p.bias + (int64_t)b * p.bias_batch_stride : nullptr;
  const int64_t row = (int64_t)blockIdx.x * BS + threadIdx.x;
  const bool valid = row < p.n;
  RegVec<D> a;
  if (valid) {
    load_row<D>(a, xb + row * p.ldx, p.x_aligned);
  } else {
    zero_row<D>(a);
  }
  // KeepFiniteVectors (lance-index utils.rs:263-286) fused: non-finite rows get no partition
  const bool finite = (MODE == 0 && p.check_finite) ? row_is_finite<D>(a) : true;
  float minv = INFINITY, mino = INFINITY;
  uint32_t mini = LANCE_HIP_NONE;
  float *mrow = (MODE == 1 && valid) ? p.matrix + ((int64_t)b * p.n + row) * p.k : nullptr;

  for (int c0 = blockIdx.z * CT; c0 < p.k; c0 += CT * gridDim.z) {
    const int ct = min(CT, p.k - c0);
    __syncthreads();
    // L2: stage -c so the inner loop is x + (-c) (see dist_exact BNEG)
    constexpr bool NEG = METRIC != METRIC_DOT;
    if (p.cent_aligned) {
      for (int i = threadIdx.x * 4; i < ct * D; i += BS * 4) {
        const f4 v = *reinterpret_cast<const f4 *>(&cb[(int64_t)c0 * D + i]);
        *reinterpret_cast<f4 *>(&tile[i]) = NEG ? -v : v;
      }
    } else {
      for (int i = threadIdx.x; i < ct * D; i += BS) tile[i] = NEG ? -cb[(int64_t)c0 * D + i] : cb[(int64_t)c0 * D + i];
    }
    __syncthreads();
    if (valid) {
      if constexpr (D <= 16) {
#pragma unroll 4
        for (int c = 0; c < ct; ++c) {
          const float v = finish_metric<METRIC>(dist_exact<D, METRIC, METRIC != METRIC_DOT>(a, &tile[c * D]));
          if constexpr (MODE == 1) {
            if constexpr (STAGE) dstage[threadIdx.x * 65 + c] = v; else mrow[c0 + c] = v;
          } else {
            const float vb = biasb ? v + biasb[c0 + c] : v;
            if (vb < minv) { minv = vb; mino = v; mini = (uint32_t)(c0 + c); }
          }
        }
      } else {
        for (int c = 0; c < ct; ++c) {
          const float v = finish_metric<METRIC>(dist_exact<D, METRIC, METRIC != METRIC_DOT>(a, &tile[c * D]));
          if constexpr (MODE == 1) {
            if constexpr (STAGE) dstage[threadIdx.x * 65 + c] = v; else mrow[c0 + c] = v;
          } else {
            const float vb = biasb ? v + biasb[c0 + c] : v;
            if (vb < minv) { minv = vb; mino = v; mini = (uint32_t)(c0 + c); }
          }
        }
      }
    }
    if constexpr (STAGE) {
      __syncthreads();
      const int64_t rbase = (int64_t)blockIdx.x * BS;
      const int nrows = (int)min<int64_t>(BS, p.n - rbase);
      if ((int)threadIdx.x < ct)
        for (int r = 0; r < nrows; ++r)
          p.matrix[((int64_t)b * p.n + rbase + r) * p.k + c0 + threadIdx.x] = dstage[r * 65 + threadIdx.x];
    }
  }
  if constexpr (MODE == 0) {
    if (valid) {
      if (!finite) { mini = LANCE_HIP_NONE; minv = INFINITY; }
      if (gridDim.z > 1) {
        const int64_t o = ((int64_t)blockIdx.z * gridDim.y + b) * p.n + row;
        p.part_vb[o] = minv; p.part_v[o] = mino; p.part_idx[o] = mini;
      } else {
        if (p.ids) p.ids[(int64_t)b * p.out_batch_stride + row] = mini;
        if (p.dists) p.dists[(int64_t)b * p.out_batch_stride + row] = mino;
        if (p.codes) p.codes[row * p.codes_ld + b] = mini == LANCE_HIP_NONE ? (uint8_t)0 : (uint8_t)mini;
      }
    }
  }
}

// merge of k-split partial argmins: smallest biased value, ties -> smallest centroid index
__global__ __launch_bounds__(256) void argmin_merge_kernel(PairwiseArgs p, int ksplit, int batches) {
  const int b = blockIdx.y;
  if (p.active && !p.active[b]) return;
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= p.n) return;
  float bvb = INFINITY, bv = INFINITY;
  uint32_t bi = LANCE_HIP_NONE;
  for (int z = 0; z < ksplit; ++z) {
    const int64_t o = ((int64_t)z * batches + b) * p.n + row;
    const uint32_t i = p.part_idx[o];
    if (i == LANCE_HIP_NONE) continue;
    const float vb = p.part_vb[o];
    if (vb < bvb || (vb == bvb && i < bi)) { bvb = vb; bv = p.part_v[o]; bi = i; }
  }
  if (p.ids) p.ids[(int64_t)b * p.out_batch_stride + row] = bi;
  if (p.dists) p.dists[(int64_t)b * p.out_batch_stride + row] = bv;
  if (p.codes) p.codes[row * p.codes_ld + b] = bi == LANCE_HIP_NONE ? (uint8_t)0 : (uint8_t)bi;
}


template <int D, int MODE, int BS>
static void launch_fixed_bs(lance_hip_ctx *ctx, const PairwiseArgs &p, int metric, int batches, int ksplit) {
  constexpr int CT = (8192 / D) > 256 ? 256 : (8192 / D);
  dim3 grid((unsigned)cdiv(p.n, BS), batches, ksplit);
  if (metric == METRIC_DOT)
    hipLaunchKernelGGL((pairwise_kernel<D, METRIC_DOT, CT, MODE, BS>), grid, dim3(BS), 0, ctx->stream, p);
  else
    hipLaunchKernelGGL((pairwise_kernel<D, METRIC_L2, CT, MODE, BS>), grid, dim3(BS), 0, ctx->stream, p);
}

// Picks the workgroup size and the centroid split so that small problems (training samples,
// query batches) still put >= 2 workgroups on every CU.
template <int D, int MODE>
static int launch_fixed(lance_hip_ctx *ctx, PairwiseArgs p, int metric, int batches) {
  constexpr int CT = (8192 / D) > 256 ? 256 : (8192 / D);
  const int ntiles = (int)cdiv(p.k, CT);
  const int64_t want = 2ll * ctx->num_cus;    // 4x and 8x measured: no gain / slower on the 65,536-row training E-step
  int bs = 256;
  if ((int64_t)cdiv(p.n, 256) * batches * ntiles < want) bs = 64;
  int ksplit = 1;
  const int64_t blocks = (int64_t)cdiv(p.n, bs) * batches;
  if (blocks < want) ksplit = (int)std::min<int64_t>(std::min(ntiles, 8), cdiv(want, blocks));
  if (MODE == 0 && ksplit > 1) {
    const size_t cnt = (size_t)ksplit * batches * p.n;
    p.part_vb = ctx->scratch_t<float>("assign.part_vb", cnt);
    p.part_v = ctx->scratch_t<float>("assign.part_v", cnt);
    p.part_idx = ctx->scratch_t<uint32_t>("assign.part_idx", cnt);
    if (!p.part_vb || !p.part_v || !p.part_idx) return LANCE_HIP_ENOMEM;
  }
  if (bs == 64) launch_fixed_bs<D, MODE, 64>(ctx, p, metric, batches, ksplit);
  else launch_fixed_bs<D, MODE, 256>(ctx, p, metric, batches, ksplit);
  if (MODE == 0 && ksplit > 1)
    hipLaunchKernelGGL(argmin_merge_kernel, dim3((unsigned)cdiv(p.n, 256), batches), dim3(256), 0, ctx->stream, p, ksplit, batches);
  return LANCE_HIP_OK;
}

template <int MODE>
int launch_wide(lance_hip_ctx *ctx, PairwiseArgs &p, int d, int metric, int batches, int *ksplit_out);   // wide.hip

template <int MODE>
static int launch_pairwise(lance_hip_ctx *ctx, PairwiseArgs p, int d, int metric, int batches) {
  if (p.n == 0 || batches == 0) return LANCE_HIP_OK;
  LH_REQUIRE(p.k > 0, "pairwise: k must be > 0");
  LH_REQUIRE(metric == METRIC_L2 || metric == METRIC_DOT, "pairwise: metric must be L2 or Dot (cosine = normalise + L2)");
  p.x_aligned = p.x != nullptr && ((reinterpret_cast<uintptr_t>(p.x) & 15) == 0) && (p.ldx % 4 == 0) && (p.x_batch_off % 4 == 0);
  if (p.x_native && p.x_dtype == LANCE_HIP_F32 && !p.x) p.x = static_cast<const float *>(p.x_native);
  p.cent_aligned = ((reinterpret_cast<uintptr_t>(p.cent) & 15) == 0) && (((int64_t)p.cent_batch_stride) % 4 == 0) && (d % 4 == 0);
  const char *tname = MODE == 0 ? "assign" : "dist_matrix";
  ScopedTimer t(ctx, tname);
  if (p.lanes32) {
    LH_REQUIRE(metric == METRIC_DOT, "internal: the 32-lane order exists for dot products only");
    if (d <= 16) p.lanes32 = false;   // up to 16 elements both orders are the same sequence of additions
  }
  if (MODE == 0 && pq_mfma_supported(p, d, metric, batches)) return launch_pq_mfma(ctx, p, d, batches);
  if (MODE == 0 && xform_assign_supported(p, d, metric, batches)) return launch_xform_assign(ctx, p, d, metric);
  if (MODE == 0 && mfma_assign_supported(p, d, batches)) return launch_assign_mfma(ctx, p, d, metric);
  LH_REQUIRE(p.x != nullptr, "internal: the exact assign kernels need the f32 view of the rows");
  bool fixed_ok = p.cent_aligned;  // LDS tile float4 reads in dist_exact need 16B-aligned rows
  if (p.lanes32) fixed_ok = false;  // f16 dot products (32 lane accumulators): the two-pass kernel of wide.hip, any dimension
  // distance matrices of query batches (find_partitions): too few rows to fill the chip with one lane per row;
  // the 32 x 64 tiles of wide.hip measured 44 us against 62 us for 10,000 x 256 x 128
  if (MODE == 1 && d >= 64 && p.n <= 65536) fixed_ok = false;
  if (fixed_ok) {
    switch (d) {
      case 4: LH_TRY((launch_fixed<4, MODE>(ctx, p, metric, batches))); goto done;
      case 8: LH_TRY((launch_fixed<8, MODE>(ctx, p, metric, batches))); goto done;
      case 16: LH_TRY((launch_fixed<16, MODE>(ctx, p, metric, batches))); goto done;
      case 32: LH_TRY((launch_fixed<32, MODE>(ctx, p, metric, batches))); goto done;
      case 64: LH_TRY((launch_fixed<64, MODE>(ctx, p, metric, batches))); goto done;
      case 96: LH_TRY((launch_fixed<96, MODE>(ctx, p, metric, batches))); goto done;
      case 128: LH_TRY((launch_fixed<128, MODE>(ctx, p, metric, batches))); goto done;
      default: break;
    }
  }
  {   // any other dimension: the 16-lanes-per-pair kernel of wide.hip
    int ks = 1;
    LH_TRY(launch_wide<MODE>(ctx, p, d, metric, batches, &ks));
    if (MODE == 0 && ks > 1)
      hipLaunchKernelGGL(argmin_merge_kernel, dim3((unsigned)cdiv(p.n, 256), batches), dim3(256), 0, ctx->stream, p, ks, batches);
  }
done:
  LH_CHECK_HIP(hipGetLastError());
  return LANCE_HIP_OK;
}

// true when launch_assign will read `x_native` directly (the MFMA path): the caller then need not widen the column
bool assign_reads_native(PairwiseArgs p, int d, int batches) {
  if (!p.x_native || p.x_dtype == LANCE_HIP_F32) return false;
  p.cent_aligned = ((reinterpret_cast<uintptr_t>(p.cent) & 15) == 0) && (((int64_t)p.cent_batch_stride) % 4 == 0) && (d % 4 == 0);
  return mfma_assign_supported(p, d, batches);
}

int launch_assign(lance_hip_ctx *ctx, const PairwiseArgs &p, int d, int metric, int batches) {
  return launch_pairwise<0>(ctx, p, d, metric, batches);
}
int launch_dist_matrix(lance_hip_ctx *ctx, const PairwiseArgs &p, int d, int metric, int batches) {
  return launch_pairwise<1>(ctx, p, d, metric, batches);
}

}  // namespace lh
