// ma_common.cuh -- pieces shared by the matrix-core coarse-quantiser kernels (mfma_assign.hip) and the single-pass
// transform kernel built on the same surrogate (xform_fused.hip): bf16 helpers, the kernel argument block, the running top-4.
#pragma once
#include <hip/hip_fp16.h>

#include "common.h"
#include "exact.cuh"
#include "kernels.h"

namespace lh {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MA_ROWS = 128;   // data rows per workgroup (4 waves x 32)
constexpr int MA_CT = 64;      // centroids per LDS tile (2 MFMA row blocks)

__device__ __forceinline__ uint32_t bf16_rne_bits(float x) {   // round-to-nearest-even bf16, NaN kept
  const uint32_t u = __float_as_uint(x);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (u >> 16) | 0x40u;
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bf16_bits_to_float(uint32_t b) { return __uint_as_float(b << 16); }

struct MaArgs {
  const void *x;         // [n][ldx] elements of the column's type (f32 / f16 / int8)
  int64_t n, ldx;
  int d, k;
  const uint16_t *chi, *clo;   // [k][d] bf16 planes
  const float *cn;             // [k] |c|^2
  const float *bias;           // [k] or NULL
  const uint32_t *maxbits;
  const float *cent;           // [k][d] f32 (exact re-check)
  uint32_t *id1, *id2, *id3;   // [n] three nearest by surrogate
  uint8_t *cls;                // [n] 0 certain, 1 / 2: two / three candidates, 3 recompute
  uint32_t *ids;               // outputs of the finalize kernel
  float *dists;
  int check_finite;
  uint32_t *fb_cnt, *fb_rows;  // rows left to ma_recompute_kernel
  const uint8_t *active;       // k-means: the (single) problem has converged -> every kernel returns at once
  // wide rows (d > 128, ma_top3_wide_kernel): the rows pre-split into bf16 planes of stride dp, and their squared norms
  const uint16_t *xhi = nullptr, *xlo = nullptr;
  const float *xn2 = nullptr;
  int dp = 0;
  // SUR = true instantiations (coarse_mfma.hip: find_partitions at query time): the surrogates go to a matrix instead of a running
  // top-4, the centroid tiles are split over blockIdx.y (small query batches would otherwise leave most CUs idle)
  float *sur = nullptr;        // [n][k] surrogate values
  float *e2 = nullptr;         // [n] 2E of the row (the select kernel's candidate margin).  Under dot every kernel adds 2^-22 to it: the reference's
                               // distance is the f32 value of 1 - x.c, so products closer than an ulp of 1 tie there (first index wins)
  int tiles_per_block = 0;     // centroid tiles (narrow: MA_CT, wide: MW_CT centroids) per blockIdx.y slice
  // SUR == 2 (find_partitions over thousands of lists): per (row, group of 16 centroids) the smallest surrogate, the member's slot in its
  // four lowest mantissa bits, and the group's second smallest; ng = 4 * centroid tiles groups per row; gkey: [n][ng / 2] 16-byte records
  // {key, key, second, second} of two groups (gsec: unused alias)
  float *gkey = nullptr, *gsec = nullptr;
  int ng = 0;
  // second stage of the f16 route: the kernels work on a COMPACTED subset -- planes, id1..3 / cls indexed 0..n-1, row i being row row_map[i] of
  // x / ids / dists (and of the recompute list)
  const uint32_t *row_map = nullptr;
};

// running four smallest (values m1 <= m2 <= m3 <= m4, centroid ids of the first three)
struct Top4 {
  float m1, m2, m3, m4;
  uint32_t i1, i2, i3;
};
__device__ __forceinline__ void top4_insert(Top4 &t, float v, uint32_t i) {
  if (v < t.m4) {
    if (v < t.m3) {
      t.m4 = t.m3;
      if (v < t.m2) {
        t.m3 = t.m2; t.i3 = t.i2;
        if (v < t.m1) { t.m2 = t.m1; t.i2 = t.i1; t.m1 = v; t.i1 = i; }
        else { t.m2 = v; t.i2 = i; }
      } else {
        t.m3 = v; t.i3 = i;
      }
    } else {
      t.m4 = v;
    }
  }
}


// mfma_assign.hip: host-side pieces the fused transform reuses
// centroid planes (hi / lo bf16, stride dp), |c|^2, maxima words [0] max|c|^2 [1] max|bias| [2] rows left to the recompute kernel [3] spare
int ma_prepare_centroids(lance_hip_ctx *ctx, const float *cent, int k, int d, int dp, const float *bias, const uint8_t *active, uint16_t **chi,
                         uint16_t **clo, float **cn, uint32_t **maxbits);
// rows the surrogate could not decide (a.fb_cnt / a.fb_rows): exact distances to every centroid, one wave per row
int ma_recompute_launch(lance_hip_ctx *ctx, const MaArgs &a, int metric, int dtype);

}  // namespace lh
