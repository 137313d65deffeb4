// mfma_assign.hip -- the IVF coarse quantiser on the matrix cores: bf16x3 MFMA candidates + exact re-check.
//
//   assign / compute_partitions   kmeans.rs:317-369, :1187-1246, ivf/transform.rs:75-137  (argmin over centroids)
//
// The reference's argmin is defined on f32 distances computed in l2_scalar order (no FMA), which no matrix instruction
// reproduces.  But the ARGMIN only needs exact arithmetic where two centroids are close: this path computes, for every
// (row, centroid), the surrogate  s(c) = |c|^2 - 2 x.c (+ bias[c])   (dot: s(c) = -x.c + bias[c])
// with x.c from v_mfma_f32_32x32x16_bf16 on a two-term bf16 split of both operands (x = xh + xl, c = ch + cl; products
// xh.ch + xl.ch + xh.cl, f32 accumulation: |error| <= 2^-14 |x||c|), keeps the four smallest per row, and classifies:
//   second - first > 2E  -> the winner is certain; its exact distance is computed once (reference order);
//   third / fourth - first > 2E -> two / three candidates: their exact distances, reference rule (smaller value, then index);
//   otherwise            -> a wave recomputes that row exactly against all k centroids.
// E = 2^-13 (|x|^2 + max|c|^2 + max|bias|) dominates the surrogate error (split + accumulation: < 2^-14 (|x|^2+|c|^2)), the f32 error of
// |c|^2 and the rounding of the reference's own distance (<= 2^-18 of its value), so every centroid the reference could
// pick is among the candidates: ids and distances are bit-equal to the exact kernels (tests: test_assign_*, k-means, encode).
// Operand roles: A = centroid tile (32 x 16 per MFMA), B = 32 data rows held in registers for the whole centroid sweep;
// D[centroid][row] puts one data row per lane (two lanes per row), so the running top-4 is a per-lane register update.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include <hip/hip_fp16.h>

#include "common.h"
#include "exact.cuh"
#include "kernels.h"
#include "ma_common.cuh"

#pragma clang fp contract(off)

namespace lh {

// ---- centroid preparation: hi / lo bf16 planes, squared norms, maxima for the error bound ------------------------
// dp = row stride of the planes (d, or d rounded up to the K-chunk of the wide kernel: the padding is zero-filled)
__global__ __launch_bounds__(64) void ma_prep_kernel(const float *__restrict__ cent, int k, int d, int dp, const float *__restrict__ bias,
                                                     uint16_t *__restrict__ chi, uint16_t *__restrict__ clo, float *__restrict__ cn,
                                                     uint32_t *__restrict__ maxbits /* [0] = max |c|^2, [1] = max |bias| (float bits) */,
                                                     const uint8_t *__restrict__ active) {
  if (active && !active[0]) return;
  const int c = blockIdx.x;
  float s = 0.0f;
  for (int e = threadIdx.x; e < dp; e += 64) {
    const float v = e < d ? cent[(int64_t)c * d + e] : 0.0f;
    const uint32_t hb = bf16_rne_bits(v);
    const float lo = v - bf16_bits_to_float(hb);
    chi[(int64_t)c * dp + e] = (uint16_t)hb;
    clo[(int64_t)c * dp + e] = (uint16_t)bf16_rne_bits(lo);
    s += v * v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (threadIdx.x == 0) {
    cn[c] = s;
    atomicMax(&maxbits[0], s == s ? __float_as_uint(fabsf(s)) : 0x7F800000u);      // a NaN centroid: an infinite bound, every row takes the exact path
    if (bias) { const float b = fabsf(bias[c]); if (b == b) atomicMax(&maxbits[1], __float_as_uint(b)); }
  }
}

// KS = d / 16 MFMA k-steps (d <= 128).  DOT: surrogate = -x.c + bias.
// SUR: 0 = assign (running top-4 per row), 1 = the surrogate matrix [n][k] (find_partitions, small k), 2 = per (row, group of 16
// centroids) the smallest surrogate with the member's slot in its four lowest mantissa bits, and the group's second smallest
// (find_partitions over thousands of lists: see coarse_select_kernel)
__device__ __forceinline__ float ma_min_f32(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float ma_med3_f32(float a, float b, float c) { float r; asm("v_med3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

template <int KS, int METRIC, typename TX, int SUR = 0>
__global__ __launch_bounds__(256, 2) void ma_top3_kernel(MaArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (p.active && !p.active[0]) return;
  constexpr int D = KS * 16;
  constexpr int XS = D + 4;          // f32 row stride of the x staging tile
  constexpr int CS = D + 8;          // bf16 row stride of a centroid plane (16-byte skew: conflict-free ds_read_b128)
  float *xs = reinterpret_cast<float *>(smem);                       // [MA_ROWS][XS]            (phase 1)
  uint16_t *cbuf = reinterpret_cast<uint16_t *>(smem);               // [2][2][MA_CT][CS] bf16   (phase 2, same bytes)
  float *cns = reinterpret_cast<float *>(smem + (size_t)2 * 2 * MA_CT * CS * 2);   // [2][2][MA_CT]: |c|^2, bias
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, g = lane >> 5;
  const int64_t row0 = (int64_t)blockIdx.x * MA_ROWS;

  // phase 1: coalesced f32 rows -> LDS -> per-lane fragments (8 consecutive dims per k-step and half), split hi / lo
  for (int idx = threadIdx.x; idx < MA_ROWS * (D / 4); idx += 256) {
    const int r = idx / (D / 4), c4 = idx - r * (D / 4);
    f4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (row0 + r < p.n) v = load4(static_cast<const TX *>(p.x) + (row0 + r) * p.ldx + 4 * c4);
    *reinterpret_cast<f4 *>(&xs[r * XS + 4 * c4]) = v;
  }
  __syncthreads();
  bf16x8 xh[KS], xl[KS];
  float xn2 = 0.0f;
  {
    const float *xr = xs + (wave * 32 + j) * XS + g * 8;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const f4 a = *reinterpret_cast<const f4 *>(xr + s * 16), b = *reinterpret_cast<const f4 *>(xr + s * 16 + 4);
      const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const uint32_t hb = bf16_rne_bits(v[e]);
        const uint32_t lb = bf16_rne_bits(v[e] - bf16_bits_to_float(hb));
        xh[s][e] = (short)hb; xl[s][e] = (short)lb;
        xn2 += v[e] * v[e];
      }
    }
  }
  xn2 += __shfl_xor(xn2, 32, 64);
  __syncthreads();   // xs is dead: the same LDS now holds centroid tiles

  int ntiles = (p.k + MA_CT - 1) / MA_CT, t_first = 0;
  if constexpr (SUR) { t_first = (int)blockIdx.y * p.tiles_per_block; ntiles = min(ntiles, t_first + p.tiles_per_block); }
  constexpr int NCH = MA_CT * D / 8;          // 16-byte chunks per plane
  constexpr int CH = (NCH + 255) / 256;       // per thread (4 at D = 128)
  uint4 ph[CH], pl[CH];
  float pcn = 0.0f, pbias = 0.0f;
  auto tile_fetch = [&](int t) {
    const int c0 = t * MA_CT;
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int ch = threadIdx.x + 256 * u;           // chunk -> (centroid, 8-element column group)
      const int cr = ch / (D / 8), cc = ch - cr * (D / 8);
      ph[u] = make_uint4(0, 0, 0, 0); pl[u] = make_uint4(0, 0, 0, 0);
      if (ch < NCH && c0 + cr < p.k) {
        ph[u] = *reinterpret_cast<const uint4 *>(p.chi + (int64_t)(c0 + cr) * D + cc * 8);
        pl[u] = *reinterpret_cast<const uint4 *>(p.clo + (int64_t)(c0 + cr) * D + cc * 8);
      }
    }
    if (threadIdx.x < MA_CT) {
      const int c = c0 + threadIdx.x;
      pcn = c < p.k ? p.cn[c] : 0.0f;
      pbias = (c < p.k && p.bias) ? p.bias[c] : 0.0f;
    }
  };
  auto tile_store = [&](int buf) {
    uint16_t *hi = cbuf + (size_t)(buf * 2 + 0) * MA_CT * CS, *lo = cbuf + (size_t)(buf * 2 + 1) * MA_CT * CS;
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int ch = threadIdx.x + 256 * u;
      const int cr = ch / (D / 8), cc = ch - cr * (D / 8);
      if (ch < NCH) {
        *reinterpret_cast<uint4 *>(hi + cr * CS + cc * 8) = ph[u];
        *reinterpret_cast<uint4 *>(lo + cr * CS + cc * 8) = pl[u];
      }
    }
    if (threadIdx.x < MA_CT) { cns[(buf * 2 + 0) * MA_CT + threadIdx.x] = pcn; cns[(buf * 2 + 1) * MA_CT + threadIdx.x] = pbias; }
  };
  Top4 tp{INFINITY, INFINITY, INFINITY, INFINITY, LANCE_HIP_NONE, LANCE_HIP_NONE, LANCE_HIP_NONE};
  tile_fetch(t_first);
  tile_store(t_first & 1);
  __syncthreads();
  for (int t = t_first; t < ntiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntiles) tile_fetch(t + 1);
    const uint16_t *hi = cbuf + (size_t)(buf * 2 + 0) * MA_CT * CS, *lo = cbuf + (size_t)(buf * 2 + 1) * MA_CT * CS;
    f32x16 acc0, acc1;
#pragma unroll
    for (int v = 0; v < 16; ++v) { acc0[v] = 0.0f; acc1[v] = 0.0f; }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const bf16x8 ah0 = *reinterpret_cast<const bf16x8 *>(hi + j * CS + s * 16 + g * 8);
      const bf16x8 al0 = *reinterpret_cast<const bf16x8 *>(lo + j * CS + s * 16 + g * 8);
      const bf16x8 ah1 = *reinterpret_cast<const bf16x8 *>(hi + (32 + j) * CS + s * 16 + g * 8);
      const bf16x8 al1 = *reinterpret_cast<const bf16x8 *>(lo + (32 + j) * CS + s * 16 + g * 8);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, xh[s], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, xh[s], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, xl[s], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, xl[s], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al0, xh[s], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al1, xh[s], acc1, 0, 0, 0);
    }
    // D[centroid i][row j]: lane (j, g) holds centroids i = (v & 3) + 8 (v >> 2) + 4 g of each 32-block
    const int c0 = t * MA_CT;
    const float *cnb = cns + (buf * 2 + 0) * MA_CT, *bib = cns + (buf * 2 + 1) * MA_CT;
    float gk[2] = {0.0f, 0.0f}, gs[2] = {0.0f, 0.0f};
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      // surrogates of this lane's 16 centroids of the block; the insertions only run when one of them is below the current
      // fourth-smallest (rare once a few tiles have been seen: with k = 4096 about one block in twenty)
      float sv[16];
      float bm = INFINITY;
      float g1 = INFINITY, g2 = INFINITY;       // SUR == 2: the group's two smallest keys
#pragma unroll
      for (int vq = 0; vq < 4; ++vq) {
        const int ib = blk * 32 + 8 * vq + 4 * g;
        const f4 cn4 = *reinterpret_cast<const f4 *>(cnb + ib), bi4 = *reinterpret_cast<const f4 *>(bib + ib);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dot = blk ? acc1[vq * 4 + e] : acc0[vq * 4 + e];
          float v = METRIC == METRIC_DOT ? -dot : __builtin_fmaf(-2.0f, dot, cn4[e]);
          v += bi4[e];
          if constexpr (SUR == 2) {
            // a padding centroid: a huge FINITE value (+inf with slot bits would read as a NaN); a NaN surrogate drops out of both
            // minima (v_min_f32 / v_med3_f32 return the other operands) -- such rows carry a non-finite 2E and take the exact path
            if (c0 + ib + e >= p.k) v = 3.0e38f;
            const float key = __uint_as_float((__float_as_uint(v) & 0xFFFFFFF0u) | (uint32_t)(vq * 4 + e));
            g2 = ma_med3_f32(g1, g2, key);
            g1 = ma_min_f32(g1, key);
          } else {
          if (c0 + ib + e >= p.k) v = INFINITY;
          sv[vq * 4 + e] = v;
          bm = fminf(bm, v);
          if constexpr (SUR == 1) {
            const int64_t srow = row0 + wave * 32 + j;
            if (srow < p.n && c0 + ib + e < p.k) p.sur[srow * p.k + c0 + ib + e] = v;
          }
          }
        }
      }
      if constexpr (SUR == 2) {
        gk[blk] = g1; gs[blk] = g2;
      }
      if constexpr (!SUR) {
      if (bm < tp.m4) {
#pragma unroll
        for (int vq = 0; vq < 4; ++vq)
#pragma unroll
          for (int e = 0; e < 4; ++e) top4_insert(tp, sv[vq * 4 + e], (uint32_t)(c0 + blk * 32 + 8 * vq + 4 * g + e));
      }
      }
    }
    if constexpr (SUR == 2) {
      // group (tile t, half g, block blk) -> index gi = 4 t + 2 g + blk; the lane's two groups leave as ONE 16-byte store {key, key, second,
      // second} at [row][2 ng] floats, offset 4 (gi >> 1): the two lanes of a row fill a whole 32-byte sector per instruction (8-byte
      // pieces into two arrays measured 1.5 ms per 10,000 x 65,536 sweep, most of it partial-sector writes)
      const int64_t srow = row0 + wave * 32 + j;
      if (srow < p.n) *reinterpret_cast<f4 *>(p.gkey + srow * 2 * p.ng + (int64_t)(t * 2 + g) * 4) = f4{gk[0], gk[1], gs[0], gs[1]};
    }
    if (t + 1 < ntiles) tile_store(buf ^ 1);
    __syncthreads();
  }
  if constexpr (SUR) {
    const int64_t srow = row0 + wave * 32 + j;
    if (g == 0 && srow < p.n && blockIdx.y == 0) {
      const float cmax2 = __uint_as_float(p.maxbits[0]), bmax = __uint_as_float(p.maxbits[1]);
      p.e2[srow] = 2.0f * 0.0001220703125f * (xn2 + cmax2 + bmax) + (METRIC == METRIC_DOT ? 4.7683716e-7f : 0.0f);   // 2E, E = 2^-13 (|x|^2 + max|c|^2 + max|bias|) (+ 2^-22 under dot: see ma_common.cuh)
    }
    return;
  }
  // the two lanes of a row hold disjoint centroid subsets: merge the partner's four
  {
    const float pm1 = __shfl_xor(tp.m1, 32, 64), pm2 = __shfl_xor(tp.m2, 32, 64), pm3 = __shfl_xor(tp.m3, 32, 64), pm4 = __shfl_xor(tp.m4, 32, 64);
    const uint32_t pi1 = __shfl_xor(tp.i1, 32, 64), pi2 = __shfl_xor(tp.i2, 32, 64), pi3 = __shfl_xor(tp.i3, 32, 64);
    top4_insert(tp, pm1, pi1);
    top4_insert(tp, pm2, pi2);
    top4_insert(tp, pm3, pi3);
    top4_insert(tp, pm4, LANCE_HIP_NONE);   // >= the three just inserted: can only land in the fourth (index-free) slot
  }
  const int64_t row = row0 + wave * 32 + j;
  if (g == 0 && row < p.n) {
    const float cmax2 = __uint_as_float(p.maxbits[0]), bmax = __uint_as_float(p.maxbits[1]);
    const float E2 = 2.0f * 0.0001220703125f * (xn2 + cmax2 + bmax) + (METRIC == METRIC_DOT ? 4.7683716e-7f : 0.0f);   // 2E, E = 2^-13 (|x|^2 + max|c|^2 + max|bias|) (+ 2^-22 under dot: see ma_common.cuh)
    uint8_t cl = 3;                       // number of exact candidates - 1; 3 = recompute against every centroid
    if (tp.m2 - tp.m1 > E2) cl = 0;
    else if (tp.m3 - tp.m1 > E2) cl = 1;
    else if (tp.m4 - tp.m1 > E2) cl = 2;
    if (tp.i1 == LANCE_HIP_NONE || !(E2 < INFINITY) || !(E2 > 7.888609052210118e-31f)) cl = 3;   // (E2 <= 2^-100: products underflow, the relative bound does not hold)   // NaN / overflow anywhere: recompute exactly
    p.id1[row] = tp.i1; p.id2[row] = tp.i2; p.id3[row] = tp.i3; p.cls[row] = cl;
  }
}

// ---- exact re-check: lane = row (vector in VGPRs), reference arithmetic --------------------------------------------
template <int D, int METRIC, typename TX, int LANES>
__global__ __launch_bounds__(256) void ma_finalize_kernel(MaArgs p) {
  if (p.active && !p.active[0]) return;
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool valid = row < p.n;
  RegVec<D> a;
#pragma unroll
  for (int i = 0; i < D / 4; ++i) a.q[i] = valid ? load4(static_cast<const TX *>(p.x) + row * p.ldx + 4 * i) : f4{0.0f, 0.0f, 0.0f, 0.0f};
  bool finite = true;
  if (p.check_finite) {
#pragma unroll
    for (int i = 0; i < D; ++i) finite &= isfinite(a.get(i));
  }
  const uint8_t cl = valid ? p.cls[row] : 0;
  uint32_t best = LANCE_HIP_NONE;
  float bestv = INFINITY;
  if (valid && cl < 3) {
    // argmin_value_float over the candidates: strictly smallest biased value, smallest index on ties, NaN never selected
    const uint32_t cand[3] = {p.id1[row], p.id2[row], p.id3[row]};
    float bestb = INFINITY;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      if (t <= cl && cand[t] != LANCE_HIP_NONE) {
        const float v = finish_metric<METRIC>(dist_exact<D, METRIC, false, LANES>(a, p.cent + (int64_t)cand[t] * D));
        const float vb = p.bias ? v + p.bias[cand[t]] : v;
        if (vb < bestb || (vb == bestb && best != LANCE_HIP_NONE && cand[t] < best)) { bestb = vb; bestv = v; best = cand[t]; }
      }
    }
  }
  // rows the surrogate could not decide go to a list; ma_recompute_kernel redoes them against all k centroids
  if (valid && cl == 3) {
    const uint32_t slot = atomicAdd(p.fb_cnt, 1u);
    p.fb_rows[slot] = (uint32_t)row;
    return;
  }
  if (valid) {
    if (!finite) { best = LANCE_HIP_NONE; bestv = INFINITY; }
    if (best == LANCE_HIP_NONE) bestv = INFINITY;
    if (p.ids) p.ids[row] = best;
    if (p.dists) p.dists[row] = bestv;
  }
}

// One wave per undecided row: exact distances to all k centroids (reference order), argmin_value_float semantics --
// strictly smallest biased value, first index on ties, NaN never selected (kernels.rs:79-111).
template <int METRIC, typename TX, int LANES>
__global__ __launch_bounds__(256) void ma_recompute_kernel(MaArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (p.active && !p.active[0]) return;
  float *wrow = reinterpret_cast<float *>(smem) + (threadIdx.x >> 6) * p.d;
  const int lane = threadIdx.x & 63;
  const uint32_t nwaves = gridDim.x * 4, w0 = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t cnt = *p.fb_cnt;
  for (uint32_t it = w0; it < cnt; it += nwaves) {
    const int64_t row = p.fb_rows[it];
    bool fin = true;
    for (int e = lane; e < p.d; e += 64) {
      const float v = ld_elem(static_cast<const TX *>(p.x) + row * p.ldx, e);   // 64-bit row offset: rows * d exceeds 2^31 at scale
      wrow[e] = v;
      fin &= isfinite(v);
    }
    const bool finite = !p.check_finite || __all(fin);
    __builtin_amdgcn_wave_barrier();
    float bb = INFINITY, bv = INFINITY;
    uint32_t bi = LANCE_HIP_NONE;
    for (int c = lane; c < p.k; c += 64) {
      const float v = finish_metric<METRIC>(dist_exact_rt<METRIC, float, LANES>(wrow, p.cent + (int64_t)c * p.d, p.d));
      const float vb = p.bias ? v + p.bias[c] : v;
      if (vb < bb) { bb = vb; bv = v; bi = (uint32_t)c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(bb, o, 64), ov = __shfl_xor(bv, o, 64);
      const uint32_t oi = __shfl_xor(bi, o, 64);
      if (oi != LANCE_HIP_NONE && (bi == LANCE_HIP_NONE || ob < bb || (ob == bb && oi < bi))) { bb = ob; bv = ov; bi = oi; }
    }
    if (!finite || bi == LANCE_HIP_NONE) { bi = LANCE_HIP_NONE; bv = INFINITY; }
    if (lane == 0) {
      if (p.ids) p.ids[row] = bi;
      if (p.dists) p.dists[row] = bv;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// LANES: lane accumulators of the exact re-check (16; 32 for the dot products of f16 columns, dot.rs:91-102) -- the surrogate's
// error bound does not depend on the summation order, only the deciding arithmetic does
template <int KS, int METRIC, typename TX, int LANES = 16>
static void ma_launch_one(lance_hip_ctx *ctx, const MaArgs &a) {
  constexpr int D = KS * 16;
  const size_t lds_x = (size_t)MA_ROWS * (D + 4) * 4;
  const size_t lds_c = (size_t)2 * 2 * MA_CT * (D + 8) * 2 + (size_t)2 * 2 * MA_CT * 4;
  const size_t lds = std::max(lds_x, lds_c);
  {
    ScopedTimer t(ctx, "ma_sweep");      // the MFMA sweep alone (bench.py: roofline_build.estep_ivf is this kernel's time, by HIP events)
    hipLaunchKernelGGL((ma_top3_kernel<KS, METRIC, TX>), dim3((unsigned)cdiv(a.n, MA_ROWS)), dim3(256), lds, ctx->stream, a);
  }
  ScopedTimer t2(ctx, "ma_recheck");       // exact re-check of the candidates inside the margin + rows recomputed against all centroids
  hipLaunchKernelGGL((ma_finalize_kernel<D, METRIC, TX, LANES>), dim3((unsigned)cdiv(a.n, 256)), dim3(256), 0, ctx->stream, a);
  hipLaunchKernelGGL((ma_recompute_kernel<METRIC, TX, LANES>), dim3(512), dim3(256), (size_t)4 * D * 4, ctx->stream, a);
}

// f32: L2 / dot; f16 columns: L2, and dot in the 32-lane order (`lanes32`: set by the callers for widened f16 operands, whether the
// rows are read as f16 or from their f32 copy); int8 columns: L2 / dot
template <int KS>
static bool ma_launch_ks(lance_hip_ctx *ctx, const MaArgs &a, int metric, int dtype, bool lanes32) {
  if (lanes32) {
    if constexpr (KS < 2) {
      return false;                       // d = 16: the callers clear the flag (both orders coincide)
    } else {
      if (metric != METRIC_DOT) return false;
      if (dtype == LANCE_HIP_F32) ma_launch_one<KS, METRIC_DOT, float, 32>(ctx, a);
      else if (dtype == LANCE_HIP_F16) ma_launch_one<KS, METRIC_DOT, __half, 32>(ctx, a);
      else return false;
      return true;
    }
  }
  if (dtype == LANCE_HIP_F32) {
    if (metric == METRIC_DOT) ma_launch_one<KS, METRIC_DOT, float>(ctx, a); else ma_launch_one<KS, METRIC_L2, float>(ctx, a);
  } else if (dtype == LANCE_HIP_F16) {
    if (metric == METRIC_DOT) {
      // f16 rows under dot arrive with lanes32 when d > 16; at d = 16 the 32-lane and the 16-lane orders are the same sequence
      // of additions (dot.rs:91-102), so the 16-lane instantiation serves (refused until the round-3 fuzz drew d = 16, f16, dot)
      if constexpr (KS == 1) ma_launch_one<1, METRIC_DOT, __half>(ctx, a);
      else return false;
    } else {
      ma_launch_one<KS, METRIC_L2, __half>(ctx, a);
    }
  } else {
    if (metric == METRIC_DOT) ma_launch_one<KS, METRIC_DOT, int8_t>(ctx, a); else ma_launch_one<KS, METRIC_L2, int8_t>(ctx, a);
  }
  return true;
}

// =====================================================================================================================
// Wide rows (d > 128: C3's 1536-dimensional embeddings).  The kernel above keeps a row's bf16 fragments in registers for the
// whole centroid sweep, which stops at d = 128.  Here the contraction is tiled over the dimension: the rows are split once
// into bf16 hi / lo planes (ma_split_rows_kernel; stride dp = d rounded up to 32, zero padded), a workgroup owns 128 rows
// and walks the centroids in super-tiles of 128 (four 32 x 32 accumulator blocks per wave = 64 VGPRs); per 32-dimension
// chunk both operands go global -> LDS (16-byte skewed rows: conflict-free ds_read_b128) -> the same fragment layout as
// above, 3 MFMAs per (fragment pair) for the two-term split.  The surrogate, the running top-4, the classification and the
// exact re-check (reference arithmetic, ma_finalize_wide_kernel / ma_recompute_kernel) are unchanged, so ids and distances
// stay bit-equal to the exact kernels.  E = 2^-12 (|x|^2 + max|c|^2 + max|bias|): the split error is as above, the f32
// accumulation over up to 4096 / 16 x 3 MFMA steps adds < 2^-15 |x||c|.
typedef _Float16 ma_f16x8 __attribute__((ext_vector_type(8)));
constexpr int MW_ROWS = 128, MW_CT = 128, MW_KC = 32, MW_LS = MW_KC + 8;   // LDS row stride in bf16 elements (80 bytes)

template <typename TX>
__global__ __launch_bounds__(256) void ma_split_rows_kernel(const TX *__restrict__ x, int64_t n, int64_t ldx, int d, int dp,
                                                            uint16_t *__restrict__ xhi, uint16_t *__restrict__ xlo, float *__restrict__ xn2,
                                                            const uint8_t *__restrict__ active, const uint32_t *__restrict__ row_map = nullptr) {
  if (active && !active[0]) return;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const int64_t src = row_map ? (int64_t)row_map[row] : row;      // (planes are written at the compact index)
  float s = 0.0f;
  for (int e = 2 * lane; e < dp; e += 128) {     // two elements per lane: one 4-byte store per plane
    const float v0 = e < d ? ld_elem(x + src * ldx, e) : 0.0f;
    const float v1 = e + 1 < d ? ld_elem(x + src * ldx, e + 1) : 0.0f;
    const uint32_t h0 = bf16_rne_bits(v0), h1 = bf16_rne_bits(v1);
    const uint32_t l0 = bf16_rne_bits(v0 - bf16_bits_to_float(h0)), l1 = bf16_rne_bits(v1 - bf16_bits_to_float(h1));
    *reinterpret_cast<uint32_t *>(xhi + row * dp + e) = (h0 & 0xFFFFu) | (h1 << 16);
    *reinterpret_cast<uint32_t *>(xlo + row * dp + e) = (l0 & 0xFFFFu) | (l1 << 16);
    s += v0 * v0 + v1 * v1;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) xn2[row] = s;
}

// F16 (round 6; unit-length rows: cosine indices): ONE binary16 plane per operand, scaled by 2^14 (rows) / 2^14 (centroids, |c_i| <= 2 checked by
// the launcher), ONE v_mfma_f32_32x32x16_f16 per fragment pair instead of three bf16 products.  Error of the surrogate: both roundings
// <= 2^-11 relative per element (binary16 subnormals -- |x_i| < 2^-28 -- are an absolute 2^-39 each), so |fl(x).fl(c) - x.c| <= 2^-10 (1 + 2^-12)
// |x||c| (Cauchy-Schwarz), the f32 accumulation of d <= 4096 exact products <= d 2^-24 |x||c| (2^-13.4 at d = 1536): s = |c|^2 - 2 x.c is
// within 2 (2^-10 + 2^-12) |x||c| <= 2^-9.7 (|x|^2 + |c|^2) / 2 ... E = 2^-9.5 (|x|^2 + max|c|^2) covers it with the f32 error of |c|^2
// and of the reference's own distance (<= 2^-12 of their values).  5.7 x the margin of the bf16 x 3 route, a third of its matrix work:
// more rows take the two- / three-candidate exact check (cheap), more take the recompute list (not cheap: measured per data set).
template <int METRIC, bool SUR = false, bool F16 = false>
__global__ __launch_bounds__(256, 2) void ma_top3_wide_kernel(MaArgs p) {
  __shared__ __attribute__((aligned(16))) uint16_t abuf[2][MW_CT][MW_LS];     // centroid chunk: hi / lo planes   20 KB
  __shared__ __attribute__((aligned(16))) uint16_t bbuf[2][MW_ROWS][MW_LS];   // row chunk: hi / lo planes        20 KB
  __shared__ __attribute__((aligned(16))) float cns[2][MW_CT];                // |c|^2, bias of the super-tile
  if (p.active && !p.active[0]) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, g = lane >> 5;
  const int64_t row0 = (int64_t)blockIdx.x * MW_ROWS;
  const int dp = p.dp;
  Top4 tp{INFINITY, INFINITY, INFINITY, INFINITY, LANCE_HIP_NONE, LANCE_HIP_NONE, LANCE_HIP_NONE};
  // chunk loads: 128 rows x 32 elements = 512 16-byte pieces per plane; thread -> (row = idx >> 2, piece = idx & 3), 2 per plane
  const int lr0 = threadIdx.x >> 2, lc = threadIdx.x & 3;
  const int nchunks = dp / MW_KC;
  int ntiles = (p.k + MW_CT - 1) / MW_CT, t_first = 0;
  if constexpr (SUR) { t_first = (int)blockIdx.y * p.tiles_per_block; ntiles = min(ntiles, t_first + p.tiles_per_block); }
  const int total = nchunks * ntiles, it_first = nchunks * t_first;
  uint4 ga[2][2], gb[2][2];
  float gcn = 0.0f, gbias = 0.0f;
  // software pipeline: the global loads of step it + 1 are issued before the MFMAs of step it and land while they run
  auto fetch = [&](int it) {
    const int c0 = (it / nchunks) * MW_CT, k0 = (it % nchunks) * MW_KC;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int r = lr0 + 64 * u;
      ga[0][u] = ga[1][u] = gb[0][u] = gb[1][u] = make_uint4(0, 0, 0, 0);
      if (c0 + r < p.k) {
        ga[0][u] = *reinterpret_cast<const uint4 *>(p.chi + (int64_t)(c0 + r) * dp + k0 + lc * 8);
        if constexpr (!F16) ga[1][u] = *reinterpret_cast<const uint4 *>(p.clo + (int64_t)(c0 + r) * dp + k0 + lc * 8);
      }
      if (row0 + r < p.n) {
        gb[0][u] = *reinterpret_cast<const uint4 *>(p.xhi + (row0 + r) * dp + k0 + lc * 8);
        if constexpr (!F16) gb[1][u] = *reinterpret_cast<const uint4 *>(p.xlo + (row0 + r) * dp + k0 + lc * 8);
      }
    }
    if (k0 == 0 && threadIdx.x < MW_CT) {
      const int c = c0 + threadIdx.x;
      gcn = c < p.k ? p.cn[c] : 0.0f;
      gbias = (c < p.k && p.bias) ? p.bias[c] : 0.0f;
    }
  };
  f32x16 acc[4];
  if (it_first < total) fetch(it_first);
  for (int it = it_first; it < total; ++it) {
    const int c0 = (it / nchunks) * MW_CT, kc = it % nchunks;
    if (kc == 0) {
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[b][v] = 0.0f;
    }
    __syncthreads();          // the previous step's fragments (and, at a tile boundary, its cns) have been read
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int r = lr0 + 64 * u;
      *reinterpret_cast<uint4 *>(&abuf[0][r][lc * 8]) = ga[0][u];
      *reinterpret_cast<uint4 *>(&bbuf[0][r][lc * 8]) = gb[0][u];
      if constexpr (!F16) {
        *reinterpret_cast<uint4 *>(&abuf[1][r][lc * 8]) = ga[1][u];
        *reinterpret_cast<uint4 *>(&bbuf[1][r][lc * 8]) = gb[1][u];
      }
    }
    if (kc == 0 && threadIdx.x < MW_CT) { cns[0][threadIdx.x] = gcn; cns[1][threadIdx.x] = gbias; }
    __syncthreads();
    if (it + 1 < total) fetch(it + 1);
#pragma unroll
    for (int s = 0; s < MW_KC / 16; ++s) {
      if constexpr (F16) {
        const ma_f16x8 xh = *reinterpret_cast<const ma_f16x8 *>(&bbuf[0][wave * 32 + j][s * 16 + g * 8]);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const ma_f16x8 ah = *reinterpret_cast<const ma_f16x8 *>(&abuf[0][b * 32 + j][s * 16 + g * 8]);
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xh, acc[b], 0, 0, 0);
        }
      } else {
      const bf16x8 xh = *reinterpret_cast<const bf16x8 *>(&bbuf[0][wave * 32 + j][s * 16 + g * 8]);
      const bf16x8 xl = *reinterpret_cast<const bf16x8 *>(&bbuf[1][wave * 32 + j][s * 16 + g * 8]);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(&abuf[0][b * 32 + j][s * 16 + g * 8]);
        const bf16x8 al = *reinterpret_cast<const bf16x8 *>(&abuf[1][b * 32 + j][s * 16 + g * 8]);
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xh, acc[b], 0, 0, 0);
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xl, acc[b], 0, 0, 0);
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, xh, acc[b], 0, 0, 0);
      }
      }
    }
    if (kc != nchunks - 1) continue;
    // D[centroid i][row j]: lane (j, g) holds centroids i = (v & 3) + 8 (v >> 2) + 4 g of each 32-block (as in ma_top3_kernel)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      float sv[16];
      float bm = INFINITY;
#pragma unroll
      for (int vq = 0; vq < 4; ++vq) {
        const int ib = b * 32 + 8 * vq + 4 * g;
        const f4 cn4 = *reinterpret_cast<const f4 *>(&cns[0][ib]), bi4 = *reinterpret_cast<const f4 *>(&cns[1][ib]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dot = F16 ? acc[b][vq * 4 + e] * 3.7252902984619140625e-9f : acc[b][vq * 4 + e];      // F16: both planes carry 2^14
          float v = METRIC == METRIC_DOT ? -dot : __builtin_fmaf(-2.0f, dot, cn4[e]);
          v += bi4[e];
          if (c0 + ib + e >= p.k) v = INFINITY;
          sv[vq * 4 + e] = v;
          bm = fminf(bm, v);
          if constexpr (SUR) {
            const int64_t srow = row0 + wave * 32 + j;
            if (srow < p.n && c0 + ib + e < p.k) p.sur[srow * p.k + c0 + ib + e] = v;
          }
        }
      }
      if constexpr (!SUR) {
      if (bm < tp.m4) {
#pragma unroll
        for (int vq = 0; vq < 4; ++vq)
#pragma unroll
          for (int e = 0; e < 4; ++e) top4_insert(tp, sv[vq * 4 + e], (uint32_t)(c0 + b * 32 + 8 * vq + 4 * g + e));
      }
      }
    }
  }
  if constexpr (SUR) {
    const int64_t srow = row0 + wave * 32 + j;
    if (g == 0 && srow < p.n && blockIdx.y == 0) {
      const float cmax2 = __uint_as_float(p.maxbits[0]), bmax = __uint_as_float(p.maxbits[1]);
      p.e2[srow] = 2.0f * (F16 ? 0.001381067932f : 0.000244140625f) * (p.xn2[srow] + cmax2 + bmax) + (METRIC == METRIC_DOT ? 4.7683716e-7f : 0.0f);   // 2E, E = 2^-12 (F16: 2^-9.5) (|x|^2 + max|c|^2 + max|bias|)
    }
    return;
  }
  {
    const float pm1 = __shfl_xor(tp.m1, 32, 64), pm2 = __shfl_xor(tp.m2, 32, 64), pm3 = __shfl_xor(tp.m3, 32, 64), pm4 = __shfl_xor(tp.m4, 32, 64);
    const uint32_t pi1 = __shfl_xor(tp.i1, 32, 64), pi2 = __shfl_xor(tp.i2, 32, 64), pi3 = __shfl_xor(tp.i3, 32, 64);
    top4_insert(tp, pm1, pi1);
    top4_insert(tp, pm2, pi2);
    top4_insert(tp, pm3, pi3);
    top4_insert(tp, pm4, LANCE_HIP_NONE);
  }
  const int64_t row = row0 + wave * 32 + j;
  if (g == 0 && row < p.n) {
    const float cmax2 = __uint_as_float(p.maxbits[0]), bmax = __uint_as_float(p.maxbits[1]);
    const float E2 = 2.0f * (F16 ? 0.001381067932f : 0.000244140625f) * (p.xn2[row] + cmax2 + bmax) + (METRIC == METRIC_DOT ? 4.7683716e-7f : 0.0f);   // 2E, E = 2^-12 (F16: 2^-9.5) (|x|^2 + max|c|^2 + max|bias|)
    uint8_t cl = 3;
    if (tp.m2 - tp.m1 > E2) cl = 0;
    else if (tp.m3 - tp.m1 > E2) cl = 1;
    else if (tp.m4 - tp.m1 > E2) cl = 2;
    if (tp.i1 == LANCE_HIP_NONE || !(E2 < INFINITY) || !(E2 > 7.888609052210118e-31f)) cl = 3;   // (E2 <= 2^-100: products underflow, the relative bound does not hold)
    p.id1[row] = tp.i1; p.id2[row] = tp.i2; p.id3[row] = tp.i3; p.cls[row] = cl;
  }
}

// exact re-check for wide rows: one wave per row, the row staged in LDS, one 16-lane group per candidate.  Lane i of a group IS
// lane accumulator i of the reference's l2_scalar / dot_scalar (sums[i] over the 16-chunks in order), the d % 16 remainder is
// summed first in element order, and the 16 partial sums are folded in lane order through shuffles -- dist_exact_rt's result,
// with the candidate's centroid read as 64-byte segments.  Lane 0 applies argmin_value_float's rule -- strictly smallest biased
// value, smallest index on ties, NaN never selected (kernels.rs:79-111).  Undecided rows go to ma_recompute_kernel's list.
template <int METRIC, typename TX, int LANES>
__global__ __launch_bounds__(256) void ma_finalize_wide_kernel(MaArgs p) {
  static_assert(LANES == 16, "16 lane accumulators: one per lane of a candidate's group");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (p.active && !p.active[0]) return;
  float *wrow = reinterpret_cast<float *>(smem) + (threadIdx.x >> 6) * p.d;
  const int lane = threadIdx.x & 63, grp = lane >> 4, i = lane & 15;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const int d = p.d, full = d / 16 * 16;
  for (int64_t crow = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); crow < p.n; crow += nwaves) {
    const uint8_t cl = p.cls[crow];
    const int64_t row = p.row_map ? (int64_t)p.row_map[crow] : crow;      // the row of x / ids / dists this (compact) slot stands for
    if (cl == 3) {       // wave-uniform
      if (lane == 0) { const uint32_t slot = atomicAdd(p.fb_cnt, 1u); p.fb_rows[slot] = (uint32_t)row; }
      continue;
    }
    bool fin = true;
    for (int e0 = 0; e0 < d; e0 += 512) {      // eight loads of the row in flight per lane
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + 64 * u + lane;
        v[u] = e < d ? ld_elem(static_cast<const TX *>(p.x) + row * p.ldx, e) : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + 64 * u + lane;
        if (e < d) { wrow[e] = v[u]; fin &= isfinite(v[u]); }
      }
    }
    const bool finite = !p.check_finite || __all(fin);
    __builtin_amdgcn_wave_barrier();
    uint32_t c = LANCE_HIP_NONE;
    if (grp <= (int)cl && grp < 3) c = grp == 0 ? p.id1[crow] : (grp == 1 ? p.id2[crow] : p.id3[crow]);
    float s = 0.0f, acc = 0.0f;
    if (c != LANCE_HIP_NONE) {
      const float *y = p.cent + (int64_t)c * d;
      if (full != d) {       // remainder first, in element order (every lane of the group computes the same value)
        float r = 0.0f;
        for (int e = full; e < d; ++e) {
          if constexpr (METRIC == METRIC_DOT) {
            r = r + wrow[e] * y[e];
          } else {
            const float diff = wrow[e] - y[e];
            r = r + diff * diff;
          }
        }
        s = r;
      }
      int ch = 0;
      for (; ch + 256 <= full; ch += 256) {      // sixteen chunks' centroid loads in flight (one dependent L2 round trip per chunk made this kernel
        float yv[16];                            // 5.5 ms per million 1536-d rows: gpurun r06u); the adds stay in chunk order
#pragma unroll
        for (int u = 0; u < 16; ++u) yv[u] = y[ch + 16 * u + i];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const float xv = wrow[ch + 16 * u + i];
          if constexpr (METRIC == METRIC_DOT) {
            acc = acc + xv * yv[u];
          } else {
            const float diff = xv - yv[u];
            acc = acc + diff * diff;
          }
        }
      }
      for (; ch < full; ch += 16) {
        const float xv = wrow[ch + i], yv = y[ch + i];
        if constexpr (METRIC == METRIC_DOT) {
          acc = acc + xv * yv;
        } else {
          const float diff = xv - yv;
          acc = acc + diff * diff;
        }
      }
    }
    float tot = 0.0f;
#pragma unroll
    for (int t = 0; t < 16; ++t) tot = tot + __shfl(acc, (lane & 48) + t, 64);     // ((0 + a0) + a1) + ... + a15
    float v = INFINITY, vb = INFINITY;
    if (c != LANCE_HIP_NONE) {
      v = finish_metric<METRIC>(s + tot);
      vb = p.bias ? v + p.bias[c] : v;
    }
    uint32_t best = LANCE_HIP_NONE;
    float bestv = INFINITY, bestb = INFINITY;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const uint32_t ct = __shfl(c, 16 * t, 64);
      const float vt = __shfl(v, 16 * t, 64), vbt = __shfl(vb, 16 * t, 64);
      if (t <= (int)cl && ct != LANCE_HIP_NONE) {
        if (vbt < bestb || (vbt == bestb && best != LANCE_HIP_NONE && ct < best)) { bestb = vbt; bestv = vt; best = ct; }
      }
    }
    if (!finite) { best = LANCE_HIP_NONE; bestv = INFINITY; }
    if (best == LANCE_HIP_NONE) bestv = INFINITY;
    if (lane == 0) {
      if (p.ids) p.ids[row] = best;
      if (p.dists) p.dists[row] = bestv;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// centroids as one binary16 plane x 2^14 (F16 route), |c|^2, and the maxima incl. [3] = max |c_i| (the route needs it <= 2)
__global__ __launch_bounds__(64) void ma_prep_f16_kernel(const float *__restrict__ cent, int k, int d, int dp, const float *__restrict__ bias,
                                                         uint16_t *__restrict__ ch16, float *__restrict__ cn, uint32_t *__restrict__ maxbits) {
  const int c = blockIdx.x;
  float s = 0.0f, am = 0.0f;
  bool bad = false;
  for (int e = threadIdx.x; e < dp; e += 64) {
    const float v = e < d ? cent[(int64_t)c * d + e] : 0.0f;
    ch16[(int64_t)c * dp + e] = __half_as_ushort(__float2half_rn(v * 16384.0f));
    s += v * v;
    am = fmaxf(am, fabsf(v));
    bad |= !(fabsf(v) < INFINITY);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); am = fmaxf(am, __shfl_xor(am, o, 64)); }
  bad = __any(bad);
  if (threadIdx.x == 0) {
    cn[c] = s;
    if (s == s) atomicMax(&maxbits[0], __float_as_uint(fabsf(s)));
    atomicMax(&maxbits[3], bad ? 0x7F800000u : __float_as_uint(am));
    if (bias) { const float b = fabsf(bias[c]); if (b == b) atomicMax(&maxbits[1], __float_as_uint(b)); }
  }
}

// Unit-length long rows, two stages.  (1) The single-f16-product sweep + exact check of up to three candidates.  Its margin is 5.7 x the
// three-term bf16 sweep's, and distances between high-dimensional unit vectors concentrate: on the C3-shaped probe a sixth of the rows
// had four or more centroids inside it, and sending those to the all-centroids recompute kernel cost more than the sweep saved (gpurun
// r06u: recompute 10.6 ms per million rows against 0.7).  (2) So the undecided rows are COMPACTED (their list is the row map) and go through
// the accurate sweep -- planes split on the fly, bf16 x 3, margin 2^-12 -- and its exact check; what even that leaves undecided (true ties:
// duplicated centroids) is recomputed against every centroid as before.  `b` = the stage-2 arguments (planes / lists sized for the subset).
template <int METRIC, typename TX>
static int ma_launch_wide_f16(lance_hip_ctx *ctx, const MaArgs &a, MaArgs b, uint16_t *chi, uint16_t *clo, float *cn, uint32_t *maxbits2) {
  hipLaunchKernelGGL((ma_top3_wide_kernel<METRIC, false, true>), dim3((unsigned)cdiv(a.n, MW_ROWS)), dim3(256), 0, ctx->stream, a);
  const unsigned fgrid = (unsigned)std::min<int64_t>(cdiv(a.n, 4), 16384);
  hipLaunchKernelGGL((ma_finalize_wide_kernel<METRIC, TX, 16>), dim3(fgrid), dim3(256), (size_t)4 * a.d * 4, ctx->stream, a);
  uint32_t nf = 0;
  LH_CHECK_HIP(hipMemcpyAsync(&nf, a.fb_cnt, 4, hipMemcpyDeviceToHost, ctx->stream));
  LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  if (nf == 0) return LANCE_HIP_OK;
  if (nf < 256) {      // a handful: straight to the exact kernel
    hipLaunchKernelGGL((ma_recompute_kernel<METRIC, TX, 16>), dim3(512), dim3(256), (size_t)4 * a.d * 4, ctx->stream, a);
    return LANCE_HIP_OK;
  }
  const size_t pl = (size_t)nf * a.dp;
  uint16_t *xhi = ctx->scratch_t<uint16_t>("ma.xhi", pl), *xlo = ctx->scratch_t<uint16_t>("ma.xlo", pl);
  float *xn2 = ctx->scratch_t<float>("ma.xn2", (size_t)nf);
  if (!xhi || !xlo || !xn2) return LANCE_HIP_ENOMEM;
  LH_CHECK_HIP(lh::memset_async(maxbits2, 0, 16, ctx->stream));
  hipLaunchKernelGGL(ma_prep_kernel, dim3((unsigned)a.k), dim3(64), 0, ctx->stream, a.cent, a.k, a.d, a.dp, (const float *)nullptr, chi, clo, cn, maxbits2,
                     (const uint8_t *)nullptr);
  hipLaunchKernelGGL((ma_split_rows_kernel<TX>), dim3((unsigned)cdiv(nf, 4)), dim3(256), 0, ctx->stream, static_cast<const TX *>(a.x), (int64_t)nf, a.ldx, a.d,
                     a.dp, xhi, xlo, xn2, (const uint8_t *)nullptr, a.fb_rows);
  b.n = nf; b.row_map = a.fb_rows; b.xhi = xhi; b.xlo = xlo; b.xn2 = xn2; b.chi = chi; b.clo = clo; b.cn = cn; b.maxbits = maxbits2;
  b.fb_cnt = maxbits2 + 2;
  hipLaunchKernelGGL((ma_top3_wide_kernel<METRIC, false, false>), dim3((unsigned)cdiv(nf, MW_ROWS)), dim3(256), 0, ctx->stream, b);
  hipLaunchKernelGGL((ma_finalize_wide_kernel<METRIC, TX, 16>), dim3((unsigned)std::min<int64_t>(cdiv(nf, 4), 16384)), dim3(256), (size_t)4 * a.d * 4, ctx->stream, b);
  MaArgs r = b;         // the rows the accurate sweep left undecided: by their global numbers
  r.n = a.n; r.row_map = nullptr;
  hipLaunchKernelGGL((ma_recompute_kernel<METRIC, TX, 16>), dim3(512), dim3(256), (size_t)4 * a.d * 4, ctx->stream, r);
  return LANCE_HIP_OK;
}

template <int METRIC, typename TX>
static void ma_launch_wide(lance_hip_ctx *ctx, const MaArgs &a, uint16_t *xhi, uint16_t *xlo, float *xn2) {
  hipLaunchKernelGGL((ma_split_rows_kernel<TX>), dim3((unsigned)cdiv(a.n, 4)), dim3(256), 0, ctx->stream, static_cast<const TX *>(a.x), a.n, a.ldx,
                     a.d, a.dp, xhi, xlo, xn2, a.active);
  hipLaunchKernelGGL((ma_top3_wide_kernel<METRIC>), dim3((unsigned)cdiv(a.n, MW_ROWS)), dim3(256), 0, ctx->stream, a);
  const unsigned fgrid = (unsigned)std::min<int64_t>(cdiv(a.n, 4), 16384);
  hipLaunchKernelGGL((ma_finalize_wide_kernel<METRIC, TX, 16>), dim3(fgrid), dim3(256), (size_t)4 * a.d * 4, ctx->stream, a);
  hipLaunchKernelGGL((ma_recompute_kernel<METRIC, TX, 16>), dim3(512), dim3(256), (size_t)4 * a.d * 4, ctx->stream, a);
}

// rows of more than 128 elements (any length up to 4096; planes padded to a multiple of 32)
static bool mfma_wide_supported(const PairwiseArgs &p, int d) {
  static const bool off = getenv("LANCE_HIP_NO_MFMA_WIDE") != nullptr;
  if (off || d <= 128 || d > 4096 || p.lanes32) return false;
  if (p.x_native && p.x_dtype != LANCE_HIP_F32) return false;   // f16 / int8 columns of this width arrive as their f32 copy
  if (p.k < 64 || p.n < 2048) return false;
  const int64_t dp = (d + MW_KC - 1) / MW_KC * MW_KC;
  if ((int64_t)p.n * dp * 4 > (16ll << 30)) return false;       // the two bf16 planes of the rows: at most 16 GiB of scratch
  return true;
}

bool mfma_assign_supported(const PairwiseArgs &p, int d, int batches) {
  static const bool off = getenv("LANCE_HIP_NO_MFMA") != nullptr;
  if (off || batches != 1 || p.codes || p.matrix) return false;
  if (d > 128) return mfma_wide_supported(p, d);
  if (d % 16 != 0 || d < 16 || d > 128) return false;
  if (p.k < 32 || p.n < 2048) return false;     // small problems: the exact kernel's fixed cost is lower
  if (p.x_native) {   // rows in the column's own element type: 4-element loads need 4 * sizeof(element) alignment
    const size_t es = p.x_dtype == LANCE_HIP_F16 ? 2 : (p.x_dtype == LANCE_HIP_I8 ? 1 : 4);
    if ((reinterpret_cast<uintptr_t>(p.x_native) % (4 * es)) || (p.ldx % 4)) return false;
    return p.cent_aligned;
  }
  if (!p.x_aligned || !p.cent_aligned) return false;
  return true;
}

int ma_prepare_centroids(lance_hip_ctx *ctx, const float *cent, int k, int d, int dp, const float *bias, const uint8_t *active, uint16_t **chi_out,
                         uint16_t **clo_out, float **cn_out, uint32_t **maxbits_out) {
  const size_t kd = (size_t)k * dp;
  uint16_t *chi = ctx->scratch_t<uint16_t>("ma.chi", kd), *clo = ctx->scratch_t<uint16_t>("ma.clo", kd);
  float *cn = ctx->scratch_t<float>("ma.cn", (size_t)k);
  uint32_t *maxbits = ctx->scratch_t<uint32_t>("ma.maxbits", 4);
  if (!chi || !clo || !cn || !maxbits) return LANCE_HIP_ENOMEM;
  LH_CHECK_HIP(lh::memset_async(maxbits, 0, 16, ctx->stream));
  hipLaunchKernelGGL(ma_prep_kernel, dim3((unsigned)k), dim3(64), 0, ctx->stream, cent, k, d, dp, bias, chi, clo, cn, maxbits, active);
  LH_CHECK_HIP(hipGetLastError());
  *chi_out = chi; *clo_out = clo; *cn_out = cn; *maxbits_out = maxbits;
  return LANCE_HIP_OK;
}

int ma_recompute_launch(lance_hip_ctx *ctx, const MaArgs &a, int metric, int dtype) {
  const size_t lds = (size_t)4 * a.d * 4;
  const dim3 grid(512), block(256);
  if (dtype == LANCE_HIP_F16) {
    if (metric == METRIC_DOT) hipLaunchKernelGGL((ma_recompute_kernel<METRIC_DOT, __half, 16>), grid, block, lds, ctx->stream, a);
    else hipLaunchKernelGGL((ma_recompute_kernel<METRIC_L2, __half, 16>), grid, block, lds, ctx->stream, a);
  } else if (dtype == LANCE_HIP_I8) {
    if (metric == METRIC_DOT) hipLaunchKernelGGL((ma_recompute_kernel<METRIC_DOT, int8_t, 16>), grid, block, lds, ctx->stream, a);
    else hipLaunchKernelGGL((ma_recompute_kernel<METRIC_L2, int8_t, 16>), grid, block, lds, ctx->stream, a);
  } else {
    if (metric == METRIC_DOT) hipLaunchKernelGGL((ma_recompute_kernel<METRIC_DOT, float, 16>), grid, block, lds, ctx->stream, a);
    else hipLaunchKernelGGL((ma_recompute_kernel<METRIC_L2, float, 16>), grid, block, lds, ctx->stream, a);
  }
  LH_CHECK_HIP(hipGetLastError());
  return LANCE_HIP_OK;
}

// ids / dists of PairwiseArgs are filled exactly as launch_assign's exact kernels fill them.
int launch_assign_mfma(lance_hip_ctx *ctx, const PairwiseArgs &p, int d, int metric) {
  const bool wide = d > 128;
  const int dp = wide ? (d + MW_KC - 1) / MW_KC * MW_KC : d;
  // unit-length long rows that come with their binary16 plane (cosine transform: build.hip): one f16 product per fragment pair, if the
  // centroids fit the plane's scale (|c_i| <= 2: centroids of unit vectors are inside the unit ball; anything else takes the bf16 route)
  if (wide && p.x_plane16 && p.x_plane_n2 && p.x_plane_dp == dp && metric == METRIC_L2 && !p.bias && !p.active && p.x) {
    uint16_t *ch16 = ctx->scratch_t<uint16_t>("ma.ch16", (size_t)p.k * dp);
    float *cn16 = ctx->scratch_t<float>("ma.cn", (size_t)p.k);
    uint32_t *mb = ctx->scratch_t<uint32_t>("ma.maxbits", 4);
    uint32_t *id1 = ctx->scratch_t<uint32_t>("ma.id1", (size_t)p.n), *id2 = ctx->scratch_t<uint32_t>("ma.id2", (size_t)p.n);
    uint32_t *id3 = ctx->scratch_t<uint32_t>("ma.id3", (size_t)p.n);
    uint8_t *cls = ctx->scratch_t<uint8_t>("ma.cls", (size_t)p.n);
    uint32_t *fb_rows = ctx->scratch_t<uint32_t>("ma.fb_rows", (size_t)p.n);
    if (!ch16 || !cn16 || !mb || !id1 || !id2 || !id3 || !cls || !fb_rows) return LANCE_HIP_ENOMEM;
    LH_REQUIRE(p.n < (1ll << 32), "assign: more than 2^32 rows per call");
    LH_CHECK_HIP(lh::memset_async(mb, 0, 16, ctx->stream));
    hipLaunchKernelGGL(ma_prep_f16_kernel, dim3((unsigned)p.k), dim3(64), 0, ctx->stream, p.cent, p.k, d, dp, p.bias, ch16, cn16, mb);
    uint32_t h[4];
    LH_CHECK_HIP(hipMemcpyAsync(h, mb, 16, hipMemcpyDeviceToHost, ctx->stream));
    LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    float cmax_el;
    memcpy(&cmax_el, &h[3], 4);
    if (cmax_el <= 2.0f) {
      MaArgs a{};
      a.x = p.x; a.n = p.n; a.ldx = p.ldx; a.d = d; a.k = p.k;
      a.chi = ch16; a.clo = nullptr; a.cn = cn16; a.bias = nullptr; a.maxbits = mb; a.cent = p.cent;
      a.fb_cnt = mb + 2; a.fb_rows = fb_rows; a.active = nullptr;
      a.id1 = id1; a.id2 = id2; a.id3 = id3; a.cls = cls; a.ids = p.ids; a.dists = p.dists; a.check_finite = p.check_finite ? 1 : 0;
      a.xhi = p.x_plane16; a.xlo = nullptr; a.xn2 = p.x_plane_n2; a.dp = dp;
      // stage 2 (the rows stage 1 leaves undecided, compacted): its own candidate arrays and recompute list -- stage 1's list is its row map
      uint16_t *chi2 = ctx->scratch_t<uint16_t>("ma.chi", (size_t)p.k * dp), *clo2 = ctx->scratch_t<uint16_t>("ma.clo", (size_t)p.k * dp);
      float *cn2 = ctx->scratch_t<float>("ma.cn2", (size_t)p.k);
      uint32_t *mb2 = ctx->scratch_t<uint32_t>("ma.maxbits2", 4);
      uint32_t *fb2 = ctx->scratch_t<uint32_t>("ma.fb_rows2", (size_t)p.n);
      uint32_t *jd1 = ctx->scratch_t<uint32_t>("ma.id1b", (size_t)p.n), *jd2 = ctx->scratch_t<uint32_t>("ma.id2b", (size_t)p.n);
      uint32_t *jd3 = ctx->scratch_t<uint32_t>("ma.id3b", (size_t)p.n);
      uint8_t *cls2 = ctx->scratch_t<uint8_t>("ma.clsb", (size_t)p.n);
      if (!chi2 || !clo2 || !cn2 || !mb2 || !fb2 || !jd1 || !jd2 || !jd3 || !cls2) return LANCE_HIP_ENOMEM;
      MaArgs b2 = a;
      b2.id1 = jd1; b2.id2 = jd2; b2.id3 = jd3; b2.cls = cls2; b2.fb_rows = fb2;
      ScopedTimer t(ctx, "ma_wide_f16");
      LH_TRY((ma_launch_wide_f16<METRIC_L2, float>(ctx, a, b2, chi2, clo2, cn2, mb2)));
      LH_CHECK_HIP(hipGetLastError());
      return LANCE_HIP_OK;
    }
  }
  uint16_t *chi, *clo;
  float *cn;
  uint32_t *maxbits;   // [0] max |c|^2, [1] max |bias|, [2] rows left to the recompute kernel
  LH_REQUIRE(p.n < (1ll << 32), "assign: more than 2^32 rows per call");
  LH_TRY(ma_prepare_centroids(ctx, p.cent, p.k, d, dp, p.bias, p.active, &chi, &clo, &cn, &maxbits));
  uint32_t *id1 = ctx->scratch_t<uint32_t>("ma.id1", (size_t)p.n), *id2 = ctx->scratch_t<uint32_t>("ma.id2", (size_t)p.n);
  uint32_t *id3 = ctx->scratch_t<uint32_t>("ma.id3", (size_t)p.n);
  uint8_t *cls = ctx->scratch_t<uint8_t>("ma.cls", (size_t)p.n);
  uint32_t *fb_rows = ctx->scratch_t<uint32_t>("ma.fb_rows", (size_t)p.n);
  if (!id1 || !id2 || !id3 || !cls || !fb_rows) return LANCE_HIP_ENOMEM;
  MaArgs a;
  a.x = p.x_native ? p.x_native : static_cast<const void *>(p.x); a.n = p.n; a.ldx = p.ldx; a.d = d; a.k = p.k;
  const int dtype = p.x_native ? p.x_dtype : LANCE_HIP_F32;
  bool ok = true;
  a.chi = chi; a.clo = clo; a.cn = cn; a.bias = p.bias; a.maxbits = maxbits; a.cent = p.cent;
  a.fb_cnt = maxbits + 2; a.fb_rows = fb_rows; a.active = p.active;
  a.id1 = id1; a.id2 = id2; a.id3 = id3; a.cls = cls; a.ids = p.ids; a.dists = p.dists; a.check_finite = p.check_finite ? 1 : 0;
  if (wide) {
    uint16_t *xhi = ctx->scratch_t<uint16_t>("ma.xhi", (size_t)p.n * dp), *xlo = ctx->scratch_t<uint16_t>("ma.xlo", (size_t)p.n * dp);
    float *xn2 = ctx->scratch_t<float>("ma.xn2", (size_t)p.n);
    if (!xhi || !xlo || !xn2) return LANCE_HIP_ENOMEM;
    a.xhi = xhi; a.xlo = xlo; a.xn2 = xn2; a.dp = dp;
    LH_REQUIRE(metric == METRIC_L2 || metric == METRIC_DOT, "assign: metric %d is not on the MFMA path", metric);
    LH_REQUIRE(dtype == LANCE_HIP_F32, "assign: wide rows reach the MFMA path as f32 (element type %d)", dtype);
    if (metric == METRIC_DOT) ma_launch_wide<METRIC_DOT, float>(ctx, a, xhi, xlo, xn2); else ma_launch_wide<METRIC_L2, float>(ctx, a, xhi, xlo, xn2);
    LH_CHECK_HIP(hipGetLastError());
    return LANCE_HIP_OK;
  }
  switch (d / 16) {
    case 1: ok = ma_launch_ks<1>(ctx, a, metric, dtype, p.lanes32); break;
    case 2: ok = ma_launch_ks<2>(ctx, a, metric, dtype, p.lanes32); break;
    case 3: ok = ma_launch_ks<3>(ctx, a, metric, dtype, p.lanes32); break;
    case 4: ok = ma_launch_ks<4>(ctx, a, metric, dtype, p.lanes32); break;
    case 5: ok = ma_launch_ks<5>(ctx, a, metric, dtype, p.lanes32); break;
    case 6: ok = ma_launch_ks<6>(ctx, a, metric, dtype, p.lanes32); break;
    case 7: ok = ma_launch_ks<7>(ctx, a, metric, dtype, p.lanes32); break;
    case 8: ok = ma_launch_ks<8>(ctx, a, metric, dtype, p.lanes32); break;
    default: return LANCE_HIP_EINVAL;
  }
  LH_REQUIRE(ok, "assign: element type %d with metric %d is not on the MFMA path", dtype, metric);
  LH_CHECK_HIP(hipGetLastError());
  return LANCE_HIP_OK;
}

// =====================================================================================================================
// find_partitions at query time on the matrix cores  (ivf/storage.rs:107-119 -> kmeans_find_partitions kmeans.rs:1134-1158)
//
// The reference computes every query's distance to all nlist centroids and keeps the nprobes smallest, ascending by
// (distance, index).  Here the [nq][nlist] matrix is the bf16x3 SURROGATE of the kernels above (SUR = true: s(c) = |c|^2 - 2 q.c,
// dot: -q.c; |s(c) + const_q - dist_ref(c)| <= E with E as in the assign path), and one wave per query turns it into the exact
// answer:
//   T0   = the nprobes-th smallest of the 64 lane minima of the row -- at least nprobes centroids have s <= T0;
//   cand = { c : s(c) <= T0 + 2E } -- every centroid of the reference's answer is in it (its dist_ref is <= the nprobes-th
//          smallest dist_ref <= T0 + const_q + E, hence s <= T0 + 2E); usually nprobes + 1..3 centroids;
//   the candidates' distances are recomputed in the reference's order (16 lane accumulators = the 16 lanes of a group,
//   remainder first, lane-ordered fold: dist_exact_rt's value), sorted by (total_cmp key, index), the first nprobes emitted.
// A row with a NaN / overflowed bound, or with more candidates than the list holds (ties: duplicate centroids), is answered by
// the same wave from exact distances to ALL centroids -- rare, slow, and still the reference's result.
//
// Thousands of lists (round 6; C4: 4096, C5: 65,536).  The matrix is [nq][nlist] floats -- 2.6 GB per 10,000-query batch at C5, written in
// 16-byte pieces at a 256 KiB stride and read twice by the select kernel: 5.1 of the batch's 8 ms (profiles/r06w_bench_c5_*).  GROUPS = true:
// the sweep keeps, per (query, group of 16 centroids = one lane's share of a 32-centroid block), the smallest surrogate with the member's
// slot in its four lowest mantissa bits (key) and the group's second smallest key -- 8 bytes per group, 1 / 8 of the matrix.  Then
//   T0   = the nprobes-th smallest of the lanes' four smallest group keys -- nprobes DISTINCT centroids have key <= T0;
//   a centroid of the reference's answer has s <= T0 + 2E + delta, key <= T0 + 2E + 2 delta =: thr (delta = 2^-19 |s| <= 2^-5 E for the cleared
//   bits; the margin is widened by 1/16): it is its group's key holder and that key is <= thr, or the group's SECOND key is <= thr --
//   cand = { key holders of groups with key <= thr }  +  { all 16 members of groups whose second key is <= thr } (one query in seven has one).
// Exact distances, sort and emission are unchanged; the exact-path fall-back writes its distances into the (otherwise untouched) matrix row.
constexpr int CS_CAP = 128;     // candidates per query (two per lane in the final sort)
constexpr int CS_CAP_G = 512;   // ... with per-group keys: a group with two members in reach brings all sixteen (trained centroids of structureless data:
                                // one query in a few thousand has ~100 centroids inside the margin, and its exact-path fall-back against 65,536
                                // centroids is one wave's work for milliseconds -- 9.5 ms per batch at C5, gpurun r06za); the final sort merges 64 at a time

template <int METRIC>
__device__ __forceinline__ float cs_group_distance(const float *wrow, const float *__restrict__ y, int d, int lane) {
  // the 16 lanes of a group: lane i = lane accumulator i of l2_scalar / dot_scalar (see ma_finalize_wide_kernel)
  const int i = lane & 15, full = d / 16 * 16;
  float s = 0.0f, acc = 0.0f;
  if (full != d) {
    float r = 0.0f;
    for (int e = full; e < d; ++e) {
      if constexpr (METRIC == METRIC_DOT) r = r + wrow[e] * y[e];
      else { const float diff = wrow[e] - y[e]; r = r + diff * diff; }
    }
    s = r;
  }
  int ch = 0;
  for (; ch + 128 <= full; ch += 128) {      // eight chunks' loads in flight; the adds stay in chunk order
    float yv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) yv[u] = y[ch + 16 * u + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float xv = wrow[ch + 16 * u + i];
      if constexpr (METRIC == METRIC_DOT) acc = acc + xv * yv[u];
      else { const float diff = xv - yv[u]; acc = acc + diff * diff; }
    }
  }
  for (; ch < full; ch += 16) {
    const float xv = wrow[ch + i], yv = y[ch + i];
    if constexpr (METRIC == METRIC_DOT) acc = acc + xv * yv;
    else { const float diff = xv - yv; acc = acc + diff * diff; }
  }
  // ((0 + a0) + a1) + ... + a15: a running sum handed down the 16-lane DPP row (row_shr:1, lane 0 receives +0.0); after step j
  // lanes 0..j hold their prefix and recomputing a settled lane changes nothing, so lane 15 ends with the reference's total
  float run = 0.0f + acc;
#pragma unroll
  for (int t = 1; t < 16; ++t) {
    const float prev = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, run), 0x111, 0xF, 0xF, false));
    run = prev + acc;
  }
  const float tot = __shfl(run, (lane & 48) + 15, 64);     // every lane of the group gets the total
  return finish_metric<METRIC>(s + tot);
}

template <int METRIC, bool GROUPS = false>
__global__ __launch_bounds__(256) void coarse_select_kernel(float *__restrict__ sur, const float *__restrict__ e2, const float *__restrict__ q,
                                                            const float *__restrict__ cent, int nq, int nlist, int d, int nprobes,
                                                            uint32_t *__restrict__ part_ids, float *__restrict__ dists,
                                                            uint32_t *__restrict__ n_exact_rows, const float *__restrict__ gkey = nullptr,
                                                            const float *__restrict__ gsec = nullptr, int ng = 0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, grp = lane >> 4;
  const int qi = blockIdx.x * 4 + wave;
  if (qi >= nq) return;                       // no workgroup-level barrier below: waves are independent
  const int dpad = (d + 3) & ~3;
  constexpr int CAP = GROUPS ? CS_CAP_G : CS_CAP;
  float *wrow = reinterpret_cast<float *>(smem) + (size_t)wave * dpad;
  unsigned long long *ck = reinterpret_cast<unsigned long long *>(smem + (size_t)4 * dpad * 4) + (size_t)wave * CAP;   // (key << 32) | centroid
  uint32_t *cl = reinterpret_cast<uint32_t *>(smem + (size_t)4 * dpad * 4 + (size_t)4 * CAP * 8) + (size_t)wave * CAP;
  const float *qv = q + (int64_t)qi * d;
  for (int e = lane; e < d; e += 64) wrow[e] = qv[e];
  float *row = sur + (int64_t)qi * nlist;
  const float E2 = GROUPS ? e2[qi] * 1.0625f : e2[qi];      // (GROUPS: + 2 delta for the keys' cleared mantissa bits)
  const float *gk = GROUPS ? gkey + (int64_t)qi * 2 * ng : nullptr;      // per pair of groups {key, key, second, second}
  // pass 1: the four smallest values of every lane (a NaN anywhere sends the row to the exact path).  The nprobes-th smallest
  // of these 256 values bounds the nprobes-th smallest of the row from above (they are distinct elements), and unlike the
  // nprobes-th smallest of 64 lane MINIMA it stays close to it when nprobes approaches 64 (r04b: at nprobes = 50 the minima
  // gave ~90 candidates per query, most rows overflowed the list and took the exact path: 4.3 ms per 1000 queries).
  float m0 = INFINITY, m1 = INFINITY, m2 = INFINITY, m3 = INFINITY;
  bool bad = !(E2 < INFINITY) || !(E2 > 7.888609052210118e-31f);      // (2 * 2^-100 scale: products in the denormal range -> exact path)
  auto ins4 = [&](float v) {
    if (v < m3) {
      if (v < m2) {
        m3 = m2;
        if (v < m1) {
          m2 = m1;
          if (v < m0) { m1 = m0; m0 = v; } else m1 = v;
        } else m2 = v;
      } else m3 = v;
    }
  };
  if constexpr (GROUPS) {
    for (int i = lane * 4; i < 2 * ng; i += 256) {      // (one 16-byte record per lane and round: two keys, two seconds)
      const f4 v = *reinterpret_cast<const f4 *>(gk + i);
      bad |= (v.x != v.x) | (v.y != v.y);
      ins4(v.x); ins4(v.y);
    }
  } else {
  for (int i = lane; i < nlist; i += 64) {
    const float v = row[i];
    bad |= v != v;
    ins4(v);
  }
  }
  bad = __any(bad);
  // bitonic network over the 256 values: element e = register e / 64 of lane e % 64 (as select_probes_wave_kernel)
  float sv4[4] = {m0, m1, m2, m3};
#pragma unroll
  for (int k2 = 2; k2 <= 256; k2 <<= 1) {
#pragma unroll
    for (int dd = k2 >> 1; dd > 0; dd >>= 1) {
      if (dd >= 64) {
        const int jd = dd >> 6;
#pragma unroll
        for (int jr = 0; jr < 4; ++jr) {
          if ((jr & jd) == 0) {
            const bool up = ((jr * 64) & k2) == 0;
            const float lo = fminf(sv4[jr], sv4[jr | jd]), hi = fmaxf(sv4[jr], sv4[jr | jd]);
            sv4[jr] = up ? lo : hi; sv4[jr | jd] = up ? hi : lo;
          }
        }
      } else {
#pragma unroll
        for (int jr = 0; jr < 4; ++jr) {
          const int e = jr * 64 + lane;
          const float o = __shfl_xor(sv4[jr], dd, 64);
          const bool up = (e & k2) == 0, lower = (lane & dd) == 0;
          sv4[jr] = (lower == up) ? fminf(sv4[jr], o) : fmaxf(sv4[jr], o);
        }
      }
    }
  }
  const float thr = __shfl(sv4[0], nprobes - 1, 64) + E2;     // nprobes <= 64 (host); +inf when the row has fewer than nprobes values
  uint32_t cnt = 0;
  if (!bad) {
    if constexpr (GROUPS) {
      // group gi = 4 t + 2 g + blk holds centroids 64 t + 32 blk + 8 (slot >> 2) + 4 g + (slot & 3), slot = 0 .. 15
      for (int base = 0; base < ng; base += 64) {
        const int gi = base + lane;
        const float kv = gi < ng ? gk[(gi >> 1) * 4 + (gi & 1)] : INFINITY, sv = gi < ng ? gk[(gi >> 1) * 4 + 2 + (gi & 1)] : INFINITY;
        const bool take = kv <= thr, whole = sv <= thr;
        const int cbase = (gi >> 2) * 64 + (gi & 1) * 32 + ((gi >> 1) & 1) * 4;
        const unsigned long long mask = __ballot(take && !whole);
        if (take && !whole) {
          const uint32_t slot = __float_as_uint(kv) & 15u;
          const int c = cbase + 8 * (int)(slot >> 2) + (int)(slot & 3u);
          const uint32_t pos = cnt + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
          if (pos < (uint32_t)CAP && c < nlist) cl[pos] = (uint32_t)c;
          if (c >= nlist) bad = true;      // (a padding slot under the threshold: only with a threshold beyond every real value)
        }
        cnt += (uint32_t)__popcll(mask);
        unsigned long long wm = __ballot(whole);      // groups with two members in reach: all sixteen, one group at a time (rare)
        while (wm) {
          const int src = __ffsll((long long)wm) - 1;
          wm &= wm - 1ull;
          const int cb = __shfl(cbase, src, 64);
          if (lane < 16) {
            const int c = cb + 8 * (lane >> 2) + (lane & 3);
            const bool in = c < nlist;
            const unsigned long long m16 = __ballot(in) & 0xFFFFull;
            if (in) {
              const uint32_t pos = cnt + (uint32_t)__popcll(m16 & ((1ull << lane) - 1ull));
              if (pos < (uint32_t)CAP) cl[pos] = (uint32_t)c;
            }
          }
          int n_in = 0;      // (the same count on every lane)
#pragma unroll
          for (int sl = 0; sl < 16; ++sl) n_in += (cb + 8 * (sl >> 2) + (sl & 3)) < nlist ? 1 : 0;
          cnt += (uint32_t)n_in;
        }
      }
      bad = __any(bad);
    } else {
    for (int base = 0; base < nlist; base += 64) {
      const int i = base + lane;
      const bool take = i < nlist && row[i] <= thr;
      const unsigned long long mask = __ballot(take);
      if (take) {
        const uint32_t pos = cnt + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        if (pos < (uint32_t)CAP) cl[pos] = (uint32_t)i;
      }
      cnt += (uint32_t)__popcll(mask);
    }
    }
    if (cnt > (uint32_t)CAP || cnt < (uint32_t)nprobes) bad = true;
  }
  __builtin_amdgcn_wave_barrier();
  if (bad) {
    // exact distances to every centroid, written over the row; then the nprobes smallest (key, index) one after the other
    if (lane == 0 && n_exact_rows) atomicAdd(n_exact_rows, 1u);
    for (int c0 = 0; c0 < nlist; c0 += 4) {
      const int c = c0 + grp;
      float v = 0.0f;
      if (c < nlist) v = cs_group_distance<METRIC>(wrow, cent + (int64_t)c * d, d, lane);
      else (void)cs_group_distance<METRIC>(wrow, cent, d, lane);          // keep the shuffles wave-uniform
      if (c < nlist && (lane & 15) == 0) row[c] = v;
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    unsigned long long last = 0ull;
    bool have_last = false;
    for (int r = 0; r < nprobes; ++r) {
      unsigned long long best = ~0ull;
      for (int i = lane; i < nlist; i += 64) {
        const unsigned long long kk = ((unsigned long long)order_key(row[i]) << 32) | (uint32_t)i;
        if ((!have_last || kk > last) && kk < best) best = kk;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { const unsigned long long ob = __shfl_xor(best, o, 64); best = ob < best ? ob : best; }
      if (lane == 0) {
        part_ids[(int64_t)qi * nprobes + r] = (uint32_t)best;
        if (dists) dists[(int64_t)qi * nprobes + r] = key_to_float((uint32_t)(best >> 32));
      }
      last = best; have_last = true;
    }
    return;
  }
  // exact distances of the candidates, four per round
  for (uint32_t c0 = 0; c0 < cnt; c0 += 4) {
    const uint32_t ci = c0 + (uint32_t)grp;
    const uint32_t c = ci < cnt ? cl[ci] : cl[0];
    const float v = cs_group_distance<METRIC>(wrow, cent + (int64_t)c * d, d, lane);
    if (ci < cnt && (lane & 15) == 0) ck[ci] = ((unsigned long long)order_key(v) << 32) | c;
  }
  __builtin_amdgcn_wave_barrier();
  // the 64 smallest (key, centroid) of the candidates, ascending, in register 0: the first 64 candidates, then 64 more at a time through a
  // bitonic network over 128 elements (element e = register e / 64 of lane e % 64) -- one pass for up to 128 candidates
  unsigned long long v2[2];
  v2[0] = (uint32_t)lane < cnt ? ck[lane] : ~0ull;
  for (uint32_t base = 64;; base += 64) {
    v2[1] = base + (uint32_t)lane < cnt ? ck[base + (uint32_t)lane] : ~0ull;
#pragma unroll
    for (int k2 = 2; k2 <= 128; k2 <<= 1) {
#pragma unroll
      for (int dd = k2 >> 1; dd > 0; dd >>= 1) {
        if (dd >= 64) {
          const unsigned long long a = v2[0], b = v2[1];      // k2 = 128: one ascending sequence
          if (a > b) { v2[0] = b; v2[1] = a; }
        } else {
#pragma unroll
          for (int jr = 0; jr < 2; ++jr) {
            const int e = jr * 64 + lane;
            const unsigned long long o = __shfl_xor(v2[jr], dd, 64);
            const bool up = (e & k2) == 0, lower = (lane & dd) == 0;
            v2[jr] = (lower == up) ? (v2[jr] < o ? v2[jr] : o) : (v2[jr] > o ? v2[jr] : o);
          }
        }
      }
    }
    if (base + 64 >= cnt) break;      // (wave-uniform)
  }
  if (lane < nprobes) {       // nprobes <= 64: the answer sits in register 0
    part_ids[(int64_t)qi * nprobes + lane] = (uint32_t)v2[0];
    if (dists) dists[(int64_t)qi * nprobes + lane] = key_to_float((uint32_t)(v2[0] >> 32));
  }
}

bool coarse_mfma_supported(int metric, int d, uint32_t nq, uint32_t nlist, uint32_t nprobes, bool lanes32, const float *q, const float *cent) {
  static const char *env = getenv("LANCE_HIP_MFMA_COARSE");      // "0": off, "1": on for every shape the kernels take
  static const bool off = env && env[0] == '0', force = env && env[0] == '1';
  if (off || lanes32 || (metric != METRIC_L2 && metric != METRIC_DOT)) return false;
  if (nprobes == 0 || nprobes > 64 || nprobes > nlist) return false;
  if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(cent)) & 15) return false;
  if (d <= 128) { if (d % 16 != 0 || d < 16 || nlist < 32) return false; }
  else if (d > 4096 || nlist < 64 || d % 4 != 0) return false;
  if (force) return true;
  // below this the exact kernel's matrix is cheaper than the split / prep launches.  Long rows (K-tiled kernel): its workgroups
  // walk the dimension in 32-element steps, 48 dependent steps at d = 1536, and a 1000-query batch fills a quarter of the CUs --
  // measured slower than the exact kernels at C3 (r04b: 0.19 vs 0.155 ms per 1000 queries), so it waits for more work per launch
  if (d > 128) return (uint64_t)nq * nlist >= (1ull << 23);
  return (uint64_t)nq * nlist * (uint64_t)d >= (1ull << 27);
}

template <int KS>
static void coarse_launch_narrow(lance_hip_ctx *ctx, const MaArgs &a, int metric, dim3 grid) {
  constexpr int D = KS * 16;
  const size_t lds_x = (size_t)MA_ROWS * (D + 4) * 4;
  const size_t lds_c = (size_t)2 * 2 * MA_CT * (D + 8) * 2 + (size_t)2 * 2 * MA_CT * 4;
  const size_t lds = std::max(lds_x, lds_c);
  if (a.gkey) {
    if (metric == METRIC_DOT) hipLaunchKernelGGL((ma_top3_kernel<KS, METRIC_DOT, float, 2>), grid, dim3(256), lds, ctx->stream, a);
    else hipLaunchKernelGGL((ma_top3_kernel<KS, METRIC_L2, float, 2>), grid, dim3(256), lds, ctx->stream, a);
    return;
  }
  if (metric == METRIC_DOT) hipLaunchKernelGGL((ma_top3_kernel<KS, METRIC_DOT, float, 1>), grid, dim3(256), lds, ctx->stream, a);
  else hipLaunchKernelGGL((ma_top3_kernel<KS, METRIC_L2, float, 1>), grid, dim3(256), lds, ctx->stream, a);
}

// part_ids [nq][nprobes], dists [nq][nprobes] or NULL; matrix: [nq][nlist] scratch of the caller.  Enqueues only.
static int coarse_groups_from() {
  static const int v = getenv("LANCE_HIP_COARSE_GROUPS") ? atoi(getenv("LANCE_HIP_COARSE_GROUPS")) : 1024;
  return v;
}
bool coarse_groups_shape(int d, uint32_t nlist) {
  static const bool ma = getenv("LANCE_HIP_COARSE_GROUPS_MA") != nullptr;
  return !ma && d <= 128 && d % 16 == 0 && coarse_groups_from() > 0 && nlist >= (uint32_t)std::max(coarse_groups_from(), 256);
}

int find_partitions_mfma(lance_hip_ctx *ctx, int metric, const float *q, uint32_t nq, int d, const float *cent, uint32_t nlist, uint32_t nprobes,
                         float *matrix, uint32_t *part_ids, float *dists, const uint16_t *cpl_ready, const uint32_t *maxbits_ready) {
  const bool wide = d > 128;
  const int dp = wide ? (d + MW_KC - 1) / MW_KC * MW_KC : d;
  const size_t kd = (size_t)nlist * dp;
  uint16_t *chi = ctx->scratch_t<uint16_t>("cq.chi", kd), *clo = ctx->scratch_t<uint16_t>("cq.clo", kd);
  float *cn = ctx->scratch_t<float>("cq.cn", (size_t)nlist);
  uint32_t *maxbits = ctx->scratch_t<uint32_t>("cq.maxbits", 4);   // [0] max |c|^2, [1] unused (no bias), [2] rows answered by the exact path
  float *e2 = ctx->scratch_t<float>("cq.e2", (size_t)nq);
  if (!chi || !clo || !cn || !maxbits || !e2) return LANCE_HIP_ENOMEM;
  // thousands of lists: per-group keys instead of the [nq][nlist] matrix (see coarse_select_kernel); LANCE_HIP_COARSE_GROUPS=0 / =n: off / from n lists
  const int groups_from = coarse_groups_from();
  const bool groups = !wide && groups_from > 0 && nlist >= (uint32_t)std::max(groups_from, 256);
  const int ng = groups ? (int)cdiv(nlist, MA_CT) * 4 : 0;
  float *gkey = nullptr, *gsec = nullptr;
  if (groups) {
    gkey = ctx->scratch_t<float>("cq.gkey", (size_t)nq * 2 * ng);      // [nq][ng / 2] records {key, key, second, second}
    gsec = gkey;
    if (!gkey) return LANCE_HIP_ENOMEM;
  }
  LH_CHECK_HIP(lh::memset_async(maxbits, 0, 16, ctx->stream));
  static const bool groups_ma = getenv("LANCE_HIP_COARSE_GROUPS_MA") != nullptr;      // A/B: the per-group keys from ma_top3_kernel<.., 2> instead of the transform kernel's sweep
  if (groups && !groups_ma) {
    ScopedTimer t(ctx, "dist_matrix");
    LH_TRY(launch_xform_sweep_groups(ctx, metric, q, nq, d, cent, nlist, maxbits, gkey, ng, e2, cpl_ready, maxbits_ready));
  } else {
    ScopedTimer t(ctx, "dist_matrix");
    hipLaunchKernelGGL(ma_prep_kernel, dim3(nlist), dim3(64), 0, ctx->stream, cent, (int)nlist, d, dp, nullptr, chi, clo, cn, maxbits, nullptr);
    MaArgs a;
    a.x = q; a.n = nq; a.ldx = d; a.d = d; a.k = (int)nlist;
    a.chi = chi; a.clo = clo; a.cn = cn; a.bias = nullptr; a.maxbits = maxbits; a.cent = cent;
    a.sur = matrix; a.e2 = e2; a.active = nullptr;
    a.gkey = gkey; a.gsec = gsec; a.ng = ng;
    const unsigned rblocks = (unsigned)cdiv(nq, wide ? MW_ROWS : MA_ROWS);
    const int ntiles = (int)cdiv(nlist, wide ? MW_CT : MA_CT);
    // enough slices to put about two workgroups on every CU
    int slices = (int)std::min<uint64_t>((uint64_t)ntiles, std::max<uint64_t>(1, ((uint64_t)2 * ctx->num_cus) / rblocks));      // (rounded DOWN: 553 workgroups on 512 slots are two rounds, the second one almost empty -- 1.42 ms instead of 0.8 for 10,000 x 65,536, gpurun r06zf)
    a.tiles_per_block = (int)cdiv((uint64_t)ntiles, (uint64_t)slices);
    slices = (int)cdiv((uint64_t)ntiles, (uint64_t)a.tiles_per_block);
    const dim3 grid(rblocks, (unsigned)slices);
    if (wide) {
      uint16_t *xhi = ctx->scratch_t<uint16_t>("cq.xhi", (size_t)nq * dp), *xlo = ctx->scratch_t<uint16_t>("cq.xlo", (size_t)nq * dp);
      float *xn2 = ctx->scratch_t<float>("cq.xn2", (size_t)nq);
      if (!xhi || !xlo || !xn2) return LANCE_HIP_ENOMEM;
      a.xhi = xhi; a.xlo = xlo; a.xn2 = xn2; a.dp = dp;
      hipLaunchKernelGGL((ma_split_rows_kernel<float>), dim3((unsigned)cdiv(nq, 4)), dim3(256), 0, ctx->stream, q, (int64_t)nq, (int64_t)d, d, dp, xhi, xlo,
                         xn2, nullptr);
      if (metric == METRIC_DOT) hipLaunchKernelGGL((ma_top3_wide_kernel<METRIC_DOT, true>), grid, dim3(256), 0, ctx->stream, a);
      else hipLaunchKernelGGL((ma_top3_wide_kernel<METRIC_L2, true>), grid, dim3(256), 0, ctx->stream, a);
    } else {
      switch (d / 16) {
        case 1: coarse_launch_narrow<1>(ctx, a, metric, grid); break;
        case 2: coarse_launch_narrow<2>(ctx, a, metric, grid); break;
        case 3: coarse_launch_narrow<3>(ctx, a, metric, grid); break;
        case 4: coarse_launch_narrow<4>(ctx, a, metric, grid); break;
        case 5: coarse_launch_narrow<5>(ctx, a, metric, grid); break;
        case 6: coarse_launch_narrow<6>(ctx, a, metric, grid); break;
        case 7: coarse_launch_narrow<7>(ctx, a, metric, grid); break;
        case 8: coarse_launch_narrow<8>(ctx, a, metric, grid); break;
        default: return LANCE_HIP_EINVAL;
      }
    }
  }
  {
    ScopedTimer t(ctx, "select_probes");
    const int dpad = (d + 3) & ~3;
    const int cap = groups ? CS_CAP_G : CS_CAP;
    const size_t lds = (size_t)4 * dpad * 4 + (size_t)4 * cap * 8 + (size_t)4 * cap * 4;
    if (groups) {
      ctx->count_stage("coarse_groups");
      if (metric == METRIC_DOT)
        hipLaunchKernelGGL((coarse_select_kernel<METRIC_DOT, true>), dim3((unsigned)cdiv(nq, 4)), dim3(256), lds, ctx->stream, matrix, e2, q, cent, (int)nq,
                           (int)nlist, d, (int)nprobes, part_ids, dists, maxbits + 2, gkey, gsec, ng);
      else
        hipLaunchKernelGGL((coarse_select_kernel<METRIC_L2, true>), dim3((unsigned)cdiv(nq, 4)), dim3(256), lds, ctx->stream, matrix, e2, q, cent, (int)nq,
                           (int)nlist, d, (int)nprobes, part_ids, dists, maxbits + 2, gkey, gsec, ng);
    } else if (metric == METRIC_DOT)
      hipLaunchKernelGGL((coarse_select_kernel<METRIC_DOT>), dim3((unsigned)cdiv(nq, 4)), dim3(256), lds, ctx->stream, matrix, e2, q, cent, (int)nq, (int)nlist,
                         d, (int)nprobes, part_ids, dists, maxbits + 2);
    else
      hipLaunchKernelGGL((coarse_select_kernel<METRIC_L2>), dim3((unsigned)cdiv(nq, 4)), dim3(256), lds, ctx->stream, matrix, e2, q, cent, (int)nq, (int)nlist,
                         d, (int)nprobes, part_ids, dists, maxbits + 2);
  }
  static const bool stats = getenv("LANCE_HIP_COARSE_STATS") != nullptr;      // diagnosis: how many queries the select kernel answered from exact distances to every centroid
  if (stats && !ctx->capturing) {
    uint32_t h[4];
    LH_CHECK_HIP(hipMemcpyAsync(h, maxbits, 16, hipMemcpyDeviceToHost, ctx->stream));
    LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    fprintf(stderr, "[coarse] nq=%u nlist=%u nprobes=%u groups=%d: %u queries took the exact path\n", nq, nlist, nprobes, groups ? 1 : 0, h[2]);
  }
  LH_CHECK_HIP(hipGetLastError());
  return LANCE_HIP_OK;
}

}  // namespace lh
