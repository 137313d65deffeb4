// wide.hip -- exact-order pairwise distances for ANY dimension (the d > 128 workhorse: C3's 1536-d rows).
//
// The reference sum (l2_scalar / dot_scalar, l2.rs:57-91, dot.rs:30-58) keeps 16 lane accumulators per
// (row, centroid) pair over the whole dimension, so a register-blocked GEMM tile cannot hold a pair in one
// register.  Here the 16 accumulators of a pair live in 16 LANES: lane (g, i) of a wave owns accumulator i
// of an 8 rows x 8 centroids micro-tile (64 VGPRs as 32 packed pairs); the four 16-lane groups of a wave
// take interleaved centroids, 8 waves make a 32 rows x 64 centroids workgroup tile.  Both operands are staged
// through LDS in 64-wide slices of the dimension (coalesced float4 global loads, centroids negated once so
// x - c is the packed add x + (-c)); per 16-chunk a lane issues 16 ds_read_b32 for 96 packed VALU ops.
// When the dimension is exhausted the 16 partial sums of a pair are transposed through LDS and added in
// lane order 0..15 by one lane -- the reference's `sums.iter().sum()` -- after the sequential remainder
// (d % 16) sum, exactly as l2_scalar orders them.  No FMA (contract off), no MFMA: see pairwise.hip.
//
//   MODE 0  argmin (+bias, first index wins, non-finite rows -> NONE)   kmeans.rs:317-369, kernels.rs:79-111
//   MODE 1  full distance matrix                                        kmeans.rs:1134-1158
//   MODE 2  flat-scan filter: "centroids" are the queries; every (row, query) distance whose (key, rowid) is
//           <= the query's threshold pair goes to that query's candidate pool (flat.hip v2).  METRIC_COSINE
//           (d % 16 == 0) accumulates x.y with FMA in the 16 lanes and reduces with the f32x16 -> f32x8 tree of
//           cosine_fast (cosine.rs:143-175, simd/f32.rs:203-218,625-644); row / query norms come precomputed.
#include <algorithm>

#include "common.h"
#include "exact.cuh"
#include "kernels.h"

#pragma clang fp contract(off)

namespace lh {

constexpr int W_ROWS = 32, W_CENTS = 64, W_DK = 64, W_LD = 80, W_BS = 512, W_RLD = 20;

// L32 (dot only): the dot products of f16 columns, dot_scalar::<f16, f32, 32> (dot.rs:91-102,138-161) -- 32 lane accumulators
// per pair.  Accumulator i < 16 of the 32 sums the EVEN 16-chunks of the dimension, accumulator 16 + i the ODD ones, so the 16
// hardware lanes of a pair make two passes over the staged slices (even chunks, then odd chunks) and the lane-ordered sum
// continues from the first pass's total: tot = fold(fold(0, even lanes 0..15), odd lanes 0..15).  The remainder (d % 32, up to
// 31 products) is summed sequentially first, in two 16-wide pieces, into its own buffer; result = remainder + tot.
template <int METRIC, int MODE, bool L32 = false>
__global__ __launch_bounds__(W_BS) void pairwise_wide_kernel(PairwiseArgs p, int d, FlatPool fp) {
  static_assert(!L32 || (METRIC == METRIC_DOT && MODE != 2), "32-lane order: dot products of f16 columns, assign / matrix modes");
  __shared__ __attribute__((aligned(16))) float xt[W_ROWS * W_LD];    // 10 KB  rows x 64-slice (+pad: bank shift per row)
  __shared__ __attribute__((aligned(16))) float ct[W_CENTS * W_LD];   // 20 KB  centroids x 64-slice; reused as transpose scratch
  __shared__ float res[W_ROWS][W_CENTS + 1];
  __shared__ float res2[L32 ? W_ROWS : 1][L32 ? W_CENTS + 1 : 1];     // L32: the sequential remainder sum of every pair
  __shared__ uint32_t nonfinite[W_ROWS];
  __shared__ uint32_t tk[MODE == 2 ? W_CENTS : 1];
  __shared__ uint64_t tr[MODE == 2 ? W_CENTS : 1];
  __shared__ float qn[MODE == 2 ? W_CENTS : 1];
  __shared__ uint64_t trid[MODE == 2 ? W_ROWS : 1];
  __shared__ float tsy[MODE == 2 ? W_ROWS : 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, i = lane & 15;
  const int wr = wave >> 1, wc = wave & 1;
  const int b = blockIdx.y;
  if (p.active && !p.active[b]) return;
  const float *xb = p.x + (int64_t)b * p.x_batch_off;
  const float *cb = p.cent + (int64_t)b * p.cent_batch_stride;
  const float *biasb = p.bias ? p.bias + (int64_t)b * p.bias_batch_stride : nullptr;
  const int64_t row0 = (int64_t)blockIdx.x * W_ROWS;
  constexpr int LW = L32 ? 32 : 16;
  const int full = d / LW * LW;
  const bool vec_ok = p.x_aligned && p.cent_aligned && (d % 4 == 0);
  constexpr bool NEG = METRIC == METRIC_L2;

  if (tid < W_ROWS) nonfinite[tid] = 0;
  if constexpr (MODE == 2) {
    if (tid < W_ROWS) {
      const int64_t row = fp.r0 + row0 + tid;
      const bool ok = row0 + tid < p.n;
      trid[tid] = !ok ? ~0ull : (fp.row_ids ? fp.row_ids[row] : (uint64_t)row);
      tsy[tid] = (METRIC == METRIC_COSINE && ok) ? fp.row_sy[row] : 1.0f;
    }
  }
  float minv = INFINITY, mino = INFINITY;      // thread t < 32 keeps the running argmin of row row0 + t
  uint32_t mini = LANCE_HIP_NONE;

  f2 acc[8][4];
  auto zero_acc = [&]() {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[r][c] = f2{0.0f, 0.0f};
  };
  // stage columns [d0, d0 + width) (width <= 64) of the row tile and of centroid block c0; columns beyond `width`
  // (up to the next multiple of 16) are zero-filled: (0 + -0)^2 = 0 and 0 * 0 = 0 add nothing to a partial sum
  auto stage = [&](int c0, int d0, int width) {
    const int wpad = (width + 15) & ~15;
    if (vec_ok && (width & 3) == 0) {
      const int e4 = tid & 15;
      {
        const int r = tid >> 4;
        f4 v = f4{0.f, 0.f, 0.f, 0.f};
        if (4 * e4 < width && row0 + r < p.n) v = *reinterpret_cast<const f4 *>(xb + (row0 + r) * p.ldx + d0 + 4 * e4);
        if (4 * e4 < wpad) *reinterpret_cast<f4 *>(&xt[r * W_LD + 4 * e4]) = v;
        if (MODE == 0 && p.check_finite && !(isfinite(v.x) && isfinite(v.y) && isfinite(v.z) && isfinite(v.w))) nonfinite[r] = 1;
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = (tid >> 4) + 32 * h;
        f4 v = f4{0.f, 0.f, 0.f, 0.f};
        if (4 * e4 < width && c0 + c < p.k) v = *reinterpret_cast<const f4 *>(cb + (int64_t)(c0 + c) * d + d0 + 4 * e4);
        if (4 * e4 < wpad) *reinterpret_cast<f4 *>(&ct[c * W_LD + 4 * e4]) = NEG ? -v : v;
      }
    } else {
      for (int idx = tid; idx < W_ROWS * wpad; idx += W_BS) {
        const int r = idx / wpad, e = idx % wpad;
        float v = 0.0f;
        if (e < width && row0 + r < p.n) v = xb[(row0 + r) * p.ldx + d0 + e];
        xt[r * W_LD + e] = v;
        if (MODE == 0 && p.check_finite && !isfinite(v)) nonfinite[r] = 1;
      }
      for (int idx = tid; idx < W_CENTS * wpad; idx += W_BS) {
        const int c = idx / wpad, e = idx % wpad;
        float v = 0.0f;
        if (e < width && c0 + c < p.k) v = cb[(int64_t)(c0 + c) * d + d0 + e];
        ct[c * W_LD + e] = NEG ? -v : v;
      }
    }
  };
  auto compute = [&](int nchunks, int ch0 = 0, int chstep = 1) {
    for (int ch = ch0; ch < nchunks; ch += chstep) {
      float xv[8];
      f2 cv[4];
#pragma unroll
      for (int r = 0; r < 8; ++r) xv[r] = xt[(wr * 8 + r) * W_LD + ch * 16 + i];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        cv[c].x = ct[(wc * 32 + g + 8 * c) * W_LD + ch * 16 + i];
        cv[c].y = ct[(wc * 32 + g + 8 * c + 4) * W_LD + ch * 16 + i];
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const f2 xs = f2{xv[r], xv[r]};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if constexpr (METRIC == METRIC_COSINE) {
            acc[r][c] = __builtin_elementwise_fma(xs, cv[c], acc[r][c]);
          } else if constexpr (METRIC == METRIC_DOT) {
            acc[r][c] = acc[r][c] + xs * cv[c];
          } else {
            const f2 df = xs + cv[c];
            acc[r][c] = acc[r][c] + df * df;
          }
        }
      }
    }
  };
  // lane-ordered sum of every pair's 16 partials: res[row][cent] (first ? = : +=) ((0 + a0) + a1) + ... + a15
  // `cont` (L32): the fold continues from the value already stored, ((v + a0) + a1) + ... + a15; `to2`: into res2
  auto reduce = [&](bool first, bool cont = false, bool to2 = false) {
    float *red = ct;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      __syncthreads();
      float *mine = red + ((wave * 4 + g) * 8) * W_RLD;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        mine[(2 * c) * W_RLD + i] = acc[r][c].x;       // pair index j = 2c (+1): centroid g + 4j within the wave's 32
        mine[(2 * c + 1) * W_RLD + i] = acc[r][c].y;
      }
      __syncthreads();
      if (i < 8) {
        const float *src = mine + i * W_RLD;
        float tot = 0.0f;
        if constexpr (L32) {
          if (cont) tot = to2 ? res2[L32 ? wr * 8 + r : 0][L32 ? wc * 32 + g + 4 * i : 0] : res[wr * 8 + r][wc * 32 + g + 4 * i];
        }
        if constexpr (METRIC == METRIC_COSINE) {
          const f4 a0 = *reinterpret_cast<const f4 *>(src), a1 = *reinterpret_cast<const f4 *>(src + 4);
          const f4 a2 = *reinterpret_cast<const f4 *>(src + 8), a3 = *reinterpret_cast<const f4 *>(src + 12);
          const f4 lo = a0 + a2, hi = a1 + a3;          // t[i] = a[i] + a[i+8]   (f32x16 -> f32x8)
          const f4 s4 = lo + hi;                         // s_i = t_i + t_{i+4}
          tot = (s4.x + s4.z) + (s4.y + s4.w);           // ((t0+t4)+(t2+t6)) + ((t1+t5)+(t3+t7))
          tot = tot + 0.0f;                              // + reduce_sum(xy8 = 0)
          tot = tot + 0.0f;                              // + dot(tail of 0 elements)
        } else {
#pragma unroll
          for (int e = 0; e < 16; e += 4) {
            const f4 v = *reinterpret_cast<const f4 *>(src + e);
            tot = tot + v.x; tot = tot + v.y; tot = tot + v.z; tot = tot + v.w;
          }
        }
        float *dst = (L32 && to2) ? &res2[L32 ? wr * 8 + r : 0][L32 ? wc * 32 + g + 4 * i : 0] : &res[wr * 8 + r][wc * 32 + g + 4 * i];
        *dst = first ? tot : *dst + tot;
      }
    }
  };

  const int nblocks = (p.k + W_CENTS - 1) / W_CENTS;
  for (int cbi = blockIdx.z; cbi < nblocks; cbi += gridDim.z) {
    const int c0 = cbi * W_CENTS;
    // remainder first (sequential sum of the d % 16 tail products): lane i holds product i, lanes >= rem hold 0
    bool first = true;
    if constexpr (MODE == 2) {
      __syncthreads();
      if (tid < W_CENTS) {
        const bool ok = c0 + tid < p.k;
        tk[tid] = ok ? fp.tkey[c0 + tid] : 0u;
        tr[tid] = ok ? fp.trid[c0 + tid] : 0ull;
        qn[tid] = (METRIC == METRIC_COSINE && ok) ? fp.q_norm[c0 + tid] : 1.0f;
      }
    }
    if constexpr (L32) {
      const int rem = d - full;
      if (rem > 0) {           // res2 = the d % 32 tail products summed in element order (two 16-wide pieces)
        zero_acc();
        __syncthreads();
        stage(c0, full, min(rem, 16));
        __syncthreads();
        compute(1);
        reduce(true, false, true);
        if (rem > 16) {
          zero_acc();
          __syncthreads();
          stage(c0, full + 16, rem - 16);
          __syncthreads();
          compute(1);
          reduce(true, true, true);          // first = true: the store is `= tot`, and tot started from res2
        }
      }
      for (int par = 0; par < 2; ++par) {   // accumulators 0..15 = even 16-chunks, 16..31 = odd 16-chunks
        zero_acc();
        for (int d0 = 0; d0 < full; d0 += W_DK) {
          const int width = min(W_DK, full - d0);   // a multiple of 32: chunk parity inside the slice = parity in the row
          __syncthreads();
          stage(c0, d0, width);
          __syncthreads();
          compute(width / 16, par, 2);
        }
        if (full == 0) __syncthreads();
        reduce(true, par == 1);              // res = fold(par ? res : 0, this pass's 16 lane sums)
      }
      __syncthreads();
      if (rem > 0) {                         // `sum + sums.iter().sum()` (dot.rs:57)
        for (int idx = tid; idx < W_ROWS * W_CENTS; idx += W_BS) res[idx >> 6][idx & 63] = res2[L32 ? idx >> 6 : 0][L32 ? idx & 63 : 0] + res[idx >> 6][idx & 63];
      }
      __syncthreads();
    } else {
    if (full != d) {
      zero_acc();
      __syncthreads();
      stage(c0, full, d - full);
      __syncthreads();
      compute(1);
      reduce(true);
      first = false;
    }
    zero_acc();
    for (int d0 = 0; d0 < full; d0 += W_DK) {
      const int width = min(W_DK, full - d0);
      __syncthreads();
      stage(c0, d0, width);
      __syncthreads();
      compute(width / 16);
    }
    if (full == 0) {           // d < 16: the whole sum is the remainder; s + 0
      __syncthreads();
    }
    reduce(first);             // res = s + tot  (l2.rs:90 `s + sums.sum()`; 0 + tot when there is no remainder)
    __syncthreads();
    }
    const int ct_n = min(W_CENTS, p.k - c0);
    if constexpr (MODE == 1) {
      for (int idx = tid; idx < W_ROWS * W_CENTS; idx += W_BS) {
        const int r = idx >> 6, c = idx & 63;
        if (row0 + r < p.n && c < ct_n)
          p.matrix[((int64_t)b * p.n + row0 + r) * p.k + c0 + c] = finish_metric<METRIC>(res[r][c]);
      }
    } else if constexpr (MODE == 2) {
      for (int idx = tid; idx < W_ROWS * W_CENTS; idx += W_BS) {
        const int r = idx >> 6, c = idx & 63;
        if (row0 + r < p.n && c < ct_n) {
          float v;
          if constexpr (METRIC == METRIC_COSINE) v = 1.0f - res[r][c] / qn[c] / tsy[r];
          else v = finish_metric<METRIC>(res[r][c]);
          const uint32_t key = order_key(v);
          const uint64_t rid = trid[r];
          if (key < tk[c] || (key == tk[c] && rid <= tr[c])) {
            const uint32_t pos = atomicAdd(&fp.cnt[c0 + c], 1u);
            if (pos < (uint32_t)fp.cap) {
              fp.pkeys[(int64_t)(c0 + c) * fp.cap + pos] = key;
              fp.prids[(int64_t)(c0 + c) * fp.cap + pos] = rid;
            }
          }
        }
      }
    } else {
      if (tid < W_ROWS && row0 + tid < p.n) {
        for (int c = 0; c < ct_n; ++c) {
          const float v = finish_metric<METRIC>(res[tid][c]);
          const float vb = biasb ? v + biasb[c0 + c] : v;
          if (vb < minv) { minv = vb; mino = v; mini = (uint32_t)(c0 + c); }
        }
      }
    }
  }
  if constexpr (MODE == 0) {
    __syncthreads();
    if (tid < W_ROWS && row0 + tid < p.n) {
      const int64_t row = row0 + tid;
      if (nonfinite[tid]) { mini = LANCE_HIP_NONE; minv = INFINITY; }
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

template <int MODE>
int launch_wide(lance_hip_ctx *ctx, PairwiseArgs &p, int d, int metric, int batches, int *ksplit_out) {
  const int nblocks = (p.k + W_CENTS - 1) / W_CENTS;
  const int64_t rblocks = (int64_t)cdiv(p.n, W_ROWS) * batches;
  const int64_t want = 2ll * ctx->num_cus;
  int ksplit = 1;
  if (rblocks < want) ksplit = (int)std::min<int64_t>(nblocks, cdiv(want, rblocks));
  if (MODE == 0 && ksplit > 1) {
    const size_t cnt = (size_t)ksplit * batches * p.n;
    p.part_vb = ctx->scratch_t<float>("assign.part_vb", cnt);
    p.part_v = ctx->scratch_t<float>("assign.part_v", cnt);
    p.part_idx = ctx->scratch_t<uint32_t>("assign.part_idx", cnt);
    if (!p.part_vb || !p.part_v || !p.part_idx) return LANCE_HIP_ENOMEM;
  }
  const dim3 grid((unsigned)cdiv(p.n, W_ROWS), batches, ksplit);
  if (p.lanes32) {
    LH_REQUIRE(metric == METRIC_DOT, "internal: the 32-lane order exists for dot products only");
    hipLaunchKernelGGL((pairwise_wide_kernel<METRIC_DOT, MODE, true>), grid, dim3(W_BS), 0, ctx->stream, p, d, FlatPool{});
  } else if (metric == METRIC_DOT)
    hipLaunchKernelGGL((pairwise_wide_kernel<METRIC_DOT, MODE>), grid, dim3(W_BS), 0, ctx->stream, p, d, FlatPool{});
  else
    hipLaunchKernelGGL((pairwise_wide_kernel<METRIC_L2, MODE>), grid, dim3(W_BS), 0, ctx->stream, p, d, FlatPool{});
  *ksplit_out = ksplit;
  return LANCE_HIP_OK;
}

template int launch_wide<0>(lance_hip_ctx *, PairwiseArgs &, int, int, int, int *);
template int launch_wide<1>(lance_hip_ctx *, PairwiseArgs &, int, int, int, int *);

// flat-scan filter over rows [fp.r0, fp.r1) for the queries of fp (any dimension; cosine needs d % 16 == 0)
int launch_wide_filter(lance_hip_ctx *ctx, const FlatPool &fp, int d, int metric) {
  PairwiseArgs p;
  p.x = fp.x + fp.r0 * (int64_t)d;
  p.n = fp.r1 - fp.r0;
  p.ldx = d;
  p.cent = fp.q;
  p.k = fp.nq;
  p.x_aligned = ((reinterpret_cast<uintptr_t>(fp.x) & 15) == 0) && (d % 4 == 0);
  p.cent_aligned = ((reinterpret_cast<uintptr_t>(fp.q) & 15) == 0) && (d % 4 == 0);
  if (p.n <= 0 || fp.nq <= 0) return LANCE_HIP_OK;
  const int nblocks = (fp.nq + W_CENTS - 1) / W_CENTS;
  const int64_t rblocks = (int64_t)cdiv(p.n, W_ROWS);
  int z = 1;
  if (rblocks < 2ll * ctx->num_cus) z = (int)std::min<int64_t>(nblocks, cdiv(2ll * ctx->num_cus, rblocks));
  const dim3 grid((unsigned)rblocks, 1, z);
  if (metric == METRIC_COSINE) {
    LH_REQUIRE(d % 16 == 0 && fp.row_sy && fp.q_norm, "wide filter: cosine needs d %% 16 == 0 and precomputed norms");
    hipLaunchKernelGGL((pairwise_wide_kernel<METRIC_COSINE, 2>), grid, dim3(W_BS), 0, ctx->stream, p, d, fp);
  } else if (metric == METRIC_DOT) {
    hipLaunchKernelGGL((pairwise_wide_kernel<METRIC_DOT, 2>), grid, dim3(W_BS), 0, ctx->stream, p, d, fp);
  } else {
    hipLaunchKernelGGL((pairwise_wide_kernel<METRIC_L2, 2>), grid, dim3(W_BS), 0, ctx->stream, p, d, fp);
  }
  return LANCE_HIP_OK;
}

}  // namespace lh
