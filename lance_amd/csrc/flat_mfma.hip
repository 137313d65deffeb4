// flat_mfma.hip -- batched flat scan on the matrix cores: bf16x3 MFMA surrogate distances filter the (query, row) pairs,
// the few that can beat a query's current threshold are recomputed exactly (reference order) before they enter its pool.
//
//   FlatDistanceCal::distance_all / compute_distance     flat/storage.rs:345-402, flat.rs:95-148   (the exact arithmetic kept)
//   final SortExec / pool selection                        scanner.rs:3386-3411                     (flat.hip: select kernel)
//
// flat.hip's v2 scan visits the rows in epochs; after each one a select kernel gives every query a threshold pair
// T = (key, rowid) of its k-th best so far, and a row only matters if (dist, rowid) <= T.  For a batch of queries the work
// is a rows x queries x d contraction: here v_mfma_f32_32x32x16_bf16 computes x.q on the two-term bf16 split of both
// operands (|error| < 2^-14 |x||q|), the surrogate |q|^2 + |x|^2 - 2 x.q (dot: 1 - x.q) is compared with T + E,
// E = 2^-12 (|x|^2 + |q|^2), and only the pairs that pass (a fraction ~ k / rows seen) are recomputed with dist_exact_rt
// and appended under the exact (key, rowid) test -- so the pools, and the answers, are the exact kernel's.
// Operand roles as in mfma_assign.hip: B = 32 data rows per wave in registers for the whole query sweep, A = query tiles.
//
// Round 3 (profiles/r02_mfma_flat_pmc.json: 8.5 VALU instructions per MFMA, matrix cores busy 14 % of the time): the epilogue
// no longer evaluates passing pairs inline.  (i) Fast reject: s = fma(-2, x.q, xk - tq) for two pairs per packed instruction,
// folded into one running minimum (v_min3) -- 1.5 VALU per pair instead of ~6, and no branch unless the minimum is <= 0;
// (ii) the pairs that pass (about k * rows_in_epoch / rows_seen per query, a wave meets one in half of its tiles) are only
// APPENDED, as row numbers, to a per-query queue; flat_mfma_eval_kernel then gives every query a workgroup that recomputes its
// queued rows exactly (one lane per row, all lanes busy) and appends to the pool under the exact (key, rowid) test -- the inline
// version ran a 128-dimension exact distance with two dependent global loads on one active lane while the wave's MFMA stream
// stood still.  A queue that overflows raises the pool-overflow flag: the caller's repair loop rescans with the exact kernels.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "exact.cuh"
#include "kernels.h"

#pragma clang fp contract(off)

namespace lh {

typedef short fm_bf16x8 __attribute__((ext_vector_type(8)));
typedef float fm_f32x16 __attribute__((ext_vector_type(16)));

#ifndef LH_FM_WAVES
#define LH_FM_WAVES 2
#endif
constexpr int FM_ROWS = 128;
constexpr int FM_QT = 64;

__device__ __forceinline__ uint32_t fm_bf16_rne(float x) {
  const uint32_t u = __float_as_uint(x);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (u >> 16) | 0x40u;
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float fm_bf16_f32(uint32_t b) { return __uint_as_float(b << 16); }

__global__ __launch_bounds__(64) void fm_query_prep_kernel(const float *__restrict__ q, int nq, int d, uint16_t *__restrict__ qhi,
                                                           uint16_t *__restrict__ qlo, float *__restrict__ qn) {
  const int i = blockIdx.x;
  float s = 0.0f;
  for (int e = threadIdx.x; e < d; e += 64) {
    const float v = q[(int64_t)i * d + e];
    const uint32_t hb = fm_bf16_rne(v);
    qhi[(int64_t)i * d + e] = (uint16_t)hb;
    qlo[(int64_t)i * d + e] = (uint16_t)fm_bf16_rne(v - fm_bf16_f32(hb));
    s += v * v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (threadIdx.x == 0) qn[i] = s;
}

struct FmArgs {
  FlatPool p;
  const uint16_t *qhi, *qlo;   // [nq][d] bf16 planes of this query chunk
  const float *qn;             // [nq] |q|^2
  uint32_t *scnt;              // [nq] queued rows of this epoch
  uint32_t *squeue;            // [nq][FM_SQ_CAP] row numbers whose surrogate distance passed
};

constexpr int FM_SQ_CAP = 2048;

template <int KS, int METRIC, typename TX>
__global__ __launch_bounds__(256, LH_FM_WAVES) void flat_filter_mfma_kernel(FmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const FlatPool &p = a.p;
  constexpr int D = KS * 16;
  constexpr int XS = D + 4, CS = D + 8;
  float *xs = reinterpret_cast<float *>(smem);                       // [FM_ROWS][XS]  (phase 1)
  uint16_t *qbuf = reinterpret_cast<uint16_t *>(smem);               // [2][2][FM_QT][CS] bf16 (phase 2)
  float *tqs = reinterpret_cast<float *>(smem + (size_t)2 * 2 * FM_QT * CS * 2);   // [2][FM_QT] threshold + query part of E
  float *qns = tqs + 2 * FM_QT;                                                     // [2][FM_QT] |q|^2 (dot: unused)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, g = lane >> 5;
  const int64_t row0 = p.r0 + (int64_t)blockIdx.x * FM_ROWS;
  for (int idx = threadIdx.x; idx < FM_ROWS * (D / 4); idx += 256) {
    const int r = idx / (D / 4), c4 = idx - r * (D / 4);
    f4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (row0 + r < p.r1) v = load4(static_cast<const TX *>(p.x_native) + (row0 + r) * D + 4 * c4);
    *reinterpret_cast<f4 *>(&xs[r * XS + 4 * c4]) = v;
  }
  __syncthreads();
  fm_bf16x8 xh[KS], xl[KS];
  float xn2 = 0.0f;
  {
    const float *xr = xs + (wave * 32 + j) * XS + g * 8;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const f4 u0 = *reinterpret_cast<const f4 *>(xr + s * 16), u1 = *reinterpret_cast<const f4 *>(xr + s * 16 + 4);
      const float v[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const uint32_t hb = fm_bf16_rne(v[e]);
        xh[s][e] = (short)hb; xl[s][e] = (short)fm_bf16_rne(v[e] - fm_bf16_f32(hb));
        xn2 += v[e] * v[e];
      }
    }
  }
  xn2 += __shfl_xor(xn2, 32, 64);
  __syncthreads();
  const int64_t row = row0 + wave * 32 + j;
  const bool rvalid = row < p.r1;
  // per-lane constant of the test  s - E <= T  <=>  (|q|^2 (1 - 2^-12) + T)'s partner: xk = |x|^2 (1 - 2^-12)
  const float xk = METRIC == METRIC_DOT ? -0.000244140625f * xn2 : xn2 - 0.000244140625f * xn2;

  const int ntiles = (p.nq + FM_QT - 1) / FM_QT;
  constexpr int NCH = FM_QT * D / 8;
  constexpr int CH = (NCH + 255) / 256;
  uint4 ph[CH], pl[CH];
  float ptq = 0.0f, pqn = 0.0f;
  auto fetch = [&](int t) {
    const int q0 = t * FM_QT;
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int ch = threadIdx.x + 256 * u;
      const int cr = ch / (D / 8), cc = ch - cr * (D / 8);
      ph[u] = make_uint4(0, 0, 0, 0); pl[u] = make_uint4(0, 0, 0, 0);
      if (ch < NCH && q0 + cr < p.nq) {
        ph[u] = *reinterpret_cast<const uint4 *>(a.qhi + (int64_t)(q0 + cr) * D + cc * 8);
        pl[u] = *reinterpret_cast<const uint4 *>(a.qlo + (int64_t)(q0 + cr) * D + cc * 8);
      }
    }
    if (threadIdx.x < FM_QT) {
      const int qi = q0 + threadIdx.x;
      ptq = -INFINITY; pqn = 0.0f;
      if (qi < p.nq) {
        const uint32_t tk = p.tkey[qi];
        const float qn = a.qn[qi];
        // no threshold yet, or a NaN threshold: every row is a candidate for the exact test
        const float T = tk >= 0xFF800000u ? INFINITY : key_to_float(tk);
        pqn = qn;
        ptq = METRIC == METRIC_DOT ? T + 0.000244140625f * qn + 4.7683716e-7f : T - (qn - 0.000244140625f * qn);      // (dot, + 2^-22: the reference's distance is the f32 value of 1 - x.q)   // s' <= ptq  with  s' = xk - 2 x.q  (dot: xk + 1 - x.q)
      }
    }
  };
  auto store = [&](int buf) {
    uint16_t *hi = qbuf + (size_t)(buf * 2 + 0) * FM_QT * CS, *lo = qbuf + (size_t)(buf * 2 + 1) * FM_QT * CS;
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int ch = threadIdx.x + 256 * u;
      const int cr = ch / (D / 8), cc = ch - cr * (D / 8);
      if (ch < NCH) {
        *reinterpret_cast<uint4 *>(hi + cr * CS + cc * 8) = ph[u];
        *reinterpret_cast<uint4 *>(lo + cr * CS + cc * 8) = pl[u];
      }
    }
    if (threadIdx.x < FM_QT) { tqs[buf * FM_QT + threadIdx.x] = ptq; qns[buf * FM_QT + threadIdx.x] = pqn; }
  };
  const int t0 = blockIdx.z;
  if (t0 >= ntiles) return;
  fetch(t0);
  store(0);
  __syncthreads();
  int it = 0;
  for (int t = t0; t < ntiles; t += gridDim.z, ++it) {
    const int buf = it & 1;
    const int tn = t + gridDim.z;
    if (tn < ntiles) fetch(tn);
    const uint16_t *hi = qbuf + (size_t)(buf * 2 + 0) * FM_QT * CS, *lo = qbuf + (size_t)(buf * 2 + 1) * FM_QT * CS;
    fm_f32x16 acc0, acc1;
#pragma unroll
    for (int v = 0; v < 16; ++v) { acc0[v] = 0.0f; acc1[v] = 0.0f; }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const fm_bf16x8 ah0 = *reinterpret_cast<const fm_bf16x8 *>(hi + j * CS + s * 16 + g * 8);
      const fm_bf16x8 al0 = *reinterpret_cast<const fm_bf16x8 *>(lo + j * CS + s * 16 + g * 8);
      const fm_bf16x8 ah1 = *reinterpret_cast<const fm_bf16x8 *>(hi + (32 + j) * CS + s * 16 + g * 8);
      const fm_bf16x8 al1 = *reinterpret_cast<const fm_bf16x8 *>(lo + (32 + j) * CS + s * 16 + g * 8);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, xh[s], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, xh[s], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, xl[s], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, xl[s], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al0, xh[s], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al1, xh[s], acc1, 0, 0, 0);
    }
    const int q0 = t * FM_QT;
    const float *tqb = tqs + buf * FM_QT;
    // fast reject: minimum over the lane's 32 pairs of  s = xk - 2 x.q - tq'  (dot: xk + 1 - x.q - tq'), two pairs per packed op.
    // A NaN s is ignored by the minimum and fails `<=` in the slow path alike (as `sp <= tq` did).
    const f2 mul2 = METRIC == METRIC_DOT ? f2{-1.0f, -1.0f} : f2{-2.0f, -2.0f};
    const float xk1 = METRIC == METRIC_DOT ? 1.0f + xk : xk;
    const f2 xk2 = {xk1, xk1};
    float mn = INFINITY;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
      for (int vq = 0; vq < 4; ++vq) {
        const int ib = blk * 32 + 8 * vq + 4 * g;
        const f4 tq4 = *reinterpret_cast<const f4 *>(tqb + ib);
        const f2 d01 = blk ? f2{acc1[vq * 4 + 0], acc1[vq * 4 + 1]} : f2{acc0[vq * 4 + 0], acc0[vq * 4 + 1]};
        const f2 d23 = blk ? f2{acc1[vq * 4 + 2], acc1[vq * 4 + 3]} : f2{acc0[vq * 4 + 2], acc0[vq * 4 + 3]};
        const f2 s01 = __builtin_elementwise_fma(mul2, d01, xk2 - f2{tq4.x, tq4.y});
        const f2 s23 = __builtin_elementwise_fma(mul2, d23, xk2 - f2{tq4.z, tq4.w});
        mn = __builtin_fminf(mn, __builtin_fminf(s01.x, s01.y));
        mn = __builtin_fminf(mn, __builtin_fminf(s23.x, s23.y));
      }
    }
    // Every pair the round-2 test `fma(-2, x.q, xk) <= tq` passed still passes: for finite operands the two forms differ by
    // 2^-24 |xk - tq|, and a pair near the boundary has |xk - tq| ~ 2 |x.q| <= |x|^2 + |q|^2, far inside the 3 * 2^-14 (|x|^2 +
    // |q|^2) the margin has to spare; tq = +inf gives s = -inf; a row whose |x|^2 overflowed (xk not finite: inf - inf = NaN,
    // which the minimum ignores) takes the slow path with the permissive comparison below.
    const bool odd = !(__builtin_fabsf(xk1) < INFINITY);
    if ((mn <= 0.0f || odd) && rvalid) {   // rare per lane (~1 % of lane-tiles): find the pairs, queue the row for each of their queries
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
        for (int vq = 0; vq < 4; ++vq) {
          const int ib = blk * 32 + 8 * vq + 4 * g;
          const f4 tq4 = *reinterpret_cast<const f4 *>(tqb + ib);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float dot = blk ? acc1[vq * 4 + e] : acc0[vq * 4 + e];
            const float sv = __builtin_fmaf(mul2.x, dot, xk1 - tq4[e]);
            if (odd ? !(sv > 0.0f) : (sv <= 0.0f)) {
              const int qi = q0 + ib + e;   // < nq: padded queries carry tq' = -inf
              const uint32_t pos = atomicAdd(&a.scnt[qi], 1u);
              if (pos < (uint32_t)FM_SQ_CAP) a.squeue[(int64_t)qi * FM_SQ_CAP + pos] = (uint32_t)(row - p.r0);
            }
          }
        }
      }
    }
    if (tn < ntiles) store(buf ^ 1);
    __syncthreads();
  }
}

// exact evaluation of the queued (query, row) pairs: one workgroup per query, one lane per queued row
template <int METRIC, typename TX>
__global__ __launch_bounds__(256) void flat_mfma_eval_kernel(FmArgs a, int d) {
  const FlatPool &p = a.p;
  const int qi = blockIdx.x;
  const uint32_t raw = a.scnt[qi];
  if (raw == 0) return;
  if (raw > (uint32_t)FM_SQ_CAP && threadIdx.x == 0) atomicOr(p.overflow, 1u);   // rows were lost: the caller's repair loop rescans
  const int c = (int)min(raw, (uint32_t)FM_SQ_CAP);
  const uint32_t tk = p.tkey[qi];
  const uint64_t tr = p.trid[qi];
  const float *qv = p.q + (int64_t)qi * d;
  for (int i = threadIdx.x; i < c; i += 256) {
    const int64_t row = p.r0 + (int64_t)a.squeue[(int64_t)qi * FM_SQ_CAP + i];
    const uint64_t rid = p.row_ids ? p.row_ids[row] : (uint64_t)row;
    // (q - x)^2 == (x - q)^2 and q*x == x*q bit for bit: the query is the f32 operand, the row is widened per element
    const float v = finish_metric<METRIC>(dist_exact_rt<METRIC, TX>(qv, static_cast<const TX *>(p.x_native) + row * d, d));
    const uint32_t key = order_key(v);
    if (key < tk || (key == tk && rid <= tr)) {
      const uint32_t pos = atomicAdd(&p.cnt[qi], 1u);
      if (pos < (uint32_t)p.cap) {
        p.pkeys[(int64_t)qi * p.cap + pos] = key;
        p.prids[(int64_t)qi * p.cap + pos] = rid;
      }
    }
  }
}

bool flat_mfma_supported(int metric, int d, int nq, const void *x, const float *q) {
  static const bool off = getenv("LANCE_HIP_NO_MFMA") != nullptr || getenv("LANCE_HIP_NO_MFMA_FLAT") != nullptr;
  if (off || (metric != LANCE_HIP_L2 && metric != LANCE_HIP_DOT)) return false;
  if (d % 16 != 0 || d < 16 || d > 128 || nq < 128) return false;
  return ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(q)) & 15) == 0;
}

int flat_mfma_prepare(lance_hip_ctx *ctx, const float *q, int nq, int d, const uint16_t **qhi, const uint16_t **qlo, const float **qn) {
  uint16_t *hi = ctx->scratch_t<uint16_t>("fm.qhi", (size_t)nq * d), *lo = ctx->scratch_t<uint16_t>("fm.qlo", (size_t)nq * d);
  float *n2 = ctx->scratch_t<float>("fm.qn", (size_t)nq);
  if (!hi || !lo || !n2) return LANCE_HIP_ENOMEM;
  hipLaunchKernelGGL(fm_query_prep_kernel, dim3(nq), dim3(64), 0, ctx->stream, q, nq, d, hi, lo, n2);
  *qhi = hi; *qlo = lo; *qn = n2;
  return LANCE_HIP_OK;
}

template <int KS>
static void fm_launch_ks(lance_hip_ctx *ctx, const FmArgs &a, int metric, dim3 grid) {
  constexpr int D = KS * 16;
  const size_t lds = std::max((size_t)FM_ROWS * (D + 4) * 4, (size_t)2 * 2 * FM_QT * (D + 8) * 2 + (size_t)4 * FM_QT * 4);
  auto go = [&](auto tag) {
    using TX = decltype(tag);
    if (metric == METRIC_DOT) hipLaunchKernelGGL((flat_filter_mfma_kernel<KS, METRIC_DOT, TX>), grid, dim3(256), lds, ctx->stream, a);
    else hipLaunchKernelGGL((flat_filter_mfma_kernel<KS, METRIC_L2, TX>), grid, dim3(256), lds, ctx->stream, a);
  };
  if (a.p.x_dtype == LANCE_HIP_F16) go(__half());
  else if (a.p.x_dtype == LANCE_HIP_I8) go(int8_t());
  else go(float());
}

static void fm_launch_eval(lance_hip_ctx *ctx, const FmArgs &a, int metric, int d) {
  auto go = [&](auto tag) {
    using TX = decltype(tag);
    if (metric == METRIC_DOT) hipLaunchKernelGGL((flat_mfma_eval_kernel<METRIC_DOT, TX>), dim3(a.p.nq), dim3(256), 0, ctx->stream, a, d);
    else hipLaunchKernelGGL((flat_mfma_eval_kernel<METRIC_L2, TX>), dim3(a.p.nq), dim3(256), 0, ctx->stream, a, d);
  };
  if (a.p.x_dtype == LANCE_HIP_F16) go(__half());
  else if (a.p.x_dtype == LANCE_HIP_I8) go(int8_t());
  else go(float());
}

// one epoch of flat.hip's v2 scan (rows [r0, r1) against the chunk's queries) -- every threshold must already be set
int launch_flat_filter_mfma(lance_hip_ctx *ctx, const FlatPool &e, int d, int metric, const uint16_t *qhi, const uint16_t *qlo, const float *qn) {
  FmArgs a;
  a.p = e; a.qhi = qhi; a.qlo = qlo; a.qn = qn;
  const int64_t rows = e.r1 - e.r0;
  if (rows <= 0) return LANCE_HIP_OK;
  LH_REQUIRE(rows < (1ll << 32), "flat scan: an epoch of %lld rows does not fit the 32-bit row queue", (long long)rows);
  a.scnt = ctx->scratch_t<uint32_t>("fm.scnt", (size_t)e.nq);
  a.squeue = ctx->scratch_t<uint32_t>("fm.squeue", (size_t)e.nq * FM_SQ_CAP);
  if (!a.scnt || !a.squeue) return LANCE_HIP_ENOMEM;
  LH_CHECK_HIP(lh::memset_async(a.scnt, 0, (size_t)e.nq * 4, ctx->stream));
  const unsigned rblocks = (unsigned)cdiv(rows, FM_ROWS);
  const int qtiles = (e.nq + FM_QT - 1) / FM_QT;
  int z = (int)std::min<int64_t>(qtiles, std::max<int64_t>(1, cdiv(2ll * ctx->num_cus, rblocks)));
  const dim3 grid(rblocks, 1, (unsigned)z);
  switch (d / 16) {
    case 1: fm_launch_ks<1>(ctx, a, metric, grid); break;
    case 2: fm_launch_ks<2>(ctx, a, metric, grid); break;
    case 3: fm_launch_ks<3>(ctx, a, metric, grid); break;
    case 4: fm_launch_ks<4>(ctx, a, metric, grid); break;
    case 5: fm_launch_ks<5>(ctx, a, metric, grid); break;
    case 6: fm_launch_ks<6>(ctx, a, metric, grid); break;
    case 7: fm_launch_ks<7>(ctx, a, metric, grid); break;
    default: fm_launch_ks<8>(ctx, a, metric, grid); break;
  }
  fm_launch_eval(ctx, a, metric, d);
  LH_CHECK_HIP(hipGetLastError());
  return LANCE_HIP_OK;
}

}  // namespace lh
