// flat_mfma_wide.hip -- batched flat scan on the matrix cores for LONG rows (d > 128: dbpedia-openai 1536, GIST 960, ...): a K-tiled
// bf16 product filters the (query, row) pairs, the few that can beat a query's current threshold are recomputed exactly.
//
//   FlatDistanceCal::distance_all / compute_distance     flat/storage.rs:345-402, flat.rs:95-148   (the exact arithmetic kept)
//   cosine                                                 lance-linalg/src/distance/cosine.rs:143-175 (cosine_fast, kept for the survivors)
//   final SortExec / pool selection                        scanner.rs:3386-3411                     (flat.hip: select kernel)
//
// flat_mfma.hip keeps a wave's 32 rows in registers for the whole query sweep, which ends at d = 128; beyond it the batched flat scan
// (the ground truth of BASELINE config 3, the refine source of every long-row index) was wide.hip's exact VALU kernel: 55 ms per 1000
// queries x 1M rows x 1536 dimensions, ~2 % of what the matrix cores can do with the same contraction.  Here the epoch's work is a plain
// tiled product  [rows x d] x [d x queries]:
//   * the rows are converted ONCE per call to a bf16 plane (round-to-nearest-even) with their |x|^2 beside it -- one read of the f32
//     column, after which every epoch and every query chunk streams 2 bytes per element;
//   * a workgroup (four waves, 2 x 2) owns a 128-row x 128-query tile, walks d in stages of 64: global -> registers (next stage in
//     flight) -> LDS (rows padded by 16 bytes: conflict-free ds_read_b128) -> v_mfma_f32_32x32x16_bf16, queries down the accumulator
//     registers and rows across the lanes as in flat_mfma.hip, so the epilogue is that kernel's: one fused multiply-add per pair against
//     the lane's row constant and the query's threshold term, folded into a running minimum, and only lanes whose minimum passes
//     look for the pairs and append the row to those queries' queues;
//   * flat_wide_eval_kernel recomputes the queued pairs in the reference's arithmetic and appends to the pool under the exact
//     (key, rowid) test -- the pools, and the answers, are the exact kernel's.
// One bf16 product (no hi / lo split).  bfloat16 keeps 8 significant bits: unit roundoff u = 2^-8 per operand, so
// |x~_i q~_i - x_i q_i| <= (2u + u^2) |x_i q_i| and by Cauchy-Schwarz |x~.q~ - x.q| <= (2^-7 + 2^-16) |x| |q| (attained when every element sits
// half an ulp from its neighbour with the errors aligned: tests/test_flat_wide_spec.py builds that case); the f32 accumulation of d <= 4096
// products adds <= d 2^-23 |x| |q| (twice the textbook bound, whatever the matrix pipe rounds inside) <= 4.9e-4 |x| |q|: together
// < EW |x| |q| with EW = 0.0084.  Hence
//   L2      s~ = |x|^2 + |q|^2 - 2 x~.q~  is within 2 EW |x||q| <= EW (|x|^2 + |q|^2) of the true value: test  |x|^2 (1 - EW) + |q|^2 (1 - EW) - 2 x~.q~ <= T
//   dot     s~ = 1 - x~.q~                is within EW |x||q| <= EW (|x|^2 + |q|^2) / 2:                 test  1 - x~.q~ - EW (|x|^2 + |q|^2) <= T
//   cosine  s~ = 1 - x~.q~ / (|x| |q|)    is within EW (the norms are the reference's own sqrt(y_norm), norm_l2(q): their f32 rounding
//           is 1e-6 relative, inside EWC = 0.0085):                                                     test  (1 - EWC - T) |q| |x| - x~.q~ <= 0
// (the reference's own f32 evaluation differs from the real-number value by ~d 2^-24 relative: inside the same margins).  A filter only
// widens: a pair that passes is decided by the exact arithmetic, a pair with true distance <= T cannot fail.  Rows or queries with
// non-finite / zero norms take the permissive comparison (everything of theirs is handed to the exact evaluation; a queue that
// overflows raises the pool-overflow flag and the caller's repair loop rescans with the exact kernel).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "exact.cuh"
#include "kernels.h"

#pragma clang fp contract(off)

namespace lh {

typedef short fw_bf16x8 __attribute__((ext_vector_type(8)));
typedef float fw_f32x16 __attribute__((ext_vector_type(16)));

// workgroup tile: 64 x BR rows by 64 x BQ queries (BR / BQ = 32-row / 32-query blocks per wave: template parameters; four waves, 2 x 2).
// Measured at C3's shape (1000 queries x 1M x 1536, filter ms per call; gpurun r05n / r05r / r05t): BR x BQ = 2 x 2 (144 VGPRs, three waves
// per SIMD) 7.7; 4 x 2 (236 VGPRs, two waves) 7.7; 2 x 1 (104 VGPRs, four waves) 9.3; 2 x 2 with 128-element stages (176 VGPRs, two waves) 14.3.
// elements of d per stage: template parameter KT (64 / 128); LDS row stride KT + 8 bf16 elements (144 / 272 bytes: conflict-free ds_read_b128)
constexpr int FW_SQ_CAP = 2048;       // queued rows per query and epoch
constexpr float FW_EW = 0.0084f;      // > 2^-7 + 2^-16 + 4096 * 2^-23 = 0.00832 (first written as 0.0045 = one operand's roundoff: the CPU spec test caught it)
constexpr float FW_EWC = 0.0085f;

__device__ __forceinline__ uint32_t fw_bf16_rne(float x) {
  const uint32_t u = __float_as_uint(x);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (u >> 16) | 0x40u;
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}

// rows -> bf16 plane + |x|^2: one wave per row
__global__ __launch_bounds__(256) void fw_rows_prep_kernel(const float *__restrict__ x, int64_t n, int d, uint16_t *__restrict__ xb,
                                                           float *__restrict__ xn2) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n) return;
  const f4 *src = reinterpret_cast<const f4 *>(x + row * d);
  uint2 *dst = reinterpret_cast<uint2 *>(xb + row * d);
  float s = 0.0f;
  for (int e4 = lane; e4 < d / 4; e4 += 64) {
    const f4 v = src[e4];
    s += v.x * v.x; s += v.y * v.y; s += v.z * v.z; s += v.w * v.w;
    dst[e4] = make_uint2(fw_bf16_rne(v.x) | (fw_bf16_rne(v.y) << 16), fw_bf16_rne(v.z) | (fw_bf16_rne(v.w) << 16));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) xn2[row] = s;
}

__global__ __launch_bounds__(64) void fw_query_prep_kernel(const float *__restrict__ q, int nq, int d, uint16_t *__restrict__ qb, float *__restrict__ qn2) {
  const int i = blockIdx.x;
  float s = 0.0f;
  for (int e = threadIdx.x; e < d; e += 64) {
    const float v = q[(int64_t)i * d + e];
    qb[(int64_t)i * d + e] = (uint16_t)fw_bf16_rne(v);
    s += v * v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (threadIdx.x == 0) qn2[i] = s;
}

struct FwArgs {
  FlatPool p;
  const uint16_t *xb;          // [n][d] bf16 rows (ALL rows of the call: indexed by absolute row)
  const float *xn2;            // [n] |x|^2
  const uint16_t *qb;          // [nq][d] bf16 queries of this chunk
  const float *qn2;            // [nq] |q|^2
  uint32_t *scnt;              // [nq] queued rows of this epoch
  uint32_t *squeue;            // [nq][FW_SQ_CAP] rows (relative to p.r0) whose surrogate distance passed
  int d;
  uint32_t nqt, nrt;           // query tiles, row tiles (one-dimensional grid, XCD-aware order: see the kernel)
};

// BR = 32-row blocks per wave (2: 128-row tiles, 64 accumulator registers; 4: 256-row tiles, 128 accumulator registers -- six operand reads
// per eight MFMAs instead of four per four, and twice the matrix work behind every stage's global loads)
// KT = elements of d per stage.  The next stage's global loads are issued when a stage starts and waited for when its MFMAs are done: one
// stage of matrix work is all that covers their latency, and the SQ counters show the waves stalled or parked 83 % of the time
// (profiles/r05q_*).  KT = 128 -- twice the work behind every load, half the barriers per MFMA, but 176 VGPRs = two waves per SIMD and 70 KB
// of LDS -- measured 1.9 x SLOWER than KT = 64 (gpurun r05r: 14.3 vs 7.6 ms at C3's shape): occupancy is what hides the latency here.
template <int METRIC, int BR, int KT, int BQ = 2>
__global__ __launch_bounds__(256, 2) void flat_filter_mfma_wide_kernel(FwArgs a) {
  constexpr int FW_KT = KT, FW_LS = KT + 8, HK = KT / 2, NU = KT / 16;      // a staging thread moves HK elements = NU 16-byte pieces of a row
  constexpr int FW_TM = 64 * BR, FW_TN = 64 * BQ;
  __shared__ __attribute__((aligned(16))) uint16_t Xs[FW_TM * FW_LS];
  __shared__ __attribute__((aligned(16))) uint16_t Qs[FW_TN * FW_LS];
  __shared__ __attribute__((aligned(16))) float tqs[FW_TN];
  const FlatPool &p = a.p;
  const int d = a.d;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, g = lane >> 5;
  const int wr = wave >> 1, wq = wave & 1;
  // XCD-aware tile order.  Workgroups are dealt to the eight XCDs round-robin (block b -> XCD b % 8), each XCD with its own L2: in plain
  // row-major order the nqt workgroups that share a row tile land on nqt DIFFERENT XCDs and every one of them pulls the same 128 x d rows
  // through the fabric (nqt = 8 at 1000 queries: 24 GB per epoch of a 3 GB plane -- the first version's bound: 8.3 ms whatever the tile
  // shape, gpurun r05n).  Here block b = 8 i + x takes row tile 8 (i / nqt) + x and query tile i % nqt: the workgroups of a row tile are
  // consecutive on ONE XCD, the tile comes from HBM once and the other query tiles hit that XCD's L2; the query plane (nq x d bf16, 3 MB)
  // is small enough to live in every L2.
  const uint32_t xi = blockIdx.x >> 3, xc = blockIdx.x & 7u;
  const uint32_t rt = (xi / a.nqt) * 8u + xc;
  if (rt >= a.nrt) return;      // (the grid is padded to whole groups of eight row tiles)
  const int q0 = (int)(xi % a.nqt) * FW_TN;
  const int64_t row0 = p.r0 + (int64_t)rt * (64 * BR);

  // per-query threshold term
  if (threadIdx.x < FW_TN) {
    const int qi = q0 + threadIdx.x;
    float t;
    if (qi < p.nq) {
      const uint32_t tk = p.tkey[qi];
      const float T = tk >= 0xFF800000u ? INFINITY : key_to_float(tk);      // no threshold yet / NaN threshold: every row is a candidate
      if constexpr (METRIC == METRIC_COSINE) {
        const float qn = p.q_norm[qi];
        t = (qn > 0.0f && qn < INFINITY) ? (1.0f - FW_EWC - T) * qn : -INFINITY;      // degenerate query: everything passes to the exact evaluation
      } else if constexpr (METRIC == METRIC_DOT) {
        t = T + FW_EW * a.qn2[qi] + 4.7683716e-7f;      // (+ 2^-22: the reference's distance is the f32 value of 1 - x.q -- products closer than an ulp of 1 tie there)
      } else {
        const float qn = a.qn2[qi];
        t = T - (qn - FW_EW * qn);
      }
    } else {
      t = METRIC == METRIC_COSINE ? INFINITY : -INFINITY;      // padded query: nothing passes (and the queue test below checks qi < nq)
    }
    tqs[threadIdx.x] = t;
  }

  // staging: thread (r, h) moves half a stage (HK elements) of query r and of the rows r, r + 128, .. per stage
  const int sr = threadIdx.x >> 1, sh = threadIdx.x & 1;
  constexpr int XR = FW_TM / 128;      // rows per thread and stage
  const bool qrow_ok = sr < FW_TN && q0 + sr < p.nq;
  const uint16_t *xsrc = a.xb + (row0 + sr) * (int64_t)d + sh * HK;
  const uint16_t *qsrc = a.qb + (int64_t)(q0 + sr) * d + sh * HK;
  uint4 px[XR][NU], pq[NU];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const bool in = k0 + sh * HK + u * 8 < d;      // d % 8 == 0: a 16-byte piece is inside or outside
#pragma unroll
      for (int xr = 0; xr < XR; ++xr)
        px[xr][u] = (row0 + sr + 128 * xr < p.r1 && in) ? *reinterpret_cast<const uint4 *>(xsrc + (int64_t)(128 * xr) * d + k0 + u * 8) : make_uint4(0, 0, 0, 0);
      pq[u] = (qrow_ok && in) ? *reinterpret_cast<const uint4 *>(qsrc + k0 + u * 8) : make_uint4(0, 0, 0, 0);
    }
  };
  auto store = [&]() {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
#pragma unroll
      for (int xr = 0; xr < XR; ++xr) *reinterpret_cast<uint4 *>(&Xs[(sr + 128 * xr) * FW_LS + sh * HK + u * 8]) = px[xr][u];
      if (sr < FW_TN) *reinterpret_cast<uint4 *>(&Qs[sr * FW_LS + sh * HK + u * 8]) = pq[u];
    }
  };

  fw_f32x16 acc[BR][BQ];
#pragma unroll
  for (int bi = 0; bi < BR; ++bi)
#pragma unroll
    for (int bj = 0; bj < BQ; ++bj)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[bi][bj][v] = 0.0f;

  fetch(0);
  store();
  __syncthreads();
  for (int k0 = 0; k0 < d; k0 += FW_KT) {
    const bool more = k0 + FW_KT < d;
    if (more) fetch(k0 + FW_KT);
#pragma unroll
    for (int s = 0; s < FW_KT / 16; ++s) {
      fw_bf16x8 aq[BQ], bx[BR];
#pragma unroll
      for (int b = 0; b < BQ; ++b) aq[b] = *reinterpret_cast<const fw_bf16x8 *>(&Qs[(wq * 32 * BQ + b * 32 + j) * FW_LS + s * 16 + g * 8]);
#pragma unroll
      for (int b = 0; b < BR; ++b) bx[b] = *reinterpret_cast<const fw_bf16x8 *>(&Xs[(wr * 32 * BR + b * 32 + j) * FW_LS + s * 16 + g * 8]);
#pragma unroll
      for (int bi = 0; bi < BR; ++bi)
#pragma unroll
        for (int bj = 0; bj < BQ; ++bj) acc[bi][bj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[bj], bx[bi], acc[bi][bj], 0, 0, 0);
    }
    __syncthreads();      // every wave has read this stage
    if (more) store();
    __syncthreads();
  }

  // epilogue: lane (j, g) of block (bi, bj) holds row j of the row block against the queries (v & 3) + 8 (v >> 2) + 4 g of the query block
#pragma unroll
  for (int bi = 0; bi < BR; ++bi) {
    const int64_t row = row0 + wr * 32 * BR + bi * 32 + j;
    const bool rvalid = row < p.r1;
    float xk = 0.0f;        // per-row constant of the test
    bool odd = false;       // row whose norm is not usable: permissive comparison
    if (rvalid) {
      if constexpr (METRIC == METRIC_COSINE) {
        xk = p.row_sy[row];      // sqrt(y_norm) of the reference
        odd = !(xk > 0.0f && xk < INFINITY);
      } else {
        const float n2 = a.xn2[row];
        xk = METRIC == METRIC_DOT ? 1.0f - FW_EW * n2 : n2 - FW_EW * n2;
        odd = !(__builtin_fabsf(xk) < INFINITY);
      }
    }
    float mn = INFINITY;
#pragma unroll
    for (int bj = 0; bj < BQ; ++bj) {
      const float *tqb = tqs + wq * 32 * BQ + bj * 32 + 4 * g;
#pragma unroll
      for (int vq = 0; vq < 4; ++vq) {
        const f4 t4 = *reinterpret_cast<const f4 *>(tqb + 8 * vq);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dot = acc[bi][bj][4 * vq + e];
          float sv;
          if constexpr (METRIC == METRIC_COSINE) sv = __builtin_fmaf(t4[e], xk, -dot);
          else if constexpr (METRIC == METRIC_DOT) sv = (xk - t4[e]) - dot;
          else sv = __builtin_fmaf(-2.0f, dot, xk - t4[e]);
          mn = __builtin_fminf(mn, sv);      // (a NaN is ignored by the minimum and fails `<=` below alike)
        }
      }
    }
    if ((mn <= 0.0f || odd) && rvalid) {      // rare per lane: find the pairs, queue the row for each of their queries
#pragma unroll
      for (int bj = 0; bj < BQ; ++bj) {
        const float *tqb = tqs + wq * 32 * BQ + bj * 32 + 4 * g;
#pragma unroll
        for (int vq = 0; vq < 4; ++vq) {
          const f4 t4 = *reinterpret_cast<const f4 *>(tqb + 8 * vq);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float dot = acc[bi][bj][4 * vq + e];
            float sv;
            if constexpr (METRIC == METRIC_COSINE) sv = __builtin_fmaf(t4[e], xk, -dot);
            else if constexpr (METRIC == METRIC_DOT) sv = (xk - t4[e]) - dot;
            else sv = __builtin_fmaf(-2.0f, dot, xk - t4[e]);
            const int qi = q0 + wq * 32 * BQ + bj * 32 + 8 * vq + 4 * g + e;
            if (qi < p.nq && (odd ? !(sv > 0.0f) : (sv <= 0.0f))) {
              const uint32_t pos = atomicAdd(&a.scnt[qi], 1u);
              if (pos < (uint32_t)FW_SQ_CAP) a.squeue[(int64_t)qi * FW_SQ_CAP + pos] = (uint32_t)(row - p.r0);
            }
          }
        }
      }
    }
  }
}

// exact evaluation of the queued (query, row) pairs: one workgroup per query, one lane per queued row
template <int METRIC>
__global__ __launch_bounds__(256) void flat_wide_eval_kernel(FwArgs a) {
  const FlatPool &p = a.p;
  const int d = a.d;
  const int qi = blockIdx.x;
  const uint32_t raw = a.scnt[qi];
  if (raw == 0) return;
  if (raw > (uint32_t)FW_SQ_CAP && threadIdx.x == 0) atomicOr(p.overflow, 1u);   // rows were lost: the caller's repair loop rescans
  const int c = (int)min(raw, (uint32_t)FW_SQ_CAP);
  const uint32_t tk = p.tkey[qi];
  const uint64_t tr = p.trid[qi];
  const float *qv = p.q + (int64_t)qi * d;
  float qnorm = 0.0f;
  if constexpr (METRIC == METRIC_COSINE) qnorm = p.q_norm[qi];
  for (int i = threadIdx.x; i < c; i += 256) {
    const int64_t row = p.r0 + (int64_t)a.squeue[(int64_t)qi * FW_SQ_CAP + i];
    const uint64_t rid = p.row_ids ? p.row_ids[row] : (uint64_t)row;
    float v;
    if constexpr (METRIC == METRIC_COSINE) v = cosine_exact_rt<float>(qv, qnorm, p.x + row * d, d);
    else v = finish_metric<METRIC>(dist_exact_rt<METRIC, float>(qv, p.x + row * d, d));
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

// Taken for f32 rows of 128 < d <= 4096 elements (and the dimensions <= 128 the fixed kernels do not instantiate), d % 16 == 0, query
// batches of at least 128, while the bf16 plane fits 16 GiB of scratch.  LANCE_HIP_NO_MFMA / LANCE_HIP_NO_MFMA_FLAT_WIDE: the exact kernel.
bool flat_mfma_wide_supported(int metric, int d, int nq, int64_t n, const float *x, const float *q) {
  static const bool off = getenv("LANCE_HIP_NO_MFMA") != nullptr || getenv("LANCE_HIP_NO_MFMA_FLAT_WIDE") != nullptr;
  if (off || !x || (metric != LANCE_HIP_L2 && metric != LANCE_HIP_DOT && metric != LANCE_HIP_COSINE)) return false;
  if (d % 16 != 0 || d < 32 || d > 4096 || nq < 128) return false;
  if ((uint64_t)n * (uint64_t)d * 2 > (16ull << 30)) return false;
  return ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(q)) & 15) == 0;
}

// once per flat_topk call: the bf16 plane and the norms of ALL rows
int flat_mfma_wide_prepare_rows(lance_hip_ctx *ctx, const float *x, int64_t n, int d, const uint16_t **xb_out, const float **xn2_out) {
  // the plane is half the column: exact size, and "no room" is reported as LH_NOT_TAKEN -- flat_topk then scans on the exact kernel
  uint16_t *xb = static_cast<uint16_t *>(ctx->scratch_exact("fw.xb", (size_t)std::max<int64_t>(n, 1) * d * 2));
  if (!xb) return LH_NOT_TAKEN;
  float *xn2 = ctx->scratch_t<float>("fw.xn2", (size_t)std::max<int64_t>(n, 1));
  if (!xn2) return LANCE_HIP_ENOMEM;
  if (n > 0) hipLaunchKernelGGL(fw_rows_prep_kernel, dim3((unsigned)cdiv((uint64_t)n, 4)), dim3(256), 0, ctx->stream, x, n, d, xb, xn2);
  *xb_out = xb; *xn2_out = xn2;
  return LANCE_HIP_OK;
}

int flat_mfma_wide_prepare_queries(lance_hip_ctx *ctx, const float *q, int nq, int d, const uint16_t **qb_out, const float **qn2_out) {
  uint16_t *qb = ctx->scratch_t<uint16_t>("fw.qb", (size_t)nq * d);
  float *qn2 = ctx->scratch_t<float>("fw.qn2", (size_t)nq);
  if (!qb || !qn2) return LANCE_HIP_ENOMEM;
  hipLaunchKernelGGL(fw_query_prep_kernel, dim3(nq), dim3(64), 0, ctx->stream, q, nq, d, qb, qn2);
  *qb_out = qb; *qn2_out = qn2;
  return LANCE_HIP_OK;
}

// one epoch of flat.hip's v2 scan (rows [r0, r1) against the chunk's queries) -- every threshold must already be set
int launch_flat_filter_mfma_wide(lance_hip_ctx *ctx, const FlatPool &e, int d, int metric, const uint16_t *xb, const float *xn2, const uint16_t *qb,
                                 const float *qn2) {
  FwArgs a;
  a.p = e; a.xb = xb; a.xn2 = xn2; a.qb = qb; a.qn2 = qn2; a.d = d;
  const int64_t rows = e.r1 - e.r0;
  if (rows <= 0 || e.nq <= 0) return LANCE_HIP_OK;
  LH_REQUIRE(rows < (1ll << 32), "flat scan: an epoch of %lld rows does not fit the 32-bit row queue", (long long)rows);
  LH_REQUIRE(metric != METRIC_COSINE || (e.row_sy && e.q_norm), "flat scan (matrix cores, long rows): cosine needs the precomputed norms");
  a.scnt = ctx->scratch_t<uint32_t>("fw.scnt", (size_t)e.nq);
  a.squeue = ctx->scratch_t<uint32_t>("fw.squeue", (size_t)e.nq * FW_SQ_CAP);
  if (!a.scnt || !a.squeue) return LANCE_HIP_ENOMEM;
  LH_CHECK_HIP(lh::memset_async(a.scnt, 0, (size_t)e.nq * 4, ctx->stream));
  static const int br_env = getenv("LANCE_HIP_FW_BR") ? atoi(getenv("LANCE_HIP_FW_BR")) : 0;      // A/B: 32-row blocks per wave (2 / 4)
  static const int bq_env = getenv("LANCE_HIP_FW_BQ") ? atoi(getenv("LANCE_HIP_FW_BQ")) : 0;      // A/B: 32-query blocks per wave (1 / 2)
  const int br = br_env == 2 || br_env == 4 ? br_env : 2;      // (4 measured no faster, gpurun r05n: the operand traffic was not the bound)
  const int bq = (bq_env == 1 || bq_env == 2) && br == 2 ? bq_env : 2;
  const uint64_t rblocks = cdiv((uint64_t)rows, (uint64_t)(64 * br));
  a.nqt = (uint32_t)cdiv((uint64_t)e.nq, (uint64_t)(64 * bq));
  a.nrt = (uint32_t)rblocks;
  const uint64_t nblocks = cdiv(rblocks, 8) * 8 * a.nqt;
  LH_REQUIRE(nblocks < (1ull << 31), "flat scan: an epoch of %lld rows x %d queries exceeds the grid of the long-row matrix-core filter",
             (long long)rows, e.nq);
  const dim3 grid((unsigned)nblocks, 1, 1);
  ScopedTimer t(ctx, "flat_mfma_wide");
  auto go = [&](auto br_tag, auto bq_tag) {
    constexpr int BR = decltype(br_tag)::value, BQ = decltype(bq_tag)::value;
    if (metric == METRIC_COSINE) {
      hipLaunchKernelGGL((flat_filter_mfma_wide_kernel<METRIC_COSINE, BR, 64, BQ>), grid, dim3(256), 0, ctx->stream, a);
      hipLaunchKernelGGL((flat_wide_eval_kernel<METRIC_COSINE>), dim3(e.nq), dim3(256), 0, ctx->stream, a);
    } else if (metric == METRIC_DOT) {
      hipLaunchKernelGGL((flat_filter_mfma_wide_kernel<METRIC_DOT, BR, 64, BQ>), grid, dim3(256), 0, ctx->stream, a);
      hipLaunchKernelGGL((flat_wide_eval_kernel<METRIC_DOT>), dim3(e.nq), dim3(256), 0, ctx->stream, a);
    } else {
      hipLaunchKernelGGL((flat_filter_mfma_wide_kernel<METRIC_L2, BR, 64, BQ>), grid, dim3(256), 0, ctx->stream, a);
      hipLaunchKernelGGL((flat_wide_eval_kernel<METRIC_L2>), dim3(e.nq), dim3(256), 0, ctx->stream, a);
    }
  };
  typedef std::integral_constant<int, 1> I1; typedef std::integral_constant<int, 2> I2; typedef std::integral_constant<int, 4> I4;
  if (br == 4) go(I4(), I2());
  else if (bq == 1) go(I2(), I1());
  else go(I2(), I2());
  LH_CHECK_HIP(hipGetLastError());
  return LANCE_HIP_OK;
}

}  // namespace lh
