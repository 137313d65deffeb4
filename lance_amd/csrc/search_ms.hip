// search_ms.hip -- the ADC filter scan on the matrix cores (8-bit PQ, L2 / cosine, d = 64 / 128; M = 16 / 32).
//
// Why: the 4-query integer scan (search_q.hip) was the dominant kernel of the C2 step (0.354 of 0.84 ms, gpurun r04g) and sat under two
// ceilings at once -- random LDS gathers (61 % of its LDS cycles are bank conflicts) and VALU issue (158 M wave instructions per launch,
// 40 % of them the table build per (query, partition) item).  Both come from evaluating
//     dist(q, row) = sum_m |r_m - c_m(code_m)|^2                 (pq/distance.rs:109-144, v2.rs:316-332: r = q - centroid_p)
// one table lookup per (row, sub-quantiser) per group of four queries.  The same number is |r - c^_row|^2 with c^_row the row's
// reconstruction (the concatenated codewords), i.e.
//     dist = |c^_row|^2 - 2 r . c^_row + |r|^2 :
// |c^_row|^2 is a constant of the stored row (4 bytes, computed once per index), |r|^2 a scalar per (query, partition) pair, and the
// cross term is a [queries x d] x [d x rows] product per partition -- matrix-core work (v_mfma_f32_32x32x16_f16).  The accumulator starts
// at |c^|^2 - limit, so the epilogue is ONE compare with zero per (row, query).
//
// It is a FILTER, exactly like the integer scan it replaces: survivors go to the same per-(query, probe) segments, with the accumulator
// value the merge kernel turns into an integer sum S ~ dist * s for its cut, and ivfpq_qmerge_kernel re-evaluates them in the reference's
// arithmetic -- ids and distances stay bit-equal to the oracle.  Soundness (no row with reference distance <= T is dropped): with
// u = 2^-11 (binary16),
//   |fl16(a) fl16(b) - a b| <= |a b| (2u + u^2)  ->  |2 r~.c~ - 2 r.c^| <= 2^-9 (1 + 2^-12) |r| |c^|      (Cauchy-Schwarz),
// the f32 accumulation of K <= 128 exact products adds <= K 2^-23 * 2 |r| |c^| (whatever the matrix pipe's internal rounding),
// the f32 evaluations of |c^|^2, |r|^2, of |c^|^2 - limit and of the reference's own table / sequential sum <= (d + M + 4) 2^-24 of their
// values (together < 2^-13 (|r|^2 + T) with |c^|^2 <= 2 |r|^2 + 2.1 T), and a row that can pass has |c^| <= |r| + sqrt(T)
// (triangle inequality); every SURVIVOR of the test has |c^| <= 1.01 (|r| + sqrt(T + E)) as well, so the same E bounds the error of
// every sum the merge kernel's histogram counts.  ms_prep_kernel evaluates
//   E = 1.05 [2^-9 1.02 |r| (|r| + sqrt T) + 2^-13 (|r|^2 + T)] + E_abs        (E_abs: f16 subnormals flushed, per element 2^-14 / sigma)
// per pair; sigma is a power of two per index that puts the largest codeword component near 2^13 (products and scales by powers
// of two are exact).  A pair whose residual overflows binary16, whose E exceeds 5 % of T, or with anything non-finite is handed
// to the exact rescan (the integer scan's overflow route); so is a segment with more than Q_CAP survivors.
//
// How the kernel got its shape (all on hardware, parity green at every step: tests/test_zz_gpu_mscan.py, tests/test_gpu_pm_scan.py, fuzz):
//   v1  256 rows x all pairs per workgroup, pairs staged global -> register -> LDS per super-block, flush with dependent atomics:
//       0.343 ms per 10k-query batch at C2 (no faster than the integer scan): a unit spent 43 us on ~4 us of arithmetic.
//   v2  LDS-DMA double buffer (global_load_lds_dwordx4, XOR-swizzled source + read), one 32-byte unit record, deferred segment stores,
//       sixteen compares hoisted in front of sixteen scalar tests: 0.267 ms.  Phase stamps (profiles/r04k_*): 29 % of a wave's life in
//       the tiles, 27 % before its first barrier, 23 % flush + DMA issue, 15 % tail flush, 6 % barriers; VALU busy 23 %, MFMA 15 %.
//       (hipcc does NOT put the vmcnt(0) an LDS-DMA needs in front of a loop-header barrier: two queries of 700 read a super-block
//       that had not landed until the wait was written out -- kept below.)
//   v3  the PAIR BLOCK resident in LDS (<= 512 pairs, 128 KiB), one persistent 512-lane workgroup per CU taking slices from a device
//       counter, eight waves working independently on 64-row chunks, no barrier inside a slice: 0.242 ms.
//   v4  (this file) the MFMA's operands trade places -- queries down the accumulator registers, ROWS across the lanes: the accumulator
//       starts from one value per lane, a wave carries one 32-row block, 128 VGPRs, sixteen waves per CU: 0.206 ms; slices taken
//       largest-first: 0.188 ms.  Timing experiments (profiles/r04n_*): without any survivor 0.139 ms, MFMA floor 0.041 ms -- a
//       32 x 32 block costs a wave ~2,600 cycles of mostly serial latency and four waves per SIMD are what the register file holds.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"
#include "exact.cuh"
#include "index.h"
#include "kernels.h"
#include "search_common.cuh"
#include "q_common.cuh"

namespace lh {

typedef _Float16 ms_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 ms_h4 __attribute__((ext_vector_type(4)));
typedef float ms_f16v __attribute__((ext_vector_type(16)));

constexpr float MS_SE = 30000.0f;   // the bound T maps to MS_SE units of the integer sums
constexpr float MS_SLACK_CAP = 1500.0f;   // largest E * s (units) the filter takes: 5 % of T
constexpr int MS_CUT_SHIFT = 6;     // merge histogram: 512 bins of 64 units
typedef __attribute__((address_space(1))) const void *ms_gptr;
typedef __attribute__((address_space(3))) void *ms_lptr;

// ---- index constants ---------------------------------------------------------------------------------------------------------
// mu (dot metric): the mean codeword of every sub-quantiser ([d], lance_hip_index::cb_mean) -- the plane holds the CENTRED codewords
__global__ __launch_bounds__(256) void ms_codebook_kernel(const float *__restrict__ cb, int64_t nwords, int sd, float scale, _Float16 *__restrict__ cbh,
                                                          float *__restrict__ cbn2, const float *__restrict__ mu) {
  const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (w >= nwords) return;
  float s = 0.0f;
  for (int u = 0; u < sd; ++u) {
    const float v = mu ? cb[w * sd + u] - mu[(w >> 8) * sd + u] : cb[w * sd + u];
    cbh[w * sd + u] = (_Float16)(v * scale);      // scale = -2 sigma
    s += v * v;
  }
  cbn2[w] = s;
}

__global__ __launch_bounds__(256) void ms_row_norm_kernel(const uint8_t *__restrict__ codes, int64_t n, int m, const float *__restrict__ cbn2, float sigma2,
                                                          float *__restrict__ row_cn2) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  const uint8_t *rc = codes + r * m;
  float s = 0.0f;
  for (int mm = 0; mm < m; ++mm) s += cbn2[mm * 256 + rc[mm]];
  row_cn2[r] = s * sigma2;
}

__global__ __launch_bounds__(256) void ms_max_kernel(const float *__restrict__ v, int64_t n, uint32_t *__restrict__ out) {      // v >= 0: float order = bit order
  float m = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, v[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

// dot metric: cmaxp[p] = upper bound of |centred reconstruction| over the rows of list p (v = sigma^2 |c'|^2 per row); one workgroup per list
__global__ __launch_bounds__(256) void ms_part_max_kernel(const float *__restrict__ v, const uint32_t *__restrict__ part_offsets, float inv_sigma,
                                                          float *__restrict__ cmaxp) {
  __shared__ float wm[4];
  const uint32_t b = part_offsets[blockIdx.x], e = part_offsets[blockIdx.x + 1];
  float m = 0.0f;
  for (uint32_t i = b + threadIdx.x; i < e; i += 256) m = fmaxf(m, v[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) cmaxp[blockIdx.x] = sqrtf(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))) * inv_sigma * 1.0001f;
}

// ---- per-pair pre-pass: f16 residual (scaled by sigma), limit, integer-sum scale --------------------------------------------------
struct MsPrepArgs {
  const float *q, *centroids;
  const uint32_t *pair_idx;     // class-A pairs grouped by partition: [0, pair_starts[nlist])
  const uint32_t *pair_starts;
  const uint32_t *probes;       // [nq][nprobes]
  const uint32_t *tbound;       // [nq]
  int d, nprobes, nlist, round_f16;
  float sigma;
  _Float16 *rh;                 // [npairs + 32][d]
  f4 *prm;                      // [npairs + 32]: {-lim sigma^2, s / sigma^2, |r|^2 s, pair (bits)}
  uint32_t *qslack;             // [nq] max over the query's pairs of ceil(E s) (zeroed before the launch)
  uint32_t *seg_cnt, *qovf, *ovf;
  uint32_t nan_slot;            // index into prm of the NaN-limit record (written here, loaded by the scan for padded pair slots)
  f2 *prm2;                     // [nq * nprobes] by PAIR: {s / sigma^2, (T' + E) s} -- what turns a survivor's accumulator value into its integer sum
  int dot = 0;                  // dot metric (no residual: the operand is q / 2 for every partition; limits and sums relative to a per-query base)
  float cmax = 0.0f;            // dot: upper bound of |CENTRED reconstruction of any stored row| (MsConst::cmax)
  const float *cmaxp = nullptr; // dot: [nlist] the same bound over the rows of one list: the pair's slack E and limit use it (the sums keep the index-wide
                                // one: all sums of a query share one base and one scale)
  float cmax_full = 0.0f;       // dot: upper bound of |reconstruction of any stored row| (the reference's own rounding terms)
  const float *mu = nullptr;    // dot: [d] the mean codeword of every sub-quantiser (lance_hip_index::cb_mean)
  int m = 0;
};

constexpr int MS_PPW = 4;      // pairs per wave of the pre-pass: their loads are in flight together (one pair per wave was 100k waves of
                               // load -> reduce -> store latency: 0.057 ms at C2 against 0.021 for the integer scan's residual pass)
__global__ __launch_bounds__(256) void ms_prep_kernel(MsPrepArgs p) {
  const uint32_t g0 = (blockIdx.x * 4u + (threadIdx.x >> 6)) * (uint32_t)MS_PPW;
  const uint32_t ng = p.pair_starts[p.nlist];
  const int lane = threadIdx.x & 63;
  if (blockIdx.x == 0 && threadIdx.x == 0) p.prm[p.nan_slot] = f4{__uint_as_float(0x7FC00000u), 0.0f, 0.0f, 0.0f};   // the record padded LDS slots load
  if (g0 >= ng) return;
  uint32_t pair[MS_PPW];
  const float *qv[MS_PPW], *cv[MS_PPW];
#pragma unroll
  for (int t = 0; t < MS_PPW; ++t) {
    const uint32_t g = min(g0 + (uint32_t)t, ng - 1u);      // the wave's surplus slots redo its last pair (results discarded)
    pair[t] = p.pair_idx[g];
  }
#pragma unroll
  for (int t = 0; t < MS_PPW; ++t) {
    qv[t] = p.q + (int64_t)(pair[t] / (uint32_t)p.nprobes) * p.d;
    cv[t] = p.centroids + (int64_t)p.probes[pair[t]] * p.d;
  }
  float n2[MS_PPW], vmax[MS_PPW], qmu[MS_PPW];
  bool bad[MS_PPW];
#pragma unroll
  for (int t = 0; t < MS_PPW; ++t) { n2[t] = 0.0f; vmax[t] = 0.0f; qmu[t] = 0.0f; bad[t] = false; }
  for (int e = lane; e < p.d; e += 64) {
    float v[MS_PPW];
    const float mue = p.dot ? p.mu[e] : 0.0f;
#pragma unroll
    for (int t = 0; t < MS_PPW; ++t) v[t] = p.dot ? qv[t][e] : qv[t][e] - cv[t][e];      // v2.rs:316-332, the subtraction of the exact path (dot: no residual)
    const float hs = p.dot ? 0.5f * p.sigma : p.sigma;      // dot: the codebook plane holds -2 sigma c, the operand is sigma q / 2 (powers of two: exact)
#pragma unroll
    for (int t = 0; t < MS_PPW; ++t) {
      if (p.round_f16 && !p.dot) v[t] = __half2float(__float2half_rn(v[t]));
      n2[t] += v[t] * v[t];
      qmu[t] += v[t] * mue;
      vmax[t] = fmaxf(vmax[t], fabsf(v[t]));
      bad[t] |= !(fabsf(v[t]) < INFINITY);
      if (g0 + (uint32_t)t < ng) p.rh[(int64_t)(g0 + t) * p.d + e] = (_Float16)(v[t] * hs);
    }
  }
#pragma unroll
  for (int t = 0; t < MS_PPW; ++t) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      n2[t] += __shfl_xor(n2[t], o, 64); qmu[t] += __shfl_xor(qmu[t], o, 64); vmax[t] = fmaxf(vmax[t], __shfl_xor(vmax[t], o, 64));
    }
    bad[t] = __any(bad[t]);
  }
  // lane t finishes pair t
  float n2l = n2[0], vml = vmax[0], qmul = qmu[0]; bool badl = bad[0]; uint32_t pairl = pair[0];
#pragma unroll
  for (int t = 1; t < MS_PPW; ++t) if (lane == t) { n2l = n2[t]; vml = vmax[t]; qmul = qmu[t]; badl = bad[t]; pairl = pair[t]; }
  if (lane >= MS_PPW || g0 + (uint32_t)lane >= ng) return;
  const uint32_t q = pairl / (uint32_t)p.nprobes;
  const float T = key_to_float(p.tbound[q]);     // class A: 0 < T < inf (dot: any finite T)
  const float sqd = sqrtf((float)p.d) + 1.0f;
  const float sig2 = p.sigma * p.sigma;
  float s, eu, lim, zsum;
  bool ok;
  if (p.dot) {
    // dist = 1 - q . c^ (pq/distance.rs:60-92, storage.rs:949-957) = (1 - q . mu) - q . c' with c' = c^ - mu the row's reconstruction from the
    // CENTRED codebook (mu: the mean codeword of every sub-quantiser) -- what L2 gets from its residuals: for rows whose components share a
    // sign |c'| is a third of |c^|, and with it every error term below.  The matrix product evaluates -(q . c')~; G >= |q . c'| for every
    // stored row (Cauchy-Schwarz, cmax = the largest |c'| of the index), base = (1 - q . mu) - G is below every distance; the sums are
    // (dist~ - base) s >= 0 with T - base + E mapped to MS_SE.  Error of dist~ against the reference's f32 table sum:
    //   binary16 operands: |fl16(a) fl16(b) - a b| <= |a b| (2u + u^2), summed: <= 2^-10 (1 + 2^-12) |q| |c'|;
    //   f32 accumulation of <= 128 exact products on top of the limit (<= 129 2^-24 (G + |lim|)), q . mu in f32 (<= 129 2^-24 |q| |mu|, and
    //   |mu| <= the longest reconstruction: |q| |mu| <= Gf), the three roundings of the limit (2^-24 (1 + |T| + |q . mu|) each), the centring
    //   c - mu in f32 (2^-24 |q| |c^|), the reference's own per-sub-vector dot products ((sd + 1) 2^-24 |q| |c^|) and the |q| |c^| part of its
    //   sequential sum of M entries (M 2^-24 |q| |c^|): <= 2^-16 (G + 1 + |T| + |q . mu| + Gf) together for sd + M <= 48;
    //   the rest of that sum (entries 1 - x_m, partial sums <= M + ..., the subtraction of M - 1): <= 2^-24 (M + 2)^2;
    //   flushed binary16 subnormals: per element 2^-14 on either operand.
    const float qn = sqrtf(n2l) * 1.000001f;
    const float G = qn * p.cmax * 1.000001f, Gf = qn * p.cmax_full * 1.000001f;
    const float e_abs = 6.1035156e-5f * sqd * (qn + 2.0f * p.cmax) / p.sigma + (float)p.d * 3.7252903e-9f / sig2;
    auto slack = [&](float g) {     // 2^-10 * 1.02, 2^-16, 2^-24
      return 1.05f * (9.9609375e-4f * g + 1.5258789e-5f * (g + 1.0f + fabsf(T) + fabsf(qmul) + Gf) + 5.9604645e-8f * (float)((p.m + 2) * (p.m + 2))) + e_abs;
    };
    const float Tq = (T - 1.0f) + qmul;         // the bound as a limit on -(q . c')
    const float Tp = (Tq + G) + slack(G);       // ... in the shifted domain (> 0: T bounds a real distance, every distance is >= base - E): per QUERY
    s = MS_SE / Tp;
    // the pair's own slack: the rows it meets are those of ONE list, whose longest centred reconstruction bounds |q . c'| for them
    const float Gp = fminf(G, qn * p.cmaxp[p.probes[pairl]] * 1.000001f);
    const float E = slack(Gp);
    eu = E * s * 1.1f + 3.0f;
    lim = Tq + E;                               // a row is kept when -(q . c')~ <= T - 1 + q . mu + E
    zsum = (lim + G) * s;                       // sum = (-(q . c')~ - lim) s + (lim + G) s = (dist~ - base) s
    ok = !badl && fabsf(T) < INFINITY && Tp > 0.0f && Tp < INFINITY && s > 0.0f && s < INFINITY && Gf < INFINITY && vml * p.sigma < 60000.0f &&
         eu <= MS_SLACK_CAP && fabsf(lim * sig2) < INFINITY && G * s < 1e30f && fabsf(zsum) < 1e30f;
  } else {
    s = MS_SE / T;
    const float rn = sqrtf(n2l) * 1.000001f, st = sqrtf(T) * 1.000001f;
    const float e_abs = 6.1035156e-5f * sqd * (3.0f * rn + 2.0f * st) / p.sigma + (float)p.d * 3.7252903e-9f / (p.sigma * p.sigma);
    const float E = 1.05f * (1.9921875e-3f * rn * (rn + st) + 1.2207031e-4f * (n2l + T)) + e_abs;     // 2^-9 * 1.02, 2^-13
    eu = E * s * 1.1f + 3.0f;
    lim = (T * 1.0000077f + E) - n2l;  // T (1 + 2^-17) + E - |r|^2; the cancellation's rounding sits inside the 2^-13 term
    zsum = (T * 1.0000077f + E) * s;
    ok = !badl && T > 0.0f && T < INFINITY && s > 0.0f && s < INFINITY && n2l < INFINITY && vml * p.sigma < 60000.0f &&
         eu <= MS_SLACK_CAP && fabsf(lim * sig2) < INFINITY && n2l * s < 1e30f;
  }
  f4 o;
  if (ok) {
    o.x = -(lim * sig2); o.y = s / sig2; o.z = n2l * s;      // MINUS the limit: the scan's accumulator starts from it
    // by pair, for the merge kernel: sum = (accumulator value, which is relative to the limit) * y + (T (1 + 2^-17) + E) s
    p.prm2[pairl] = f2{o.y, zsum};
    atomicMax(&p.qslack[q], (uint32_t)ceilf(eu));
  } else {
    // this (query, partition) pair goes to the exact rescan (ivfpq_qrescan_kernel), as an overflowed segment does.  The limit is a
    // NaN: no accumulator value compares <= to it, whatever the pair's (possibly infinite) f16 residual made of the products
    o.x = __uint_as_float(0x7FC00000u); o.y = 0.0f; o.z = 0.0f;
    p.seg_cnt[pairl] = (uint32_t)Q_CAP + 1u;
    p.qovf[q] = 1u;
    p.ovf[1u + atomicAdd(&p.ovf[0], 1u)] = pairl;
  }
  o.w = __uint_as_float(pairl);
  p.prm[g0 + (uint32_t)lane] = o;
}

// ---- slices: (partition, <= MS3_PB of its pairs, <= rows-per-slice of its rows) -----------------------------------------------------------
#ifndef LH_MS3_PB
#define LH_MS3_PB 512
#endif
constexpr int MS3_PB = LH_MS3_PB;   // pairs resident in LDS (128 KiB at d = 128); -DLH_MS3_PB=256: a variant build (scripts/build_variant.sh) for A/B runs
constexpr int MS3_RS = 3072;        // rows per slice (96 chunks of 32 for sixteen waves): the block's DMA + two barriers + the wait for the slowest wave are paid per slice.
                                    // 2048 until the end of round 6; on the kernel with the halved survivor branch 3072 is ahead on every batch shape tried (C2 bench:
                                    // scan 0.166 -> 0.155 ms, 24.7 -> 25.1 M q/s; nprobes 25, 4,000-query batches, 2M rows: +0.7 .. +2.3 %; 1024 / 1536 / 2560 / 2816 /
                                    // 3328 / 3584 / 4096 / 6144 / 8192 measured beside it: gpurun r06zzx, r06zzy).  Lists of ~3,900 rows become a 3,072-row slice and a
                                    // small one, and the small ones fill the largest-first schedule's tail.
static int ms_rows_per_slice() {    // LANCE_HIP_MS_RS: A/B of the slice height (multiple of 64)
  static const int v = [] { const char *e = getenv("LANCE_HIP_MS_RS"); const int x = e ? atoi(e) : MS3_RS; return x >= 64 ? (x / 64) * 64 : MS3_RS; }();
  return v;
}
// LANCE_HIP_MS_RS2: height of a partition's slices AFTER its first one (multiple of 64; default: the same height).  Smaller later slices
// give the largest-first schedule more small jobs for its tail.
static int ms_rows_per_slice2() {
  static const int v = [] { const char *e = getenv("LANCE_HIP_MS_RS2"); const int x = e ? atoi(e) : 0; return x >= 64 ? (x / 64) * 64 : ms_rows_per_slice(); }();
  return v;
}
struct MsSlice { uint32_t off, np, row_begin, row_count, gs, qp, pad0, pad1; };
// row slice `rs` of a partition of np rows: the first one of rs1 rows, the following ones of rs2
__host__ __device__ __forceinline__ uint32_t ms_nrs(uint32_t np, uint32_t rs1, uint32_t rs2) { return np <= rs1 ? (np ? 1u : 0u) : 1u + (np - rs1 + rs2 - 1u) / rs2; }
__device__ __forceinline__ uint32_t ms_row_begin(uint32_t rs, uint32_t rs1, uint32_t rs2) { return rs ? rs1 + (rs - 1u) * rs2 : 0u; }
__device__ __forceinline__ uint32_t ms_row_count(uint32_t rs, uint32_t np, uint32_t rs1, uint32_t rs2) {
  return rs ? min(rs2, np - ms_row_begin(rs, rs1, rs2)) : min(rs1, np);
}

// slice_start[p] = exclusive scan of (pair blocks of partition p) x (row slices of partition p); cls_cursor[c] = first position, in the
// largest-first order, of work class c (rows x pairs in 32 classes, class 0 the largest).  The persistent workgroups take slices from a
// counter, and a large slice taken last would be the kernel's tail (simulation on the bench's partition sizes and probe counts:
// makespan / mean 1.37 in index order, 1.15 largest-first at 722 slices on 256 CUs; measured: scan 0.211 -> 0.188 ms).
__device__ __forceinline__ uint32_t ms_work_class(uint32_t rows, uint32_t pairs, uint32_t rs_rows) {
  const uint64_t w = (uint64_t)rows * pairs;      // <= rs_rows * MS3_PB
  return 31u - (uint32_t)min<uint64_t>(31u, w * 32u / ((uint64_t)rs_rows * MS3_PB));
}
struct MsSplit { uint32_t npb, blk, nrs; };
__device__ __forceinline__ MsSplit ms_split(uint32_t qp, uint32_t np, uint32_t rs_rows, uint32_t rs2_rows) {
  MsSplit r;
  r.npb = (qp + MS3_PB - 1) / MS3_PB;
  r.blk = r.npb ? ((((qp + r.npb - 1) / r.npb) + 31u) & ~31u) : 0u;      // equal pair blocks, whole tiles of 32 (<= MS3_PB)
  r.nrs = ms_nrs(np, rs_rows, rs2_rows);
  return r;
}

__global__ __launch_bounds__(256) void ms_slice_table_kernel(const uint32_t *__restrict__ pair_starts, const uint32_t *__restrict__ part_offsets, int nlist,
                                                             uint32_t rs_rows, uint32_t rs2_rows, uint32_t cls_rows, uint32_t *__restrict__ slice_start,
                                                             uint32_t *__restrict__ slice_ctr, uint32_t *__restrict__ cls_cursor) {
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t carry_s;
  __shared__ uint32_t hist[32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) { carry_s = 0; slice_ctr[0] = 0u; }
  if (threadIdx.x < 32) hist[threadIdx.x] = 0u;
  __syncthreads();
  for (int base = 0; base < nlist; base += 256) {
    const int i = base + threadIdx.x;
    uint32_t v = 0;
    if (i < nlist) {
      const uint32_t qp = pair_starts[i + 1] - pair_starts[i], np = part_offsets[i + 1] - part_offsets[i];
      const MsSplit sp = ms_split(qp, np, rs_rows, rs2_rows);
      v = sp.npb * sp.nrs;
      // at most six distinct (rows, pairs) shapes per partition: first / middle / last row slice x full / last pair block
      for (uint32_t pb = 0; pb < sp.npb; pb += max(1u, sp.npb - 1u)) {
        const uint32_t pairs = min(sp.blk, qp - pb * sp.blk), npb_same = (pb + 1 == sp.npb) ? 1u : sp.npb - 1u;
        if (sp.nrs > 0) atomicAdd(&hist[ms_work_class(ms_row_count(0u, np, rs_rows, rs2_rows), pairs, cls_rows)], npb_same);
        if (sp.nrs > 2) atomicAdd(&hist[ms_work_class(rs2_rows, pairs, cls_rows)], npb_same * (sp.nrs - 2u));
        if (sp.nrs > 1) atomicAdd(&hist[ms_work_class(ms_row_count(sp.nrs - 1u, np, rs_rows, rs2_rows), pairs, cls_rows)], npb_same);
        if (sp.npb == 1) break;
      }
    }
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(incl, off, 64);
      if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    const uint32_t carry = carry_s;
    if (i < nlist) slice_start[i] = carry + woff + incl - v;
    __syncthreads();
    if (threadIdx.x == 255) carry_s = carry + woff + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    slice_start[nlist] = carry_s;
    uint32_t run = 0;
    for (int c = 0; c < 32; ++c) { cls_cursor[c] = run; run += hist[c]; }
  }
}

__global__ __launch_bounds__(256) void ms_slice_desc_kernel(const uint32_t *__restrict__ slice_start, const uint32_t *__restrict__ pair_starts,
                                                            const uint32_t *__restrict__ part_offsets, int nlist, uint32_t rs_rows, uint32_t rs2_rows, uint32_t cls_rows,
                                                            MsSlice *__restrict__ slices, uint32_t *__restrict__ cls_cursor, uint32_t *__restrict__ order) {
  const int part = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (part >= nlist) return;
  const uint32_t s0 = slice_start[part], s1 = slice_start[part + 1];
  if (s1 == s0) return;
  const uint32_t off = part_offsets[part], np = part_offsets[part + 1] - off;
  const uint32_t gs = pair_starts[part], qp = pair_starts[part + 1] - gs;
  const MsSplit sp = ms_split(qp, np, rs_rows, rs2_rows);
  // row slices vary fastest: the slices of one pair block are taken by different CUs at about the same time (its residuals come from L2)
  for (uint32_t t = (uint32_t)lane; t < s1 - s0; t += 64u) {
    const uint32_t pb = t / sp.nrs, rs = t - pb * sp.nrs;
    MsSlice u;
    u.off = off; u.np = np; u.row_begin = ms_row_begin(rs, rs_rows, rs2_rows); u.row_count = ms_row_count(rs, np, rs_rows, rs2_rows);
    u.gs = gs + pb * sp.blk; u.qp = min(sp.blk, qp - pb * sp.blk); u.pad0 = u.pad1 = 0u;
    slices[s0 + t] = u;
    // the slice's place in the largest-first order (inside a class: whatever the atomics give -- only the schedule depends on it)
    order[atomicAdd(&cls_cursor[ms_work_class(u.row_count, u.qp, cls_rows)], 1u)] = s0 + t;
  }
}

struct MscanArgs {
  const uint32_t *order;        // [slices] by decreasing work
  const MsSlice *slices;
  const uint32_t *slice_start;  // [nlist + 1]: slice_start[nlist] = number of slices
  uint32_t *slice_ctr;          // [1] next slice (zeroed by ms_slice_table_kernel)
  const uint8_t *codes;
  const _Float16 *cbh;          // [m][256][sd] = f16(-2 sigma c)
  const float *row_cn2;         // [n] sigma^2 |c^_row|^2
  const _Float16 *rh;           // [pairs][d]
  const f4 *prm;                // [pairs] grouped order: {-limit sigma^2, -, -, pair}
  const f2 *prm2;               // [nq * nprobes] by pair (read by the merge kernel; here only passed along)
  uint32_t nan_slot;
  int nlist, nprobes;
  uint32_t *seg_cnt, *seg_pos;
  uint2 *seg_pv = nullptr;      // [nq * nprobes][Q_CAP] the survivors: {storage position, accumulator value (bits)} -- ONE 8-byte store per survivor
                                // (two arrays were two partial cache lines per survivor: 53 bytes of HBM writes per 8-byte record, r05 PMC); the merge
                                // kernel scales the values into integer sums
  uint32_t *ovf;
  const uint32_t *allow;
#ifdef LH_TIMING_EXPERIMENTS
  int dbg = 0;                          // LANCE_HIP_MS_DBG (timing experiments, results WRONG): 1 = the flush drops its entries, 2 = every limit a NaN (nothing passes)
#endif
  unsigned long long *prof_slices = nullptr;   // LANCE_HIP_MS_PROF=1: [slices] ticks a workgroup spent on the slice (taken order)
  unsigned long long *prof = nullptr;   // LANCE_HIP_MS_PROF=1: [0] stage [1] gather [2] tiles [3] flush [4] life [5] waves [6] chunks [7] longest life
};

// Timing experiments that make the results WRONG (LANCE_HIP_MS_DBG) exist only in builds with -DLH_TIMING_EXPERIMENTS
// (scripts/build_variant.sh NAME search_ms.hip -- -DLH_TIMING_EXPERIMENTS): the product library cannot be switched into them.
#ifdef LH_TIMING_EXPERIMENTS
#define MS_DBG(p) ((p).dbg)
#else
#define MS_DBG(p) 0
#endif

// ---- the scan -------------------------------------------------------------------------------------------------------------------------
// One persistent 1024-lane workgroup per CU.  It takes a slice from the device counter (largest first), brings the block's f16 residuals
// into LDS once (LDS-DMA, lane-linear destination, XOR-swizzled SOURCE address and the same swizzle on the ds_read_b128 address), limits
// and pair ids beside them, and then its sixteen waves work independently: a wave takes 32-row chunks from an LDS counter, gathers the
// chunk's reconstruction into registers (the MFMA's B operand: lane (j, g) = row j, k-slice g) and runs it against every 32-query tile
// of the block: D = Q x R^T, queries down the accumulator registers, rows across the lanes, accumulator started at |c^|^2 - limit.
// Sixteen compares with zero -> sixteen lane masks -> scalar tests; a survivor is queued (pair slot, row, accumulator value); a full
// queue half requests its segment slots (atomicAdd, nothing waits) while the other half fills, and writes behind them a few tiles later.
constexpr int MS_QH = 64;          // entries per queue half (one per lane)
template <int SD, int KS, bool PROF = false>
__global__ __launch_bounds__(1024, 4) void ivfpq_mscan_kernel(MscanArgs p) {
  constexpr int D = KS * 16;
  constexpr int M = D / SD;
  constexpr int RB = D * 2;                 // bytes of a pair's f16 residual
  constexpr int CPR = RB / 16;              // 16-byte chunks per pair row (16 / 8)
  constexpr int RPK = 256 / RB;             // pair rows per 256 bytes (1 / 2): the swizzle key is (row / RPK) & (CPR - 1)
  constexpr int SPI = 16384 / RB;           // pair slots one 1024-lane DMA pass covers (64 / 128)
  __shared__ __attribute__((aligned(16))) char sB[MS3_PB * RB];
  __shared__ __attribute__((aligned(16))) float sLim[MS3_PB];
  __shared__ __attribute__((aligned(16))) uint32_t sPair[MS3_PB];
  __shared__ __attribute__((aligned(8))) uint2 sQ[16][2][MS_QH + 1];      // per wave: two halves (fill one while the other's atomics are in flight)
  __shared__ uint32_t s_slice, s_chunk;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // (scalar: the queue's base address stays out of the VGPRs)
  const int j = lane & 31, g = lane >> 5;
  const uint32_t nslices = p.slice_start[p.nlist];

  uint2 *const qw = &sQ[wave][0][0];
  uint32_t cur = 0;     // wave-uniform: offset (entries) of the half being filled; the other half's segment slots have been requested
  uint32_t qn = 0;      // wave-uniform: entries in the half being filled
  uint32_t pd_n = 0, pd_base = 0;      // wave-uniform: entries of the other half in flight (this lane's slot in pd_k), their chunk's first position
  uint32_t pd_k = 0;
  float ncn2v = 0.0f;   // minus |c^|^2 of the lane's row in the current chunk (flush_begin finishes the survivors' values with it)
  long long pc_t0 = 0, pc_stage = 0, pc_gather = 0, pc_tiles = 0, pc_flush = 0, pct = 0, pc_nchunk = 0;      // PROF: s_memtime stamps
  if constexpr (PROF) { pc_t0 = clock64(); pct = pc_t0; }
  // Flush in two steps, one queue half apart: flush_begin requests a segment slot per entry of the full half (atomicAdd, nothing waits)
  // and the halves trade places; flush_end -- called before the NEXT flush_begin, several tiles later -- writes position and value behind
  // the slots.  One register of pending state per lane (the first rows-on-lanes build kept pair / position / value / scale there and
  // spilled; a synchronous flush measured 33 % of a wave's life, gpurun r04n).  A queue entry names the pair's LDS slot, so both steps
  // run while the slice's block is resident.
  auto flush_end = [&]() {
    if (pd_n) {
      if ((uint32_t)lane < pd_n && pd_k != 0xFFFFFFFFu) {
        const uint2 ent = qw[((uint32_t)(MS_QH + 1) - cur) + (uint32_t)lane];
        const uint32_t pair = sPair[ent.x >> 8];
        if (pd_k < (uint32_t)Q_CAP) {
          p.seg_pv[(int64_t)pair * Q_CAP + pd_k] = make_uint2(pd_base + (ent.x & 255u), ent.y);
        } else if (pd_k == (uint32_t)Q_CAP) {      // the segment lost survivors from here on: exact rescan of this (query, probe)
          p.ovf[1u + atomicAdd(&p.ovf[0], 1u)] = pair;      // (the rescan kernel flags the query)
        }
      }
      pd_n = 0;
    }
  };
  auto flush_begin = [&](uint32_t pos_base) {      // (after flush_end)
    pd_k = 0xFFFFFFFFu;
    // The tile loop leaves RAW entries: x = (tile << 10) | (accumulator register << 6) | lane, y = the accumulator itself.  Here, with all
    // lanes busy and once per 64 survivors, they become {(pair slot << 8) | row, acc + |c^|^2}: the survivor branch -- executed with one
    // or two lanes active, ~6 times per tile -- no longer assembles slot and row from the lane number nor forms the value (round 6,
    // second half: 14 -> 9 VALU instructions per nonzero compare mask).  Row j's |c^|^2 sits in lane j of this chunk.
    {
      const uint2 raw = qw[cur + (uint32_t)lane];
      const uint32_t ln_e = raw.x & 63u, vv = (raw.x >> 6) & 15u, jbb = raw.x >> 10;
      const float nv = __shfl(ncn2v, (int)(ln_e & 31u), 64);
      const uint2 ent = make_uint2(((jbb * 32u + 4u * (ln_e >> 5) + (vv & 3u) + 8u * (vv >> 2)) << 8) | (ln_e & 31u), __float_as_uint(__uint_as_float(raw.y) - nv));
      if ((uint32_t)lane < qn && !(MS_DBG(p) & 1)) {
        qw[cur + (uint32_t)lane] = ent;
        if (row_allowed(p.allow, pos_base + (ent.x & 255u))) pd_k = atomicAdd(&p.seg_cnt[sPair[ent.x >> 8]], 1u);
      }
    }
    pd_n = qn; pd_base = pos_base; qn = 0;
    cur = (uint32_t)(MS_QH + 1) - cur;
  };

  for (;;) {
    __syncthreads();      // every wave is done with the previous slice's block (and with s_slice / s_chunk)
    if (threadIdx.x == 0) { s_slice = atomicAdd(p.slice_ctr, 1u); s_chunk = 0u; }
    __syncthreads();
    const uint32_t slice = s_slice;
    if (slice >= nslices) break;
    long long pc_slice0 = 0;
    if constexpr (PROF) pc_slice0 = clock64();
    const MsSlice U = p.slices[p.order[slice]];
    const int Qp = (int)U.qp;
    const int nslots = ((Qp + 31) >> 5) << 5;
    {
      // LDS-DMA of the pair block: lane-linear destination, swizzled source; padded slots keep whatever was there (their limit is a NaN)
      const char *src = reinterpret_cast<const char *>(p.rh + (int64_t)U.gs * D);
#pragma unroll 4
      for (int it = 0; it < MS3_PB / SPI; ++it) {
        const int slot = it * SPI + wave * (SPI / 16) + lane / CPR, k = lane % CPR;
        if (it * SPI < nslots && slot < Qp)
          __builtin_amdgcn_global_load_lds((ms_gptr)(src + (int64_t)slot * RB + ((k ^ ((slot / RPK) & (CPR - 1))) << 4)),
                                           (ms_lptr)(&sB[(it * SPI + wave * (SPI / 16)) * RB]), 16, 0, 0);
      }
      // limits (waves 0-7) and pair ids (waves 8-15): one dword per slot out of the {limit, -, -, pair} records
      const int slot = (wave & 7) * 64 + lane;
      if (slot < nslots) {
        const float *ps = reinterpret_cast<const float *>(slot < Qp && !(MS_DBG(p) & 2) ? p.prm + (int64_t)U.gs + slot : p.prm + p.nan_slot);
        if (wave < 8) __builtin_amdgcn_global_load_lds((ms_gptr)ps, (ms_lptr)(&sLim[(wave & 7) * 64]), 4, 0, 0);
        else __builtin_amdgcn_global_load_lds((ms_gptr)(ps + 3), (ms_lptr)(&sPair[(wave & 7) * 64]), 4, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the DMA is ordered for the readers by the issuer's vmcnt(0) + the barrier
    __syncthreads();
    if constexpr (PROF) { const long long t = clock64(); pc_stage += t - pct; pct = t; }
    const uint32_t nchunks = (U.row_count + 31u) >> 5;
    const int nblk = nslots >> 5;
    const int row_end = (int)(U.row_begin + U.row_count);      // <= np

    for (;;) {
      uint32_t c = 0;
      if (lane == 0) c = atomicAdd(&s_chunk, 1u);
      c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
      if (c >= nchunks) break;
      const int row0 = (int)(U.row_begin + c * 32u);

      // the f16 reconstruction of the wave's 32 rows as the MFMA's B operand (lane (j, g): row j, k-slice g), |c^|^2 of row j
      ms_h8 rw[KS];
      {
        const int rowc = min(row0 + j, row_end - 1);
        uint32_t cw[M / 4];
        {
          const uint4 *rc4 = reinterpret_cast<const uint4 *>(p.codes + ((int64_t)U.off + rowc) * M);      // rows of 16 / 32 bytes, 16-byte aligned
#pragma unroll
          for (int w = 0; w < M / 16; ++w) { const uint4 t = rc4[w]; cw[4 * w] = t.x; cw[4 * w + 1] = t.y; cw[4 * w + 2] = t.z; cw[4 * w + 3] = t.w; }
        }
        ncn2v = row0 + j < row_end ? -p.row_cn2[(int64_t)U.off + rowc] : -INFINITY;      // minus |c^|^2; a padded row never passes (nothing is <= -inf)
        auto code = [&](int mm) -> uint32_t { return (cw[mm >> 2] >> (8 * (mm & 3))) & 255u; };
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          if constexpr (SD == 8) {
            const uint32_t c0 = code(2 * s), c1 = code(2 * s + 1);
            const int mm = 2 * s + g;
            rw[s] = *reinterpret_cast<const ms_h8 *>(reinterpret_cast<const char *>(p.cbh) + (uint32_t)(((uint32_t)mm * 256u + (g ? c1 : c0)) * 16u));
          } else {
            static_assert(SD == 4, "sub-dimension 4 / 8");
            const uint32_t c0 = g ? code(4 * s + 2) : code(4 * s), c1 = g ? code(4 * s + 3) : code(4 * s + 1);
            const int mm = 4 * s + 2 * g;
            const ms_h4 lo = *reinterpret_cast<const ms_h4 *>(reinterpret_cast<const char *>(p.cbh) + (uint32_t)(((uint32_t)mm * 256u + c0) * 8u));
            const ms_h4 hi = *reinterpret_cast<const ms_h4 *>(reinterpret_cast<const char *>(p.cbh) + (uint32_t)(((uint32_t)(mm + 1) * 256u + c1) * 8u));
            rw[s] = ms_h8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          }
        }
      }
      const uint32_t pos_base = U.off + (uint32_t)row0;
      // The gathers must have landed before the first MFMA anyway: wait for them HERE, once, and hand the registers to the tile loop
      // through empty asm statements -- which makes them VALU-defined values for the compiler's wait-count pass.  Without this it put
      // `s_waitcnt vmcnt(8) .. vmcnt(1)` in front of the eight MFMAs of EVERY tile (it cannot prove, inside the loop, that the chunk's
      // loads are done), and since the counter is in-order those waits also drained the flush's atomics and stores that were meant to
      // stay in flight across the following tiles (round 5: read off the gfx950 assembly; the r04 stamps had the time in "tiles").
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(rw[s]));
      asm volatile("" : "+v"(ncn2v));
      if constexpr (PROF) {
        const long long t = clock64(); pc_gather += t - pct; pct = t;
      }

      // A: a tile's 32 queries (lane (i, g): query i, k-slice g); D[query i][row j]: lane (j, g) holds queries i = (v & 3) + 8 (v >> 2) + 4 g.
      // The loop is rotated: tile jb + 1's operand is requested right behind tile jb's MFMAs (into the same registers -- the chain has
      // read them by then), so its LDS round trip runs under the compare / queue work instead of in front of the next chain.
      ms_h8 qa[KS];
      auto load_qa = [&](int jbx) {
        const int sl = jbx * 32 + j;
        const char *br = &sB[sl * RB];
        const int key = (sl / RPK) & (CPR - 1);
#pragma unroll
        for (int s = 0; s < KS; ++s) qa[s] = *reinterpret_cast<const ms_h8 *>(br + (((2 * s + g) ^ key) << 4));
      };
      load_qa(0);
      for (int jb = 0; jb < nblk; ++jb) {
        const int slot = jb * 32 + j;
        // The accumulator starts at MINUS the query's limit, read straight from LDS into the accumulator registers (the pre-pass stores
        // the limits negated), and the row's |c^|^2 enters at the compare: acc <= -|c^|^2.  Round 4 started it at |c^|^2 - limit and
        // compared with zero: sixteen v_sub per tile in front of the MFMA chain and, in the generated code, a sixteen-register splat of
        // |c^|^2 held across the tile loop (the spills).  Same test up to one f32 rounding fewer (the compare itself is exact); the
        // survivor's value is acc + |c^|^2, formed in the survivor branch.
        ms_f16v acc;
        {
          const f4 *lp = reinterpret_cast<const f4 *>(&sLim[jb * 32 + 4 * g]);
#pragma unroll
          for (int vq = 0; vq < 4; ++vq) {
            const f4 t = lp[2 * vq];
            acc[4 * vq] = t.x; acc[4 * vq + 1] = t.y; acc[4 * vq + 2] = t.z; acc[4 * vq + 3] = t.w;
          }
        }
        // (the queue entry's slot / row bits are put together inside the survivor branch from the loop counter and the lane number:
        // kept in a register across the tile loop the compiler turned them into an induction variable it SPILLED, and the reload -- a
        // scratch load -- put an `s_waitcnt vmcnt(0)` into every survivor branch, i.e. each first survivor after a flush waited for
        // the flush's atomics to come back)
        // room for a tile's usual yield; a burst beyond the queue is handled after the tile
        if (qn > (uint32_t)(MS_QH - 32)) {      // room for a tile's usual yield (6-7 survivors at C2); a burst beyond the half is handled after the tile
          if constexpr (PROF) { const long long t = clock64(); pc_tiles += t - pct; pct = t; }
          flush_end(); flush_begin(pos_base);
          if constexpr (PROF) { const long long t = clock64(); pc_flush += t - pct; pct = t; }
        }
        uint32_t qraw = qn;      // wave-uniform: entries the tile wanted (qn stays clamped to the queue)
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(qa[s], rw[s], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        load_qa(min(jb + 1, nblk - 1));      // (the last tile re-reads itself: harmless)
        __builtin_amdgcn_sched_barrier(0);
        uint64_t mk[16];
#pragma unroll
        for (int v = 0; v < 16; ++v) mk[v] = __ballot(acc[v] <= ncn2v);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          if (mk[v]) {
            const uint32_t idx = min(qraw + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk[v] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk[v], 0u)),
                                     (uint32_t)MS_QH);      // entry MS_QH: the bin of a burst
            if (acc[v] <= ncn2v) {     // (the same compare: the compiler reuses its lane mask as the exec mask)
#ifdef MS_LN_RECOMPUTE
              uint32_t ln;            // lane number, recomputed here (two VALU operations, no register held across the loop)
              asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
#else
              const uint32_t ln = (uint32_t)lane;
#endif
              qw[cur + idx] = make_uint2((((uint32_t)jb << 10) | (uint32_t)(v << 6)) | ln, __float_as_uint(acc[v]));      // raw: flush_begin decodes
            }
            qraw += (uint32_t)__popcll(mk[v]);
          }
        }
        qn = min(qraw, (uint32_t)MS_QH);
        if (qraw > (uint32_t)MS_QH) {
          // more than the half's room in one tile (>= 2 % of its cells pass -- these pairs' segments would overflow anyway):
          // survivors were dropped, so every pair of the tile is handed to the exact rescan: the count jumps past Q_CAP, and whoever
          // crosses it lists the pair
          if (g == 0 && slot < Qp) {
            const uint32_t pair = sPair[slot];
            const uint32_t k = atomicAdd(&p.seg_cnt[pair], (uint32_t)Q_CAP + 1u);
            if (k <= (uint32_t)Q_CAP) p.ovf[1u + atomicAdd(&p.ovf[0], 1u)] = pair;      // (the rescan kernel flags the query)
          }
        }
      }
      if constexpr (PROF) { const long long t = clock64(); pc_tiles += t - pct; pct = t; }
      flush_end(); flush_begin(pos_base);      // (positions are relative to the chunk: a half never spans two chunks)
      if constexpr (PROF) { const long long t = clock64(); pc_flush += t - pct; pct = t; ++pc_nchunk; }
    }
    flush_end();      // before the block (and its pair ids) leaves LDS
    if constexpr (PROF) {
      __syncthreads();
      if (threadIdx.x == 0 && p.prof_slices) p.prof_slices[slice] = (unsigned long long)(clock64() - pc_slice0);
    }
  }
  if constexpr (PROF) {
    if (lane == 0) {
      const long long t1 = clock64();
      atomicAdd(&p.prof[0], (unsigned long long)pc_stage);    // slice counter, DMA of the block, barriers, waiting for the slowest wave
      atomicAdd(&p.prof[1], (unsigned long long)pc_gather);   // codes -> codeword gathers, |c^|^2
      atomicAdd(&p.prof[2], (unsigned long long)pc_tiles);    // LDS reads, MFMA, compares, queue
      atomicAdd(&p.prof[3], (unsigned long long)pc_flush);    // queue -> atomics -> stores
      atomicAdd(&p.prof[4], (unsigned long long)(t1 - pc_t0));
      atomicAdd(&p.prof[5], 1ull);
      atomicAdd(&p.prof[6], (unsigned long long)pc_nchunk);
      atomicMax(&p.prof[7], (unsigned long long)(t1 - pc_t0));
    }
  }
}

// ---- the bound pass on the matrix cores -----------------------------------------------------------------------------------------------
// Every query needs an upper bound T of its final k*refine-th ADC distance before the filter (search_q.hip, "integer bound pass": a
// 512-bin histogram of the nearest partition's sums, four queries per LDS gather, a 16 x 256 table built per item: 0.080 ms per 10,000
// queries at C2 alone on the device, 0.128 ms when it shares the device with the other engine contexts' kernels -- as much as the
// refine).  The same histogram from the matrix product of the scan above: one workgroup per (partition, block of <= 64 of the queries
// whose nearest partition it is), the block's f16 residuals AND the whole f16 codebook (64 KiB) resident in LDS -- so a chunk's
// reconstruction gather is eight ds_read_b128 instead of a dependent global round trip --, sixteen waves taking 32-row chunks: per
// 32-query tile 8 x v_mfma_f32_32x32x16_f16 from a zero accumulator, then per (row, query) cell
//     bin = floor(((acc + sigma^2 |c^|^2) / sigma^2 + |r|^2) * sb) = floor(dist~ * sb),     sb = 496 / (mean distance of a random code),
// one FMA, one compare, one LDS atomic on a histogram of 512 u16 bins per query (two bins per word; partitions of >= 65,536 rows keep
// the integer pass).  If at least k*refine allowed rows have dist~ < (b + 1) / sb =: Ta then -- by the error bound of the header
// (every row that passes a test with threshold Ta has |c^| <= 1.01 (|r| + sqrt(Ta + E)), so |dist~ - dist_reference| <= E(Ta)) --
// at least k*refine reference distances are <= Ta + E(Ta):   T = (Ta + 1.1 E(Ta)) (1 + 2^-16)   is a valid bound.  The epilogue's
// own three f32 roundings (the add of |c^|^2, the FMA) are ~2^-19 (|r|^2 + T) and sit inside E's 2^-13 term.  A query whose residual
// overflows binary16, with a non-finite value anywhere, or with fewer than k*refine countable rows gets no bound here (class B: the
// exact pair kernel), exactly as in the integer pass.  T only has to be AN upper bound: ids and distances do not depend on which pass
// produced it (tests/test_zz_gpu_msbound.py compares both against the oracle).
constexpr int MSB_BQ = 64;           // queries per workgroup (two tiles)
constexpr int MSB_BINS = 512;        // bins per query, two per LDS word
constexpr float MSB_MEAN_BIN = 496.0f;   // the mean distance of a random code lands here (the integer pass: SE / 8)
struct MsBoundArgs {
  const float *q, *centroids, *cb_mean;      // cb_mean: [d] mean codeword of every sub-quantiser, [d] = sum over m of the mean |c|^2
  const uint32_t *pair_idx0;    // [nq] query indices grouped by nearest partition
  const uint32_t *item_start;   // [nlist + 1] blocks of MSB_BQ queries
  const int4 *desc;             // {partition, first grouped query, queries (1 .. MSB_BQ), -}
  const uint32_t *part_offsets;
  const uint8_t *codes;
  const _Float16 *cbh;          // [m][256][sd] f16(-2 sigma c)
  const float *row_cn2;         // [n] sigma^2 |c^_row|^2
  int d, nlist, keff, round_f16;
  float sigma;
  uint32_t *tglobal;            // [nq] bound key (atomicMin)
  const uint32_t *allow;
  int dot = 0;                  // dot metric: operand q / 2, no residual, centred codebook plane; bins of (dist~ - base), base = (1 - q . mu) - |q| cmax (ms_prep_kernel)
  float cmax = 0.0f, cmax_full = 0.0f;
  const float *cmaxp = nullptr; // dot: [nlist] per-list bound of |centred reconstruction| (the workgroup's list: bins and slack use it)
  uint32_t nb = 1;              // entries of pair_idx0 are query * nb + b: the query's b-th nearest list (every list gives a valid bound; atomicMin keeps the best)
};

template <int SD, int KS>
__global__ __launch_bounds__(1024) void ms_bound_kernel(MsBoundArgs p) {
  constexpr int D = KS * 16;
  constexpr int M = D / SD;
  constexpr int RB = D * 2;
  constexpr int CPR = RB / 16;
  constexpr int RPK = 256 / RB;
  constexpr int CBB = 256 * D * 2;          // bytes of the f16 codebook
  __shared__ __attribute__((aligned(16))) char sCB[CBB];
  __shared__ __attribute__((aligned(16))) char sB[MSB_BQ * RB];
  __shared__ __attribute__((aligned(16))) uint32_t sH[MSB_BQ * (MSB_BINS / 2)];
  __shared__ __attribute__((aligned(16))) float sPa[MSB_BQ];      // sb / sigma^2 (NaN: the query takes no bound from this pass)
  __shared__ __attribute__((aligned(16))) float sPb[MSB_BQ];      // |r|^2 sb
  __shared__ float sN2[MSB_BQ], sSb[MSB_BQ], sRmu[MSB_BQ];
  __shared__ uint32_t s_chunk;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, g = lane >> 5;
  const uint32_t item = blockIdx.x;
  if (item >= p.item_start[p.nlist]) return;
  const int4 dsc = p.desc[item];
  const int part = dsc.x, i0 = dsc.y, cnt = dsc.z;
  const uint32_t off = p.part_offsets[part];
  const int np_all = (int)(p.part_offsets[part + 1] - off);
  if (np_all < p.keff) return;      // uniform: fewer rows than k*refine -> no bound from this partition
  // u16 bins: a bin never holds more rows than were counted.  A bound only needs k*refine rows under it, so the first 65,535 rows of a larger
  // partition give a valid (slightly looser) one -- the dot metric's lists are as uneven as the rows' norms, and it has no integer pass to fall back to
  const int np = min(np_all, 65535);
  const int nslots = ((cnt + 31) >> 5) << 5;

  // the codebook: 16 bytes per lane and pass
  for (int b = (int)threadIdx.x * 16; b < CBB; b += 1024 * 16)
    *reinterpret_cast<uint4 *>(&sCB[b]) = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(p.cbh) + b);
  for (int i = threadIdx.x; i < MSB_BQ * (MSB_BINS / 2); i += 1024) sH[i] = 0u;
  if (threadIdx.x == 0) s_chunk = 0u;
  // residuals: wave w stages slots w, w + 16, w + 32, w + 48 -- their loads in flight together; a lane owns two neighbouring elements
  {
    constexpr int SPW = MSB_BQ / 16;
    const int e = 2 * lane;
    uint32_t qi[SPW];
#pragma unroll
    for (int t = 0; t < SPW; ++t) { const int sl = wave + 16 * t; qi[t] = sl < cnt ? p.pair_idx0[i0 + sl] / p.nb : 0u; }
    f2 qv[SPW], cv = {0.0f, 0.0f}, mu = {0.0f, 0.0f};
    if (e < D) {
      if (!p.dot) cv = *reinterpret_cast<const f2 *>(p.centroids + (int64_t)part * D + e);
      mu = *reinterpret_cast<const f2 *>(p.cb_mean + e);
    }
    const float hs = p.dot ? 0.5f * p.sigma : p.sigma;
#pragma unroll
    for (int t = 0; t < SPW; ++t) {
      qv[t] = f2{0.0f, 0.0f};
      if (wave + 16 * t < cnt && e < D) qv[t] = *reinterpret_cast<const f2 *>(p.q + (int64_t)qi[t] * D + e);
    }
#pragma unroll
    for (int t = 0; t < SPW; ++t) {
      const int sl = wave + 16 * t;
      if (sl >= nslots) break;      // uniform
      const int key = (sl / RPK) & (CPR - 1);
      float n2 = 0.0f, vmax = 0.0f, rmu = 0.0f;
      bool bad = false;
      uint32_t packed = 0u;
      if (sl < cnt && e < D) {
        float v0 = qv[t].x - cv.x, v1 = qv[t].y - cv.y;      // v2.rs:316-332, the subtraction of the exact path
        if (p.round_f16 && !p.dot) { v0 = __half2float(__float2half_rn(v0)); v1 = __half2float(__float2half_rn(v1)); }
        n2 = v0 * v0 + v1 * v1;
        vmax = fmaxf(fabsf(v0), fabsf(v1));
        rmu = v0 * mu.x + v1 * mu.y;
        bad = !(fabsf(v0) < INFINITY) || !(fabsf(v1) < INFINITY);
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const h2 hv = {(_Float16)(v0 * hs), (_Float16)(v1 * hs)};
        packed = __builtin_bit_cast(uint32_t, hv);
      }
      if (e < D) *reinterpret_cast<uint32_t *>(&sB[sl * RB + ((((e >> 3) ^ key)) << 4) + ((e & 7) << 1)]) = packed;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        n2 += __shfl_xor(n2, o, 64); rmu += __shfl_xor(rmu, o, 64); vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
      }
      bad = __any(bad);
      if (lane == 0) {
        float a = __uint_as_float(0x7FC00000u), b = 0.0f, sb = 0.0f;
        if (sl < cnt) {
          // L2: sum over m of the mean table entry = the distance of a random code.  dot: the same minus base = (1 - q . mu) - G, G = |q| cmax >=
          // |q . c'| (ms_prep_kernel; the centred codewords average to zero): mean - base = G; the bins count (dist~ - base) sb = (acc / sigma^2 + G) sb
          const float G = p.dot ? sqrtf(n2) * 1.000001f * fminf(p.cmax, p.cmaxp[part]) * 1.000001f : 0.0f;      // (this list's rows only)
          const float mean = p.dot ? G : n2 - 2.0f * rmu + p.cb_mean[D];
          const float off = p.dot ? G : n2;
          sb = MSB_MEAN_BIN / mean;
          const float sig2 = p.sigma * p.sigma;
          const bool ok = !bad && off < INFINITY && vmax * p.sigma < 60000.0f && mean > 0.0f && mean < INFINITY && sb > 0.0f && sb < INFINITY &&
                          off * sb < 1e30f && sb / sig2 > 0.0f && sb / sig2 < INFINITY;
          if (ok) { a = sb / sig2; b = off * sb; }
        }
        sPa[sl] = a; sPb[sl] = b; sN2[sl] = n2; sSb[sl] = sb; sRmu[sl] = rmu;
      }
    }
  }
  for (int sl = nslots + (int)threadIdx.x; sl < MSB_BQ; sl += 1024) { sPa[sl] = __uint_as_float(0x7FC00000u); sPb[sl] = 0.0f; }
  __syncthreads();

  const uint32_t nchunks = ((uint32_t)np + 31u) >> 5;
  const int nblk = nslots >> 5;
  // a chunk's codes and |c^|^2 are requested one chunk ahead (the codeword gather itself is LDS)
  auto take = [&]() -> uint32_t {
    uint32_t c = 0;
    if (lane == 0) c = atomicAdd(&s_chunk, 1u);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
  };
  uint32_t cwn[M / 4];
  float cn2n = INFINITY;
  auto request = [&](uint32_t c) {
    const int row0 = (int)(c * 32u);
    const int rowc = min(row0 + j, np - 1);
    const uint4 *rc4 = reinterpret_cast<const uint4 *>(p.codes + ((int64_t)off + rowc) * M);
#pragma unroll
    for (int w = 0; w < M / 16; ++w) { const uint4 t = rc4[w]; cwn[4 * w] = t.x; cwn[4 * w + 1] = t.y; cwn[4 * w + 2] = t.z; cwn[4 * w + 3] = t.w; }
    const bool live = row0 + j < np && row_allowed(p.allow, off + (uint32_t)rowc);
    cn2n = live ? p.row_cn2[(int64_t)off + rowc] : INFINITY;      // a padded / filtered row lands in no bin
  };
  uint32_t c = take();
  if (c < nchunks) request(c);
  while (c < nchunks) {
    const uint32_t cnext = take();
    uint32_t cw[M / 4];
#pragma unroll
    for (int w = 0; w < M / 4; ++w) cw[w] = cwn[w];
    const float cn2v = cn2n;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int w = 0; w < M / 4; ++w) asm volatile("" : "+v"(cw[w]));
    if (cnext < nchunks) request(cnext);
    ms_h8 rw[KS];
    {
      auto code = [&](int mm) -> uint32_t { return (cw[mm >> 2] >> (8 * (mm & 3))) & 255u; };
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        if constexpr (SD == 8) {
          const uint32_t c0 = code(2 * s), c1 = code(2 * s + 1);
          const int mm = 2 * s + g;
          rw[s] = *reinterpret_cast<const ms_h8 *>(&sCB[((uint32_t)mm * 256u + (g ? c1 : c0)) * 16u]);
        } else {
          static_assert(SD == 4, "sub-dimension 4 / 8");
          const uint32_t c0 = g ? code(4 * s + 2) : code(4 * s), c1 = g ? code(4 * s + 3) : code(4 * s + 1);
          const int mm = 4 * s + 2 * g;
          const ms_h4 lo = *reinterpret_cast<const ms_h4 *>(&sCB[((uint32_t)mm * 256u + c0) * 8u]);
          const ms_h4 hi = *reinterpret_cast<const ms_h4 *>(&sCB[((uint32_t)(mm + 1) * 256u + c1) * 8u]);
          rw[s] = ms_h8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
      }
    }
    for (int jb = 0; jb < nblk; ++jb) {
      ms_h8 qa[KS];
      {
        const int sl = jb * 32 + j;
        const char *br = &sB[sl * RB];
        const int key = (sl / RPK) & (CPR - 1);
#pragma unroll
        for (int s = 0; s < KS; ++s) qa[s] = *reinterpret_cast<const ms_h8 *>(br + (((2 * s + g) ^ key) << 4));
      }
      ms_f16v acc;
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
#pragma unroll
      for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(qa[s], rw[s], acc, 0, 0, 0);
      // lane (j, g) holds row j against the queries (v & 3) + 8 (v >> 2) + 4 g of the tile
      const f4 *pa = reinterpret_cast<const f4 *>(&sPa[jb * 32 + 4 * g]);
      const f4 *pb = reinterpret_cast<const f4 *>(&sPb[jb * 32 + 4 * g]);
#pragma unroll
      for (int vq = 0; vq < 4; ++vq) {
        const f4 a4 = pa[2 * vq], b4 = pb[2 * vq];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float t = __builtin_fmaf(acc[4 * vq + e] + cn2v, a4[e], b4[e]);
          if (t < (float)MSB_BINS) {      // (false for a NaN scale and for padded rows)
            const int bin = max((int)t, 0);
            const int slot = jb * 32 + 8 * vq + 4 * g + e;
            atomicAdd(&sH[slot * (MSB_BINS / 2) + (bin >> 1)], 1u << (16 * (bin & 1)));
          }
        }
      }
    }
    c = cnext;
  }
  __syncthreads();
  // wave w: queries w, w + 16, ..: the first bin where the cumulative count reaches k*refine
  for (int sl = wave; sl < cnt; sl += 16) {
    const float a = sPa[sl];
    if (!(a > 0.0f)) continue;      // uniform: no bound for this query
    const uint4 hw = *reinterpret_cast<const uint4 *>(&sH[sl * (MSB_BINS / 2) + lane * 4]);      // bins 8 lane .. 8 lane + 7
    const uint32_t loc[8] = {hw.x & 0xFFFFu, hw.x >> 16, hw.y & 0xFFFFu, hw.y >> 16, hw.z & 0xFFFFu, hw.z >> 16, hw.w & 0xFFFFu, hw.w >> 16};
    uint32_t tot = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) tot += loc[i];
    uint32_t incl = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    uint32_t run = incl - tot;
    int found = -1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      run += loc[i];
      if (found < 0 && run >= (uint32_t)p.keff) found = lane * 8 + i;
    }
    const uint64_t mask = __ballot(found >= 0);
    if (mask) {
      const int leader = __ffsll((long long)mask) - 1;
      const int bin = __shfl(found, leader, 64);
      if (lane == 0) {
        const float n2 = sN2[sl], sb = sSb[sl];
        const float Ta = (float)(bin + 1) / sb * 1.000001f;      // every counted row has dist~ < (bin + 1) / sb
        const float rn = sqrtf(n2) * 1.000001f, st = sqrtf(Ta) * 1.000001f;
        const float sqd = sqrtf((float)D) + 1.0f;
        if (p.dot) {
          // every counted row has dist~ - base < Ta; its reference distance is <= base + Ta + E (ms_prep_kernel's E with |T| <= |base| + Ta)
          const float G = rn * fminf(p.cmax, p.cmaxp[part]) * 1.000001f, Gf = rn * p.cmax_full * 1.000001f, qmu = sRmu[sl];
          const float tmag = fabsf(1.0f - qmu) + G + Ta;
          const float e_abs = 6.1035156e-5f * sqd * (rn + 2.0f * p.cmax) / p.sigma + (float)D * 3.7252903e-9f / (p.sigma * p.sigma);
          const float E = 1.05f * (9.9609375e-4f * G + 1.5258789e-5f * (G + 1.0f + tmag + fabsf(qmu) + Gf) + 5.9604645e-8f * (float)((M + 2) * (M + 2))) + e_abs;
          const float T = (((1.0f - qmu) - G) + Ta) + (1.1f * E + 1.5258789e-5f * tmag);
          if (fabsf(T) < INFINITY) atomicMin(&p.tglobal[p.pair_idx0[i0 + sl] / p.nb], order_key(T));
        } else {
        const float e_abs = 6.1035156e-5f * sqd * (3.0f * rn + 2.0f * st) / p.sigma + (float)D * 3.7252903e-9f / (p.sigma * p.sigma);
        const float E = 1.05f * (1.9921875e-3f * rn * (rn + st) + 1.2207031e-4f * (n2 + Ta)) + e_abs;      // ms_prep_kernel's E at T = Ta
        const float T = (Ta + 1.1f * E) * 1.0000153f;
        if (T > 0.0f && T < INFINITY) atomicMin(&p.tglobal[p.pair_idx0[i0 + sl] / p.nb], order_key(T));
        }
      }
    }
  }
}

// ---- host --------------------------------------------------------------------------------------------------------------------
static bool ms_shape(const lance_hip_index *ix, int *sd_out, int *ks_out) {
  if (!ix || ix->m == 0 || ix->nbits != 8 || ix->d % ix->m != 0) return false;
  const int d = (int)ix->d, m = (int)ix->m, sd = d / m;
  if (!(m == 16 || m == 32)) return false;
  const bool ok = (d == 128 && (sd == 8 || sd == 4)) || (d == 64 && sd == 4);
  if (!ok) return false;
  if (sd_out) *sd_out = sd;
  if (ks_out) *ks_out = d / 16;
  return true;
}

// The matrix-core scan pays once a partition sees a few tiles' worth of queries: a wave's 32-row chunk costs a gather (codes -> 8
// codewords -> |c^|^2) worth ~4 tiles before its first MFMA.  Measured (gpurun r04v): 390 pairs per partition (C2) 0.188 + 0.047 ms
// against 0.354 + 0.021 for the integer scan; 24 pairs per partition (C4 shape, nlist 4096, 10k x 10) 0.463 + 0.110 against
// 0.324 + 0.023 -- slower.  Per (row, query) cell the model (G + t C) / (1024 t) crosses the integer scan's cost near t = 3 tiles.
bool mscan_batch_shape(const lance_hip_index *ix, uint32_t nq, uint32_t nprobes) {
  static const bool off = getenv("LANCE_HIP_NO_MSCAN") != nullptr;
  static const uint32_t minq = getenv("LANCE_HIP_MSCAN_MINQ") ? (uint32_t)std::max(1, atoi(getenv("LANCE_HIP_MSCAN_MINQ"))) : 96u;
  if (off || !ms_shape(ix, nullptr, nullptr)) return false;
  if (ix->metric != LANCE_HIP_L2 && ix->metric != LANCE_HIP_COSINE && ix->metric != LANCE_HIP_DOT) return false;
  return (uint64_t)nq * nprobes >= (uint64_t)minq * ix->nlist;
}

static bool ms_dot_skew_ok(const lance_hip_index *ix) {      // (see mscan_dot_ready)
  static const double skew = getenv("LANCE_HIP_DOT_FLOW_SKEW") ? atof(getenv("LANCE_HIP_DOT_FLOW_SKEW")) : 1e18;
  return (double)ix->max_part * (double)ix->nlist <= skew * (double)ix->n;
}

// dot metric: the quantised flow is the matrix-core bound pass + scan or nothing (the integer tables need entries >= 0).  What can only be
// known inside the launchers (an all-zero codebook, unaligned query rows) makes them return LH_NOT_TAKEN and the caller keeps the exact pair scan.
bool mscan_dot_ready(const lance_hip_index *ix, uint32_t nq, uint32_t nprobes) {
  static const bool off = getenv("LANCE_HIP_NO_MSBOUND") != nullptr || getenv("LANCE_HIP_NO_DOT_FLOW") != nullptr;
  if (off || !mscan_batch_shape(ix, nq, nprobes) || !ix->cb_mean) return false;
  // Lists as uneven as the rows' norms (dot over unnormalised rows with components of one sign: at the C2 shape the largest of 256 lists held 82,424
  // of the 10^6 rows and was every query's nearest): with the bound of the NEAREST list alone a fifth of the segments overflowed into exact rescans of
  // whole lists and the flow measured slower than the exact pair scan (3.94 against 3.46 ms per 10,000-query batch, gpurun r06zu); with the bound pass
  // over three lists it is 1.88 (gpurun r06zzg).  LANCE_HIP_DOT_FLOW_SKEW keeps a guard for A/B runs: the largest list / mean list ratio up to which
  // the flow is taken (default: no limit).
  if (!ms_dot_skew_ok(ix)) return false;
  lance_hip_index *mix = const_cast<lance_hip_index *>(ix);      // (the constants are published under this lock by the first search of any context)
  std::lock_guard<std::mutex> lk(mix->lazy_mu);
  return !mix->ms || mix->ms->usable;
}

bool mscan_supported(const lance_hip_index *ix, uint32_t nq, uint32_t nprobes) {
  return mscan_batch_shape(ix, nq, nprobes) && qscan_supported(ix, nq, nprobes);
}

static int mscan_prepare(lance_hip_ctx *ctx, lance_hip_index *ix) {
  std::lock_guard<std::mutex> lk(ix->lazy_mu);   // the first search of any context builds the constants, the others wait for it
  if (ix->ms) return LANCE_HIP_OK;
  const int d = (int)ix->d, m = (int)ix->m, sd = d / m;
  const int64_t nwords = (int64_t)m * 256;
  std::vector<float> cb((size_t)nwords * sd);
  LH_CHECK_HIP(hipMemcpyAsync(cb.data(), ix->codebook, cb.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
  LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  const bool dot = ix->metric == LANCE_HIP_DOT;
  std::vector<float> mu;      // dot: the plane holds the codewords minus their sub-quantiser's mean (ms_prep_kernel) -- the device's own cb_mean, so that
  if (dot && ix->cb_mean) {   // the centring here and the q . mu of the pre-pass use the same numbers
    mu.resize((size_t)d);
    LH_CHECK_HIP(hipMemcpyAsync(mu.data(), ix->cb_mean, (size_t)d * 4, hipMemcpyDeviceToHost, ctx->stream));
    LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  }
  auto *mc = new lance_hip_index::MsConst();      // (no early return between here and its publication / drop())
  if (dot) {
    if (!ix->cb_mean) { mc->usable = false; ix->ms = mc; return LANCE_HIP_OK; }
    // |reconstruction of any row|^2 <= sum over the sub-quantisers of their longest codeword's |c|^2 (only the reference's rounding terms use it)
    double tot = 0.0;
    for (int mm = 0; mm < m; ++mm) {
      double best = 0.0;
      for (int c = 0; c < 256; ++c) {
        double sq = 0.0;
        for (int u = 0; u < sd; ++u) { const double v = cb[((size_t)mm * 256 + c) * sd + u]; sq += v * v; }
        best = std::max(best, sq);
      }
      tot += best;
    }
    mc->cmax_full = (float)(std::sqrt(tot) * 1.00001);
  }
  float cbmax = 0.0f;
  for (size_t i = 0; i < cb.size(); ++i) cbmax = std::max(cbmax, std::fabs(dot ? cb[i] - mu[(i / ((size_t)256 * sd)) * sd + i % sd] : cb[i]));
  if (!(cbmax > 0.0f) || !std::isfinite(cbmax) || ix->n == 0) {      // an all-zero codebook: nothing to scale by -- the integer scan serves this index
    mc->usable = false;
    ix->ms = mc;
    return LANCE_HIP_OK;
  }
  int e = 0;
  (void)std::frexp(cbmax, &e);      // cbmax in [2^(e-1), 2^e)
  mc->sigma = std::ldexp(1.0f, 13 - e);      // 2 sigma cbmax in [2^13, 2^14)
  bool ok = hipMalloc(reinterpret_cast<void **>(&mc->cbh), (size_t)nwords * sd * 2) == hipSuccess;
  ok = ok && hipMalloc(reinterpret_cast<void **>(&mc->cbn2), (size_t)nwords * 4) == hipSuccess;
  ok = ok && hipMalloc(reinterpret_cast<void **>(&mc->row_cn2), (size_t)ix->n * 4) == hipSuccess;
  if (dot) ok = ok && hipMalloc(reinterpret_cast<void **>(&mc->cmaxp), (size_t)std::max<uint32_t>(ix->nlist, 1u) * 4) == hipSuccess;
  auto drop = [&]() { (void)hipFree(mc->cbh); (void)hipFree(mc->cbn2); (void)hipFree(mc->row_cn2); (void)hipFree(mc->cmaxp); delete mc; };
  if (!ok) { drop(); set_error("matrix-core scan: out of device memory for the index constants"); return LANCE_HIP_ENOMEM; }
  hipLaunchKernelGGL(ms_codebook_kernel, dim3((unsigned)cdiv((uint64_t)nwords, 256)), dim3(256), 0, ctx->stream, ix->codebook, nwords, sd,
                     -2.0f * mc->sigma, reinterpret_cast<_Float16 *>(mc->cbh), mc->cbn2, dot ? ix->cb_mean : nullptr);
  hipLaunchKernelGGL(ms_row_norm_kernel, dim3((unsigned)cdiv(ix->n, 256)), dim3(256), 0, ctx->stream, ix->codes, (int64_t)ix->n, m, mc->cbn2,
                     mc->sigma * mc->sigma, mc->row_cn2);
  uint32_t rowmax_bits = 0;
  if (dot) {
    // the largest |centred reconstruction| of any stored row (the dot metric's G), then the row term itself is ZERO: the same kernels test
    // -(q . c')~ against the limit
    uint32_t *mx = reinterpret_cast<uint32_t *>(mc->cbn2);      // (the per-codeword norms are not needed after the row norms: word 0 is reused)
    (void)lh::memset_async(mx, 0, 4, ctx->stream);
    hipLaunchKernelGGL(ms_max_kernel, dim3(256), dim3(256), 0, ctx->stream, mc->row_cn2, (int64_t)ix->n, mx);
    (void)hipMemcpyAsync(&rowmax_bits, mx, 4, hipMemcpyDeviceToHost, ctx->stream);
    hipLaunchKernelGGL(ms_part_max_kernel, dim3(ix->nlist), dim3(256), 0, ctx->stream, mc->row_cn2, ix->part_offsets, 1.0f / mc->sigma, mc->cmaxp);
    (void)lh::memset_async(mc->row_cn2, 0, (size_t)ix->n * 4, ctx->stream);
  }
  for (uint32_t pid = 0; pid < ix->nlist; ++pid) {
    const uint32_t rs = ms_nrs((uint32_t)(ix->part_offsets_h[pid + 1] - ix->part_offsets_h[pid]), (uint32_t)ms_rows_per_slice(), (uint32_t)ms_rows_per_slice2());
    mc->sum_rs += rs; mc->max_rs = std::max(mc->max_rs, rs);
  }
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {   // other contexts (streams) search the same index
    drop();
    set_error("matrix-core scan: building the index constants failed");
    return LANCE_HIP_ERUNTIME;
  }
  if (dot) {
    float rm; memcpy(&rm, &rowmax_bits, 4);      // sigma^2 max |c'|^2 (the stream was synchronised above)
    mc->cmax = (float)(std::sqrt((double)rm) / (double)mc->sigma * 1.0001);
    if (!(mc->cmax > 0.0f) || !std::isfinite(mc->cmax)) mc->cmax = mc->cmax_full;      // (every row on the mean: any positive bound will do)
  }
  mc->usable = true;
  ix->ms = mc;      // published complete; lance_hip_index's destructor frees it
  return LANCE_HIP_OK;
}

// -1: this index cannot take the matrix-core scan (the caller falls back to the integer scan); otherwise a status code.
// Replaces qscan_launch: same segment outputs (seg_cnt / seg_pos / qovf / ovf) with the survivors' accumulator values (seg_val) and the
// per-pair scale that turns them into integer sums (seg_scale), plus qslack for the merge kernel's cut.
int mscan_launch(lance_hip_ctx *ctx, const lance_hip_index *ix_c, const float *qs, uint32_t nq, uint32_t nprobes, const uint32_t *probes,
                 const uint32_t *pair_starts, const uint32_t *pair_idx, const uint32_t *tbound, uint32_t *seg_cnt, uint32_t *seg_pos,
                 uint32_t *qovf, const uint32_t *allow, uint32_t **qslack_out, float **seg_val_out, float **seg_scale_out) {
  lance_hip_index *ix = const_cast<lance_hip_index *>(ix_c);   // the constants are a cache attached to the index
  int sd = 0, ks = 0;
  if (!ms_shape(ix, &sd, &ks)) return LH_NOT_TAKEN;
  LH_TRY(mscan_prepare(ctx, ix));
  if (!ix->ms->usable) return LH_NOT_TAKEN;
  const int d = (int)ix->d, nlist = (int)ix->nlist;
  const size_t npairs = (size_t)nq * nprobes;
  // slices: at most sum over the partitions of (row slices) x (pair blocks), pair blocks <= pairs / MS3_PB + 1 per partition
  const uint64_t cap = (uint64_t)ix->ms->sum_rs + (uint64_t)ix->ms->max_rs * cdiv(npairs, MS3_PB) + 8;
  _Float16 *rh = reinterpret_cast<_Float16 *>(ctx->scratch("ms.rh", (npairs + 32) * (size_t)d * 2));
  f4 *prm = reinterpret_cast<f4 *>(ctx->scratch("ms.prm", (npairs + 32) * 16));
  f2 *prm2 = reinterpret_cast<f2 *>(ctx->scratch("ms.prm2", npairs * 8));
  uint32_t *qslack = ctx->scratch_t<uint32_t>("ms.qslack", nq);
  uint32_t *slice_start = ctx->scratch_t<uint32_t>("ms.slice_start", (size_t)nlist + 2 + 32);      // [nlist + 1], the work counter, 32 class cursors
  MsSlice *slices = reinterpret_cast<MsSlice *>(ctx->scratch("ms.slices", cap * sizeof(MsSlice)));
  uint32_t *order = ctx->scratch_t<uint32_t>("ms.order", cap);
  float *seg_val = ctx->scratch_t<float>("ms.seg_val", npairs * Q_CAP * 2);      // uint2 records {position, value}
  uint32_t *ovf = ctx->scratch_t<uint32_t>("q.ovf", npairs + 1);                // the merge launcher asks for the same slot
  if (!rh || !prm || !prm2 || !qslack || !slice_start || !slices || !order || !seg_val || !ovf) return LANCE_HIP_ENOMEM;
  uint32_t *slice_ctr = slice_start + nlist + 1, *cls_cursor = slice_ctr + 1;
  const uint32_t nan_slot = (uint32_t)npairs + 1u;      // inside prm's 32 records of padding
  {
    ScopedTimer t(ctx, "q_residual");
    LH_CHECK_HIP(lh::memset_multi(ctx->stream, {{seg_cnt, 0, npairs * 4}, {qovf, 0, (size_t)nq * 4}, {qslack, 0, (size_t)nq * 4}, {ovf, 0, 4}}));
    MsPrepArgs pa;
    pa.q = qs; pa.centroids = ix->centroids; pa.pair_idx = pair_idx; pa.pair_starts = pair_starts; pa.probes = probes; pa.tbound = tbound;
    pa.d = d; pa.nprobes = (int)nprobes; pa.nlist = nlist; pa.round_f16 = ix->dtype == LANCE_HIP_F16 ? 1 : 0;
    pa.sigma = ix->ms->sigma; pa.rh = rh; pa.prm = prm; pa.qslack = qslack; pa.seg_cnt = seg_cnt; pa.qovf = qovf; pa.ovf = ovf;
    pa.nan_slot = nan_slot; pa.prm2 = prm2;
    pa.dot = ix->metric == LANCE_HIP_DOT ? 1 : 0; pa.cmax = ix->ms->cmax; pa.cmax_full = ix->ms->cmax_full; pa.cmaxp = ix->ms->cmaxp; pa.mu = ix->cb_mean; pa.m = (int)ix->m;
    hipLaunchKernelGGL(ms_prep_kernel, dim3((unsigned)cdiv(npairs, 4 * MS_PPW)), dim3(256), 0, ctx->stream, pa);
    const uint32_t rs_rows = (uint32_t)ms_rows_per_slice(), rs2_rows = (uint32_t)ms_rows_per_slice2();
    static const bool no_order = getenv("LANCE_HIP_MS_NOORDER") != nullptr;      // A/B: one work class = slices in (roughly) index order
    const uint32_t cls_rows = no_order ? 0x40000000u : rs_rows;
    hipLaunchKernelGGL(ms_slice_table_kernel, dim3(1), dim3(256), 0, ctx->stream, pair_starts, ix->part_offsets, nlist, rs_rows, rs2_rows, cls_rows, slice_start,
                       slice_ctr, cls_cursor);
    hipLaunchKernelGGL(ms_slice_desc_kernel, dim3((unsigned)cdiv((uint64_t)nlist, 4)), dim3(256), 0, ctx->stream, slice_start, pair_starts, ix->part_offsets,
                       nlist, rs_rows, rs2_rows, cls_rows, slices, cls_cursor, order);
  }
  ScopedTimer t(ctx, "ivfpq_scan_c1");
  ScopedTimer tm(ctx, "ivfpq_mscan");      // the same launch under its own name: tests assert the matrix-core scan was the one taken
  MscanArgs a;
  a.order = order; a.slices = slices; a.slice_start = slice_start; a.slice_ctr = slice_ctr; a.codes = ix->codes;
  a.cbh = reinterpret_cast<const _Float16 *>(ix->ms->cbh); a.row_cn2 = ix->ms->row_cn2; a.rh = rh; a.prm = prm; a.prm2 = prm2;
  a.nan_slot = nan_slot; a.nlist = nlist; a.nprobes = (int)nprobes;
  a.seg_cnt = seg_cnt; a.seg_pos = seg_pos; a.seg_pv = reinterpret_cast<uint2 *>(seg_val); a.ovf = ovf; a.allow = allow;
#ifdef LH_TIMING_EXPERIMENTS
  static const int dbg = [] {
    const int v = getenv("LANCE_HIP_MS_DBG") ? atoi(getenv("LANCE_HIP_MS_DBG")) : 0;
    if (v) fprintf(stderr, "lance_hip: LANCE_HIP_MS_DBG=%d -- timing experiment, search RESULTS ARE WRONG\n", v);
    return v;
  }();
  a.dbg = dbg;
#endif
  // persistent: one workgroup per CU (151 KiB of LDS each).  LANCE_HIP_MS_GRID: fewer workgroups leave CUs to the latency-bound kernels of
  // other engine contexts (merge, refine, bound pass) while this one runs -- an A/B knob
  static const int grid_env = getenv("LANCE_HIP_MS_GRID") ? atoi(getenv("LANCE_HIP_MS_GRID")) : 0;
  const unsigned grid = (unsigned)std::min<uint64_t>((uint64_t)(grid_env > 0 ? std::min(grid_env, ctx->num_cus) : ctx->num_cus), cap);
  static const bool prof = getenv("LANCE_HIP_MS_PROF") != nullptr;      // s_memtime phase stamps (d = 128, M = 16), printed per launch; plain path only
  if (prof && sd == 8 && ks == 8 && !ctx->capturing) {
    a.prof = ctx->scratch_t<unsigned long long>("ms.prof", 8);
    if (!a.prof) return LANCE_HIP_ENOMEM;
    LH_CHECK_HIP(lh::memset_async(a.prof, 0, 64, ctx->stream));
    a.prof_slices = ctx->scratch_t<unsigned long long>("ms.prof_slices", cap);
    if (!a.prof_slices) return LANCE_HIP_ENOMEM;
    LH_CHECK_HIP(lh::memset_async(a.prof_slices, 0, cap * 8, ctx->stream));
    hipLaunchKernelGGL((ivfpq_mscan_kernel<8, 8, true>), dim3(grid), dim3(1024), 0, ctx->stream, a);
    unsigned long long h[8];
    LH_CHECK_HIP(hipMemcpyAsync(h, a.prof, 64, hipMemcpyDeviceToHost, ctx->stream));
    LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    {   // per-slice durations in taken order, with the slice's shape: which slices make the tail
      std::vector<unsigned long long> st(cap);
      std::vector<MsSlice> sl(cap);
      std::vector<uint32_t> ord(cap), ss((size_t)nlist + 1);
      (void)hipMemcpy(st.data(), a.prof_slices, cap * 8, hipMemcpyDeviceToHost);
      (void)hipMemcpy(sl.data(), slices, cap * sizeof(MsSlice), hipMemcpyDeviceToHost);
      (void)hipMemcpy(ord.data(), order, cap * 4, hipMemcpyDeviceToHost);
      (void)hipMemcpy(ss.data(), slice_start, ((size_t)nlist + 1) * 4, hipMemcpyDeviceToHost);
      const uint32_t ns = ss[nlist];
      std::vector<uint32_t> idx(ns);
      for (uint32_t i = 0; i < ns; ++i) idx[i] = i;
      std::sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { return st[x] > st[y]; });
      double tot = 0, totw = 0;
      for (uint32_t i = 0; i < ns; ++i) { tot += (double)st[i]; totw += (double)sl[ord[i]].row_count * sl[ord[i]].qp; }
      fprintf(stderr, "[ms prof] %u slices, mean %.0f ticks, mean ticks per 1000 cells %.2f; slowest (taken#, ticks, rows, pairs, ticks per 1000 cells):", ns, tot / ns, tot / totw * 1e3);
      for (uint32_t t = 0; t < std::min<uint32_t>(ns, 12); ++t) {
        const uint32_t i = idx[t];
        fprintf(stderr, " (%u, %llu, %u, %u, %.2f)", i, st[i], sl[ord[i]].row_count, sl[ord[i]].qp, (double)st[i] / ((double)sl[ord[i]].row_count * sl[ord[i]].qp) * 1e3);
      }
      // and the last 12 taken
      fprintf(stderr, " | last taken:");
      for (uint32_t i = ns > 12 ? ns - 12 : 0; i < ns; ++i)
        fprintf(stderr, " (%u, %llu, %u, %u)", i, st[i], sl[ord[i]].row_count, sl[ord[i]].qp);
      fprintf(stderr, "\n");
    }
    if (h[5])
      fprintf(stderr, "[ms prof] waves=%llu chunks/wave %.2f | s_memtime ticks per wave: stage+barriers %.0f | gather %.0f | tiles %.0f | flush %.0f | life %.0f (longest %llu)\n",
              h[5], (double)h[6] / h[5], (double)h[0] / h[5], (double)h[1] / h[5], (double)h[2] / h[5], (double)h[3] / h[5], (double)h[4] / h[5], h[7]);
  } else if (sd == 8 && ks == 8) hipLaunchKernelGGL((ivfpq_mscan_kernel<8, 8>), dim3(grid), dim3(1024), 0, ctx->stream, a);
  else if (sd == 4 && ks == 8) hipLaunchKernelGGL((ivfpq_mscan_kernel<4, 8>), dim3(grid), dim3(1024), 0, ctx->stream, a);
  else if (sd == 4 && ks == 4) hipLaunchKernelGGL((ivfpq_mscan_kernel<4, 4>), dim3(grid), dim3(1024), 0, ctx->stream, a);
  else { set_error("matrix-core scan: unsupported shape (d=%d, sd=%d)", d, sd); return LANCE_HIP_EINVAL; }
  LH_CHECK_HIP(hipGetLastError());
  *qslack_out = qslack; *seg_val_out = seg_val; *seg_scale_out = reinterpret_cast<float *>(prm2);
  return LANCE_HIP_OK;
}

// The bound pass of a batch the matrix-core scan serves (search_pm.hip).  pair_starts0 / pair_idx0: the nq (query, nearest partition) pairs
// grouped by partition.  -1: not taken (the caller runs the integer pass); otherwise a status code.
int msbound_launch(lance_hip_ctx *ctx, const lance_hip_index *ix_c, const float *qs, uint32_t nq, uint32_t keff, const uint32_t *pair_starts0,
                   const uint32_t *pair_idx0, uint32_t *item_start, int4 *desc, uint32_t max_items, uint32_t *tglobal, const uint32_t *allow,
                   uint32_t nb) {
  static const bool off = getenv("LANCE_HIP_NO_MSBOUND") != nullptr;      // A/B switch: the integer histogram pass (search_q.hip)
  lance_hip_index *ix = const_cast<lance_hip_index *>(ix_c);
  int sd = 0, ks = 0;
  const bool dot = ix->metric == LANCE_HIP_DOT;
  // u16 bins: a bin never holds more rows than the partition has (L2: larger lists keep the integer pass; dot: the kernel counts the first 65,535 rows)
  if (off || !ms_shape(ix, &sd, &ks) || !ix->cb_mean || (!dot && ix->max_part >= 65536u)) return LH_NOT_TAKEN;
  if (((reinterpret_cast<uintptr_t>(qs) | reinterpret_cast<uintptr_t>(ix->centroids)) & 7) != 0) return LH_NOT_TAKEN;
  LH_TRY(mscan_prepare(ctx, ix));
  if (!ix->ms->usable) return LH_NOT_TAKEN;
  const int nlist = (int)ix->nlist;
  LH_TRY(qscan_items(ctx, pair_starts0, nlist, MSB_BQ, item_start, desc, max_items));
  if (!((sd == 8 && ks == 8) || (sd == 4 && (ks == 8 || ks == 4)))) return LH_NOT_TAKEN;      // before the stage is counted
  ScopedTimer t(ctx, "ivfpq_msbound");      // the launch under its own name: tests assert which bound pass ran
  MsBoundArgs a;
  a.q = qs; a.centroids = ix->centroids; a.cb_mean = ix->cb_mean; a.pair_idx0 = pair_idx0; a.item_start = item_start; a.desc = desc;
  a.part_offsets = ix->part_offsets; a.codes = ix->codes; a.cbh = reinterpret_cast<const _Float16 *>(ix->ms->cbh); a.row_cn2 = ix->ms->row_cn2;
  a.d = (int)ix->d; a.nlist = nlist; a.keff = (int)keff; a.round_f16 = ix->dtype == LANCE_HIP_F16 ? 1 : 0; a.sigma = ix->ms->sigma;
  a.tglobal = tglobal; a.allow = allow;
  a.dot = ix->metric == LANCE_HIP_DOT ? 1 : 0; a.cmax = ix->ms->cmax; a.cmax_full = ix->ms->cmax_full; a.cmaxp = ix->ms->cmaxp;
  a.nb = std::max<uint32_t>(nb, 1u);
  const unsigned grid = (unsigned)std::min<uint64_t>(max_items, (uint64_t)nq * a.nb / MSB_BQ + (uint64_t)nlist + 1);      // sum over partitions of ceil(queries / MSB_BQ)
  if (sd == 8 && ks == 8) hipLaunchKernelGGL((ms_bound_kernel<8, 8>), dim3(grid), dim3(1024), 0, ctx->stream, a);
  else if (sd == 4 && ks == 8) hipLaunchKernelGGL((ms_bound_kernel<4, 8>), dim3(grid), dim3(1024), 0, ctx->stream, a);
  else hipLaunchKernelGGL((ms_bound_kernel<4, 4>), dim3(grid), dim3(1024), 0, ctx->stream, a);
  LH_CHECK_HIP(hipGetLastError());
  return LANCE_HIP_OK;
}

int mscan_prewarm(lance_hip_ctx *ctx, const lance_hip_index *ix_c) {
  lance_hip_index *ix = const_cast<lance_hip_index *>(ix_c);
  static const bool off = getenv("LANCE_HIP_NO_MSCAN") != nullptr;
  if (off || !ms_shape(ix, nullptr, nullptr) || !ix->model_finite) return LANCE_HIP_OK;
  if (ix->metric == LANCE_HIP_DOT && (!ms_dot_skew_ok(ix) || getenv("LANCE_HIP_NO_DOT_FLOW"))) return LANCE_HIP_OK;      // the index will not take the flow: no constants
  return mscan_prepare(ctx, ix);
}

void mscan_cut_params(int *cut_shift, uint32_t *cut_slack) { *cut_shift = MS_CUT_SHIFT; *cut_slack = 2u; }

}  // namespace lh
