// q_common.cuh -- constants, argument blocks and the packed table-entry arithmetic shared by the quantised filter scan kernels
// (search_q.hip: M = 16 / 32, whole table in LDS, packed u16 sums; search_qt.hip: M = 48 / 64 / 96, table tiled over the
// sub-quantisers, 32-bit sums).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "exact.cuh"
#include "kernels.h"

#pragma clang fp contract(off)

namespace lh {

constexpr int Q_BS = 512;      // lanes per scan workgroup
constexpr int Q_G = 4;         // queries per work item (4 x u16 = one ds_read_b64)
#ifndef LH_Q_WAVES
#define LH_Q_WAVES 8
#endif
constexpr int Q_WAVES = LH_Q_WAVES;   // launch-bounds hint for M = 16, sub-dimension <= 8: waves per SIMD (8 = FOUR 512-lane workgroups per CU, <= 64 VGPRs: fits without spills once the table build is not unrolled -- main pass 0.52 -> 0.42 ms; with the build unrolled by 2 it spilled and lost 25 %)
constexpr int Q_CAP = QSCAN_SEG_CAP;   // survivors kept per (query, probe); more -> that partition is rescanned exactly for the query
#ifndef LH_Q_LUT_UNROLL
#define LH_Q_LUT_UNROLL 1
#endif
#ifndef LH_Q_MPF
#define LH_Q_MPF 4
#endif
constexpr int Q_MPF = LH_Q_MPF;        // merge kernel: codebook entries fetched together per candidate row (registers vs round trips)

struct QscanArgs {
  const f4 *rq;                 // [items][d] x 4 queries: negated residuals (q_residual_kernel)
  const uint32_t *pair_idx;     // grouped pair indices (pair = q * nprobes + rank)
  const uint32_t *item_start;   // [nlist+1]: items of class A
  const int4 *desc;
  const float *centroids, *codebook;
  const uint32_t *part_offsets;
  const uint8_t *codes;
  int d, nprobes, nlist, round_f16;
  const uint32_t *tbound;       // [nq] bound key per query (class A: 0 < T < inf)
  uint32_t *seg_cnt;            // [nq * nprobes] survivors of (query, probe) -- zeroed before the launch
  uint32_t *seg_pos;            // [nq * nprobes][Q_CAP] storage positions
  uint16_t *seg_sum;            // [nq * nprobes][Q_CAP] the survivors' integer sums (<= LIM < 65536): the merge kernel derives a tighter
                                // cut from them before it re-evaluates anything exactly
  uint32_t *qovf;               // [nq] set when a segment of the query overflowed -- zeroed before the launch
  uint32_t *ovf;                // [1 + nq * nprobes] count (zeroed before the launch) + the overflowed segments: the rescan kernel's work list
  const uint32_t *allow;        // prefilter bitmap over storage positions or NULL
  unsigned long long *prof = nullptr;   // -DLH_QT_PROF builds only (tiled kernel): [0] build clocks [1] scan [2] emit [3] items
};


// ---- integer table build, shared by the filter scan and the bound pass ------------------------------------------------------
// rq4[dim] = the four queries' NEGATED residual components of that dimension, so one packed add + one packed FMA advance two
// queries by one dimension and the accumulators come out as {L_q0, L_q1}, {L_q2, L_q3}: no horizontal adds, and the
// quantisation is packed too: z = L * (s / 65535) (v_pk_mul_f32), clamped to [0, CAPE / 65535] (v_med3_f32: a NaN becomes 0,
// the row then survives the filter and the exact pass decides), v_cvt_pknorm_u16_f32 turns two of them into the two u16
// halves of a table word.  Whatever rounding the conversion uses, |e - L * s| <= 1 for unsaturated entries; the users'
// limits carry M units for it (floor would need none: 0.4 % of the range).
// RP: `const f4 *` (search_q.hip: the compiler proves the loads uniform and unclobbered and selects s_load by itself) or the
// same pointer cast to the CONSTANT address space (search_qt.hip: inside loops with barriers that proof fails, the cast
// states it -- the residuals were written by an earlier kernel and nobody writes them during this one).
typedef const __attribute__((address_space(4))) f4 *cf4_ptr;
template <int SD, typename RP>
__device__ __forceinline__ void q_entry_acc(RP rq4m, const float *__restrict__ cbp, f2 &acc01, f2 &acc23) {
  constexpr int QV = SD / 4;
  f4 cb[QV];
#pragma unroll
  for (int u = 0; u < QV; ++u) cb[u] = reinterpret_cast<const f4 *>(cbp)[u];
  acc01 = f2{0.0f, 0.0f};
  acc23 = f2{0.0f, 0.0f};
#pragma unroll
  for (int u = 0; u < SD; ++u) {
    const f4 r4 = rq4m[u];
    const float cv = cb[u >> 2][u & 3];
    const f2 cc = {cv, cv};
    const f2 d01 = f2{r4.x, r4.y} + cc;
    const f2 d23 = f2{r4.z, r4.w} + cc;
    acc01 = __builtin_elementwise_fma(d01, d01, acc01);
    acc23 = __builtin_elementwise_fma(d23, d23, acc23);
  }
}

template <uint32_t CAPE>
__device__ __forceinline__ uint2 q_entry_quantise(f2 acc01, f2 acc23, f2 s01, f2 s23) {
  constexpr float CAPZ = (float)CAPE / 65535.0f;
  const f2 z01 = acc01 * s01, z23 = acc23 * s23;   // s = scale / 65535
  typedef unsigned short us2 __attribute__((ext_vector_type(2)));
  const us2 e01 = __builtin_amdgcn_cvt_pknorm_u16(__builtin_amdgcn_fmed3f(z01.x, 0.0f, CAPZ), __builtin_amdgcn_fmed3f(z01.y, 0.0f, CAPZ));
  const us2 e23 = __builtin_amdgcn_cvt_pknorm_u16(__builtin_amdgcn_fmed3f(z23.x, 0.0f, CAPZ), __builtin_amdgcn_fmed3f(z23.y, 0.0f, CAPZ));
  return make_uint2(__builtin_bit_cast(uint32_t, e01), __builtin_bit_cast(uint32_t, e23));
}

// Bound pass scale: the sum over m of the MEAN table entry (the expected distance of a random code) without building the table:
//   sum_m mean_c |r_m - c|^2 = |r|^2 - 2 r . mu + nu,   mu = the mean codeword of every sub-quantiser (d values), nu = sum_m mean_c |c|^2
// -- constants of the index (lance_hip_index::cb_mean).  rq holds the NEGATED residuals: |r|^2 + 2 (-r) . mu.  Any scale is sound
// (it only sets how tight T comes out), so the f32 rounding of this form against the summed table is immaterial.
// Leaves the four queries' sums in sums[0..3] (LDS, zeroed by the caller before a barrier); the caller adds cb_mean[d].
template <int BS>
__device__ __forceinline__ void q_mean_entry_sums(const f4 *__restrict__ rq_item, const float *__restrict__ cb_mean, int d, float *sums) {
  f4 t = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int e = threadIdx.x; e < d; e += BS) {
    const f4 r4 = rq_item[e];
    const float mu2 = 2.0f * cb_mean[e];
    t += r4 * (r4 + f4{mu2, mu2, mu2, mu2});
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float v = t[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0 && (int)(threadIdx.x & ~63u) < d) atomicAdd(&sums[j], v);
  }
}

constexpr int QB_BINS = 512;      // histogram bins per query
constexpr int QB_SHIFT = 3;       // bin width 8: sums 0 .. 4095
struct QboundArgs {
  const f4 *rq;                 // [items][d] x 4 queries: negated residuals (q_residual_kernel)
  const uint32_t *pair_idx;     // nearest-partition pairs grouped by partition: entries are query indices
  const uint32_t *item_start;   // [nlist+1], groups of 4
  const int4 *desc;
  const float *centroids, *codebook;
  const float *cb_mean;         // [d] mean codeword per sub-quantiser dimension, [d] = sum over m of the mean |c|^2 (lance_hip_index::cb_mean)
  const uint32_t *part_offsets;
  const uint8_t *codes;
  int d, nlist, keff, round_f16;
  uint32_t *tglobal;            // [nq] bound key (atomicMin)
  const uint32_t *allow;
};


// search_qt.hip
int qscan_pt_launch(lance_hip_ctx *ctx, const lance_hip_index *ix, const QscanArgs &a, const float *qs, uint32_t nq, const uint32_t *probes,
                    unsigned grid);
bool qscan_tiled_launch(lance_hip_ctx *ctx, const QscanArgs &a, int m, int sd, unsigned grid);
bool qbound_tiled_launch(lance_hip_ctx *ctx, const QboundArgs &a, int m, int sd, unsigned grid);

// floor() variant of q_entry_quantise over the full u16 range: e = rint(L * s * 65535 - 0.5) <= L * s * 65535 whatever the
// conversion's rounding mode, so a row whose distance is <= T can never exceed the limit because of the quantisation (the
// tiled kernels' limits carry no per-entry unit).  s = scale / 65535 as above; the -0.5 rides in the FMA.
__device__ __forceinline__ uint2 q_entry_quantise_floor(f2 acc01, f2 acc23, f2 s01, f2 s23) {
  constexpr float HALF = 0.5f / 65535.0f;
  const f2 nh = {-HALF, -HALF};
  const f2 z01 = __builtin_elementwise_fma(acc01, s01, nh), z23 = __builtin_elementwise_fma(acc23, s23, nh);
  typedef unsigned short us2 __attribute__((ext_vector_type(2)));
  const us2 e01 = __builtin_amdgcn_cvt_pknorm_u16(__builtin_amdgcn_fmed3f(z01.x, 0.0f, 1.0f), __builtin_amdgcn_fmed3f(z01.y, 0.0f, 1.0f));
  const us2 e23 = __builtin_amdgcn_cvt_pknorm_u16(__builtin_amdgcn_fmed3f(z23.x, 0.0f, 1.0f), __builtin_amdgcn_fmed3f(z23.y, 0.0f, 1.0f));
  return make_uint2(__builtin_bit_cast(uint32_t, e01), __builtin_bit_cast(uint32_t, e23));
}

}  // namespace lh
