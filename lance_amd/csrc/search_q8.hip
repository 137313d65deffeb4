// search_q8.hip -- the filter scan with EIGHT queries per LDS gather (M = 16; round 3, behind LANCE_HIP_Q8=1 until it has met
// hardware -- see DESIGN.md section 8).
//
// Why: ivfpq_qscan_kernel (search_q.hip) runs at 0.77 of the measured random-gather ceiling of the LDS with 4 queries per
// ds_read_b64; the next factor is in the number of queries a gather serves.  Here a table word holds eight 8-bit entries:
//   * e[m][c][j] = min(floor-like(L_j[m][c] * SE8 / T_j), 63) for the item's queries j = 0..7 -- bytes 0..3 in .x, 4..7 in .y;
//   * four sub-quantisers are summed inside the bytes with plain 32-bit adds (4 x 63 = 252 cannot carry), then the two
//     registers are widened into four u16-pair registers (and / shift-and + add): 13 VALU per (query, row) against 18 in the
//     4-query kernel, and HALF the gathers;
//   * a row whose reference ADC distance is <= T_j has sum_j <= SE8 (floor-like entries only lower a sum, saturation too; the
//     f32 rounding terms are < 0.01 units at this scale): limit SE8 + 1.
// What it costs: SE8 = 378 instead of 3968 levels for T -> ~20 % more survivors (scripts/sim/q8_selectivity.py: x 1.21 of the rows
// with distance <= T, the u16 table x 1.03), and an entry can saturate BELOW the limit, so a survivor's sum is only a LOWER
// bound of its distance: the merge kernel's integer-sum cut is replaced by a two-phase form that takes its upper bound from
// exact distances (ivfpq_qmerge_kernel, cut_mode 1).  Everything after the filter is unchanged: survivors are re-evaluated in
// the reference's arithmetic, ids and distances stay bit-equal to the oracle.
// Reference behaviour preserved: pq/distance.rs:109-144, pq/storage.rs:921-960, flat/index.rs:94-126, v2.rs:316-332.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "common.h"
#include "exact.cuh"
#include "index.h"
#include "kernels.h"
#include "search_common.cuh"
#include "q_common.cuh"

#pragma clang fp contract(off)

namespace lh {

#ifndef LH_Q8_WAVES
#define LH_Q8_WAVES 6          // 44 KiB of LDS per workgroup: three 512-lane workgroups per CU -> <= 80 VGPRs
#endif

// rq8[item][dim] = the item's eight queries' NEGATED residual components: two float4 (queries 0..3, 4..7)
__global__ __launch_bounds__(256) void q_residual8_kernel(const float *__restrict__ q, const uint32_t *__restrict__ pair_idx,
                                                          const uint32_t *__restrict__ item_start, const int4 *__restrict__ desc,
                                                          const float *__restrict__ centroids, int d, int nlist, int pdiv, int round_f16,
                                                          f4 *__restrict__ rq) {
  const uint32_t item = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (item >= item_start[nlist]) return;
  const int lane = threadIdx.x & 63;
  const int4 dsc = desc[item];
  const int part = dsc.x, i0 = dsc.y, cnt = dsc.z;
  uint32_t qj[Q8_G];
#pragma unroll
  for (int j = 0; j < Q8_G; ++j) qj[j] = pair_idx[i0 + (j < cnt ? j : 0)] / (uint32_t)pdiv;
  for (int dim = lane; dim < d; dim += 64) {
    const float cen = centroids[(int64_t)part * d + dim];
    float r[Q8_G];
#pragma unroll
    for (int j = 0; j < Q8_G; ++j) {
      float v = q[(int64_t)qj[j] * d + dim] - cen;      // v2.rs:316-332, same subtraction as the exact path
      if (round_f16) v = __half2float(__float2half_rn(v));
      r[j] = -v;
    }
    rq[((int64_t)item * d + dim) * 2 + 0] = f4{r[0], r[1], r[2], r[3]};
    rq[((int64_t)item * d + dim) * 2 + 1] = f4{r[4], r[5], r[6], r[7]};
  }
}

// one table word: eight queries' entries of (sub-quantiser, codeword); acc[k] = {L_2k, L_2k+1}
__device__ __forceinline__ uint2 q8_entry_quantise(const f2 (&acc)[4], const f2 (&s)[4]) {
  // z = L * s - 0.5, clamped to [0, 63] (a NaN becomes 0: the row survives and the exact pass decides), converted to u8.
  // Whatever rounding v_cvt_pk_u8_f32 uses, e <= L * s (+ 4e-6 from the FMA) and e > L * s - 1.5 for unsaturated entries.
  uint32_t w[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const f2 half = {-0.5f, -0.5f};
    const f2 z0 = __builtin_elementwise_fma(acc[2 * h], s[2 * h], half), z1 = __builtin_elementwise_fma(acc[2 * h + 1], s[2 * h + 1], half);
    uint32_t v = 0u;
    v = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_amdgcn_fmed3f(z0.x, 0.0f, (float)Q8_CAP_E), 0u, v);
    v = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_amdgcn_fmed3f(z0.y, 0.0f, (float)Q8_CAP_E), 1u, v);
    v = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_amdgcn_fmed3f(z1.x, 0.0f, (float)Q8_CAP_E), 2u, v);
    v = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_amdgcn_fmed3f(z1.y, 0.0f, (float)Q8_CAP_E), 3u, v);
    w[h] = v;
  }
  return make_uint2(w[0], w[1]);
}

template <int SD>
__global__ __launch_bounds__(Q_BS, LH_Q8_WAVES) void ivfpq_qscan8_kernel(QscanArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int M = 16;
  constexpr uint32_t LIM = Q8_SE + 1u;
  static_assert(4 * Q8_CAP_E <= 255 && 16 * Q8_CAP_E < 65536, "byte / u16 field sums must not carry");
  // [M][256] x (8 x u8) in STATIC LDS at offset 0: the gather address is one shift of the code byte plus an immediate
  __shared__ __attribute__((aligned(16))) uint2 lutq[M * 256];
  uint32_t *cand = reinterpret_cast<uint32_t *>(smem);                // [8][Q_CAP]
  uint32_t *misc = cand + Q8_G * Q_CAP;                               // [0..7] survivor counts
  float *sc = reinterpret_cast<float *>(misc + Q8_G);                 // [8] SE8 / T (1e30: no such query in this item)
  uint16_t *csum = reinterpret_cast<uint16_t *>(sc + Q8_G);           // [8][Q_CAP] the survivors' integer sums

  const uint32_t item = blockIdx.x;
  if (item >= p.item_start[p.nlist]) return;
  const int4 dsc = p.desc[item];
  const int part = dsc.x, i0 = dsc.y, cnt = dsc.z;
  const uint32_t off = p.part_offsets[part];
  const int np = (int)(p.part_offsets[part + 1] - off);
  if (np == 0) return;   // uniform; seg_cnt stays 0
  if (threadIdx.x < Q8_G) {
    misc[threadIdx.x] = 0;
    float s = 1e30f;   // absent query: every non-zero entry saturates
    if ((int)threadIdx.x < cnt) {
      const uint32_t pr = p.pair_idx[i0 + threadIdx.x];
      const float T = key_to_float(p.tbound[pr / (uint32_t)p.nprobes]);    // 0 < T < inf (class A)
      s = fminf((float)Q8_SE / T, 1e30f);
    }
    sc[threadIdx.x] = s;
  }
  __syncthreads();
  {
    // lane (c = tid & 255, half = tid >> 8) fills sub-quantisers [half * 8, half * 8 + 8)
    const f4 *rq8 = p.rq + (int64_t)item * p.d * 2;
    const int c = threadIdx.x & 255, half = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
    const f4 sa = *reinterpret_cast<const f4 *>(sc), sb = *reinterpret_cast<const f4 *>(sc + 4);
    const f2 s[4] = {{sa.x, sa.y}, {sa.z, sa.w}, {sb.x, sb.y}, {sb.z, sb.w}};
    constexpr int QV = SD / 4;
#pragma unroll 1
    for (int i = 0; i < M / 2; ++i) {
      const int mm = half * (M / 2) + i;
      const float *cbp = p.codebook + ((int64_t)mm * 256 + c) * SD;
      f4 cb[QV];
#pragma unroll
      for (int u = 0; u < QV; ++u) cb[u] = reinterpret_cast<const f4 *>(cbp)[u];
      f2 acc[4] = {{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}};
#pragma unroll
      for (int u = 0; u < SD; ++u) {
        const f4 ra = rq8[(mm * SD + u) * 2 + 0], rb = rq8[(mm * SD + u) * 2 + 1];   // wave-uniform: scalar loads
        const float cv = cb[u >> 2][u & 3];
        const f2 cc = {cv, cv};
        const f2 d0 = f2{ra.x, ra.y} + cc, d1 = f2{ra.z, ra.w} + cc, d2 = f2{rb.x, rb.y} + cc, d3 = f2{rb.z, rb.w} + cc;
        acc[0] = __builtin_elementwise_fma(d0, d0, acc[0]);
        acc[1] = __builtin_elementwise_fma(d1, d1, acc[1]);
        acc[2] = __builtin_elementwise_fma(d2, d2, acc[2]);
        acc[3] = __builtin_elementwise_fma(d3, d3, acc[3]);
      }
      lutq[mm * 256 + c] = q8_entry_quantise(acc, s);
    }
  }
  __syncthreads();
  {
    const uint8_t *pcodes = p.codes + (int64_t)off * M;
    uint4 cwn = make_uint4(0, 0, 0, 0);
    if ((int)threadIdx.x < np) cwn = *reinterpret_cast<const uint4 *>(pcodes + (int64_t)threadIdx.x * M);
    for (int base = 0; base < np; base += Q_BS) {
      const int row = base + threadIdx.x;
      const uint4 cw = cwn;
      if (row + Q_BS < np) cwn = *reinterpret_cast<const uint4 *>(pcodes + (int64_t)(row + Q_BS) * M);
      if (row < np) {
        const uint32_t cws[4] = {cw.x, cw.y, cw.z, cw.w};
        uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;   // u16 pairs: queries (0, 2), (1, 3), (4, 6), (5, 7)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          uint32_t a0 = 0, a1 = 0;                 // bytes: queries 0..3 / 4..7, four sub-quantisers each
#pragma unroll
          for (int bb = 0; bb < 4; ++bb) {
            const uint2 v = lutq[(e * 4 + bb) * 256 + ((cws[e] >> (8 * bb)) & 255u)];
            a0 += v.x; a1 += v.y;
          }
          w0 += a0 & 0x00FF00FFu; w1 += (a0 >> 8) & 0x00FF00FFu;
          w2 += a1 & 0x00FF00FFu; w3 += (a1 >> 8) & 0x00FF00FFu;
        }
        const uint32_t sj[Q8_G] = {w0 & 0xFFFFu, w1 & 0xFFFFu, w0 >> 16, w1 >> 16, w2 & 0xFFFFu, w3 & 0xFFFFu, w2 >> 16, w3 >> 16};
        bool any = false;
#pragma unroll
        for (int j = 0; j < Q8_G; ++j) any |= sj[j] <= LIM;
        if (any && row_allowed(p.allow, off + (uint32_t)row)) {
          const uint32_t pos = off + (uint32_t)row;
#pragma unroll
          for (int j = 0; j < Q8_G; ++j) {
            if (sj[j] <= LIM) {
              const uint32_t slot = atomicAdd(&misc[j], 1u);
              if (slot < (uint32_t)Q_CAP) { cand[j * Q_CAP + slot] = pos; csum[j * Q_CAP + slot] = (uint16_t)sj[j]; }
            }
          }
        }
      }
    }
  }
  __syncthreads();
#pragma unroll 1
  for (int j = 0; j < cnt; ++j) {
    const uint32_t pr = p.pair_idx[i0 + j];
    const uint32_t raw = misc[j];
    const uint32_t n = min(raw, (uint32_t)Q_CAP);
    const int64_t seg = (int64_t)pr;      // pair = query * nprobes + rank = the segment index
    if (threadIdx.x == 0) {
      p.seg_cnt[seg] = raw;   // raw > Q_CAP: survivors were lost -> the rescan kernel redoes this (query, probe) exactly
      if (raw > (uint32_t)Q_CAP) { p.qovf[pr / (uint32_t)p.nprobes] = 1u; p.ovf[1u + atomicAdd(&p.ovf[0], 1u)] = (uint32_t)seg; }
    }
    for (uint32_t i = threadIdx.x; i < n; i += Q_BS) {
      p.seg_pos[seg * Q_CAP + i] = cand[j * Q_CAP + i];
      p.seg_sum[seg * Q_CAP + i] = csum[j * Q_CAP + i];
    }
  }
}

// ---- host -------------------------------------------------------------------------------------------------------------------
bool qscan8_enabled(int m, int sd) {
  static const bool on = getenv("LANCE_HIP_Q8") != nullptr;
  return on && m == 16 && (sd == 4 || sd == 8 || sd == 16);
}

size_t qscan8_lds_bytes() { return (size_t)Q8_G * Q_CAP * 4 + Q8_G * 4 + Q8_G * 4 + (size_t)Q8_G * Q_CAP * 2; }

int qscan8_residual(lance_hip_ctx *ctx, const float *qs, const uint32_t *pair_idx, const uint32_t *item_start, const int4 *desc,
                    const float *centroids, int d, int nlist, int nprobes, int round_f16, uint32_t max_items, f4 *rq) {
  hipLaunchKernelGGL(q_residual8_kernel, dim3((unsigned)cdiv(max_items, 4)), dim3(256), 0, ctx->stream, qs, pair_idx, item_start, desc, centroids, d,
                     nlist, nprobes, round_f16, rq);
  return LANCE_HIP_OK;
}

bool qscan8_launch(lance_hip_ctx *ctx, const QscanArgs &a, int sd, unsigned grid) {
  const size_t lds = qscan8_lds_bytes();
  if (sd == 4) hipLaunchKernelGGL((ivfpq_qscan8_kernel<4>), dim3(grid), dim3(Q_BS), lds, ctx->stream, a);
  else if (sd == 8) hipLaunchKernelGGL((ivfpq_qscan8_kernel<8>), dim3(grid), dim3(Q_BS), lds, ctx->stream, a);
  else if (sd == 16) hipLaunchKernelGGL((ivfpq_qscan8_kernel<16>), dim3(grid), dim3(Q_BS), lds, ctx->stream, a);
  else return false;
  return true;
}

}  // namespace lh
