// search_q.hip -- quantised filter scan: FOUR queries share every LDS gather, survivors are re-evaluated exactly.
//
// Why: rocprofv3 counters of the exact partition-major pair scan (search_pm.hip; profiles/r02_*): 84 % of the VALU issue
// slots and 68 % of the LDS cycles are busy -- the scan is instruction-issue bound, and what it issues is (a) the exact f32
// LUT build (35 % of a work item) and (b) one ds_read_b64 + two f32 adds + two address ops per (row, sub-quantiser) per PAIR of
// queries.  Bit-exact results only need the exact arithmetic for rows that can reach the output.  So the main pass becomes a
// FILTER with a rigorous lower bound, in integer arithmetic:
//   * the bound pass (unchanged, exact) gives every query an upper bound T of its final k-th distance;
//   * per (partition, 4 queries): e[m][c] = min(floor(L[m][c] * SE / T), CAPE) as u16, four queries packed in 8 bytes
//     ([m][256] x uint2 at LDS offset 0: one ds_read_b64 serves four queries); L >= 0 (squared L2), CAPE = 65535 / M, so the
//     sum of M entries cannot carry between the packed u16 fields and two v_add_u32 add all four queries;
//   * a row can only have ADC distance <= T if sum_m e[m][code_m] <= SE + 2 (floor() only lowers, the head-room covers
//     the f32 rounding of the reference's sequential sum and of the FMA-evaluated L): everything else is dropped without ever
//     touching f32;
//   * the survivors (a few hundred per query) go to per-(query, probe) segments; ivfpq_qmerge_kernel recomputes THEIR
//     distances exactly as the reference does (l2_scalar order per sub-vector, sequential-m sum), keeps key <= T, and runs the
//     same (dist, rowid) selection / tie check as the exact path -- so ids and distances stay bit-equal to the oracle.
// The quantised LUT may be computed with FMAs (it is a bound, not a result): half the VALU of the exact build.
// Queries without a usable bound (fewer than k*refine rows in the nearest partition, T = 0, NaN) take the exact pair kernel.
// Reference behaviour preserved: pq/distance.rs:109-144, flat/index.rs:94-126, scanner.rs:3440-3468, v2.rs:316-332.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "common.h"
#include "exact.cuh"
#include "index.h"
#include "kernels.h"
#include "search_common.cuh"
#include "pm_common.cuh"
#include "q_common.cuh"

#pragma clang fp contract(off)

namespace lh {

// ---- grouping by (partition, bound class) ---------------------------------------------------------------------
// class A (virtual partition = partition): the query has a usable bound 0 < T < inf -> quantised scan;
// class B (virtual partition = nlist + partition): exact pair kernel.  tbound[q] = T for class A, 0xFFFFFFFF for class B.
__global__ __launch_bounds__(256) void q_tclass_keys_kernel(const uint32_t *__restrict__ probes, int64_t npairs, int nprobes, int nlist,
                                                            const uint32_t *__restrict__ tglobal, uint32_t *__restrict__ keys,
                                                            uint32_t *__restrict__ tbound, int dot) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= npairs) return;
  const int64_t q = i / nprobes;
  const uint32_t T = tglobal[q];
  // order_key(+0.0) < T < order_key(+inf); dot distances (1 - q . c^) have either sign: order_key(-inf) < T < order_key(+inf)
  const bool usable = T > (dot ? 0x007FFFFFu : 0x80000000u) && T < 0xFF800000u;
  keys[i] = probes[i] + (usable ? 0u : (uint32_t)nlist);
  if (i % nprobes == 0) tbound[q] = usable ? T : 0xFFFFFFFFu;
}

// keys[q * nb + b] = the query's b-th nearest partition, b < nb (nb = 1: the nearest; the dot metric's bound pass takes the two nearest)
__global__ __launch_bounds__(256) void q_nearest_keys_kernel(const uint32_t *__restrict__ probes, int nq, int nprobes, int nb, uint32_t *__restrict__ keys) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < nq * nb) keys[i] = probes[(int64_t)(i / nb) * nprobes + (i % nb)];
}

// item_start[vp] = exclusive scan of ceil(c_vp / G) over the first nvp virtual partitions
__global__ __launch_bounds__(256) void q_item_table_kernel(const uint32_t *__restrict__ pair_starts, int nvp, int G, uint32_t *__restrict__ item_start) {
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t carry_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < nvp; base += 256) {
    const int i = base + threadIdx.x;
    const uint32_t v = i < nvp ? (pair_starts[i + 1] - pair_starts[i] + G - 1) / G : 0;
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
    if (i < nvp) item_start[i] = carry + woff + incl - v;
    __syncthreads();
    if (threadIdx.x == 255) carry_s = carry + woff + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) item_start[nvp] = carry_s;
}

// desc[item] = {partition, first grouped pair, number of pairs (1..G), 0}
__global__ __launch_bounds__(256) void q_item_desc_kernel(const uint32_t *__restrict__ item_start, const uint32_t *__restrict__ pair_starts,
                                                          int nvp, int G, uint32_t max_items, int4 *__restrict__ desc) {
  const uint32_t item = blockIdx.x * 256 + threadIdx.x;
  if (item >= max_items) return;
  int4 dsc = make_int4(-1, 0, 0, 0);
  if (item < item_start[nvp]) {
    int vp = (int)find_partition_dev(item_start, nvp, item);
    while (item_start[vp + 1] <= item) ++vp;   // empty ranges share their successor's start
    const uint32_t g = item - item_start[vp];
    const uint32_t ps = pair_starts[vp], pe = pair_starts[vp + 1];
    const uint32_t i0 = ps + (uint32_t)G * g;
    dsc.x = vp;
    dsc.y = (int)i0;
    dsc.z = (int)min((uint32_t)G, pe - i0);
  }
  desc[item] = dsc;
}

// ---- residual pre-pass ------------------------------------------------------------------------------------------------------
// The four queries of an item share every table entry's arithmetic, and their residual components are the same in all 512
// lanes.  Kept in LDS they cost one broadcast ds_read_b128 per (entry, dimension): 38 % of the scan kernel's LDS-pipe time
// (PMC: LDS busy 73 %, and 20 % fewer VALU instructions in the table build changed nothing).  Written once to global memory
// by this kernel they come back through the SCALAR cache (s_load_dwordx8/16 into SGPRs, wave-uniform addresses) and the LDS
// pipe is left to the gathers.  Layout: rq[item][dim] = float4 of the 4 queries' NEGATED residual components.
__global__ __launch_bounds__(256) void q_residual_kernel(const float *__restrict__ q, const uint32_t *__restrict__ pair_idx,
                                                         const uint32_t *__restrict__ item_start, const int4 *__restrict__ desc,
                                                         const float *__restrict__ centroids, int d, int nlist, int pdiv, int round_f16,
                                                         f4 *__restrict__ rq) {
  const uint32_t item = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (item >= item_start[nlist]) return;
  const int lane = threadIdx.x & 63;
  const int4 dsc = desc[item];
  const int part = dsc.x, i0 = dsc.y, cnt = dsc.z;
  uint32_t qj[Q_G];
#pragma unroll
  for (int j = 0; j < Q_G; ++j) qj[j] = pair_idx[i0 + (j < cnt ? j : 0)] / (uint32_t)pdiv;
  for (int dim = lane; dim < d; dim += 64) {
    const float cen = centroids[(int64_t)part * d + dim];
    f4 r4;
#pragma unroll
    for (int j = 0; j < Q_G; ++j) {
      float v = q[(int64_t)qj[j] * d + dim] - cen;      // v2.rs:316-332, same subtraction as the exact path
      if (round_f16) v = __half2float(__float2half_rn(v));
      r4[j] = -v;   // NEGATED: (r - c)^2 is evaluated as (c + (-r))^2 so that the add packs (v_pk_add_f32)
    }
    rq[(int64_t)item * d + dim] = r4;
  }
}

// (An MFMA-built table -- [256 x 8] x [8 x 4] per sub-quantiser on v_mfma_f32_16x16x32_bf16, bf16 hi / lo split -- was parity-green in
// round 4 and SLOWER than the packed-VALU build below (scan 0.405 vs 0.347 ms per 10k-query batch at C2, gpurun r04e: eight tiles per
// wave are a chain of L2 round trips the 44-VALU build does not have); so was an 8-queries-per-gather variant with 8-bit entries
// (round 3, profiles/r03_q8_variant.txt).  Both were removed in round 5; the matrix-core scan of search_ms.hip took their place.)
template <int SD, int MU>
__global__ __launch_bounds__(Q_BS, (MU == 1 ? (SD <= 8 ? Q_WAVES : 6) : 4)) void ivfpq_qscan_kernel(QscanArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int M = MU * 16;
  constexpr uint32_t CAPE = 65535u / M;        // largest entry: M of them cannot overflow a u16 field
  constexpr uint32_t SE = CAPE - CAPE / 32;     // the bound T maps to SE; ~3 % head-room below CAPE
  constexpr uint32_t LIM = SE + M + 2;          // one unit per entry for the conversion's rounding; +2 covers the f32 rounding terms (see header)
  static_assert(LIM < CAPE - 1, "a saturated entry must put the row above the limit");
  // [M][256] x (4 x u16) in STATIC LDS at offset 0: the gather address is one SDWA shift of the code byte plus an immediate
  __shared__ __attribute__((aligned(16))) uint2 lutq[M * 256];
  uint32_t *cand = reinterpret_cast<uint32_t *>(smem);                // [4][Q_CAP]
  uint32_t *misc = cand + 4 * Q_CAP;                                  // [0..3] survivor counts
  float *sc = reinterpret_cast<float *>(misc + 4);                    // [4] SE / T / 65535 (1e30: no such query in this item)
  uint16_t *csum = reinterpret_cast<uint16_t *>(sc + 4);              // [4][Q_CAP] the survivors' integer sums

  // one item per workgroup and no loop: nothing is stored to global memory before the residual loads, so the compiler may
  // (and does) turn them into scalar loads
  const uint32_t item = blockIdx.x;
  if (item >= p.item_start[p.nlist]) return;
  {
    const int4 dsc = p.desc[item];
    const int part = dsc.x, i0 = dsc.y, cnt = dsc.z;
    const uint32_t off = p.part_offsets[part];
    const int np = (int)(p.part_offsets[part + 1] - off);
    if (np == 0) return;   // uniform; seg_cnt stays 0
    uint32_t qj[Q_G], rk[Q_G];
#pragma unroll
    for (int j = 0; j < Q_G; ++j) {
      const uint32_t pr = p.pair_idx[i0 + (j < cnt ? j : 0)];
      qj[j] = pr / (uint32_t)p.nprobes;
      rk[j] = pr % (uint32_t)p.nprobes;
    }
    if (threadIdx.x < Q_G) {
      misc[threadIdx.x] = 0;
      float s = 1e30f;   // absent query: every non-zero entry saturates
      if ((int)threadIdx.x < cnt) {
        const float T = key_to_float(p.tbound[qj[threadIdx.x]]);    // 0 < T < inf (class A)
        s = fminf((float)SE / T, 1e30f);
      }
      sc[threadIdx.x] = s * (1.0f / 65535.0f);
    }
    __syncthreads();
    const f4 *rq4 = p.rq + (int64_t)item * p.d;
    const uint32_t lim4[Q_G] = {LIM, LIM, LIM, LIM};
    // quantised LUT: lane (c = tid & 255, half = tid >> 8) fills sub-quantisers [half * M/2, (half+1) * M/2).  A bound, not a
    // result (q_entry_acc / q_entry_quantise above).
    {
      const int c = threadIdx.x & 255, half = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
      constexpr int MH = M / 2;
      const f4 s4 = *reinterpret_cast<const f4 *>(sc);
      const f2 s01 = {s4.x, s4.y}, s23 = {s4.z, s4.w};
#pragma unroll LH_Q_LUT_UNROLL
      for (int i = 0; i < MH; ++i) {
        const int mm = half * MH + i;
        f2 acc01, acc23;
        q_entry_acc<SD>(rq4 + mm * SD, p.codebook + ((int64_t)mm * 256 + c) * SD, acc01, acc23);
        lutq[mm * 256 + c] = q_entry_quantise<CAPE>(acc01, acc23, s01, s23);
      }
    }
    __syncthreads();
    // scan: no barrier inside; survivors go to the per-query LDS lists
    {
      const uint8_t *pcodes = p.codes + (int64_t)off * M;
      uint4 cwn[MU];
      if ((int)threadIdx.x < np) {
#pragma unroll
        for (int w = 0; w < MU; ++w) cwn[w] = *reinterpret_cast<const uint4 *>(pcodes + (int64_t)threadIdx.x * M + w * 16);
      }
      for (int base = 0; base < np; base += Q_BS) {
        const int row = base + threadIdx.x;
        uint4 cw[MU];
#pragma unroll
        for (int w = 0; w < MU; ++w) cw[w] = cwn[w];
        if (row + Q_BS < np) {
#pragma unroll
          for (int w = 0; w < MU; ++w) cwn[w] = *reinterpret_cast<const uint4 *>(pcodes + (int64_t)(row + Q_BS) * M + w * 16);
        }
        if (row < np) {
          uint32_t a0 = 0, a1 = 0;
#pragma unroll
          for (int w = 0; w < MU; ++w) {
            const uint32_t cws[4] = {cw[w].x, cw[w].y, cw[w].z, cw[w].w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int bb = 0; bb < 4; ++bb) {
                const uint2 v = lutq[(w * 16 + e * 4 + bb) * 256 + ((cws[e] >> (8 * bb)) & 255u)];
                a0 += v.x; a1 += v.y;
              }
          }
          const bool p0 = (a0 & 0xFFFFu) <= lim4[0], p1 = (a0 >> 16) <= lim4[1], p2 = (a1 & 0xFFFFu) <= lim4[2], p3 = (a1 >> 16) <= lim4[3];
          if ((p0 | p1 | p2 | p3) && row_allowed(p.allow, off + (uint32_t)row)) {
            const uint32_t pos = off + (uint32_t)row;
            if (p0) { const uint32_t slot = atomicAdd(&misc[0], 1u); if (slot < (uint32_t)Q_CAP) { cand[0 * Q_CAP + slot] = pos; csum[0 * Q_CAP + slot] = (uint16_t)(a0 & 0xFFFFu); } }
            if (p1) { const uint32_t slot = atomicAdd(&misc[1], 1u); if (slot < (uint32_t)Q_CAP) { cand[1 * Q_CAP + slot] = pos; csum[1 * Q_CAP + slot] = (uint16_t)(a0 >> 16); } }
            if (p2) { const uint32_t slot = atomicAdd(&misc[2], 1u); if (slot < (uint32_t)Q_CAP) { cand[2 * Q_CAP + slot] = pos; csum[2 * Q_CAP + slot] = (uint16_t)(a1 & 0xFFFFu); } }
            if (p3) { const uint32_t slot = atomicAdd(&misc[3], 1u); if (slot < (uint32_t)Q_CAP) { cand[3 * Q_CAP + slot] = pos; csum[3 * Q_CAP + slot] = (uint16_t)(a1 >> 16); } }
          }
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < Q_G; ++j) {
      if (j < cnt) {
        const uint32_t raw = misc[j];
        const uint32_t n = raw > (uint32_t)Q_CAP ? (uint32_t)Q_CAP : raw;
        const int64_t seg = (int64_t)qj[j] * p.nprobes + rk[j];
        if (threadIdx.x == 0) {
          p.seg_cnt[seg] = raw;   // raw > Q_CAP: survivors were lost -> the rescan kernel redoes this (query, probe) exactly
          if (raw > (uint32_t)Q_CAP) { p.qovf[qj[j]] = 1u; p.ovf[1u + atomicAdd(&p.ovf[0], 1u)] = (uint32_t)seg; }
        }
        for (uint32_t i = threadIdx.x; i < n; i += Q_BS) {
          p.seg_pos[seg * Q_CAP + i] = cand[j * Q_CAP + i];
          p.seg_sum[seg * Q_CAP + i] = csum[j * Q_CAP + i];
        }
      }
    }
  }
}

// ---- integer bound pass ---------------------------------------------------------------------------------------------------
// Every query needs an upper bound T of its final k*refine-th distance before the filter scan.  Round 1 got it from an exact
// f32 scan of the query's nearest partition (0.17 ms per 10,000 queries).  A bound does not need exact distances either:
// with e[m][c] = min(floor(L * s), CAPE) a row's integer sum S satisfies  dist * s < S + M  as long as no entry saturated
// (S <= HMAX < CAPE guarantees that), so if at least k*refine rows of the partition have S <= B then the k*refine-th smallest
// ADC distance is below (B + M + 1) / s -- a valid T from a HISTOGRAM of integer sums, four queries per gather, no f32 scan.
// The scale s only affects tightness: it is set from the mean table entry (the expected distance of a random code), which puts
// the nearest 1-3 % of a partition's rows around a quarter of the histogram range (slack ~3 % on T: ~8 % more survivors).
// A query whose partition has fewer than k*refine countable rows gets no bound (class B: exact pair kernel), as before.
template <int SD, int MU>
__global__ __launch_bounds__(Q_BS, (MU == 1 ? 6 : 4)) void ivfpq_qbound_kernel(QboundArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int M = MU * 16;
  constexpr uint32_t CAPE = 65535u / M;
  constexpr uint32_t SE = CAPE - CAPE / 32;
  constexpr uint32_t HMAX = (uint32_t)(QB_BINS << QB_SHIFT) - 1u < CAPE - 1u ? (uint32_t)(QB_BINS << QB_SHIFT) - 1u : CAPE - 1u;
  __shared__ __attribute__((aligned(16))) uint2 lutq[M * 256];
  uint32_t *hist = reinterpret_cast<uint32_t *>(smem);                   // [4][QB_BINS]
  float *sums = reinterpret_cast<float *>(hist + 4 * QB_BINS);            // [4] sum of all table entries
  float *sc = sums + 4;                                                   // [4] scale
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t item = blockIdx.x;
  if (item >= p.item_start[p.nlist]) return;
  {
    const int4 dsc = p.desc[item];
    const int part = dsc.x, i0 = dsc.y, cnt = dsc.z;
    const uint32_t off = p.part_offsets[part];
    const int np = (int)(p.part_offsets[part + 1] - off);
    if (np < p.keff) return;   // uniform: fewer rows than k*refine -> no bound from this partition
    uint32_t qj[Q_G];
#pragma unroll
    for (int j = 0; j < Q_G; ++j) qj[j] = p.pair_idx[i0 + (j < cnt ? j : 0)];
    const f4 *rq4 = p.rq + (int64_t)item * p.d;
    for (int i = threadIdx.x; i < 4 * QB_BINS; i += Q_BS) hist[i] = 0u;
    if (threadIdx.x < 4) sums[threadIdx.x] = 0.0f;
    __syncthreads();
    const int c = threadIdx.x & 255, half = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
    constexpr int MH = M / 2;
    q_mean_entry_sums<Q_BS>(p.rq + (int64_t)item * p.d, p.cb_mean, p.d, sums);   // round 3: was a full table build
    __syncthreads();
    if (threadIdx.x < Q_G) {
      const float mean = sums[threadIdx.x] + p.cb_mean[p.d];     // sum over m of the mean entry: distance of a random code
      float s = 1e30f;                                           // absent query / degenerate mean: everything saturates
      if ((int)threadIdx.x < cnt && mean > 0.0f && mean < INFINITY) s = fminf((float)SE / mean, 1e30f);
      sc[threadIdx.x] = s;
    }
    __syncthreads();
    {
      const f4 s4 = *reinterpret_cast<const f4 *>(sc) * (1.0f / 65535.0f);
      const f2 s01 = {s4.x, s4.y}, s23 = {s4.z, s4.w};
#pragma unroll 1
      for (int i = 0; i < MH; ++i) {
        const int mm = half * MH + i;
        f2 acc01, acc23;
        q_entry_acc<SD>(rq4 + mm * SD, p.codebook + ((int64_t)mm * 256 + c) * SD, acc01, acc23);
        lutq[mm * 256 + c] = q_entry_quantise<CAPE>(acc01, acc23, s01, s23);
      }
    }
    __syncthreads();
    {
      const uint8_t *pcodes = p.codes + (int64_t)off * M;
      for (int row = threadIdx.x; row < np; row += Q_BS) {
        uint32_t a0 = 0, a1 = 0;
#pragma unroll
        for (int w = 0; w < MU; ++w) {
          const uint4 cw = *reinterpret_cast<const uint4 *>(pcodes + (int64_t)row * M + w * 16);
          const uint32_t cws[4] = {cw.x, cw.y, cw.z, cw.w};
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
              const uint2 v = lutq[(w * 16 + e * 4 + bb) * 256 + ((cws[e] >> (8 * bb)) & 255u)];
              a0 += v.x; a1 += v.y;
            }
        }
        const uint32_t sj[4] = {a0 & 0xFFFFu, a0 >> 16, a1 & 0xFFFFu, a1 >> 16};
        if (row_allowed(p.allow, off + (uint32_t)row)) {
#pragma unroll
          for (int j = 0; j < Q_G; ++j)
            if (sj[j] <= HMAX) atomicAdd(&hist[j * QB_BINS + (sj[j] >> QB_SHIFT)], 1u);
        }
      }
    }
    __syncthreads();
    // wave j (< cnt) finds the first bin where the cumulative count reaches keff
    if (wave < cnt) {
      const uint32_t *h = hist + wave * QB_BINS;
      constexpr int PER = QB_BINS / 64;
      uint32_t loc[PER], tot = 0;
#pragma unroll
      for (int i = 0; i < PER; ++i) { loc[i] = h[lane * PER + i]; tot += loc[i]; }
      uint32_t incl = tot;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
      }
      uint32_t run = incl - tot;
      int found = -1;
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        run += loc[i];
        if (found < 0 && run >= (uint32_t)p.keff) found = lane * PER + i;
      }
      const uint64_t mask = __ballot(found >= 0);
      if (mask) {
        const int leader = __ffsll((long long)mask) - 1;
        const int bin = __shfl(found, leader, 64);
        if (lane == 0) {
          const float B = (float)(((uint32_t)bin + 1u) << QB_SHIFT);     // every counted row has S <= B - 1
          const float T = (B + (float)M) / sc[wave] * 1.00001f;           // dist * s <= S + M; margin >> (SD + M) * 2^-24 for the f32 / FMA rounding terms
          if (T > 0.0f && T < INFINITY) atomicMin(&p.tglobal[qj[wave]], order_key(T));
        }
      }
    }
  }
}

// ---- exact re-evaluation + merge --------------------------------------------------------------------------------
constexpr int QM_G = 16;       // probes whose residuals are staged together (nprobes <= 16: one staging, no second pass)

struct QmergeArgs {
  const float *q;
  const uint32_t *probes;        // [nq][nprobes]
  const float *centroids, *codebook;
  const uint8_t *codes;
  const uint64_t *row_ids;
  int d, nprobes, round_f16;
  int qm_g;                      // probes whose residuals are staged together (<= QM_G; fewer for long rows: LDS = occupancy)
  int vec4;                      // d % 4 == 0 and 16-byte aligned query / centroid rows: staged with float4 loads
  const uint16_t *seg_sum;       // [nq * nprobes][Q_CAP] integer sums of the survivors (scan kernels)
  int cut_shift;                 // histogram bin = sum >> cut_shift (512 bins cover 0 .. LIM)
  uint32_t cut_slack;            // a survivor whose sum exceeds (upper edge of the keff-th bin) + cut_slack cannot reach the top keff
  const uint32_t *qslack;        // search_ms.hip: [nq] per-query bound of |sum - dist * s| (units); the cut carries twice that on top of cut_slack
  const float *seg_val;          // search_ms.hip (rows-on-lanes kernel): != NULL -> the survivors come as {position, accumulator value} records
  const uint2 *seg_pv;           //   [nq * nprobes][Q_CAP] (the same pointer) instead of seg_pos / seg_sum;
  const f2 *seg_scale;           //   sum = rint(val * seg_scale[pair].x + seg_scale[pair].y), clamped to 0 .. 65535
#ifdef LH_TIMING_EXPERIMENTS      // builds with -DLH_TIMING_EXPERIMENTS only (scripts/build_variant.sh): the product library has no such switch
  int dbg;                       // LANCE_HIP_QM_DBG (timing experiments, results WRONG): the kernel returns after 1: the cut, 2: staging the residuals,
                                 // 3: compaction, 4: exact re-evaluation, 5: the sort
#define QM_DBG_RETURN(a, n) do { if ((a).dbg == (n)) return; } while (0)
#else
#define QM_DBG_RETURN(a, n) do { } while (0)
#endif
  const uint32_t *tbound;        // class per query (0xFFFFFFFF: class B -> pool)
  uint32_t *tglobal;             // class B: running bound of the exact pair kernel
  const uint32_t *seg_cnt, *seg_pos;
  uint32_t *pool_key, *pool_pos, *pool_cnt;
  int pool_cap;
  const uint32_t *qovf;          // [nq] != 0: some segment of the query overflowed (its rows come through the pool)
  const uint32_t *ovf;           // [1 + nq * nprobes] count, then the overflowed segments (query * nprobes + rank): the rescan kernel's work list
  const uint32_t *allow;         // prefilter bitmap (rescan)
  SelectOut o;
};

// Segments that lost survivors (more than Q_CAP rows under the bound: it was loose for this query) are rescanned with the
// exact f32 table; what stays under the (tightened) threshold is appended to the query's pool, which the merge kernel reads
// besides the segments.  Rare for M = 16 / 32 (a few queries per 10,000); the tiled shapes (M >= 48) see it for 1-2 % of the
// segments -- small partitions give loose bounds, and a query with a loose bound overflows in MOST of its probes.
// Round 3: one workgroup per overflowed SEGMENT, taken from a device-side list the scan kernels fill (ovf[0] = count, then the
// segment indices), by a fixed grid of looping workgroups.  The earlier one-workgroup-per-query form walked such a query's
// 10-50 partitions one after the other (table build + scan each): 0.2-0.4 ms of single-workgroup latency in front of every
// merge at C3, whatever the workgroup size (256 lanes: 0.25 ms, 1024 lanes: 0.22 ms).
// BS: 256 lanes where the table is small (M = 16 / 32); 1024 for the tiled shapes (48-96 KiB of exact table per workgroup).
template <int SD, int MU, int BS, bool DOT = false>
__global__ __launch_bounds__(BS) void ivfpq_qrescan_kernel(QmergeArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int M = MU * 16;
  constexpr int QV = SD / 4;
  constexpr int QMET = DOT ? METRIC_DOT : METRIC_L2;
  constexpr int CAP = BS > 256 ? 2 * BS : 1024;
  __shared__ uint32_t ckey[CAP], cpos[CAP], sorted[BS], misc[8];
  const SelectOut &o = a.o;
  const int dpad = (a.d + 3) & ~3;
  float *r = reinterpret_cast<float *>(smem);   // [dpad]
  float *lutx = r + dpad;                       // [M][256] exact table
  CandBuf b{ckey, cpos, &misc[0], &misc[1]};
  const uint32_t nseg = a.ovf[0];
  for (uint32_t it = blockIdx.x; it < nseg; it += gridDim.x) {
    const uint32_t seg = a.ovf[1 + it];
    const int q = (int)(seg / (uint32_t)a.nprobes);
    const uint32_t part = a.probes[seg];
    const float *qv = a.q + (int64_t)q * a.d;
    __syncthreads();   // the previous segment's buffers are done with
    // any value of tglobal is an upper bound of the query's final keff-th distance (the filter's bound, lowered by whoever
    // finished a segment of this query first)
    const uint32_t t_start = min(a.tbound[q], __hip_atomic_load(&a.tglobal[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if (threadIdx.x == 0) {
      misc[0] = 0; misc[1] = t_start; misc[3] = 0;
      // every listed segment's rows come through the pool: the merge kernel reads it for queries flagged here (the scan kernels set the
      // flag themselves; search_ms.hip's burst path only lists the segment)
      const_cast<uint32_t *>(a.qovf)[q] = 1u;
    }
    for (int e = threadIdx.x; e < a.d; e += BS) {
      float v = qv[e];
      if constexpr (!DOT) {      // (dot: no residual, pq/distance.rs:60-92)
        v = v - a.centroids[(int64_t)part * a.d + e];
        if (a.round_f16) v = __half2float(__float2half_rn(v));
      }
      r[e] = v;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < M * 256; idx += BS) {
      const int mm = idx >> 8;
      RegVec<SD> av;
#pragma unroll
      for (int u = 0; u < QV; ++u) av.q[u] = *reinterpret_cast<const f4 *>(&r[mm * SD + 4 * u]);
      lutx[idx] = finish_metric<QMET>(dist_exact<SD, QMET>(av, a.codebook + (int64_t)idx * SD));
    }
    __syncthreads();
    const uint32_t off = o.part_offsets[part];
    const int np = (int)(o.part_offsets[part + 1] - off);
    for (int base = 0; base < np; base += BS) {
      const bool need_tighten = (int)misc[0] > CAP - BS;   // read, barrier, decide (lanes past this point append at once)
      __syncthreads();
      if (need_tighten) tighten_bs<BS, CAP>(b, o.keff, sorted, &misc[2]);
      const uint32_t T = misc[1];
      const int row = base + threadIdx.x;
      if (row < np) {
        const uint8_t *rc = a.codes + ((int64_t)off + row) * M;
        float dist = 0.0f;
#pragma unroll
        for (int w = 0; w < MU; ++w) {
          const uint4 cw = *reinterpret_cast<const uint4 *>(rc + w * 16);
          const uint32_t cws[4] = {cw.x, cw.y, cw.z, cw.w};
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) dist += lutx[(w * 16 + e * 4 + bb) * 256 + ((cws[e] >> (8 * bb)) & 255u)];
        }
        if constexpr (DOT) dist = dist - ((float)M - 1.0f);      // pq/storage.rs:949-957
        const uint32_t kk = order_key(dist);
        if (kk <= T && row_allowed(a.allow, off + (uint32_t)row)) {
          const uint32_t slot = atomicAdd(&misc[0], 1u);
          if (slot < (uint32_t)CAP) { ckey[slot] = kk; cpos[slot] = off + (uint32_t)row; } else misc[3] = 1u;   // an entry was lost
        }
      }
      __syncthreads();
    }
    // publish: about keff rows per segment (<= BS entries left -> one per lane -> the bound is the exact keff-th smallest)
    for (int iter = 0; iter < 3; ++iter) {
      const bool more = (int)misc[0] > o.keff + 28;   // read, barrier (inside tighten_bs), decide
      __syncthreads();
      if (!more) break;
      tighten_bs<BS, CAP>(b, o.keff, sorted, &misc[2]);
    }
    __syncthreads();
    const int c = min((int)misc[0], CAP);
    if (threadIdx.x == 0) {
      uint32_t basep = 0xFFFFFFFFu;
      if (!misc[3] && c > 0) {
        if (misc[1] < t_start) atomicMin(&a.tglobal[q], misc[1]);
        basep = atomicAdd(&a.pool_cnt[q], (uint32_t)c);
      }
      // more rows tied under the bound than the pool holds (or an entry lost above): the exact kernel replays the query
      if (misc[3] || (c > 0 && basep + (uint32_t)c > (uint32_t)a.pool_cap)) { atomicOr(&o.flags[q], FLAG_OVERFLOW); basep = 0xFFFFFFFFu; }
      misc[4] = basep;
    }
    __syncthreads();
    const uint32_t basep = misc[4];
    if (basep != 0xFFFFFFFFu) {
      for (int i = threadIdx.x; i < c; i += BS) {
        a.pool_key[(int64_t)q * a.pool_cap + basep + i] = ckey[i];
        a.pool_pos[(int64_t)q * a.pool_cap + basep + i] = cpos[i];
      }
    }
  }
}

// One workgroup of BS lanes per query.  The kernel is latency-bound (dependent position -> code -> codebook loads, selection
// rounds), so what counts is the number of queries in flight, not lanes per query: BS = 128 is the smallest group the
// threshold machinery allows (the k-th smallest of one value per lane needs BS >= k * refine, at most 128 here).
#ifndef LH_QM_WAVES
#define LH_QM_WAVES 6
#endif
#if LH_QM_WAVES > 0
#define LH_QM_BOUNDS(BS) __launch_bounds__(BS, (SD <= 8 && MU == 1 ? LH_QM_WAVES : 4))
#else
#define LH_QM_BOUNDS(BS) __launch_bounds__(BS)
#endif
template <int SD, int MU, int BS, bool DOT = false>
__global__ LH_QM_BOUNDS(BS) void ivfpq_qmerge_kernel(QmergeArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int M = MU * 16;
  constexpr int QV = SD / 4;
  constexpr int QMET = DOT ? METRIC_DOT : METRIC_L2;
  constexpr int CAP = BS == 128 ? 512 : 1024;   // (key, pos) entries under selection
  static_assert(BS >= SCAN_MAX_KEFF, "tighten_bs selects among one value per lane");
  __shared__ uint32_t ckey[CAP], cpos[CAP], sorted[BS], misc[8];
  // the sort buffers of the last phase live in the dynamic region, which holds the staged residuals until then
  // (qmerge_lds_bytes): less LDS per workgroup = more queries in flight on a CU, which is what this kernel is bound by
  uint64_t *rid = reinterpret_cast<uint64_t *>(smem);                    // [SCAN_LCAP]
  uint32_t *skey = reinterpret_cast<uint32_t *>(rid + SCAN_LCAP);        // [SCAN_LCAP]
  uint32_t *spos = skey + SCAN_LCAP;                                     // [SCAN_LCAP]
  __shared__ uint32_t s_cnt[QM_G + 1], s_pre[QM_G + 1];
  __shared__ f2 s_yz[QM_G];
  __shared__ int s_amb;
  // Round 3: the survivors arrive with their integer sums S.  For one query all sums share one scale s, and
  // S - lo <= dist * s <= S + hi (lo, hi = the rounding slacks of the table encoding), so once keff survivors have S <= B every
  // survivor with S > B + hi + lo is farther than keff others and cannot be in the answer: a 512-bin histogram of the sums
  // gives B, and only the survivors under the cut (about 1.1 x keff of the ~2.5 x keff) are re-evaluated exactly -- compacted
  // first, so that the lanes of a round are all busy.
  constexpr int QM_LC = 512;                       // compacted survivors per chunk
  __shared__ uint32_t s_hist[512], l_pos[QM_LC], s_cut, l_cnt;
  __shared__ uint8_t l_rr[QM_LC];
  const SelectOut &o = a.o;
  const int q = blockIdx.x;
  if (o.flags[q] & FLAG_OVERFLOW) return;   // the exact kernel recomputes this query
  const uint32_t tb = a.tbound[q];
  const bool class_a = tb != 0xFFFFFFFFu;
  if (threadIdx.x == 0) { misc[0] = 0; misc[1] = a.tglobal[q]; misc[3] = 0; s_amb = 0; }   // class A: the bound (lowered by a rescan, if any)
  __syncthreads();
  CandBuf b{ckey, cpos, &misc[0], &misc[1]};
  // rows with exact keys already: class B's pool (exact pair kernel) or the rescan of overflowed segments
  if (!class_a || a.qovf[q]) {
    const int n = min((int)a.pool_cnt[q], a.pool_cap);
    const uint32_t *pk = a.pool_key + (int64_t)q * a.pool_cap, *pp = a.pool_pos + (int64_t)q * a.pool_cap;
    for (int base = 0; base < n; base += BS) {
      const bool need_tighten = (int)misc[0] > CAP - BS;   // read, barrier, decide
      __syncthreads();
      if (need_tighten) tighten_bs<BS, CAP>(b, o.keff, sorted, &misc[2]);
      const uint32_t T = misc[1];
      const int i = base + threadIdx.x;
      if (i < n) {
        const uint32_t kk = pk[i];
        if (kk <= T) {
          const uint32_t slot = atomicAdd(&misc[0], 1u);
          if (slot < (uint32_t)CAP) { ckey[slot] = kk; cpos[slot] = pp[i]; } else misc[3] = 1u;
        }
      }
      __syncthreads();
    }
  }
  if (class_a) {
    const int dpad = (a.d + 3) & ~3;
    float *r = reinterpret_cast<float *>(smem);   // [QM_G][dpad]
    const float *qv = a.q + (int64_t)q * a.d;
    // ---- pass A: histogram of the survivors' sums -> cut   (cut_shift < 0: LANCE_HIP_NO_QCUT, every survivor is re-evaluated)
    for (int i = threadIdx.x; i < 512; i += BS) s_hist[i] = 0u;
    if (threadIdx.x == 0) s_cut = 0xFFFFFFFFu;
    for (int g0 = 0; a.cut_shift >= 0 && g0 < a.nprobes; g0 += QM_G) {
      const int ng = min(QM_G, a.nprobes - g0);
      __syncthreads();
      if ((int)threadIdx.x < ng) {
        const uint32_t c = a.seg_cnt[(int64_t)q * a.nprobes + g0 + threadIdx.x];
        s_cnt[threadIdx.x] = c > (uint32_t)Q_CAP ? 0u : c;
        if (a.seg_val) s_yz[threadIdx.x] = a.seg_scale[(int64_t)q * a.nprobes + g0 + threadIdx.x];
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int i = 0; i < QM_G; ++i) { s_pre[i] = run; run += i < ng ? s_cnt[i] : 0u; }
        s_pre[QM_G] = run;
      }
      __syncthreads();
      const int total = (int)s_pre[QM_G];
      for (int t = threadIdx.x; t < total; t += BS) {
        int rr = 0, st = 0;
#pragma unroll
        for (int i = 1; i < QM_G; ++i) {
          const int pi = (int)s_pre[i];
          if (t >= pi) { rr = i; st = pi; }
        }
        const int64_t e = ((int64_t)q * a.nprobes + g0 + rr) * Q_CAP + (t - st);
        const uint32_t sv = a.seg_val ? (uint32_t)__builtin_amdgcn_fmed3f(rintf(__builtin_fmaf(__uint_as_float(a.seg_pv[e].y), s_yz[rr].x, s_yz[rr].y)), 0.0f, 65535.0f)
                                      : (uint32_t)a.seg_sum[e];
        atomicAdd(&s_hist[min(511u, sv >> a.cut_shift)], 1u);
      }
    }
    __syncthreads();
    if (a.cut_shift >= 0 && threadIdx.x < 64) {   // first bin where the cumulative count reaches keff
      const int lane = threadIdx.x;
      uint32_t loc[8], tot = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) { loc[i] = s_hist[lane * 8 + i]; tot += loc[i]; }
      uint32_t incl = tot;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t tt = __shfl_up(incl, off, 64);
        if (lane >= off) incl += tt;
      }
      uint32_t run = incl - tot;
      int found = -1;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        run += loc[i];
        if (found < 0 && run >= (uint32_t)o.keff) found = lane * 8 + i;
      }
      const uint64_t mask = __ballot(found >= 0);
      if (mask) {
        const int leader = __ffsll((long long)mask) - 1;
        const int bin = __shfl(found, leader, 64);
        // bin 511 also holds the sums beyond the histogram's range: no upper edge there -> no cut
        if (lane == 0 && bin < 511) s_cut = (((uint32_t)bin + 1u) << a.cut_shift) - 1u + a.cut_slack + (a.qslack ? 2u * a.qslack[q] : 0u);
      }
    }
    __syncthreads();
    QM_DBG_RETURN(a, 1);
    const uint32_t cut = s_cut;
    // ---- pass B: exact re-evaluation of the survivors under the cut, probe group by probe group
    for (int g0 = 0; g0 < a.nprobes; g0 += a.qm_g) {
      const int ng = min(a.qm_g, a.nprobes - g0);
      __syncthreads();
      if ((int)threadIdx.x < ng) {
        const uint32_t c = a.seg_cnt[(int64_t)q * a.nprobes + g0 + threadIdx.x];
        s_cnt[threadIdx.x] = c > (uint32_t)Q_CAP ? 0u : c;    // an overflowed segment comes through the pool (rescan kernel)
        if (a.seg_val) s_yz[threadIdx.x] = a.seg_scale[(int64_t)q * a.nprobes + g0 + threadIdx.x];
      }
      if constexpr (DOT) {      // no residual (pq/distance.rs:60-92): the query itself, staged once for every probe
        for (int t = threadIdx.x; t < a.d; t += BS) r[t] = qv[t];
      } else if (a.vec4) {
        // long rows (C3: 5 probes x 1536 elements per group): 16-byte loads, four independent ones in flight per lane -- the
        // element-at-a-time loop below was a chain of 60 dependent L2 round trips per group at d = 1536
        const int d4 = a.d >> 2, tot4 = ng * d4;
#pragma unroll 4
        for (int t = threadIdx.x; t < tot4; t += BS) {
          const int rr = t / d4, e4 = t - rr * d4;
          const uint32_t part = a.probes[(int64_t)q * a.nprobes + g0 + rr];
          const f4 qq = reinterpret_cast<const f4 *>(qv)[e4];
          const f4 cc = reinterpret_cast<const f4 *>(a.centroids + (int64_t)part * a.d)[e4];
          f4 v = qq - cc;      // element-wise IEEE subtraction: the same values as the scalar loop
          if (a.round_f16) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __half2float(__float2half_rn(v[e]));
          }
          *reinterpret_cast<f4 *>(&r[rr * dpad + 4 * e4]) = v;
        }
      } else {
        for (int t = threadIdx.x; t < ng * a.d; t += BS) {
          const int rr = t / a.d, e = t - rr * a.d;
          const uint32_t part = a.probes[(int64_t)q * a.nprobes + g0 + rr];
          float v = qv[e] - a.centroids[(int64_t)part * a.d + e];
          if (a.round_f16) v = __half2float(__float2half_rn(v));
          r[rr * dpad + e] = v;
        }
      }
      __syncthreads();
      QM_DBG_RETURN(a, 2);
      // prefix of the segment sizes, kept in LDS (17 registers less per lane: occupancy is what this kernel lives on)
      if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int i = 0; i < QM_G; ++i) { s_pre[i] = run; run += i < ng ? s_cnt[i] : 0u; }
        s_pre[QM_G] = run;
      }
      __syncthreads();
      const int total = (int)s_pre[QM_G];
      for (int chunk0 = 0; chunk0 < total; chunk0 += QM_LC) {
        __syncthreads();
        if (threadIdx.x == 0) l_cnt = 0u;
        __syncthreads();
        const int cend = min(total, chunk0 + QM_LC);
        for (int t = chunk0 + threadIdx.x; t < cend; t += BS) {
          int rr = 0, st = 0;
#pragma unroll
          for (int i = 1; i < QM_G; ++i) {
            const int pi = (int)s_pre[i];
            if (t >= pi) { rr = i; st = pi; }   // s_pre[] is non-decreasing and t < s_pre[QM_G]: the last hit is the segment
          }
          const int64_t e = ((int64_t)q * a.nprobes + g0 + rr) * Q_CAP + (t - st);
          uint2 pv = make_uint2(0u, 0u);
          if (a.seg_val) pv = a.seg_pv[e];
          const uint32_t sv = a.seg_val ? (uint32_t)__builtin_amdgcn_fmed3f(rintf(__builtin_fmaf(__uint_as_float(pv.y), s_yz[rr].x, s_yz[rr].y)), 0.0f, 65535.0f)
                                        : (uint32_t)a.seg_sum[e];
          if (sv <= cut) {
            const uint32_t slot = atomicAdd(&l_cnt, 1u);
            l_pos[slot] = a.seg_val ? pv.x : a.seg_pos[e]; l_rr[slot] = (uint8_t)rr;
          }
        }
        __syncthreads();
        QM_DBG_RETURN(a, 3);
        const int nl = (int)l_cnt;
        for (int base = 0; base < nl; base += BS) {
          const bool need_tighten = (int)misc[0] > CAP - BS;   // read, barrier, decide
          __syncthreads();
          if (need_tighten) tighten_bs<BS, CAP>(b, o.keff, sorted, &misc[2]);
          const uint32_t T = misc[1];
          const int t = base + threadIdx.x;
          if (t < nl) {
            const uint32_t pos = l_pos[t];
            const int rr = (int)l_rr[t];
            const uint8_t *rc = a.codes + (int64_t)pos * M;
            const float *rres = r + (DOT ? 0 : rr * dpad);
            float dist = 0.0f;   // pq/distance.rs:128-141: += table[code] for m = 0..M-1 -- the table entry is recomputed here
#pragma unroll
            for (int w = 0; w < MU; ++w) {
              const uint4 cw = *reinterpret_cast<const uint4 *>(rc + w * 16);
              const uint32_t cws[4] = {cw.x, cw.y, cw.z, cw.w};
#pragma unroll
              for (int hh = 0; hh < 16 / Q_MPF; ++hh) {
                // the codebook entries of Q_MPF sub-quantisers are requested together (one L2 round trip for all of them)
                f4 cbv[Q_MPF][QV];
#pragma unroll
                for (int t8 = 0; t8 < Q_MPF; ++t8) {
                  const int mi = hh * Q_MPF + t8, mm = w * 16 + mi;
                  const uint32_t code = (cws[mi >> 2] >> (8 * (mi & 3))) & 255u;
                  const f4 *src = reinterpret_cast<const f4 *>(a.codebook + ((int64_t)mm * 256 + code) * SD);
#pragma unroll
                  for (int u = 0; u < QV; ++u) cbv[t8][u] = src[u];
                }
#pragma unroll
                for (int t8 = 0; t8 < Q_MPF; ++t8) {
                  const int mm = w * 16 + hh * Q_MPF + t8;
                  RegVec<SD> av;
#pragma unroll
                  for (int u = 0; u < QV; ++u) av.q[u] = *reinterpret_cast<const f4 *>(&rres[mm * SD + 4 * u]);
                  dist += finish_metric<QMET>(dist_exact<SD, QMET>(av, reinterpret_cast<const float *>(&cbv[t8][0])));
                }
              }
            }
            if constexpr (DOT) dist = dist - ((float)M - 1.0f);      // pq/storage.rs:949-957
            const uint32_t kk = order_key(dist);
            if (kk <= T) {
              const uint32_t slot = atomicAdd(&misc[0], 1u);
              if (slot < (uint32_t)CAP) { ckey[slot] = kk; cpos[slot] = pos; } else misc[3] = 1u;   // an entry was lost (ties at the bound)
            }
          }
          __syncthreads();
        }
      }
    }
    __syncthreads();
  }
  QM_DBG_RETURN(a, 4);
  for (int iter = 0; iter < 8 && (int)misc[0] > SCAN_LCAP; ++iter) tighten_bs<BS, CAP>(b, o.keff, sorted, &misc[2]);
  __syncthreads();   // also: every lane is done with the staged residuals, the region is reused below
  int c = min((int)misc[0], CAP);
  if (c > SCAN_LCAP || misc[3]) {
    if (threadIdx.x == 0) atomicOr(&o.flags[q], FLAG_OVERFLOW);
    c = min(c, SCAN_LCAP);
  }
  for (int i = threadIdx.x; i < SCAN_LCAP; i += BS) {
    if (i < c) { skey[i] = ckey[i]; spos[i] = cpos[i]; rid[i] = a.row_ids[cpos[i]]; }
    else { skey[i] = 0xFFFFFFFFu; spos[i] = 0; rid[i] = ~0ull; }
  }
  __syncthreads();
  int Pq = 64;
  while (Pq < c) Pq <<= 1;
  bitonic_sort_kr<BS>(skey, rid, spos, Pq);
  QM_DBG_RETURN(a, 5);
  select_and_emit<BS>(o, q, skey, rid, spos, c, &s_amb);
}

// ---- merge, one residual group (nprobes <= qm_g): fewer dependent memory round trips -------------------------------------------------
// The kernel above is bound by the number of queries in flight x the length of a query's DEPENDENT chain of memory round trips
// (gpurun r04zb: 0.166 ms for 10,000 queries at C2, ~40 us per workgroup; the arithmetic is a few thousand lane-operations).  Its chain:
// flags / bounds -> segment counts -> survivors' sums (histogram) -> [cut] -> counts again -> probes -> centroids (staging) -> sums AND
// positions again (compaction) -> codes -> 4 x codebook -> row ids: thirteen trips.  Here everything that does not depend on the cut is
// requested TOGETHER after the one trip that yields the segment counts and the probed partitions: each lane takes its survivors' (value,
// position) pairs into registers -- four per lane cover 512 survivors, a query has ~250 -- and its share of the residual staging loads
// right behind them; the histogram, the cut and the compaction then run out of registers and LDS.  Chain: counts / probes -> survivors +
// residual rows -> codes -> 4 x codebook -> row ids: eight trips.  More than 512 survivors: further chunks are re-read (rare).
// Same arithmetic, same selection machinery, same outputs as the kernel above (which keeps the batches with more probes than one
// staging holds).
template <int SD, int MU, int BS, bool DOT = false>
__global__ LH_QM_BOUNDS(BS) void ivfpq_qmerge1g_kernel(QmergeArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int M = MU * 16;
  constexpr int QV = SD / 4;
  constexpr int QMET = DOT ? METRIC_DOT : METRIC_L2;
  constexpr int CAP = BS == 128 ? 512 : 1024;   // (key, pos) entries under selection
  static_assert(BS >= SCAN_MAX_KEFF, "tighten_bs selects among one value per lane");
  __shared__ uint32_t ckey[CAP], cpos[CAP], sorted[BS], misc[8];
  uint64_t *rid = reinterpret_cast<uint64_t *>(smem);                    // [SCAN_LCAP]  (the dynamic region holds the staged residuals first)
  uint32_t *skey = reinterpret_cast<uint32_t *>(rid + SCAN_LCAP);        // [SCAN_LCAP]
  uint32_t *spos = skey + SCAN_LCAP;                                     // [SCAN_LCAP]
  constexpr int QM_LC = 512;                       // survivors per chunk
  constexpr int SPL = QM_LC / BS;                  // survivors per lane per chunk (registers)
  __shared__ uint32_t s_cnt[QM_G], s_part[QM_G];
  __shared__ f2 s_yz[QM_G];
  __shared__ int s_amb;
  __shared__ uint32_t s_hist[512], l_pos[QM_LC], s_cut, l_cnt;
  __shared__ uint8_t l_rr[QM_LC];
  const SelectOut &o = a.o;
  const int q = blockIdx.x;
  const int np = a.nprobes;                        // <= a.qm_g <= QM_G (launcher)
  // ---- trip 1: everything a lane can ask for knowing only the query's number
  const uint32_t qflags = o.flags[q];
  const uint32_t tb = a.tbound[q];
  uint32_t c_l = 0, part_l = 0;
  f2 yz_l = {0.0f, 0.0f};
  if ((int)threadIdx.x < np) {
    c_l = a.seg_cnt[(int64_t)q * np + threadIdx.x];
    part_l = a.probes[(int64_t)q * np + threadIdx.x];
    if (a.seg_val) yz_l = a.seg_scale[(int64_t)q * np + threadIdx.x];
  }
  const uint32_t qov = a.qovf[q];
  uint32_t tg0 = 0;
  if (threadIdx.x == 0) tg0 = a.tglobal[q];
  if (qflags & FLAG_OVERFLOW) return;   // the exact kernel recomputes this query
  const bool class_a = tb != 0xFFFFFFFFu;
  if (threadIdx.x == 0) { misc[0] = 0; misc[1] = tg0; misc[3] = 0; s_amb = 0; s_cut = 0xFFFFFFFFu; l_cnt = 0u; }   // class A: the bound (lowered by a rescan, if any)
  for (int i = threadIdx.x; i < 512; i += BS) s_hist[i] = 0u;
  if ((int)threadIdx.x < QM_G) {
    s_cnt[threadIdx.x] = ((int)threadIdx.x < np && c_l <= (uint32_t)Q_CAP) ? c_l : 0u;    // an overflowed segment comes through the pool (rescan kernel)
    s_part[threadIdx.x] = part_l;
    s_yz[threadIdx.x] = yz_l;
  }
  __syncthreads();
  CandBuf b{ckey, cpos, &misc[0], &misc[1]};
  // rows with exact keys already: class B's pool (exact pair kernel) or the rescan of overflowed segments
  if (!class_a || qov) {
    const int n = min((int)a.pool_cnt[q], a.pool_cap);
    const uint32_t *pk = a.pool_key + (int64_t)q * a.pool_cap, *pp = a.pool_pos + (int64_t)q * a.pool_cap;
    for (int base = 0; base < n; base += BS) {
      const bool need_tighten = (int)misc[0] > CAP - BS;   // read, barrier, decide
      __syncthreads();
      if (need_tighten) tighten_bs<BS, CAP>(b, o.keff, sorted, &misc[2]);
      const uint32_t T = misc[1];
      const int i = base + threadIdx.x;
      if (i < n) {
        const uint32_t kk = pk[i];
        if (kk <= T) {
          const uint32_t slot = atomicAdd(&misc[0], 1u);
          if (slot < (uint32_t)CAP) { ckey[slot] = kk; cpos[slot] = pp[i]; } else misc[3] = 1u;
        }
      }
      __syncthreads();
    }
  }
  if (class_a) {
    const int dpad = (a.d + 3) & ~3;
    float *r = reinterpret_cast<float *>(smem);   // [np][dpad]
    const float *qv = a.q + (int64_t)q * a.d;
    // prefix of the segment sizes: every lane folds the <= 16 counts itself (LDS broadcast reads; no barrier, no serial lane)
    uint32_t total = 0;
    for (int i = 0; i < np; ++i) total += s_cnt[i];
    const int nchunks = ((int)total + QM_LC - 1) / QM_LC;
    uint32_t sv_pos[SPL], sv_sum[SPL];      // this lane's survivors of the chunk in registers: position, integer sum (0xFFFFFFFF: none)
    uint32_t sv_rr[SPL];
    auto load_chunk = [&](int ch) {
#pragma unroll
      for (int j = 0; j < SPL; ++j) {
        const int t = ch * QM_LC + j * BS + (int)threadIdx.x;
        sv_sum[j] = 0xFFFFFFFFu; sv_pos[j] = 0u; sv_rr[j] = 0u;
        if (t < (int)total) {
          int rr = 0, st = 0, run = 0;
          for (int i = 0; i < np; ++i) {      // s_cnt[] >= 0 and t < total: the last segment whose start is <= t
            if (t >= run) { rr = i; st = run; }
            run += (int)s_cnt[i];
          }
          const int64_t e = ((int64_t)q * np + rr) * Q_CAP + (t - st);
          sv_rr[j] = (uint32_t)rr;
          if (a.seg_val) { const uint2 pv = a.seg_pv[e]; sv_pos[j] = pv.x; sv_sum[j] = pv.y; }      // (the value's conversion waits until it is used)
          else { sv_pos[j] = a.seg_pos[e]; sv_sum[j] = (uint32_t)a.seg_sum[e]; }
        }
      }
    };
    // ---- trip 2: the first chunk of survivors and the residual rows, requested together
    bool have[SPL];
    auto mark = [&](int ch) {
#pragma unroll
      for (int j = 0; j < SPL; ++j) have[j] = ch * QM_LC + j * BS + (int)threadIdx.x < (int)total;
    };
    if (nchunks > 0) load_chunk(0);
    mark(0);
    if constexpr (DOT) {      // no residual (pq/distance.rs:60-92): the query itself, staged once for every probe
      for (int t = threadIdx.x; t < a.d; t += BS) r[t] = qv[t];
    } else if (a.vec4) {
      const int d4 = a.d >> 2, tot4 = np * d4;
#pragma unroll 4
      for (int t = threadIdx.x; t < tot4; t += BS) {
        const int rr = t / d4, e4 = t - rr * d4;
        const uint32_t part = s_part[rr];
        const f4 qq = reinterpret_cast<const f4 *>(qv)[e4];
        const f4 cc = reinterpret_cast<const f4 *>(a.centroids + (int64_t)part * a.d)[e4];
        f4 v = qq - cc;      // element-wise IEEE subtraction (v2.rs:316-332)
        if (a.round_f16) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = __half2float(__float2half_rn(v[e]));
        }
        *reinterpret_cast<f4 *>(&r[rr * dpad + 4 * e4]) = v;
      }
    } else {
      for (int t = threadIdx.x; t < np * a.d; t += BS) {
        const int rr = t / a.d, e = t - rr * a.d;
        float v = qv[e] - a.centroids[(int64_t)s_part[rr] * a.d + e];
        if (a.round_f16) v = __half2float(__float2half_rn(v));
        r[rr * dpad + e] = v;
      }
    }
    auto to_sums = [&]() {
      if (a.seg_val) {
#pragma unroll
        for (int j = 0; j < SPL; ++j)
          if (have[j]) {
            const f2 yz = s_yz[sv_rr[j]];
            sv_sum[j] = (uint32_t)__builtin_amdgcn_fmed3f(rintf(__builtin_fmaf(__uint_as_float(sv_sum[j]), yz.x, yz.y)), 0.0f, 65535.0f);
          }
      }
    };
    // ---- pass A: histogram of the survivors' sums -> cut   (cut_shift < 0: LANCE_HIP_NO_QCUT, every survivor is re-evaluated)
    if (a.cut_shift >= 0) {
      for (int ch = 0; ch < nchunks; ++ch) {
        if (ch > 0) { load_chunk(ch); mark(ch); }
        to_sums();
#pragma unroll
        for (int j = 0; j < SPL; ++j)
          if (have[j]) atomicAdd(&s_hist[min(511u, sv_sum[j] >> a.cut_shift)], 1u);
      }
    } else if (nchunks > 0) {
      to_sums();
    }
    __syncthreads();      // histogram complete; residuals staged
    if (a.cut_shift >= 0 && threadIdx.x < 64) {   // first bin where the cumulative count reaches keff
      const int lane = threadIdx.x;
      uint32_t loc[8], tot = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) { loc[i] = s_hist[lane * 8 + i]; tot += loc[i]; }
      uint32_t incl = tot;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t tt = __shfl_up(incl, off, 64);
        if (lane >= off) incl += tt;
      }
      uint32_t run = incl - tot;
      int found = -1;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        run += loc[i];
        if (found < 0 && run >= (uint32_t)o.keff) found = lane * 8 + i;
      }
      const uint64_t mask = __ballot(found >= 0);
      if (mask) {
        const int leader = __ffsll((long long)mask) - 1;
        const int bin = __shfl(found, leader, 64);
        // bin 511 also holds the sums beyond the histogram's range: no upper edge there -> no cut
        if (lane == 0 && bin < 511) s_cut = (((uint32_t)bin + 1u) << a.cut_shift) - 1u + a.cut_slack + (a.qslack ? 2u * a.qslack[q] : 0u);
      }
    }
    __syncthreads();
    QM_DBG_RETURN(a, 1);
    const uint32_t cut = s_cut;
    // ---- pass B: compaction under the cut (registers -> LDS), exact re-evaluation
    for (int ch = 0; ch < nchunks; ++ch) {
      if (nchunks > 1) {      // (one chunk: the registers still hold it)
        __syncthreads();
        if (threadIdx.x == 0) l_cnt = 0u;
        __syncthreads();
        load_chunk(ch); mark(ch); to_sums();
      }
#pragma unroll
      for (int j = 0; j < SPL; ++j)
        if (have[j] && sv_sum[j] <= cut) {
          const uint32_t slot = atomicAdd(&l_cnt, 1u);
          l_pos[slot] = sv_pos[j]; l_rr[slot] = (uint8_t)sv_rr[j];
        }
      __syncthreads();
      QM_DBG_RETURN(a, 3);
      const int nl = (int)l_cnt;
      for (int base = 0; base < nl; base += BS) {
        const bool need_tighten = (int)misc[0] > CAP - BS;   // read, barrier, decide
        __syncthreads();
        if (need_tighten) tighten_bs<BS, CAP>(b, o.keff, sorted, &misc[2]);
        const uint32_t T = misc[1];
        const int t = base + threadIdx.x;
        if (t < nl) {
          const uint32_t pos = l_pos[t];
          const int rr = (int)l_rr[t];
          const uint8_t *rc = a.codes + (int64_t)pos * M;
          const float *rres = r + (DOT ? 0 : rr * dpad);
          float dist = 0.0f;   // pq/distance.rs:128-141: += table[code] for m = 0..M-1 -- the table entry is recomputed here
#pragma unroll
          for (int w = 0; w < MU; ++w) {
            const uint4 cw = *reinterpret_cast<const uint4 *>(rc + w * 16);
            const uint32_t cws[4] = {cw.x, cw.y, cw.z, cw.w};
#pragma unroll
            for (int hh = 0; hh < 16 / Q_MPF; ++hh) {
              // the codebook entries of Q_MPF sub-quantisers are requested together (one L2 round trip for all of them)
              f4 cbv[Q_MPF][QV];
#pragma unroll
              for (int t8 = 0; t8 < Q_MPF; ++t8) {
                const int mi = hh * Q_MPF + t8, mm = w * 16 + mi;
                const uint32_t code = (cws[mi >> 2] >> (8 * (mi & 3))) & 255u;
                const f4 *src = reinterpret_cast<const f4 *>(a.codebook + ((int64_t)mm * 256 + code) * SD);
#pragma unroll
                for (int u = 0; u < QV; ++u) cbv[t8][u] = src[u];
              }
#pragma unroll
              for (int t8 = 0; t8 < Q_MPF; ++t8) {
                const int mm = w * 16 + hh * Q_MPF + t8;
                RegVec<SD> av;
#pragma unroll
                for (int u = 0; u < QV; ++u) av.q[u] = *reinterpret_cast<const f4 *>(&rres[mm * SD + 4 * u]);
                dist += finish_metric<QMET>(dist_exact<SD, QMET>(av, reinterpret_cast<const float *>(&cbv[t8][0])));
              }
            }
          }
          if constexpr (DOT) dist = dist - ((float)M - 1.0f);      // pq/storage.rs:949-957
          const uint32_t kk = order_key(dist);
          if (kk <= T) {
            const uint32_t slot = atomicAdd(&misc[0], 1u);
            if (slot < (uint32_t)CAP) { ckey[slot] = kk; cpos[slot] = pos; } else misc[3] = 1u;   // an entry was lost (ties at the bound)
          }
        }
        __syncthreads();
      }
    }
    __syncthreads();
  }
  QM_DBG_RETURN(a, 4);
  if (o.refine) {
    // Refine follows: it ranks the candidates by their exact distances, so all it needs from here is the SET of the keff best by
    // (key, row id), in any order.  Once at most one entry per lane is left, tighten_bs returns the exact keff-th smallest key and keeps
    // key <= it: if exactly keff entries remain (no tie across the boundary) they ARE that set -- no (key, rowid) sort of up to 256
    // entries through LDS (36 barrier-separated passes: 0.034 of this kernel's 0.117 ms at C2, gpurun r04zb), no tie check.  A tie across
    // the boundary, a lost entry, or more entries than lanes after the tightening rounds take the sort below as before.
    for (int iter = 0; iter < 8; ++iter) {
      __syncthreads();
      const bool more = (int)misc[0] > BS;      // read, barrier (inside tighten_bs), decide
      if (!more) break;      // uniform
      tighten_bs<BS, CAP>(b, o.keff, sorted, &misc[2]);
    }
    __syncthreads();
    int c1 = (int)misc[0];
    const bool lost = misc[3] != 0u;
    __syncthreads();
    if (!lost && c1 <= BS) {
      if (c1 >= o.keff) {
        tighten_bs<BS, CAP>(b, o.keff, sorted, &misc[2]);      // c1 <= BS: exact threshold, entries with key <= it kept (compacted)
        c1 = (int)misc[0];
      }
      if (c1 <= o.keff) {      // uniform
        for (int i = threadIdx.x; i < o.keff; i += BS) o.cand_rid[(int64_t)q * o.keff + i] = i < c1 ? a.row_ids[cpos[i]] : ~0ull;
        if (threadIdx.x == 0) o.cand_cnt[q] = (uint32_t)c1;
        return;
      }
    }
  }
  for (int iter = 0; iter < 8 && (int)misc[0] > SCAN_LCAP; ++iter) tighten_bs<BS, CAP>(b, o.keff, sorted, &misc[2]);
  __syncthreads();   // also: every lane is done with the staged residuals, the region is reused below
  int c = min((int)misc[0], CAP);
  if (c > SCAN_LCAP || misc[3]) {
    if (threadIdx.x == 0) atomicOr(&o.flags[q], FLAG_OVERFLOW);
    c = min(c, SCAN_LCAP);
  }
  for (int i = threadIdx.x; i < SCAN_LCAP; i += BS) {
    if (i < c) { skey[i] = ckey[i]; spos[i] = cpos[i]; rid[i] = a.row_ids[cpos[i]]; }
    else { skey[i] = 0xFFFFFFFFu; spos[i] = 0; rid[i] = ~0ull; }
  }
  __syncthreads();
  int Pq = 64;
  while (Pq < c) Pq <<= 1;
  bitonic_sort_kr<BS>(skey, rid, spos, Pq);
  QM_DBG_RETURN(a, 5);
  select_and_emit<BS>(o, q, skey, rid, spos, c, &s_amb);
}

// ---- host -------------------------------------------------------------------------------------------------------
bool qscan_supported(const lance_hip_index *ix, uint32_t nq, uint32_t nprobes) {
  static const bool off = getenv("LANCE_HIP_NO_QSCAN") != nullptr;
  if (off || !ix->model_finite) return false;      // NaN / infinite centroids or codewords: the exact kernels decide (index.h)
  const int scan_metric = ix->metric == LANCE_HIP_COSINE ? LANCE_HIP_L2 : ix->metric;
  // the integer tables need entries >= 0 (squared L2); the dot metric has a quantised flow where the matrix-core scan and its bound pass
  // serve the batch (search_ms.hip: limits and sums relative to a per-query base distance) -- every other dot batch keeps the exact pair scan
  if (scan_metric == LANCE_HIP_DOT) { if (!mscan_dot_ready(ix, nq, nprobes)) return false; }
  else if (scan_metric != LANCE_HIP_L2) return false;
  {
    const int m = (int)ix->m, sd = (int)(ix->d / ix->m);
    if (!((m == 16 || m == 32) && (sd == 4 || sd == 8 || sd == 16)) && !qscan_tiled_shape(m, sd)) return false;
  }
  if ((uint64_t)nq * nprobes * Q_CAP * 4 > (2ull << 30)) return false;   // survivor segments: at most 2 GiB of scratch
  // item residuals likewise -- where the main pass reads them: neither the per-query-table filter (search_qt.hip) nor the matrix-core
  // scan (search_ms.hip) does.  (This check used to apply to both: a 10,000-query batch at nprobes = 50 on the C3 shape fell back to the
  // query-major kernel, 62 ms instead of ~6: gpurun r04v.)
  const bool needs_rq = !qscan_pt_enabled(ix) && !mscan_batch_shape(ix, nq, nprobes);
  if (needs_rq && ((uint64_t)nq * nprobes / Q_G + ix->nlist + 1) * ix->d * 16 > (2ull << 30)) return false;
  return true;
}

size_t qscan_lds_bytes(int d, int m) {   // dynamic part (the quantised LUT is static LDS)
  (void)d; (void)m;
  return (size_t)4 * Q_CAP * 4 + 8 * 4 + (size_t)4 * Q_CAP * 2;   // positions, counters + scales, sums
}

int qscan_group(lance_hip_ctx *ctx, const uint32_t *probes, uint32_t nq, uint32_t nprobes, int nlist, const uint32_t *tglobal,
                uint32_t *keys, uint32_t *tbound, uint32_t *pair_starts, uint32_t *pair_idx, uint32_t *item_start4, int4 *desc4,
                uint32_t max_items4, int G, int dot) {
  const size_t npairs = (size_t)nq * nprobes;
  hipLaunchKernelGGL(q_tclass_keys_kernel, dim3((unsigned)cdiv(npairs, 256)), dim3(256), 0, ctx->stream, probes, (int64_t)npairs, (int)nprobes,
                     nlist, tglobal, keys, tbound, dot);
  LH_TRY(stable_group(ctx, keys, (int64_t)npairs, (int64_t)npairs, 2 * nlist, 1, pair_starts, pair_idx, (int64_t)npairs, nullptr));
  hipLaunchKernelGGL(q_item_table_kernel, dim3(1), dim3(256), 0, ctx->stream, pair_starts, nlist, G, item_start4);
  hipLaunchKernelGGL(q_item_desc_kernel, dim3((unsigned)cdiv(max_items4, 256)), dim3(256), 0, ctx->stream, item_start4, pair_starts, nlist, G,
                     max_items4, desc4);
  return LANCE_HIP_OK;
}

// item_start / desc of the (partition, G queries) items of a grouping (used by the bound passes)
int qscan_item_tables(lance_hip_ctx *ctx, const uint32_t *pair_starts, int nlist, int G, uint32_t *item_start, int4 *desc, uint32_t max_items) {
  hipLaunchKernelGGL(q_item_table_kernel, dim3(1), dim3(256), 0, ctx->stream, pair_starts, nlist, G, item_start);
  hipLaunchKernelGGL(q_item_desc_kernel, dim3((unsigned)cdiv(max_items, 256)), dim3(256), 0, ctx->stream, item_start, pair_starts, nlist, G, max_items, desc);
  return LANCE_HIP_OK;
}

int qscan_nearest_keys(lance_hip_ctx *ctx, const uint32_t *probes, uint32_t nq, uint32_t nprobes, uint32_t *keys, uint32_t nb) {
  hipLaunchKernelGGL(q_nearest_keys_kernel, dim3((unsigned)cdiv((uint64_t)nq * nb, 256)), dim3(256), 0, ctx->stream, probes, (int)nq, (int)nprobes, (int)nb, keys);
  return LANCE_HIP_OK;
}

template <int SD>
static bool launch_qscan_sd(lance_hip_ctx *ctx, const QscanArgs &a, int m, unsigned grid, size_t lds) {
  if (m == 16) { hipLaunchKernelGGL((ivfpq_qscan_kernel<SD, 1>), dim3(grid), dim3(Q_BS), lds, ctx->stream, a); return true; }
  if (m == 32) { hipLaunchKernelGGL((ivfpq_qscan_kernel<SD, 2>), dim3(grid), dim3(Q_BS), lds, ctx->stream, a); return true; }
  return false;
}

int qscan_launch(lance_hip_ctx *ctx, const lance_hip_index *ix, const float *qs, uint32_t nq, uint32_t nprobes, const uint32_t *pair_idx,
                 const uint32_t *item_start4, const int4 *desc4, uint32_t max_items4, const uint32_t *tbound, uint32_t *seg_cnt,
                 uint32_t *seg_pos, uint32_t *qovf, const uint32_t *allow, const uint32_t *probes) {
  const int d = (int)ix->d, m = (int)ix->m, sd = d / m;
  const bool pt = probes != nullptr && qscan_pt_enabled(ix);   // per-query tables: no residual pre-pass, no table build in the scan
  f4 *rq = pt ? nullptr : reinterpret_cast<f4 *>(ctx->scratch_t<float>("qscan.rq", (size_t)max_items4 * d * 4));
  if (!pt && !rq) return LANCE_HIP_ENOMEM;
  {
    ScopedTimer t(ctx, "q_residual");
    if (!pt)
      hipLaunchKernelGGL(q_residual_kernel, dim3((unsigned)cdiv(max_items4, 4)), dim3(256), 0, ctx->stream, qs, pair_idx, item_start4, desc4,
                         ix->centroids, d, (int)ix->nlist, (int)nprobes, ix->dtype == LANCE_HIP_F16 ? 1 : 0, rq);
    LH_CHECK_HIP(lh::memset_async(seg_cnt, 0, (size_t)nq * nprobes * 4, ctx->stream));
    LH_CHECK_HIP(lh::memset_async(qovf, 0, (size_t)nq * 4, ctx->stream));
  }
  ScopedTimer t(ctx, "ivfpq_scan_c1");
  QscanArgs a;
  a.rq = rq; a.pair_idx = pair_idx; a.item_start = item_start4; a.desc = desc4;
  a.centroids = ix->centroids; a.codebook = ix->codebook; a.part_offsets = ix->part_offsets; a.codes = ix->codes;
  a.d = d; a.nprobes = (int)nprobes; a.nlist = (int)ix->nlist; a.round_f16 = ix->dtype == LANCE_HIP_F16 ? 1 : 0;
  a.tbound = tbound; a.seg_cnt = seg_cnt; a.seg_pos = seg_pos; a.qovf = qovf; a.allow = allow;
  a.seg_sum = ctx->scratch_t<uint16_t>("q.seg_sum", (size_t)nq * nprobes * Q_CAP);   // the merge launcher asks for the same slot
  a.ovf = ctx->scratch_t<uint32_t>("q.ovf", (size_t)nq * nprobes + 1);                // likewise (and the class-B conversion)
  if (!a.seg_sum || !a.ovf) return LANCE_HIP_ENOMEM;
  LH_CHECK_HIP(lh::memset_async(a.ovf, 0, 4, ctx->stream));
  const size_t lds = qscan_lds_bytes(d, m);
  const unsigned grid = max_items4;   // one workgroup per item (persistent workgroups looping over items measured no faster)
  bool ok = false;
#ifdef LH_QT_PROF
  static const bool qt_prof = getenv("LANCE_HIP_QT_PROF") != nullptr;
  if (qt_prof && qscan_tiled_shape(m, sd)) {
    a.prof = ctx->scratch_t<unsigned long long>("qt.prof", 8);
    if (a.prof) (void)lh::memset_async(a.prof, 0, 64, ctx->stream);
  }
#endif
  if (pt) { LH_TRY(qscan_pt_launch(ctx, ix, a, qs, nq, probes, grid)); ok = true; }
  else if (qscan_tiled_shape(m, sd)) ok = qscan_tiled_launch(ctx, a, m, sd, grid);
  else if (sd == 4) ok = launch_qscan_sd<4>(ctx, a, m, grid, lds);
  else if (sd == 8) ok = launch_qscan_sd<8>(ctx, a, m, grid, lds);
  else if (sd == 16) ok = launch_qscan_sd<16>(ctx, a, m, grid, lds);
  LH_REQUIRE(ok, "quantised scan: unsupported shape (m=%d, sd=%d)", m, sd);
#ifdef LH_QT_PROF
  if (a.prof) {
    unsigned long long h[8];
    (void)hipMemcpyAsync(h, a.prof, 64, hipMemcpyDeviceToHost, ctx->stream);
    (void)hipStreamSynchronize(ctx->stream);
    if (h[3]) fprintf(stderr, "[qt prof] items=%llu  100 MHz clocks per item: table build %.0f | row scan %.0f\n", h[3], (double)h[0] / h[3], (double)h[1] / h[3]);
  }
#endif
  return LANCE_HIP_OK;
}

template <int SD>
static bool launch_qbound_sd(lance_hip_ctx *ctx, const QboundArgs &a, int m, unsigned grid, size_t lds) {
  if (m == 16) { hipLaunchKernelGGL((ivfpq_qbound_kernel<SD, 1>), dim3(grid), dim3(Q_BS), lds, ctx->stream, a); return true; }
  if (m == 32) { hipLaunchKernelGGL((ivfpq_qbound_kernel<SD, 2>), dim3(grid), dim3(Q_BS), lds, ctx->stream, a); return true; }
  return false;
}

// lance_hip_index::cb_mean -- one workgroup per sub-quantiser, one lane per codeword
__global__ __launch_bounds__(256) void q_codebook_mean_kernel(const float *__restrict__ codebook, int sd, int d, float *__restrict__ out) {
  __shared__ float part[4];
  const int mm = blockIdx.x, c = threadIdx.x, lane = c & 63, wave = c >> 6;
  const float *cw = codebook + ((int64_t)mm * 256 + c) * sd;
  float sq = 0.0f;
  for (int e = 0; e <= sd; ++e) {
    float v;
    if (e < sd) { v = cw[e]; sq += v * v; } else v = sq;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if (lane == 0) part[wave] = v;
    __syncthreads();
    if (c == 0) {
      const float tot = (part[0] + part[1] + part[2] + part[3]) * (1.0f / 256.0f);
      if (e < sd) out[mm * sd + e] = tot; else atomicAdd(&out[d], tot);
    }
  }
}

__global__ __launch_bounds__(256) void q_model_finite_kernel(const float *__restrict__ a, int64_t na, const float *__restrict__ b, int64_t nb,
                                                             uint32_t *__restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  bool bad = false;
  if (i < na) bad = !isfinite(a[i]);
  else if (i < na + nb) bad = !isfinite(b[i - na]);
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

int qscan_index_constants(lance_hip_ctx *ctx, lance_hip_index *ix) {
  const int d = (int)ix->d, m = (int)ix->m;
  if (!ix->cb_mean) LH_CHECK_HIP(hipMalloc(reinterpret_cast<void **>(&ix->cb_mean), (size_t)(d + 2) * 4));
  LH_CHECK_HIP(lh::memset_async(ix->cb_mean, 0, (size_t)(d + 2) * 4, ctx->stream));
  hipLaunchKernelGGL(q_codebook_mean_kernel, dim3((unsigned)m), dim3(256), 0, ctx->stream, ix->codebook, d / m, d, ix->cb_mean);
  const int64_t nc = (int64_t)ix->nlist * d, ncb = (int64_t)m * 256 * (d / m);
  hipLaunchKernelGGL(q_model_finite_kernel, dim3((unsigned)cdiv((uint64_t)(nc + ncb), 256)), dim3(256), 0, ctx->stream, ix->centroids, nc, ix->codebook,
                     ncb, reinterpret_cast<uint32_t *>(ix->cb_mean + d + 1));
  LH_CHECK_HIP(hipGetLastError());
  return LANCE_HIP_OK;
}

// item_start[vp] = exclusive scan of ceil(pairs of vp / G), desc[item] = {virtual partition, first grouped pair, pairs (1..G), 0}
int qscan_items(lance_hip_ctx *ctx, const uint32_t *pair_starts, int nvp, int G, uint32_t *item_start, int4 *desc, uint32_t max_items) {
  hipLaunchKernelGGL(q_item_table_kernel, dim3(1), dim3(256), 0, ctx->stream, pair_starts, nvp, G, item_start);
  hipLaunchKernelGGL(q_item_desc_kernel, dim3((unsigned)cdiv(max_items, 256)), dim3(256), 0, ctx->stream, item_start, pair_starts, nvp, G, max_items, desc);
  LH_CHECK_HIP(hipGetLastError());
  return LANCE_HIP_OK;
}

// bound pass on the integer table: pair_starts0 / pair_idx0 = the nq (query, nearest partition) pairs grouped by partition
int qbound_launch(lance_hip_ctx *ctx, const lance_hip_index *ix, const float *qs, uint32_t nq, uint32_t keff, const uint32_t *pair_starts0,
                  const uint32_t *pair_idx0, uint32_t *item_start, int4 *desc, uint32_t max_items, uint32_t *tglobal, const uint32_t *allow) {
  const int d = (int)ix->d, m = (int)ix->m, sd = d / m, nlist = (int)ix->nlist;
  hipLaunchKernelGGL(q_item_table_kernel, dim3(1), dim3(256), 0, ctx->stream, pair_starts0, nlist, Q_G, item_start);
  hipLaunchKernelGGL(q_item_desc_kernel, dim3((unsigned)cdiv(max_items, 256)), dim3(256), 0, ctx->stream, item_start, pair_starts0, nlist, Q_G,
                     max_items, desc);
  f4 *rq = reinterpret_cast<f4 *>(ctx->scratch_t<float>("qbound.rq", (size_t)max_items * d * 4));
  if (!rq) return LANCE_HIP_ENOMEM;
  hipLaunchKernelGGL(q_residual_kernel, dim3((unsigned)cdiv(max_items, 4)), dim3(256), 0, ctx->stream, qs, pair_idx0, item_start, desc, ix->centroids, d,
                     nlist, 1, ix->dtype == LANCE_HIP_F16 ? 1 : 0, rq);
  QboundArgs a;
  a.rq = rq; a.pair_idx = pair_idx0; a.item_start = item_start; a.desc = desc;
  a.centroids = ix->centroids; a.codebook = ix->codebook; a.part_offsets = ix->part_offsets; a.codes = ix->codes;
  a.d = d; a.nlist = nlist; a.keff = (int)keff; a.round_f16 = ix->dtype == LANCE_HIP_F16 ? 1 : 0;
  a.tglobal = tglobal; a.allow = allow;
  LH_REQUIRE(ix->cb_mean, "integer bound pass: the index carries no codebook means (8-bit PQ only)");
  a.cb_mean = ix->cb_mean;
  const size_t lds = (size_t)4 * QB_BINS * 4 + 8 * 4;
  bool ok = false;
  if (qscan_tiled_shape(m, sd)) ok = qbound_tiled_launch(ctx, a, m, sd, max_items);
  else if (sd == 4) ok = launch_qbound_sd<4>(ctx, a, m, max_items, lds);
  else if (sd == 8) ok = launch_qbound_sd<8>(ctx, a, m, max_items, lds);
  else if (sd == 16) ok = launch_qbound_sd<16>(ctx, a, m, max_items, lds);
  LH_REQUIRE(ok, "integer bound pass: unsupported shape (m=%d, sd=%d)", m, sd);
  (void)nq;
  return LANCE_HIP_OK;
}

template <int SD, int MU, bool DOT = false>
static void launch_qmerge_mu(lance_hip_ctx *ctx, const QmergeArgs &a, unsigned nq, int bs) {
  const int dpad = (a.d + 3) & ~3;
  const size_t lds_rescan = (size_t)dpad * 4 + (size_t)MU * 16 * 256 * 4;
  // 1024 lanes per overflowed segment for every shape (round 4; M = 16 / 32 had 256: 0.1666 -> 0.1628 ms for rescan + merge at C2 -- the few
  // segments a batch lists, ~5 per 10,000 queries there, are single-workgroup latency in front of the merge)
  const unsigned rgrid = (unsigned)std::min<uint64_t>((uint64_t)nq * a.nprobes, (uint64_t)ctx->num_cus);
  hipLaunchKernelGGL((ivfpq_qrescan_kernel<SD, MU, 1024, DOT>), dim3(rgrid), dim3(1024), lds_rescan, ctx->stream, a);
  // staged residuals of min(QM_G, nprobes) probes; later the (rowid, key, position) sort buffers
  const size_t lds = std::max((size_t)std::min<int>(a.qm_g, a.nprobes) * dpad * 4, (size_t)SCAN_LCAP * 16);
  // one residual group (nprobes <= qm_g <= 16): the kernel with the short dependent chain; LANCE_HIP_QMERGE_V1=1 keeps the general one (A/B)
  static const bool v1 = getenv("LANCE_HIP_QMERGE_V1") != nullptr;
  if (!v1 && a.nprobes <= a.qm_g) {
    if (bs == 256) hipLaunchKernelGGL((ivfpq_qmerge1g_kernel<SD, MU, 256, DOT>), dim3(nq), dim3(256), lds, ctx->stream, a);
    else hipLaunchKernelGGL((ivfpq_qmerge1g_kernel<SD, MU, 128, DOT>), dim3(nq), dim3(128), lds, ctx->stream, a);
    return;
  }
  if (bs == 256) hipLaunchKernelGGL((ivfpq_qmerge_kernel<SD, MU, 256, DOT>), dim3(nq), dim3(256), lds, ctx->stream, a);
  else hipLaunchKernelGGL((ivfpq_qmerge_kernel<SD, MU, 128, DOT>), dim3(nq), dim3(128), lds, ctx->stream, a);
}

int qmerge_launch(lance_hip_ctx *ctx, const lance_hip_index *ix, const float *qs, uint32_t nq, const uint32_t *probes, uint32_t nprobes,
                  const uint32_t *tbound, uint32_t *tglobal, const uint32_t *seg_cnt, const uint32_t *seg_pos, const uint32_t *qovf,
                  uint32_t *pool_key, uint32_t *pool_pos, uint32_t *pool_cnt, int pool_cap, const SelectOut &o, const uint32_t *allow,
                  const uint32_t *qslack, const float *seg_val, const float *seg_scale) {
  const int d = (int)ix->d, m = (int)ix->m, sd = d / m;
  QmergeArgs a;
  a.qslack = qslack; a.seg_val = seg_val; a.seg_pv = reinterpret_cast<const uint2 *>(seg_val); a.seg_scale = reinterpret_cast<const f2 *>(seg_scale);
#ifdef LH_TIMING_EXPERIMENTS
  {
    static const int dbg = [] {
      const int v = getenv("LANCE_HIP_QM_DBG") ? atoi(getenv("LANCE_HIP_QM_DBG")) : 0;
      if (v) fprintf(stderr, "lance_hip: LANCE_HIP_QM_DBG=%d -- timing experiment, search RESULTS ARE WRONG\n", v);
      return v;
    }();
    a.dbg = dbg;
  }
#endif
  a.q = qs; a.probes = probes; a.centroids = ix->centroids; a.codebook = ix->codebook; a.codes = ix->codes; a.row_ids = ix->row_ids;
  a.d = d; a.nprobes = (int)nprobes; a.round_f16 = ix->dtype == LANCE_HIP_F16 ? 1 : 0;
  a.tbound = tbound; a.tglobal = tglobal; a.seg_cnt = seg_cnt; a.seg_pos = seg_pos; a.qovf = qovf;
  a.pool_key = pool_key; a.pool_pos = pool_pos; a.pool_cnt = pool_cnt; a.pool_cap = pool_cap; a.allow = allow;
  a.o = o;
  {   // staged residuals: at most 32 KiB per workgroup (d = 1536: 5 probes at a time)
    const int dpad = (d + 3) & ~3;
    a.qm_g = std::max(1, std::min(QM_G, 32768 / (dpad * 4)));
    a.vec4 = (d % 4 == 0) && ((reinterpret_cast<uintptr_t>(qs) | reinterpret_cast<uintptr_t>(ix->centroids)) & 15) == 0;
  }
  {
    // the survivors' sums (written by the scan launcher into the same scratch slot) and the cut's parameters.  Encodings:
    //   M = 16 / 32 (search_q.hip):    |e - L s| <= 1 per entry, +2 for the f32 terms: S - (M + 2) <= dist * s <= S + (M + 2);
    //                                  LIM = SE + M + 2 with SE ~ 65535 / M: 3986 / 2018 -> bins of 8 / 4
    //   M = 48 / 64 / 96 (search_qt.hip): floor-like entries over the full u16 range: S - 4 <= dist * s <= S + 1.5 M + 4;
    //                                  LIM = 61444 -> bins of 128
    static const bool no_cut = getenv("LANCE_HIP_NO_QCUT") != nullptr;
    a.seg_sum = ctx->scratch_t<uint16_t>("q.seg_sum", (size_t)nq * nprobes * Q_CAP);
    a.ovf = ctx->scratch_t<uint32_t>("q.ovf", (size_t)nq * nprobes + 1);
    if (!a.seg_sum || !a.ovf) return LANCE_HIP_ENOMEM;
    const bool tiled = qscan_tiled_shape(m, sd);
    a.cut_shift = no_cut ? -1 : (tiled ? 7 : (m == 16 ? 3 : 2));
    a.cut_slack = tiled ? (uint32_t)(2 * m + 8) : (uint32_t)(2 * m + 4);
    if (qslack) {   // search_ms.hip's sums: rint(dist~ * s) with |dist~ - dist| s <= qslack[q]; LIM ~ 30000 + slack -> bins of 64
      int sh = 0; uint32_t sl = 0;
      mscan_cut_params(&sh, &sl);
      a.cut_shift = no_cut ? -1 : sh; a.cut_slack = sl;
    }
  }
  static const int bs = getenv("LANCE_HIP_QMERGE_BS") ? atoi(getenv("LANCE_HIP_QMERGE_BS")) : 128;
  bool ok = true;
  if (ix->metric == LANCE_HIP_DOT) {      // the dot flow exists for the matrix-core scan's shapes only (search_ms.hip: ms_shape)
    if (m == 16 && sd == 4) launch_qmerge_mu<4, 1, true>(ctx, a, nq, bs);
    else if (m == 16 && sd == 8) launch_qmerge_mu<8, 1, true>(ctx, a, nq, bs);
    else if (m == 32 && sd == 4) launch_qmerge_mu<4, 2, true>(ctx, a, nq, bs);
    else ok = false;
  } else if (m == 16) {
    if (sd == 4) launch_qmerge_mu<4, 1>(ctx, a, nq, bs);
    else if (sd == 8) launch_qmerge_mu<8, 1>(ctx, a, nq, bs);
    else if (sd == 16) launch_qmerge_mu<16, 1>(ctx, a, nq, bs);
    else ok = false;
  } else if (m == 32) {
    if (sd == 4) launch_qmerge_mu<4, 2>(ctx, a, nq, bs);
    else if (sd == 8) launch_qmerge_mu<8, 2>(ctx, a, nq, bs);
    else if (sd == 16) launch_qmerge_mu<16, 2>(ctx, a, nq, bs);
    else ok = false;
  } else if (m == 48) {
    if (sd == 4) launch_qmerge_mu<4, 3>(ctx, a, nq, bs);
    else if (sd == 8) launch_qmerge_mu<8, 3>(ctx, a, nq, bs);
    else if (sd == 16) launch_qmerge_mu<16, 3>(ctx, a, nq, bs);
    else ok = false;
  } else if (m == 64) {
    if (sd == 4) launch_qmerge_mu<4, 4>(ctx, a, nq, bs);
    else if (sd == 8) launch_qmerge_mu<8, 4>(ctx, a, nq, bs);
    else if (sd == 16) launch_qmerge_mu<16, 4>(ctx, a, nq, bs);
    else ok = false;
  } else if (m == 96) {
    if (sd == 4) launch_qmerge_mu<4, 6>(ctx, a, nq, bs);
    else if (sd == 8) launch_qmerge_mu<8, 6>(ctx, a, nq, bs);
    else if (sd == 16) launch_qmerge_mu<16, 6>(ctx, a, nq, bs);
    else ok = false;
  } else ok = false;
  LH_REQUIRE(ok, "quantised scan merge: unsupported shape (m=%d, sd=%d)", m, sd);
  return LANCE_HIP_OK;
}

}  // namespace lh
