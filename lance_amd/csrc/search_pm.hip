// search_pm.hip -- partition-major ADC scan: two queries share every LDS gather.
//
// Why: the query-major scan (search.hip) is bound by random 4-byte LDS gathers -- 32 lanes hitting
// 32 banks collide ~3.5 deep, which measures 9.15 lookups/clk/CU on gfx950 (scripts/ubench/lds_gather.hip).
// An 8-byte gather costs the same LDS cycles, so interleaving the LUTs of TWO queries that probe the SAME
// partition ([m][256][2] floats, one ds_read_b64 per (row, m)) doubles the useful lookups per LDS cycle
// (17.9 values/clk/CU measured), halves the code-byte loads and halves the codebook reads of the LUT build.
//
// Pipeline (all on the stream, no host round trip):
//   1. (query, probe) pairs are grouped by partition with the stable counting sort of group.hip;
//   2. item table: partition p with c_p probing queries contributes ceil(c_p / 2) work items;
//   3. scan: one 512-lane workgroup per item builds the interleaved LUT pair, streams the partition once
//      and keeps, per query, every row with key <= T in an LDS buffer (T tightened exactly as in the
//      query-major kernel).  Each query also owns a global bound Tglobal[q] = min over its finished
//      partitions of their k-th-smallest bound -- a valid upper bound of the final k-th distance, so later
//      partitions of the same query prune with it.  Survivors are appended to a per-query pool;
//   4. merge: per query, the pool is reduced with the same threshold machinery, sorted by (dist, rowid),
//      tie-checked and emitted (identical outputs to the query-major merge kernel).
// The union of the pool always contains every scanned row with dist <= final T, whatever order the items
// ran in, so results are deterministic and equal to the reference's SortExec over per-partition heaps.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstdlib>
#include <vector>

#include "common.h"
#include "exact.cuh"
#include "index.h"
#include "kernels.h"
#include "search_common.cuh"
#include "pm_common.cuh"

#pragma clang fp contract(off)

namespace lh {

#ifndef LH_PM_BS
#define LH_PM_BS 512
#endif
#ifndef LH_PM_STATIC_LUT
#define LH_PM_STATIC_LUT 0
#endif
constexpr int PM_BS = LH_PM_BS;            // lanes per workgroup (512: 3 workgroups x 8 waves per CU)
constexpr int PM_CAP = PM_BS + 256;        // candidate buffer entries per query, pruned class (one round + slack)
constexpr int PM_CAP_BOUND = 16;           // the bound pass keeps no candidates: its LDS is the LUT pair only (4 workgroups per CU)
#ifndef LH_PM_RPL1
#define LH_PM_RPL1 2
#endif
constexpr int PM_RPL1 = LH_PM_RPL1;  // rows per lane per round for the pruned class (4 measured slower: > 80 VGPRs drop a workgroup)

struct PmArgs {
  const float *q;               // [nq][d]
  const uint32_t *probes;       // [nq*nprobes] partition of each (query, probe) pair
  const uint32_t *pair_starts;  // [2*nlist+1] pairs grouped by virtual partition = cls*nlist + partition
  const uint32_t *pair_idx;     // [nq*nprobes] pair index (q = idx / nprobes), grouped
  const uint32_t *item_start;   // [2*nlist+1] exclusive scan of ceil(c_vp / 2) over virtual partitions
  int cls;                      // which items this launch covers: 0 = nearest-partition pairs, 1 = the rest, 2 = all
  int unbounded;                // class-1 launch whose queries start without a bound (quantised flow, class B)
  int loop;                     // 1: workgroups loop over the items of their class (item count known on the device only)
  const uint32_t *allow;        // prefilter bitmap over storage positions or NULL
  int bound_pass;               // class 0 only: compute Tglobal bounds, keep no candidates (RPL = 0 instantiation)
  const int4 *desc;             // [items] {partition, q0, q1 (-1 = none), 0}: filled by pm_item_desc_kernel
  const float *centroids, *codebook;
  const uint32_t *part_offsets;
  const uint8_t *codes;
  int d, m, nprobes, nlist, keff;
  int residual, round_f16;
  uint32_t *tglobal;            // [nq] running upper bound of the keff-th distance (key)
  uint32_t *pool_key, *pool_pos, *pool_cnt;  // [nq][pool_cap], [nq]
  int pool_cap;                 // pool entries per query
  uint32_t *flags;
  unsigned long long *prof;     // optional [8]: summed shader clocks per phase of the scan kernel (thread 0 of every item)
};

// ---- grouping keys --------------------------------------------------------------------------------------
// Class 0 = the pair whose partition is the query's nearest centroid (probe rank 0), class 1 = all others.
// Class 0 is scanned first (own launch): it fixes a tight Tglobal[q] for every query, so the other 90 % of the
// pairs prune with it from their first row and almost never have to tighten.
__global__ __launch_bounds__(256) void pm_keys_kernel(const uint32_t *__restrict__ probes, int64_t npairs, int nprobes, int nlist,
                                                      uint32_t *__restrict__ keys) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= npairs) return;
  keys[i] = probes[i] + ((i % nprobes) == 0 ? 0u : (uint32_t)nlist);
}

// ---- item table ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pm_item_table_kernel(const uint32_t *__restrict__ pair_starts, int nlist,
                                                            uint32_t *__restrict__ item_start) {
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t carry_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < nlist; base += 256) {
    const int i = base + threadIdx.x;
    const uint32_t v = i < nlist ? (pair_starts[i + 1] - pair_starts[i] + 1) / 2 : 0;
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
    if (i < nlist) item_start[i] = carry + woff + incl - v;
    __syncthreads();
    if (threadIdx.x == 255) carry_s = carry + woff + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) item_start[nlist] = carry_s;
}

// one lane per item: resolve (virtual partition, query pair) once, in parallel, instead of a serial binary search
// at the head of every scan workgroup
__global__ __launch_bounds__(256) void pm_item_desc_kernel(const uint32_t *__restrict__ item_start, const uint32_t *__restrict__ pair_starts,
                                                           const uint32_t *__restrict__ pair_idx, int nvp, int nlist, int nprobes, uint32_t max_items,
                                                           int4 *__restrict__ desc) {
  const uint32_t item = blockIdx.x * 256 + threadIdx.x;
  if (item >= max_items) return;
  int4 dsc = make_int4(-1, -1, -1, 0);
  if (item < item_start[nvp]) {
    int vp = (int)find_partition_dev(item_start, nvp, item);
    while (item_start[vp + 1] <= item) ++vp;  // empty ranges share their successor's start
    const uint32_t g = item - item_start[vp];
    const uint32_t ps = pair_starts[vp], pe = pair_starts[vp + 1];
    const uint32_t i0 = ps + 2 * g;
    dsc.x = vp % nlist;
    dsc.y = (int)(pair_idx[i0] / (uint32_t)nprobes);
    dsc.z = i0 + 1 < pe ? (int)(pair_idx[i0 + 1] / (uint32_t)nprobes) : -1;
  }
  desc[item] = dsc;
}

// ---- scan -----------------------------------------------------------------------------------------------
// RPL = rows per lane per round.  Class 0 (threshold starts at +inf, every row is a candidate) uses 1 so that
// the buffer can never overflow between two capacity checks; class 1 starts from a tight bound, appends are
// rare, so it streams 4 rows per lane between barriers (an overflow there is flagged and the query replayed
// by the exact kernel).
template <int SD, int METRIC, int MU, int RPL, int CAP>
__device__ __forceinline__ void pm_scan_item(const PmArgs &p, const uint32_t item) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int m = MU * 16;
  constexpr int Q = SD / 4;
  const int dpad = (p.d + 3) & ~3;
  float *r0 = reinterpret_cast<float *>(smem);
  float *r1 = r0 + dpad;
#if LH_PM_STATIC_LUT
  // [m][256] pairs in STATIC LDS: its base is a compile-time constant, so the gather address is one SDWA shift of the
  // code byte plus an immediate offset instead of bfe + lshl_add(base SGPR) (scripts/ubench/scan_addr.hip)
  __shared__ __attribute__((aligned(16))) f2 lut2[m * 256];
  uint32_t *ck0 = reinterpret_cast<uint32_t *>(r1 + dpad);
#else
  f2 *lut2 = reinterpret_cast<f2 *>(r1 + dpad);  // [m][256] pairs
  uint32_t *ck0 = reinterpret_cast<uint32_t *>(lut2 + m * 256);
#endif
  uint32_t *cp0 = ck0 + CAP;
  uint32_t *ck1 = cp0 + CAP;
  uint32_t *cp1 = ck1 + CAP;
  uint32_t *sorted = cp1 + CAP;   // [PM_BS]
  uint32_t *misc = sorted + PM_BS;   // [0]=cnt0 [1]=T0 [2]=cnt1 [3]=T1 [4]=tnew [5]=flags
  __shared__ int s_part, s_q0, s_q1, s_valid;

  // Stage A (independent of the work item): request the first half of this lane's codebook entries now, so
  // their L2 latency overlaps the descriptor / query / centroid loads below.
  const int cb_c = threadIdx.x & 255, cb_half = threadIdx.x >> 8;   // cb_half = which 256-lane part of the workgroup
  constexpr int MH = m / (PM_BS / 256), MA = MH / 2;
  f4 cbA[MA][Q];
#pragma unroll
  for (int i = 0; i < MA; ++i) {
    const f4 *src = reinterpret_cast<const f4 *>(p.codebook + ((int64_t)(cb_half * MH + i) * 256 + cb_c) * SD);
#pragma unroll
    for (int u = 0; u < Q; ++u) cbA[i][u] = src[u];
  }

  if (threadIdx.x == 0) {
    // direct mode (one workgroup per grid slot): the slot may lie beyond this class's items
    const uint32_t it = p.loop ? item : item + (p.cls == 1 ? p.item_start[p.nlist] : 0u);
    const int valid = p.loop ? 1 : (it < p.item_start[(p.cls == 0 ? 1 : 2) * p.nlist]);
    int4 dsc = make_int4(0, -1, -1, 0);
    if (valid) dsc = p.desc[it];
    const int part = dsc.x, q0 = dsc.y, q1 = dsc.z;
    s_valid = valid; s_part = part; s_q0 = q0; s_q1 = q1;
    misc[0] = 0; misc[2] = 0; misc[5] = 0;
  }
  const long long pt0 = p.prof ? clock64() : 0;
  __syncthreads();
  if (!s_valid) return;
  const long long pt1 = p.prof ? clock64() : 0;
  const int part = s_part, q0 = s_q0, q1 = s_q1;
  const bool has1 = q1 >= 0;
  // Stage B: everything that depends only on the descriptor is requested together (one memory round trip):
  // query / centroid elements, partition bounds, the queries' global bounds
  float qa_v = 0.0f, qb_v = 0.0f, cen_v = 0.0f;
  if ((int)threadIdx.x < p.d) {
    qa_v = p.q[(int64_t)q0 * p.d + threadIdx.x];
    qb_v = p.q[(int64_t)(has1 ? q1 : q0) * p.d + threadIdx.x];
    cen_v = p.residual ? p.centroids[(int64_t)part * p.d + threadIdx.x] : 0.0f;
  }
  if (threadIdx.x == 0) {
    misc[1] = p.tglobal[q0];
    misc[3] = has1 ? p.tglobal[q1] : 0u;
    misc[6] = misc[1]; misc[7] = misc[3];     // the bounds this item started from (publish skips the atomicMin if unchanged)
  }
  const uint32_t off = p.part_offsets[part];
  const int np = (int)(p.part_offsets[part + 1] - off);
  if (np == 0) return;
  CandBuf b0{ck0, cp0, &misc[0], &misc[1]}, b1{ck1, cp1, &misc[2], &misc[3]};

  // residual queries (v2.rs:316-332); d = M*SD <= 512 lanes for every supported shape
  if ((int)threadIdx.x < p.d) {
    float a = p.residual ? qa_v - cen_v : qa_v;
    float bq = p.residual ? qb_v - cen_v : qb_v;
    if (p.round_f16 && p.residual) { a = __half2float(__float2half_rn(a)); bq = __half2float(__float2half_rn(bq)); }
    // L2: the LDS copy is NEGATED so that the table build evaluates (r - c)^2 as (c + (-r))^2 -- the same bits (IEEE negation
    // and subtraction are exact mirror images, the square drops the sign) with packed adds (dist_exact BNEG)
    constexpr bool NEG = METRIC != METRIC_DOT;
    r0[threadIdx.x] = NEG ? -a : a; r1[threadIdx.x] = NEG ? -bq : bq;
  }
  __syncthreads();
  const long long pt2 = p.prof ? clock64() : 0;
  // LUT pair: lane (c = tid & 255, half = tid >> 8) fills sub-quantisers [half*m/2, (half+1)*m/2); each
  // codebook entry is fetched once and used for both residuals.  Second half of the entries is requested
  // before the first half is consumed.
  {
    f4 cbB[MH - MA][Q];
#pragma unroll
    for (int i = MA; i < MH; ++i) {
      const f4 *src = reinterpret_cast<const f4 *>(p.codebook + ((int64_t)(cb_half * MH + i) * 256 + cb_c) * SD);
#pragma unroll
      for (int u = 0; u < Q; ++u) cbB[i - MA][u] = src[u];
    }
#pragma unroll
    for (int i = 0; i < MH; ++i) {
      const int mm = cb_half * MH + i;
      RegVec<SD> cv;
#pragma unroll
      for (int u = 0; u < Q; ++u) cv.q[u] = i < MA ? cbA[i < MA ? i : 0][u] : cbB[i >= MA ? i - MA : 0][u];
      f2 v;
      v.x = finish_metric<METRIC>(dist_exact<SD, METRIC, METRIC != METRIC_DOT>(cv, &r0[mm * SD]));
      v.y = finish_metric<METRIC>(dist_exact<SD, METRIC, METRIC != METRIC_DOT>(cv, &r1[mm * SD]));
      lut2[mm * 256 + cb_c] = v;
    }
  }
  __syncthreads();

  const long long pt3 = p.prof ? clock64() : 0;
  const uint8_t *pcodes = p.codes + (int64_t)off * m;
  if constexpr (RPL == 0) {
    // BOUND pass (class 0): no candidates are kept.  Every lane tracks the smallest key among its rows; the
    // keff-th smallest of the 512 lane minima -- keys of keff distinct rows of this query -- is an upper bound of
    // the query's final keff-th distance (within a few ranks of the partition's exact keff-th: two of the best
    // keff rows rarely share a lane).  It seeds Tglobal, so the main pass prunes every partition, this one
    // included, from its first row, and nobody pays for selecting among unfiltered rows.
    uint32_t mn0 = 0xFFFFFFFFu, mn1 = 0xFFFFFFFFu;
    uint4 cwb[MU];
    if ((int)threadIdx.x < np) {
#pragma unroll
      for (int w = 0; w < MU; ++w) cwb[w] = *reinterpret_cast<const uint4 *>(pcodes + (int64_t)threadIdx.x * m + w * 16);
    }
    for (int base = 0; base < np; base += PM_BS) {
      const int row = base + threadIdx.x;
      uint4 cwc[MU];
#pragma unroll
      for (int w = 0; w < MU; ++w) cwc[w] = cwb[w];
      if (row + PM_BS < np) {
#pragma unroll
        for (int w = 0; w < MU; ++w) cwb[w] = *reinterpret_cast<const uint4 *>(pcodes + (int64_t)(row + PM_BS) * m + w * 16);
      }
      if (row < np) {
        float d0 = 0.0f, d1 = 0.0f;
#pragma unroll
        for (int w = 0; w < MU; ++w) {
          const uint32_t cws[4] = {cwc[w].x, cwc[w].y, cwc[w].z, cwc[w].w};
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
              const f2 v = lut2[(w * 16 + e * 4 + bb) * 256 + ((cws[e] >> (8 * bb)) & 255u)];
              d0 += v.x; d1 += v.y;
            }
        }
        if constexpr (METRIC == METRIC_DOT) { d0 = d0 - ((float)m - 1.0f); d1 = d1 - ((float)m - 1.0f); }
        if (row_allowed(p.allow, off + (uint32_t)row)) {   // a filtered row cannot be one of the k the bound stands on
          mn0 = min(mn0, order_key(d0));
          mn1 = min(mn1, order_key(d1));
        }
      }
    }
    kth_smallest_bs<PM_BS>(mn0, p.keff - 1, sorted, &misc[4]);
    if (threadIdx.x == 0 && misc[4] != 0xFFFFFFFFu) atomicMin(&p.tglobal[q0], misc[4]);
    __syncthreads();
    if (has1) {
      kth_smallest_bs<PM_BS>(mn1, p.keff - 1, sorted, &misc[4]);
      if (threadIdx.x == 0 && misc[4] != 0xFFFFFFFFu) atomicMin(&p.tglobal[q1], misc[4]);
    }
    return;
  }
  constexpr int RP = RPL == 0 ? 1 : RPL;
  // software pipeline: the code bytes of round r+1 are requested before round r's gathers, so the L2/HBM
  // latency of the (only) global load in the loop overlaps LDS work
  constexpr int ROUND = PM_BS * RP;
  uint4 cwn[RP][MU];
#pragma unroll
  for (int u = 0; u < RP; ++u) {
    const int row = u * PM_BS + threadIdx.x;
    if (row < np) {
#pragma unroll
      for (int w = 0; w < MU; ++w) cwn[u][w] = *reinterpret_cast<const uint4 *>(pcodes + (int64_t)row * m + w * 16);
    }
  }
  constexpr uint32_t ROUND_OVF = 0x100u;   // round-local overflow marker in misc[5]
  auto scan_rows = [&](int base, int u, const uint4 (&cw)[MU], uint32_t T0, uint32_t T1) {
    const int row = base + u * PM_BS + threadIdx.x;
    if (row >= np) return;
    float d0 = 0.0f, d1 = 0.0f;  // pq/distance.rs:128-141: += table[code] for m = 0..M-1, per query
#pragma unroll
    for (int w = 0; w < MU; ++w) {
      const uint32_t cws[4] = {cw[w].x, cw[w].y, cw[w].z, cw[w].w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
          const f2 v = lut2[(w * 16 + e * 4 + bb) * 256 + ((cws[e] >> (8 * bb)) & 255u)];
          d0 += v.x; d1 += v.y;
        }
    }
    if constexpr (METRIC == METRIC_DOT) { d0 = d0 - ((float)m - 1.0f); d1 = d1 - ((float)m - 1.0f); }
    const uint32_t k0 = order_key(d0), k1 = order_key(d1);
    if ((k0 <= T0 || (has1 && k1 <= T1)) && !row_allowed(p.allow, off + (uint32_t)row)) return;
    if (k0 <= T0) {
      const uint32_t slot = atomicAdd(&misc[0], 1u);
      if (slot < CAP) { ck0[slot] = k0; cp0[slot] = off + (uint32_t)row; } else misc[5] = misc[5] | ROUND_OVF;
    }
    if (has1 && k1 <= T1) {
      const uint32_t slot = atomicAdd(&misc[2], 1u);
      if (slot < CAP) { ck1[slot] = k1; cp1[slot] = off + (uint32_t)row; } else misc[5] = misc[5] | ROUND_OVF;
    }
  };
  for (int base = 0; base < np; base += ROUND) {
    constexpr int LIMIT = RP == 1 ? CAP - PM_BS : CAP / 2;
    // counts are read by every lane BEFORE a barrier: lanes past it append to them at once (read, barrier, decide -- the
    // unguarded read let a fast wave's appends change a slow wave's decision: divergent barriers, see search.hip)
    uint32_t cnt0_before = misc[0], cnt1_before = misc[2];
    __syncthreads();
    if ((int)cnt0_before > LIMIT || (int)cnt1_before > LIMIT) {   // uniform
      if ((int)cnt0_before > LIMIT) tighten_bs<PM_BS, CAP>(b0, p.keff, sorted, &misc[4]);
      if ((int)cnt1_before > LIMIT) tighten_bs<PM_BS, CAP>(b1, p.keff, sorted, &misc[4]);
      cnt0_before = misc[0]; cnt1_before = misc[2];
      __syncthreads();
    }
    const uint32_t T0 = misc[1], T1 = misc[3];
    uint4 cwc[RP][MU];
#pragma unroll
    for (int u = 0; u < RP; ++u)
#pragma unroll
      for (int w = 0; w < MU; ++w) cwc[u][w] = cwn[u][w];
#pragma unroll
    for (int u = 0; u < RP; ++u) {
      const int rown = base + ROUND + u * PM_BS + threadIdx.x;
      if (rown < np) {
#pragma unroll
        for (int w = 0; w < MU; ++w) cwn[u][w] = *reinterpret_cast<const uint4 *>(pcodes + (int64_t)rown * m + w * 16);
      }
    }
#pragma unroll
    for (int u = 0; u < RP; ++u) scan_rows(base, u, cwc[u], T0, T1);
    __syncthreads();
    if constexpr (RP > 1) {
      // optimistic round overflowed a buffer (more than CAP/2 rows under the bound in 2 sub-rounds): roll the
      // round back and replay it one sub-round at a time with a capacity check before each (cannot overflow
      // unless more than 256 rows tie at the bound, which is flagged for the exact kernel)
      if (misc[5] & ROUND_OVF) {
        __syncthreads();
        if (threadIdx.x == 0) { misc[0] = cnt0_before; misc[2] = cnt1_before; misc[5] &= ~ROUND_OVF; }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < RP; ++u) {
          const uint32_t r0 = misc[0], r1 = misc[2];
          __syncthreads();
          if ((int)r0 > CAP - PM_BS) tighten_bs<PM_BS, CAP>(b0, p.keff, sorted, &misc[4]);
          if ((int)r1 > CAP - PM_BS) tighten_bs<PM_BS, CAP>(b1, p.keff, sorted, &misc[4]);
          scan_rows(base, u, cwc[u], misc[1], misc[3]);
          __syncthreads();
        }
        if (misc[5] & ROUND_OVF) {
          __syncthreads();
          if (threadIdx.x == 0) misc[5] = (misc[5] & ~ROUND_OVF) | FLAG_OVERFLOW;
          __syncthreads();
        }
      }
    } else {
      if (misc[5] & ROUND_OVF) {
        __syncthreads();
        if (threadIdx.x == 0) misc[5] = (misc[5] & ~ROUND_OVF) | FLAG_OVERFLOW;
        __syncthreads();
      }
    }
  }
  const long long pt4 = p.prof ? clock64() : 0;
  // publish: shrink each buffer once, lower the query's global bound, append the survivors to its pool.  The two
  // queries go through it TOGETHER so that their global atomics (the only long-latency operations here) are in
  // flight at the same time, and a query whose threshold was never tightened locally skips the atomicMin round trip
  // (its T is still the value read from Tglobal at item start: stale at worst, never invalid).
  __shared__ uint32_t s_tg[2], s_base[2], s_tot[2], s_wr[2];
  for (int j = 0; j < 2; ++j) {
    if (j == 1 && !has1) break;
    const CandBuf &b = j ? b1 : b0;
    // <= PM_BS entries -> one entry per lane -> the bound is the exact keff-th smallest: publish ~keff rows
    if ((int)*b.cnt > p.keff + 32) tighten_bs<PM_BS, CAP>(b, p.keff, sorted, &misc[4]);
    if ((int)*b.cnt > PM_BS) tighten_bs<PM_BS, CAP>(b, p.keff, sorted, &misc[4]);
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    const int j = threadIdx.x;
    const CandBuf &b = j ? b1 : b0;
    const bool on = j == 0 || has1;
    uint32_t tg = on ? *b.T : 0u;
    // *b.T is always a valid upper bound of this query's final keff-th distance: it is either the value read
    // from tglobal or the keff-th-smallest bound of >= keff real rows of this query
    if (on && *b.cnt > 0 && tg < misc[6 + j]) tg = min(atomicMin(&p.tglobal[j ? q1 : q0], tg), tg);
    s_tg[j] = tg; s_tot[j] = 0; s_wr[j] = 0;
  }
  __syncthreads();
  const int c0n = min((int)*b0.cnt, CAP), c1n = has1 ? min((int)*b1.cnt, CAP) : 0;
  {
    uint32_t m0 = 0, m1 = 0;
    const uint32_t tg0 = s_tg[0], tg1 = s_tg[1];
    for (int i = threadIdx.x; i < c0n; i += PM_BS) m0 += ck0[i] <= tg0 ? 1u : 0u;
    for (int i = threadIdx.x; i < c1n; i += PM_BS) m1 += ck1[i] <= tg1 ? 1u : 0u;
    if (m0) atomicAdd(&s_tot[0], m0);
    if (m1) atomicAdd(&s_tot[1], m1);
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    const int j = threadIdx.x;
    const uint32_t tot = s_tot[j];
    s_base[j] = tot ? atomicAdd(&p.pool_cnt[j ? q1 : q0], tot) : 0u;
  }
  __syncthreads();
  for (int j = 0; j < 2; ++j) {
    const uint32_t tot = s_tot[j], basep = s_base[j], tg = s_tg[j];
    if (!tot) continue;
    const int qj = j ? q1 : q0;
    if (basep + tot > (uint32_t)p.pool_cap) {
      if (threadIdx.x == 0) atomicOr(&p.flags[qj], FLAG_OVERFLOW);
      continue;
    }
    const uint32_t *bk = j ? ck1 : ck0, *bp = j ? cp1 : cp0;
    const int c = j ? c1n : c0n;
    for (int i = threadIdx.x; i < c; i += PM_BS) {
      if (bk[i] <= tg) {
        const uint32_t slot = basep + atomicAdd(&s_wr[j], 1u);
        p.pool_key[(int64_t)qj * p.pool_cap + slot] = bk[i];
        p.pool_pos[(int64_t)qj * p.pool_cap + slot] = bp[i];
      }
    }
  }
  if (p.prof && threadIdx.x == 0) {
    const long long pt5 = clock64();
    atomicAdd(&p.prof[0], (unsigned long long)(pt1 - pt0)); atomicAdd(&p.prof[1], (unsigned long long)(pt2 - pt1));
    atomicAdd(&p.prof[2], (unsigned long long)(pt3 - pt2)); atomicAdd(&p.prof[3], (unsigned long long)(pt4 - pt3));
    atomicAdd(&p.prof[4], (unsigned long long)(pt5 - pt4)); atomicAdd(&p.prof[5], 1ull);
  }
  if (threadIdx.x == 0 && (misc[5] & FLAG_OVERFLOW)) {
    atomicOr(&p.flags[q0], FLAG_OVERFLOW);
    if (has1) atomicOr(&p.flags[q1], FLAG_OVERFLOW);
  }
}

template <int SD, int METRIC, int MU, int RPL, int CAP>
__global__ __launch_bounds__(PM_BS) void ivfpq_scan_pm_kernel(PmArgs p) {
  pm_scan_item<SD, METRIC, MU, RPL, CAP>(p, blockIdx.x);
}

// Class-B launch of the quantised flow: the item count is only known on the device, so a fixed grid of workgroups loops
// over the items.  A separate kernel: the loop costs registers (107 instead of 62 VGPRs) that the hot launches must not pay.
template <int SD, int METRIC, int MU, int RPL, int CAP>
__global__ __launch_bounds__(PM_BS) void ivfpq_scan_pm_loop_kernel(PmArgs p) {
  const uint32_t lo = p.cls == 1 ? p.item_start[p.nlist] : 0u;
  const uint32_t hi = p.item_start[(p.cls == 0 ? 1 : 2) * p.nlist];
  for (uint32_t item = lo + blockIdx.x; item < hi; item += gridDim.x) {
    pm_scan_item<SD, METRIC, MU, RPL, CAP>(p, item);
    __syncthreads();   // the next item re-initialises the LDS state
  }
}

// ---- merge: pool -> top keff by (dist, rowid) ---------------------------------------------------------
constexpr int PMM_CAP = 1024;
__global__ __launch_bounds__(256) void ivfpq_merge_pm_kernel(const uint32_t *__restrict__ pool_key, const uint32_t *__restrict__ pool_pos,
                                                             const uint32_t *__restrict__ pool_cnt, const uint32_t *__restrict__ tglobal,
                                                             const uint64_t *__restrict__ row_ids, int pool_cap, SelectOut o) {
  __shared__ uint32_t ckey[PMM_CAP], cpos[PMM_CAP], sorted[256], misc[8];
  __shared__ uint64_t rid[SCAN_LCAP];
  __shared__ uint32_t skey[SCAN_LCAP], spos[SCAN_LCAP];
  __shared__ int s_amb;
  const int q = blockIdx.x;
  if (o.flags[q] & FLAG_OVERFLOW) return;  // pool incomplete: the exact kernel recomputes this query
  const int n = min((int)pool_cnt[q], pool_cap);
  if (threadIdx.x == 0) { misc[0] = 0; misc[1] = tglobal[q]; s_amb = 0; }
  __syncthreads();
  CandBuf b{ckey, cpos, &misc[0], &misc[1]};
  const uint32_t *pk = pool_key + (int64_t)q * pool_cap, *pp = pool_pos + (int64_t)q * pool_cap;
  for (int base = 0; base < n; base += 512) {
    const bool need_tighten = (int)misc[0] > PMM_CAP - 512;   // read, barrier, decide
    __syncthreads();
    if (need_tighten) tighten_bs<256, PMM_CAP>(b, o.keff, sorted, &misc[2]);
    const uint32_t T = misc[1];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = base + u * 256 + threadIdx.x;
      if (i < n) {
        const uint32_t kk = pk[i];
        if (kk <= T) {
          const uint32_t slot = atomicAdd(&misc[0], 1u);
          if (slot < PMM_CAP) { ckey[slot] = kk; cpos[slot] = pp[i]; }
        }
      }
    }
    __syncthreads();
  }
  for (int iter = 0; iter < 8 && (int)misc[0] > SCAN_LCAP; ++iter) tighten_bs<256, PMM_CAP>(b, o.keff, sorted, &misc[2]);
  __syncthreads();
  int c = min((int)misc[0], PMM_CAP);
  if (c > SCAN_LCAP) {
    if (threadIdx.x == 0) atomicOr(&o.flags[q], FLAG_OVERFLOW);
    c = SCAN_LCAP;
  }
  for (int i = threadIdx.x; i < SCAN_LCAP; i += 256) {
    if (i < c) { skey[i] = ckey[i]; spos[i] = cpos[i]; rid[i] = row_ids[cpos[i]]; }
    else { skey[i] = 0xFFFFFFFFu; spos[i] = 0; rid[i] = ~0ull; }
  }
  __syncthreads();
  int Pq = 64;
  while (Pq < c) Pq <<= 1;
  bitonic_sort_kr<256>(skey, rid, spos, Pq);
  select_and_emit<256>(o, q, skey, rid, spos, c, &s_amb);
}

// ---- host ---------------------------------------------------------------------------------------------
template <int SD, int METRIC>
static bool launch_pm_mu(lance_hip_ctx *ctx, const PmArgs &a, unsigned grid, size_t lds) {
  const int mu = a.m / 16;
  if (a.cls == 0 && a.bound_pass) {
    if (mu == 1) { hipLaunchKernelGGL((ivfpq_scan_pm_kernel<SD, METRIC, 1, 0, PM_CAP_BOUND>), dim3(grid), dim3(PM_BS), lds, ctx->stream, a); return true; }
    if (mu == 2) { hipLaunchKernelGGL((ivfpq_scan_pm_kernel<SD, METRIC, 2, 0, PM_CAP_BOUND>), dim3(grid), dim3(PM_BS), lds, ctx->stream, a); return true; }
  } else if (a.cls == 0) {   // LANCE_HIP_PM_NOBOUND=1: the earlier two-class flow (class 0 selects among unfiltered rows)
    if (mu == 1) { hipLaunchKernelGGL((ivfpq_scan_pm_kernel<SD, METRIC, 1, 1, PM_CAP>), dim3(grid), dim3(PM_BS), lds, ctx->stream, a); return true; }
    if (mu == 2) { hipLaunchKernelGGL((ivfpq_scan_pm_kernel<SD, METRIC, 2, 1, PM_CAP>), dim3(grid), dim3(PM_BS), lds, ctx->stream, a); return true; }
  } else if (a.unbounded) {   // queries without a bound: every row is a candidate at first -> one row per lane per round
    if (mu == 1) { hipLaunchKernelGGL((ivfpq_scan_pm_loop_kernel<SD, METRIC, 1, 1, PM_CAP>), dim3(grid), dim3(PM_BS), lds, ctx->stream, a); return true; }
    if (mu == 2) { hipLaunchKernelGGL((ivfpq_scan_pm_loop_kernel<SD, METRIC, 2, 1, PM_CAP>), dim3(grid), dim3(PM_BS), lds, ctx->stream, a); return true; }
  } else {
    if (mu == 1) { hipLaunchKernelGGL((ivfpq_scan_pm_kernel<SD, METRIC, 1, PM_RPL1, PM_CAP>), dim3(grid), dim3(PM_BS), lds, ctx->stream, a); return true; }
    if (mu == 2) { hipLaunchKernelGGL((ivfpq_scan_pm_kernel<SD, METRIC, 2, 1, PM_CAP>), dim3(grid), dim3(PM_BS), lds, ctx->stream, a); return true; }
  }
  return false;
}

template <int METRIC>
static bool launch_pm_sd(lance_hip_ctx *ctx, const PmArgs &a, int sd, unsigned grid, size_t lds) {
  if (sd == 4) return launch_pm_mu<4, METRIC>(ctx, a, grid, lds);
  if (sd == 8) return launch_pm_mu<8, METRIC>(ctx, a, grid, lds);
  if (sd == 16) return launch_pm_mu<16, METRIC>(ctx, a, grid, lds);
  return false;
}

// A/B switches, read once per process
static bool pm_nobound() { static const bool v = getenv("LANCE_HIP_PM_NOBOUND") != nullptr; return v; }

bool pm_supported(const lance_hip_index *ix, uint32_t keff, int has_range, uint32_t nq, uint32_t nprobes) {
  const int d = (int)ix->d, m = (int)ix->m;
  if (ix->nbits != 8 || has_range || keff > (uint32_t)SCAN_MAX_KEFF) return false;
  if (m == 0 || d % m != 0) return false;
  const int sd = d / m;
  if (sd != 4 && sd != 8 && sd != 16) return false;
  if ((reinterpret_cast<uintptr_t>(ix->codebook) & 15) || (reinterpret_cast<uintptr_t>(ix->codes) & 15)) return false;
  if (qscan_tiled_shape(m, sd))    // M = 48 / 64 / 96: only the quantised flow exists (the exact pair table would not fit in LDS)
    return !pm_nobound() && qscan_supported(ix, nq, nprobes);
  if (m % 16 != 0 || m / 16 > 2) return false;
  return true;
}

static size_t pm_lds_base(int d, int m) {
  const int dpad = (d + 3) & ~3;
#if LH_PM_STATIC_LUT
  return (size_t)dpad * 8 + PM_BS * 4 + 8 * 4;   // the LUT pair table is static LDS
#else
  return (size_t)dpad * 8 + (size_t)m * 256 * 8 + PM_BS * 4 + 8 * 4;
#endif
}

// Quantised flow (L2 / cosine): exact bound pass over every query's nearest partition -> 4-query integer filter scan of all
// probed partitions (search_q.hip) -> exact re-evaluation of the survivors + (dist, rowid) selection.  Queries whose bound
// pass found fewer than keff rows have no bound: they go through the exact pair kernel (class B) and its pool.
static int ivfpq_scan_merge_q(lance_hip_ctx *ctx, const lance_hip_index *ix, const float *qs, uint32_t nq, const uint32_t *probes,
                              uint32_t nprobes, uint32_t keff, uint32_t k, bool do_refine, uint64_t *ids, float *dists,
                              uint64_t *cand_rid, uint32_t *cand_cnt, uint32_t *flags, const uint32_t *allow) {
  const int d = (int)ix->d, m = (int)ix->m, sd = d / m, nlist = (int)ix->nlist;
  const size_t npairs = (size_t)nq * nprobes;
  const uint32_t max_items2 = (uint32_t)(npairs / 2 + 2 * nlist + 2);
  const uint32_t max_items4 = (uint32_t)(npairs / 4 + nlist + 2);
  uint32_t *keys = ctx->scratch_t<uint32_t>("pm.keys", npairs);
  uint32_t *pair_starts = ctx->scratch_t<uint32_t>("pm.pair_starts", (size_t)2 * nlist + 1);
  uint32_t *pair_idx = ctx->scratch_t<uint32_t>("pm.pair_idx", npairs);
  uint32_t *item_start = ctx->scratch_t<uint32_t>("pm.item_start", (size_t)2 * nlist + 1);
  int4 *desc = ctx->scratch_t<int4>("pm.desc", max_items2);
  uint32_t *pair_starts0 = ctx->scratch_t<uint32_t>("q.pair_starts0", (size_t)nlist + 1);
  // dot: the bound pass runs over each query's THREE nearest lists and keeps the smallest bound -- without residuals the list a query's best rows sit in
  // is often not the one with the largest centroid score, and the nearest list's k*refine-th distance then lets hundreds of that list's rows through.
  // C2 shape, ms per 10,000-query batch with 1 / 2 / 3 / 4 lists (gpurun r06zzf, r06zzg): unit-normalised rows 0.951 / 0.854 / 0.860 / 0.867 (overflowed
  // segments 5,311 -> 2,852 with two), centred rows 1.150 / 0.911 / 0.900 / 0.910, SIFT-like rows as they are (largest list 82,424 of 10^6 rows) 3.94 / 2.29 /
  // 1.88 against 3.46 on the exact pair scan.  LANCE_HIP_DOT_BOUND_LISTS: A/B
  static const uint32_t dot_nb_env = getenv("LANCE_HIP_DOT_BOUND_LISTS") ? (uint32_t)std::max(1, atoi(getenv("LANCE_HIP_DOT_BOUND_LISTS"))) : 3u;
  // L2 / cosine: LANCE_HIP_BOUND_LISTS = 2 .. 4 is an A/B switch (one list by default); only batches whose bound pass runs on the matrix cores take it
  static const uint32_t l2_nb_env = getenv("LANCE_HIP_BOUND_LISTS") ? (uint32_t)std::max(1, atoi(getenv("LANCE_HIP_BOUND_LISTS"))) : 1u;
  static const bool no_msbound_env = getenv("LANCE_HIP_NO_MSBOUND") != nullptr || getenv("LANCE_HIP_EXACT_BOUND") != nullptr;
  const uint32_t l2_nb = (!no_msbound_env && l2_nb_env > 1 && mscan_supported(ix, nq, nprobes) && qscan_pt_mode(ix) != 2) ? l2_nb_env : 1u;
  const uint32_t nb0 = std::min<uint32_t>(std::min<uint32_t>(ix->metric == LANCE_HIP_DOT ? dot_nb_env : l2_nb, 4u), nprobes);
  const uint32_t max_items0 = (uint32_t)((size_t)nq * nb0 / 2 + nlist + 2);
  uint32_t *pair_idx0 = ctx->scratch_t<uint32_t>("q.pair_idx0", (size_t)nq * nb0);
  uint32_t *item_start0 = ctx->scratch_t<uint32_t>("q.item_start0", (size_t)nlist + 1);
  int4 *desc0 = ctx->scratch_t<int4>("q.desc0", max_items0);
  uint32_t *item_start4 = ctx->scratch_t<uint32_t>("q.item_start4", (size_t)nlist + 1);
  int4 *desc4 = ctx->scratch_t<int4>("q.desc4", max_items4);
  uint32_t *tglobal = ctx->scratch_t<uint32_t>("pm.tglobal", (size_t)nq * 4);
  uint32_t *seg_cnt = ctx->scratch_t<uint32_t>("q.seg_cnt", npairs);
  uint32_t *seg_pos = ctx->scratch_t<uint32_t>("q.seg_pos", npairs * QSCAN_SEG_CAP);
  // tiled shapes (M >= 48) have no exact pair kernel: EVERY segment of a query without a bound goes through the rescan kernel and
  // publishes ~keff rows to the pool, so the pool follows nprobes there (nprobes = 100 with refine 10 overflowed the
  // 8192-entry pool of every such query and sent it to the query-major replay), within 1 GiB of scratch
  uint64_t pool_max = qscan_tiled_shape(m, sd) ? 65536 : 8192;
  while (pool_max > 8192 && (uint64_t)nq * pool_max * 8 > (1ull << 30)) pool_max /= 2;
  int pool_cap = (int)std::min<uint64_t>(pool_max, std::max<uint64_t>(512, (uint64_t)nprobes * (keff + 28)));
  pool_cap = (pool_cap + 255) & ~255;
  uint32_t *pool_key = ctx->scratch_t<uint32_t>("pm.pool_key", (size_t)nq * pool_cap);
  uint32_t *pool_pos = ctx->scratch_t<uint32_t>("pm.pool_pos", (size_t)nq * pool_cap);
  if (!keys || !pair_starts || !pair_idx || !item_start || !desc || !pair_starts0 || !pair_idx0 || !item_start0 || !desc0 || !item_start4 ||
      !desc4 || !tglobal || !seg_cnt || !seg_pos || !pool_key || !pool_pos)
    return LANCE_HIP_ENOMEM;
  uint32_t *pool_cnt = tglobal + nq, *tbound = tglobal + 2 * (size_t)nq, *qovf = tglobal + 3 * (size_t)nq;
  LH_CHECK_HIP(lh::memset_multi(ctx->stream, {{tglobal, 0xFF, (size_t)nq * 4}, {pool_cnt, 0, (size_t)nq * 4}}));
  PmArgs a;
  a.q = qs; a.probes = probes;
  a.centroids = ix->centroids; a.codebook = ix->codebook; a.part_offsets = ix->part_offsets; a.codes = ix->codes;
  a.d = d; a.m = m; a.nprobes = (int)nprobes; a.nlist = nlist; a.keff = (int)keff;
  const bool dot = ix->metric == LANCE_HIP_DOT;      // (qscan_supported: only batches the matrix-core bound pass + scan serve)
  a.residual = dot ? 0 : 1;
  a.round_f16 = ix->dtype == LANCE_HIP_F16 ? 1 : 0;
  a.prof = nullptr;
  a.tglobal = tglobal; a.pool_key = pool_key; a.pool_pos = pool_pos; a.pool_cnt = pool_cnt; a.pool_cap = pool_cap; a.flags = flags;
  a.unbounded = 0; a.loop = 0; a.allow = allow;
  const bool tiled = qscan_tiled_shape(m, sd);
  static const bool exact_bound_env = getenv("LANCE_HIP_EXACT_BOUND") != nullptr;
  const bool exact_bound = exact_bound_env && !tiled && !dot;   // the exact pair kernel has no M > 32 instantiation
  {
    // bound pass: the nq (query, nearest partition) pairs grouped by partition
    ScopedTimer t(ctx, "pm_group");
    LH_TRY(qscan_nearest_keys(ctx, probes, nq, nprobes, keys, nb0));
    LH_TRY(stable_group(ctx, keys, (int64_t)nq * nb0, (int64_t)nq * nb0, nlist, 1, pair_starts0, pair_idx0, (int64_t)nq * nb0, nullptr));
    if (exact_bound) {
      hipLaunchKernelGGL(pm_item_table_kernel, dim3(1), dim3(256), 0, ctx->stream, pair_starts0, nlist, item_start0);
      hipLaunchKernelGGL(pm_item_desc_kernel, dim3((unsigned)cdiv(max_items0, 256)), dim3(256), 0, ctx->stream, item_start0, pair_starts0, pair_idx0,
                         nlist, nlist, 1, max_items0, desc0);
    }
  }
  if (exact_bound) {   // round-1 bound: exact f32 pair scan of the nearest partition (two queries per item)
    ScopedTimer t(ctx, "ivfpq_scan_c0");
    a.pair_starts = pair_starts0; a.pair_idx = pair_idx0; a.item_start = item_start0; a.desc = desc0;
    a.cls = 0; a.bound_pass = 1;
    const size_t lds = pm_lds_base(d, m) + (size_t)PM_CAP_BOUND * 16;
    const bool ok = launch_pm_sd<METRIC_L2>(ctx, a, sd, (unsigned)(nq / 2 + nlist + 1), lds);
    LH_REQUIRE(ok, "partition-major scan: unsupported shape (m=%d, sd=%d)", m, sd);
  } else if (!dot && qscan_pt_mode(ix) == 2) {   // per-query tables built before the bound pass and shared with the main pass (search_qt.hip)
    ScopedTimer t(ctx, "ivfpq_scan_c0");
    LH_TRY(qbound_pt_launch(ctx, ix, qs, nq, nprobes, keff, probes, pair_starts0, pair_idx0, item_start0, desc0, (uint32_t)(nq / 4 + nlist + 2), tglobal,
                            allow));
  } else {
    ScopedTimer t(ctx, "ivfpq_scan_c0");
    // a batch the matrix-core scan serves gets its bounds from the same matrix product (search_ms.hip: ms_bound_kernel) ...
    int mb_rc = LH_NOT_TAKEN;
    if (mscan_supported(ix, nq, nprobes))
      mb_rc = msbound_launch(ctx, ix, qs, nq, keff, pair_starts0, pair_idx0, item_start0, desc0, (uint32_t)((size_t)nq * nb0 / 4 + nlist + 2), tglobal, allow, nb0);
    if (mb_rc < 0) return mb_rc;          // a real failure (every LANCE_HIP_E* code is negative): never a silent fall-back
    if (mb_rc == LH_NOT_TAKEN && dot) return LH_NOT_TAKEN;      // no integer pass for dot: the caller runs the exact pair scan (nothing but scratch was written)
    // ... every other one from the integer histogram, four queries per gather (search_q.hip)
    if (mb_rc == LH_NOT_TAKEN) LH_TRY(qbound_launch(ctx, ix, qs, nq, keff, pair_starts0, pair_idx0, item_start0, desc0, (uint32_t)(nq / 4 + nlist + 2), tglobal, allow));
  }
  {
    // main pass grouping: class A (bounded) pairs by partition for the filter scan, class B for the exact pair kernel
    ScopedTimer t(ctx, "pm_group");
    LH_TRY(qscan_group(ctx, probes, nq, nprobes, nlist, tglobal, keys, tbound, pair_starts, pair_idx, item_start4, desc4, max_items4, 4, dot ? 1 : 0));      // four queries per work item (q_common.cuh: Q_G)
    hipLaunchKernelGGL(pm_item_table_kernel, dim3(1), dim3(256), 0, ctx->stream, pair_starts, 2 * nlist, item_start);
    hipLaunchKernelGGL(pm_item_desc_kernel, dim3((unsigned)cdiv(max_items2, 256)), dim3(256), 0, ctx->stream, item_start, pair_starts, pair_idx,
                       2 * nlist, nlist, (int)nprobes, max_items2, desc);
  }
  // timers inside: "q_residual" (memsets + residual pre-pass) and "ivfpq_scan_c1" (the filter scan kernel alone)
  // search_ms.hip: the filter as a matrix product per partition (its own pre-pass; same segment outputs + a per-query slack for the merge cut)
  uint32_t *qslack = nullptr;
  float *seg_val = nullptr, *seg_scale = nullptr;      // rows-on-lanes kernel: the survivors' accumulator values + the per-pair scale of their sums
  int ms_rc = LH_NOT_TAKEN;
  if (mscan_supported(ix, nq, nprobes))
    ms_rc = mscan_launch(ctx, ix, qs, nq, nprobes, probes, pair_starts, pair_idx, tbound, seg_cnt, seg_pos, qovf, allow, &qslack, &seg_val, &seg_scale);
  if (ms_rc < 0) return ms_rc;
  if (ms_rc == LH_NOT_TAKEN && dot) return LH_NOT_TAKEN;
  if (ms_rc == LH_NOT_TAKEN) {
    qslack = nullptr; seg_val = nullptr; seg_scale = nullptr;
    LH_TRY(qscan_launch(ctx, ix, qs, nq, nprobes, pair_idx, item_start4, desc4, max_items4, tbound, seg_cnt, seg_pos, qovf, allow, probes));
  }
  static const bool q_stats = getenv("LANCE_HIP_Q_STATS") != nullptr;
  if (q_stats) {   // diagnosis: how many rows survive the integer filter
    std::vector<uint32_t> sc(npairs), tb(nq);
    (void)hipMemcpyAsync(sc.data(), seg_cnt, npairs * 4, hipMemcpyDeviceToHost, ctx->stream);
    (void)hipMemcpyAsync(tb.data(), tbound, (size_t)nq * 4, hipMemcpyDeviceToHost, ctx->stream);
    (void)hipStreamSynchronize(ctx->stream);
    uint64_t tot = 0, ovf = 0, mx = 0, nb = 0, r0 = 0;
    for (size_t i = 0; i < npairs; ++i) { tot += std::min<uint32_t>(sc[i], QSCAN_SEG_CAP); ovf += sc[i] > (uint32_t)QSCAN_SEG_CAP; mx = std::max<uint64_t>(mx, sc[i]); if (i % nprobes == 0) r0 += sc[i]; }
    for (uint32_t i = 0; i < nq; ++i) nb += tb[i] == 0xFFFFFFFFu;
    fprintf(stderr, "[qscan] nq=%u nprobes=%u keff=%u survivors/query %.1f (nearest partition %.1f) max segment %llu overflowed segments %llu class-B queries %llu\n",
            nq, nprobes, keff, (double)tot / nq, (double)r0 / nq, (unsigned long long)mx, (unsigned long long)ovf, (unsigned long long)nb);
  }
  if (tiled) {   // no exact pair kernel at these sizes: class-B queries are scanned by the rescan kernel (exact f32 table)
    ScopedTimer t(ctx, "ivfpq_scan_cb");
    LH_TRY(qscan_classb_to_rescan(ctx, tbound, nq, nprobes, seg_cnt, qovf));
  } else {
    ScopedTimer t(ctx, "ivfpq_scan_cb");
    a.pair_starts = pair_starts; a.pair_idx = pair_idx; a.item_start = item_start; a.desc = desc;
    a.cls = 1; a.bound_pass = 0; a.unbounded = 1; a.loop = 1;
    const size_t lds = pm_lds_base(d, m) + (size_t)PM_CAP * 16;
    const unsigned grid = (unsigned)std::min<uint64_t>((uint64_t)ctx->num_cus * 3, (uint64_t)max_items2);
    const bool ok = dot ? launch_pm_sd<METRIC_DOT>(ctx, a, sd, grid, lds) : launch_pm_sd<METRIC_L2>(ctx, a, sd, grid, lds);
    LH_REQUIRE(ok, "partition-major scan: unsupported shape (m=%d, sd=%d)", m, sd);
  }
  {
    SelectOut o;
    o.keff = (int)keff; o.k = (int)k; o.refine = do_refine ? 1 : 0;
    o.out_ids = ids; o.out_dists = dists; o.cand_rid = cand_rid; o.cand_cnt = cand_cnt; o.flags = flags;
    o.part_offsets = ix->part_offsets; o.nlist = nlist;
    ScopedTimer t(ctx, "ivfpq_merge");
    LH_TRY(qmerge_launch(ctx, ix, qs, nq, probes, nprobes, tbound, tglobal, seg_cnt, seg_pos, qovf, pool_key, pool_pos, pool_cnt, pool_cap, o, allow, qslack, seg_val, seg_scale));
  }
  LH_CHECK_HIP(hipGetLastError());
  return LANCE_HIP_OK;
}

// scan + merge for the whole batch; outputs as the query-major path (ids/dists or refine candidates)
int ivfpq_scan_merge_pm(lance_hip_ctx *ctx, const lance_hip_index *ix, const float *qs, uint32_t nq, const uint32_t *probes,
                        uint32_t nprobes, uint32_t keff, uint32_t k, bool do_refine, uint64_t *ids, float *dists,
                        uint64_t *cand_rid, uint32_t *cand_cnt, uint32_t *flags, const uint32_t *allow) {
  const int d = (int)ix->d, m = (int)ix->m, sd = d / m, nlist = (int)ix->nlist;
  const int scan_metric = ix->metric == LANCE_HIP_COSINE ? LANCE_HIP_L2 : ix->metric;
  if (!pm_nobound() && qscan_supported(ix, nq, nprobes)) {
    const int rc = ivfpq_scan_merge_q(ctx, ix, qs, nq, probes, nprobes, keff, k, do_refine, ids, dists, cand_rid, cand_cnt, flags, allow);
    if (rc != LH_NOT_TAKEN) return rc;      // (dot: the matrix-core passes did not take the batch after all -- the exact pair scan below)
  }
  const size_t npairs = (size_t)nq * nprobes;
  uint32_t *pair_starts = ctx->scratch_t<uint32_t>("pm.pair_starts", (size_t)2 * nlist + 1);
  uint32_t *pair_idx = ctx->scratch_t<uint32_t>("pm.pair_idx", npairs);
  uint32_t *item_start = ctx->scratch_t<uint32_t>("pm.item_start", (size_t)2 * nlist + 1);
  uint32_t *tglobal = ctx->scratch_t<uint32_t>("pm.tglobal", (size_t)nq * 2);
  // every (query, partition) publishes at most ~keff (+ ties) rows, usually far fewer once Tglobal is tight
  int pool_cap = (int)std::min<uint64_t>(8192, std::max<uint64_t>(512, (uint64_t)nprobes * (keff + 28)));
  pool_cap = (pool_cap + 255) & ~255;
  uint32_t *pool_key = ctx->scratch_t<uint32_t>("pm.pool_key", (size_t)nq * pool_cap);
  uint32_t *pool_pos = ctx->scratch_t<uint32_t>("pm.pool_pos", (size_t)nq * pool_cap);
  if (!pair_starts || !pair_idx || !item_start || !tglobal || !pool_key || !pool_pos) return LANCE_HIP_ENOMEM;
  uint32_t *pool_cnt = tglobal + nq;
  LH_CHECK_HIP(lh::memset_multi(ctx->stream, {{tglobal, 0xFF, (size_t)nq * 4}, {pool_cnt, 0, (size_t)nq * 4}}));
  uint32_t *keys = ctx->scratch_t<uint32_t>("pm.keys", npairs);
  const uint32_t max_items = (uint32_t)(npairs / 2 + 2 * nlist + 2);   // >= sum over virtual partitions of ceil(c / 2)
  int4 *desc = ctx->scratch_t<int4>("pm.desc", max_items);
  if (!keys || !desc) return LANCE_HIP_ENOMEM;
  {
    ScopedTimer t(ctx, "pm_group");
    hipLaunchKernelGGL(pm_keys_kernel, dim3((unsigned)cdiv(npairs, 256)), dim3(256), 0, ctx->stream, probes, (int64_t)npairs, (int)nprobes,
                       nlist, keys);
    LH_TRY(stable_group(ctx, keys, (int64_t)npairs, (int64_t)npairs, 2 * nlist, 1, pair_starts, pair_idx, (int64_t)npairs, nullptr));
    hipLaunchKernelGGL(pm_item_table_kernel, dim3(1), dim3(256), 0, ctx->stream, pair_starts, 2 * nlist, item_start);
    hipLaunchKernelGGL(pm_item_desc_kernel, dim3((unsigned)cdiv(max_items, 256)), dim3(256), 0, ctx->stream, item_start, pair_starts, pair_idx,
                       2 * nlist, nlist, (int)nprobes, max_items, desc);
  }
  PmArgs a;
  a.q = qs; a.probes = probes; a.pair_starts = pair_starts; a.pair_idx = pair_idx; a.item_start = item_start;
  a.centroids = ix->centroids; a.codebook = ix->codebook; a.part_offsets = ix->part_offsets; a.codes = ix->codes;
  a.d = d; a.m = m; a.nprobes = (int)nprobes; a.nlist = nlist; a.keff = (int)keff;
  a.residual = scan_metric == LANCE_HIP_L2 ? 1 : 0;
  a.round_f16 = ix->dtype == LANCE_HIP_F16 ? 1 : 0;
  a.desc = desc;
  a.unbounded = 0; a.loop = 0; a.allow = allow;
  a.prof = nullptr;
  static const bool pm_prof = getenv("LANCE_HIP_PM_PROF") != nullptr;
  if (pm_prof) {
    a.prof = ctx->scratch_t<unsigned long long>("pm.prof", 8);
    if (a.prof) (void)lh::memset_async(a.prof, 0, 64, ctx->stream);
  }
  a.tglobal = tglobal; a.pool_key = pool_key; a.pool_pos = pool_pos; a.pool_cnt = pool_cnt; a.pool_cap = pool_cap; a.flags = flags;
  const size_t lds_base = pm_lds_base(d, m);
  {
    // pass 0 (bound): every query's nearest partition is streamed once to seed Tglobal[q]; pass 1 (main): all
    // (query, probe) pairs, nearest partition included, prune with that bound from their first row.
    const bool nobound = pm_nobound();
    for (int pass = 0; pass < 2; ++pass) {
      if (nobound && pass == 1 && nprobes == 1) break;
      ScopedTimer t(ctx, pass == 0 ? "ivfpq_scan_c0" : "ivfpq_scan_c1");
      a.cls = pass == 0 ? 0 : (nobound ? 1 : 2);
      a.bound_pass = (pass == 0 && !nobound) ? 1 : 0;
      const size_t lds = lds_base + (size_t)(a.bound_pass ? PM_CAP_BOUND : PM_CAP) * 16;
      // upper bound of sum ceil(c_vp / 2) over the covered virtual partitions; surplus workgroups exit at once
      const size_t cpairs = a.cls == 0 ? (size_t)nq : (a.cls == 1 ? (size_t)nq * (nprobes - 1) : (size_t)nq * nprobes);
      const unsigned grid = (unsigned)(cpairs / 2 + (a.cls == 2 ? 2 : 1) * nlist + 1);
      bool ok = false;
      if (scan_metric == LANCE_HIP_DOT) {
        if (sd == 4) ok = launch_pm_mu<4, METRIC_DOT>(ctx, a, grid, lds);
        else if (sd == 8) ok = launch_pm_mu<8, METRIC_DOT>(ctx, a, grid, lds);
        else if (sd == 16) ok = launch_pm_mu<16, METRIC_DOT>(ctx, a, grid, lds);
      } else {
        if (sd == 4) ok = launch_pm_mu<4, METRIC_L2>(ctx, a, grid, lds);
        else if (sd == 8) ok = launch_pm_mu<8, METRIC_L2>(ctx, a, grid, lds);
        else if (sd == 16) ok = launch_pm_mu<16, METRIC_L2>(ctx, a, grid, lds);
      }
      LH_REQUIRE(ok, "partition-major scan: unsupported shape (m=%d, sd=%d)", m, sd);
      if (a.prof && pass == 1) {
        unsigned long long h[8];
        (void)hipMemcpyAsync(h, a.prof, 64, hipMemcpyDeviceToHost, ctx->stream);
        (void)hipStreamSynchronize(ctx->stream);
        if (h[5]) fprintf(stderr, "[pm prof] items=%llu clocks/item: desc %.0f | q,centroid,residual %.0f | LUT %.0f | scan %.0f | publish %.0f\n", h[5],
                          (double)h[0] / h[5], (double)h[1] / h[5], (double)h[2] / h[5], (double)h[3] / h[5], (double)h[4] / h[5]);
        (void)lh::memset_async(a.prof, 0, 64, ctx->stream);
      }
    }
  }
  {
    SelectOut o;
    o.keff = (int)keff; o.k = (int)k; o.refine = do_refine ? 1 : 0;
    o.out_ids = ids; o.out_dists = dists; o.cand_rid = cand_rid; o.cand_cnt = cand_cnt; o.flags = flags;
    o.part_offsets = ix->part_offsets; o.nlist = nlist;
    ScopedTimer t(ctx, "ivfpq_merge");
    hipLaunchKernelGGL(ivfpq_merge_pm_kernel, dim3(nq), dim3(256), 0, ctx->stream, pool_key, pool_pos, pool_cnt, tglobal, ix->row_ids, pool_cap, o);
  }
  LH_CHECK_HIP(hipGetLastError());
  return LANCE_HIP_OK;
}

}  // namespace lh
