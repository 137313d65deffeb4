// search_qt.hip -- the quantised filter scan for PQ shapes whose 4-query table does not fit in LDS at once: M = 48 / 64 / 96
// sub-quantisers (BASELINE config 3, dbpedia: M = 96, sub-dimension 16 -> [96][256] x 8 bytes = 192 KiB against 160 KiB of LDS).
//
// Same idea as search_q.hip (the filter is a rigorous lower bound in integer arithmetic, survivors are re-evaluated in the
// reference's order by ivfpq_qmerge_kernel) with two changes:
//   * the table is TILED over the sub-quantisers: NT tiles of MT = M / NT sub-quantisers (48 -> 96 KiB, 32 -> 64 KiB); a lane
//     keeps the partial sums of its QT_R rows in registers while the workgroup rebuilds the table for the next tile, so the codes
//     are still read once and nothing but survivors leaves the CU.  Partitions of more than QT_BS * QT_R rows are walked in
//     row blocks, each with its own table builds (the build is 3x the scan of 1000 rows at sub-dimension 16: the blocks are
//     sized so that it amortises);
//   * with M = 96 a packed u16 sum would leave 65535 / 96 = 682 levels per entry.  For the scan that costs ~30 % more survivors
//     (scripts/sim/qfilter_selectivity.py), for the BOUND pass -- dist <= (S + M) / s needs one unit per entry -- it would
//     loosen T by 24 %.  Entries therefore use the full u16 range and the four queries' sums are accumulated in four 32-bit
//     registers (two more VALU per gather; the kernel is bound by the table build, 64 packed VALU per entry at sub-dimension 16).
//     Quantisation is floor-like (q_entry_quantise_floor): dist <= T implies sum <= SE exactly, the limit carries only the
//     f32 rounding head-room.
// Soundness of the limit: e_m <= L'_m * s with L'_m the FMA-evaluated entry; |L'_m - L_m| <= (SD + 1) * 2^-24 * L_m, the
// reference's sequential f32 sum differs from the exact sum by <= M * 2^-24 relative, the f32 evaluation of L' * s - 0.5 by
// <= 2^-23 * 65535 per entry: together < 1.3 units at M = 96, SD = 16, SE = 61440.  LIM = SE + 4.
// Queries without a usable bound (class B) have no exact pair kernel at these sizes (its f32 pair table would be 192 KiB
// too): their segments are marked overflowed and ivfpq_qrescan_kernel scans them with the exact f32 table (96 KiB).
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

#ifndef LH_QT_BS
#define LH_QT_BS 512
#endif
#ifndef LH_QT_R
#define LH_QT_R 4
#endif
#ifndef LH_QT_NT96
#define LH_QT_NT96 3          // tiles at M = 96 (2 x 48 sub-quantisers = 96 KiB; 3 x 32 = 64 KiB; 6 x 16 = 32 KiB) -- A/B builds
#endif
#ifndef LH_QT_NT64
#define LH_QT_NT64 2
#endif
// Shape measured on the C3 probe (profiles/r03_c3_tile_shapes.txt): 512 lanes x 4 rows, three 64 KiB tiles at M = 96 -> two
// workgroups per CU, one building its table while the other gathers: scan 0.378 ms against 0.56 ms for 1024 lanes x 2 rows x
// two 96 KiB tiles (one workgroup per CU, build and gather phases strictly alternating).
constexpr int QT_BS = LH_QT_BS;    // lanes per workgroup
constexpr int QT_R = LH_QT_R;      // rows per lane between table rebuilds (2048 rows per block; C3's partitions average 977)
constexpr int QT_PARTS = QT_BS / 256;
constexpr uint32_t QT_SE = 61440u;          // the bound T maps to SE; entries saturate at 65535 (L > 1.067 T: such a row is out anyway)
constexpr uint32_t QT_LIM = QT_SE + 4u;
constexpr uint32_t QT_SEB = 32768u;         // bound pass: the mean table sum (distance of a random code) maps to SEB
constexpr int QT_BSHIFT = 7;                // histogram bin width 128: 512 bins cover sums 0 .. 65535

// four 32-bit sums from the four u16 fields of one gathered entry
// (v_dot2_u32_u16 with a {1, 0} / {0, 1} selector: one VALU per field instead of mask / shift + add)
__device__ __forceinline__ void qt_add(uint32_t (&acc)[4], uint2 v) {
  typedef unsigned short us2 __attribute__((ext_vector_type(2)));
  const us2 lo = {1, 0}, hi = {0, 1};
  const us2 x = __builtin_bit_cast(us2, v.x), y = __builtin_bit_cast(us2, v.y);
  acc[0] = __builtin_amdgcn_udot2(x, lo, acc[0], false); acc[1] = __builtin_amdgcn_udot2(x, hi, acc[1], false);
  acc[2] = __builtin_amdgcn_udot2(y, lo, acc[2], false); acc[3] = __builtin_amdgcn_udot2(y, hi, acc[3], false);
}

// one tile of the table: lane (c, part) fills sub-quantisers [part * MTP, (part + 1) * MTP) of the tile
template <int SD, int MT>
__device__ __forceinline__ void qt_build_tile(uint2 *lutq, cf4_ptr rq4, const float *__restrict__ codebook, int tile, int c,
                                              int part, f2 s01, f2 s23) {
  constexpr int MTP = MT / QT_PARTS;
#pragma unroll 1
  for (int i = 0; i < MTP; ++i) {
    const int ml = part * MTP + i, mm = tile * MT + ml;
    f2 acc01, acc23;
#if LH_QT_ABLATE == 1      // perf experiment: no arithmetic -- one codeword load and one residual load per entry stay
    { const f4 cv = *reinterpret_cast<const f4 *>(codebook + ((int64_t)mm * 256 + c) * SD); const f4 rv = rq4[mm * SD];
      acc01 = f2{cv.x + rv.x, cv.y + rv.y}; acc23 = f2{cv.z + rv.z, cv.w + rv.w}; }
#elif LH_QT_ABLATE == 2    // perf experiment: no codeword loads (the lane's own index stands in for the codeword)
    { float fake[SD]; for (int u = 0; u < SD; ++u) fake[u] = (float)(c + u + mm); q_entry_acc<SD>(rq4 + mm * SD, fake, acc01, acc23); }
#elif LH_QT_ABLATE == 3    // perf experiment: no residual loads (constants instead): codeword loads + arithmetic
    { acc01 = f2{0.f, 0.f}; acc23 = f2{0.f, 0.f}; const float *cbp = codebook + ((int64_t)mm * 256 + c) * SD;
      for (int u = 0; u < SD; ++u) { const float cv = cbp[u]; const f2 cc = {cv, cv}; const f2 d01 = f2{1.0f + u, 2.0f} + cc, d23 = f2{3.0f, 4.0f + mm} + cc;
        acc01 = __builtin_elementwise_fma(d01, d01, acc01); acc23 = __builtin_elementwise_fma(d23, d23, acc23); } }
#else
    q_entry_acc<SD>(rq4 + mm * SD, codebook + ((int64_t)mm * 256 + c) * SD, acc01, acc23);
#endif
    lutq[ml * 256 + c] = q_entry_quantise_floor(acc01, acc23, s01, s23);
  }
}

// partial sums of one row over one tile: MT / 16 16-byte code words
template <int MT>
__device__ __forceinline__ void qt_row_tile(const uint2 *lutq, const uint8_t *__restrict__ rc, uint32_t (&acc)[4]) {
  constexpr int WT = MT / 16;
  uint4 cw[WT];
#pragma unroll
  for (int w = 0; w < WT; ++w) cw[w] = *reinterpret_cast<const uint4 *>(rc + w * 16);
#pragma unroll
  for (int w = 0; w < WT; ++w) {
    const uint32_t cws[4] = {cw[w].x, cw[w].y, cw[w].z, cw[w].w};
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) qt_add(acc, lutq[(w * 16 + e * 4 + bb) * 256 + ((cws[e] >> (8 * bb)) & 255u)]);
  }
}

#ifndef LH_QT_ABLATE
#define LH_QT_ABLATE 0
#endif
#ifdef LH_QT_PROF
#define QT_PROF_MARK(v) const long long v = wall_clock64()
#define QT_PROF_ADD(slot, dv) do { if (threadIdx.x == 0 && p.prof) atomicAdd(&p.prof[slot], (unsigned long long)(dv)); } while (0)
#else
#define QT_PROF_MARK(v) do { } while (0)
#define QT_PROF_ADD(slot, dv) do { } while (0)
#endif

template <int SD, int MU, int NT>
__global__ __launch_bounds__(QT_BS) void ivfpq_qscan_tiled_kernel(QscanArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int M = MU * 16, MT = M / NT;
  static_assert(M % NT == 0 && MT % 16 == 0 && MT % QT_PARTS == 0, "tile shape");
  static_assert((size_t)MT * 256 * 8 <= 96 * 1024, "table tile must leave LDS for the survivor lists");
  __shared__ __attribute__((aligned(16))) uint2 lutq[MT * 256];
  uint32_t *cand = reinterpret_cast<uint32_t *>(smem);                // [4][Q_CAP]
  uint32_t *misc = cand + 4 * Q_CAP;                                  // [0..3] survivor counts
  float *sc = reinterpret_cast<float *>(misc + 4);                    // [4] SE / T / 65535 (1e30: no such query in this item)
  uint16_t *csum = reinterpret_cast<uint16_t *>(sc + 4);              // [4][Q_CAP] the survivors' integer sums (<= QT_LIM < 65536)
  const uint32_t item = blockIdx.x;
  if (item >= p.item_start[p.nlist]) return;
  const int4 dsc = p.desc[item];
  const int part_id = dsc.x, i0 = dsc.y, cnt = dsc.z;
  const uint32_t off = p.part_offsets[part_id];
  const int np = (int)(p.part_offsets[part_id + 1] - off);
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
      s = fminf((float)QT_SE / T, 1e30f);
    }
    sc[threadIdx.x] = s * (1.0f / 65535.0f);
  }
  __syncthreads();
  const cf4_ptr rq4 = (cf4_ptr)(p.rq + (int64_t)item * p.d);   // scalar loads (see q_entry_acc)
  const int c = threadIdx.x & 255, part = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
  const f4 s4 = *reinterpret_cast<const f4 *>(sc);
  const f2 s01 = {s4.x, s4.y}, s23 = {s4.z, s4.w};
  const uint8_t *pcodes = p.codes + (int64_t)off * M;
  QT_PROF_ADD(3, 1);
  for (int row0 = 0; row0 < np; row0 += QT_BS * QT_R) {
    uint32_t acc[QT_R][4];
#pragma unroll
    for (int r = 0; r < QT_R; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[r][j] = 0u;
#pragma unroll 1
    for (int tile = 0; tile < NT; ++tile) {
      if (tile > 0 || row0 > 0) __syncthreads();   // every lane is done with the previous tile's table
      QT_PROF_MARK(t0);
      qt_build_tile<SD, MT>(lutq, rq4, p.codebook, tile, c, part, s01, s23);
      __syncthreads();
      QT_PROF_MARK(t1);
#pragma unroll
      for (int r = 0; r < QT_R; ++r) {
        const int row = row0 + r * QT_BS + (int)threadIdx.x;
        if (row < np) qt_row_tile<MT>(lutq, pcodes + (int64_t)row * M + tile * MT, acc[r]);
      }
      QT_PROF_ADD(0, t1 - t0);
      QT_PROF_MARK(t2);
      QT_PROF_ADD(1, t2 - t1);
    }
#pragma unroll
    for (int r = 0; r < QT_R; ++r) {
      const int row = row0 + r * QT_BS + (int)threadIdx.x;
      if (row < np) {
        const bool p0 = acc[r][0] <= QT_LIM, p1 = acc[r][1] <= QT_LIM, p2 = acc[r][2] <= QT_LIM, p3 = acc[r][3] <= QT_LIM;
        if ((p0 | p1 | p2 | p3) && row_allowed(p.allow, off + (uint32_t)row)) {
          const uint32_t pos = off + (uint32_t)row;
          if (p0) { const uint32_t slot = atomicAdd(&misc[0], 1u); if (slot < (uint32_t)Q_CAP) { cand[0 * Q_CAP + slot] = pos; csum[0 * Q_CAP + slot] = (uint16_t)acc[r][0]; } }
          if (p1) { const uint32_t slot = atomicAdd(&misc[1], 1u); if (slot < (uint32_t)Q_CAP) { cand[1 * Q_CAP + slot] = pos; csum[1 * Q_CAP + slot] = (uint16_t)acc[r][1]; } }
          if (p2) { const uint32_t slot = atomicAdd(&misc[2], 1u); if (slot < (uint32_t)Q_CAP) { cand[2 * Q_CAP + slot] = pos; csum[2 * Q_CAP + slot] = (uint16_t)acc[r][2]; } }
          if (p3) { const uint32_t slot = atomicAdd(&misc[3], 1u); if (slot < (uint32_t)Q_CAP) { cand[3 * Q_CAP + slot] = pos; csum[3 * Q_CAP + slot] = (uint16_t)acc[r][3]; } }
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < Q_G; ++j) {
    if (j < cnt) {
      const uint32_t raw = misc[j];
      const uint32_t n = min(raw, (uint32_t)Q_CAP);
      const int64_t seg = (int64_t)qj[j] * p.nprobes + rk[j];
      if (threadIdx.x == 0) {
        p.seg_cnt[seg] = raw;   // raw > Q_CAP: survivors were lost -> the rescan kernel redoes this (query, probe) exactly
        if (raw > (uint32_t)Q_CAP) { p.qovf[qj[j]] = 1u; p.ovf[1u + atomicAdd(&p.ovf[0], 1u)] = (uint32_t)seg; }
      }
      for (uint32_t i = threadIdx.x; i < n; i += QT_BS) {
        p.seg_pos[seg * Q_CAP + i] = cand[j * Q_CAP + i];
        p.seg_sum[seg * Q_CAP + i] = csum[j * Q_CAP + i];
      }
    }
  }
}

// ---- bound pass, tiled -------------------------------------------------------------------------------------------------------
// As ivfpq_qbound_kernel: a histogram of integer sums over the query's nearest partition gives an upper bound of the
// k*refine-th smallest ADC distance.  Any subset of the partition's rows gives a valid bound, so only the first QT_BS * QT_R
// rows are histogrammed (one table build per tile).  With floor-like entries dist * s <= S + 2 per entry whatever the
// conversion's rounding mode (e >= L * s - 1.5), so T = (B + 2 M + 4) / s with B the upper edge of the bin where the count
// reaches k*refine; a row is only counted when S < 65535, which rules out a saturated entry.
template <int SD, int MU, int NT>
__global__ __launch_bounds__(QT_BS) void ivfpq_qbound_tiled_kernel(QboundArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int M = MU * 16, MT = M / NT;
  __shared__ __attribute__((aligned(16))) uint2 lutq[MT * 256];
  uint32_t *hist = reinterpret_cast<uint32_t *>(smem);                   // [4][QB_BINS]
  float *sums = reinterpret_cast<float *>(hist + 4 * QB_BINS);            // [4] sum of all table entries
  float *sc = sums + 4;                                                   // [4] scale
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t item = blockIdx.x;
  if (item >= p.item_start[p.nlist]) return;
  const int4 dsc = p.desc[item];
  const int part_id = dsc.x, i0 = dsc.y, cnt = dsc.z;
  const uint32_t off = p.part_offsets[part_id];
  const int np = (int)(p.part_offsets[part_id + 1] - off);
  if (np < p.keff) return;   // uniform: fewer rows than k*refine -> no bound from this partition
  uint32_t qj[Q_G];
#pragma unroll
  for (int j = 0; j < Q_G; ++j) qj[j] = p.pair_idx[i0 + (j < cnt ? j : 0)];
  const cf4_ptr rq4 = (cf4_ptr)(p.rq + (int64_t)item * p.d);   // scalar loads (see q_entry_acc)
  for (int i = threadIdx.x; i < 4 * QB_BINS; i += QT_BS) hist[i] = 0u;
  if (threadIdx.x < 4) sums[threadIdx.x] = 0.0f;
  __syncthreads();
  const int c = threadIdx.x & 255, part = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
  q_mean_entry_sums<QT_BS>(p.rq + (int64_t)item * p.d, p.cb_mean, p.d, sums);   // the scale without a table build (q_common.cuh)
  __syncthreads();
  if (threadIdx.x < Q_G) {
    const float mean = sums[threadIdx.x] + p.cb_mean[p.d];     // sum over m of the mean entry: distance of a random code
    float s = 1e30f;                                           // absent query / degenerate mean: everything saturates
    if ((int)threadIdx.x < cnt && mean > 0.0f && mean < INFINITY) s = fminf((float)QT_SEB / mean, 1e30f);
    sc[threadIdx.x] = s;
  }
  __syncthreads();
  const f4 s4 = *reinterpret_cast<const f4 *>(sc) * (1.0f / 65535.0f);
  const f2 s01 = {s4.x, s4.y}, s23 = {s4.z, s4.w};
  const uint8_t *pcodes = p.codes + (int64_t)off * M;
  const int nrows = min(np, QT_BS * QT_R);
  uint32_t acc[QT_R][4];
#pragma unroll
  for (int r = 0; r < QT_R; ++r)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[r][j] = 0u;
#pragma unroll 1
  for (int tile = 0; tile < NT; ++tile) {
    if (tile > 0) __syncthreads();
    qt_build_tile<SD, MT>(lutq, rq4, p.codebook, tile, c, part, s01, s23);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < QT_R; ++r) {
      const int row = r * QT_BS + (int)threadIdx.x;
      if (row < nrows) qt_row_tile<MT>(lutq, pcodes + (int64_t)row * M + tile * MT, acc[r]);
    }
  }
#pragma unroll
  for (int r = 0; r < QT_R; ++r) {
    const int row = r * QT_BS + (int)threadIdx.x;
    if (row < nrows && row_allowed(p.allow, off + (uint32_t)row)) {
#pragma unroll
      for (int j = 0; j < Q_G; ++j)
        if (acc[r][j] < 65535u) atomicAdd(&hist[j * QB_BINS + (acc[r][j] >> QT_BSHIFT)], 1u);
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
        const float B = (float)(((uint32_t)bin + 1u) << QT_BSHIFT);             // every counted row has S <= B - 1
        const float T = (B + (float)(2 * M + 4)) / sc[wave] * 1.00001f;           // margin >> (SD + M) * 2^-24 for the f32 / FMA rounding terms
        if (T > 0.0f && T < INFINITY) atomicMin(&p.tglobal[qj[wave]], order_key(T));
      }
    }
  }
}

// class B at these sizes: every segment of a query without a bound is handed to the rescan kernel
__global__ __launch_bounds__(256) void q_classb_to_rescan_kernel(const uint32_t *__restrict__ tbound, int64_t npairs, int nprobes,
                                                                  uint32_t *__restrict__ seg_cnt, uint32_t *__restrict__ qovf,
                                                                  uint32_t *__restrict__ ovf) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= npairs) return;
  const int64_t q = i / nprobes;
  if (tbound[q] != 0xFFFFFFFFu) return;
  seg_cnt[i] = 0xFFFFFFFFu;      // class-B pairs were never scanned: they cannot be on the list already
  ovf[1u + atomicAdd(&ovf[0], 1u)] = (uint32_t)i;
  if (i % nprobes == 0) qovf[q] = 1u;
}

// ---- host ---------------------------------------------------------------------------------------------------------------------
bool qscan_tiled_shape(int m, int sd) {
  return (m == 48 || m == 64 || m == 96) && (sd == 4 || sd == 8 || sd == 16);
}

template <int SD, int MU, int NT>
static void launch_qscan_tiled(lance_hip_ctx *ctx, const QscanArgs &a, unsigned grid) {
  const size_t lds = (size_t)4 * Q_CAP * 4 + 8 * 4 + (size_t)4 * Q_CAP * 2;
  hipLaunchKernelGGL((ivfpq_qscan_tiled_kernel<SD, MU, NT>), dim3(grid), dim3(QT_BS), lds, ctx->stream, a);
}

template <int SD>
static bool launch_qscan_tiled_sd(lance_hip_ctx *ctx, const QscanArgs &a, int m, unsigned grid) {
  if (m == 48) { launch_qscan_tiled<SD, 3, 1>(ctx, a, grid); return true; }
  if (m == 64) { launch_qscan_tiled<SD, 4, LH_QT_NT64>(ctx, a, grid); return true; }
  if (m == 96) { launch_qscan_tiled<SD, 6, LH_QT_NT96>(ctx, a, grid); return true; }
  return false;
}

bool qscan_tiled_launch(lance_hip_ctx *ctx, const QscanArgs &a, int m, int sd, unsigned grid) {
  if (sd == 4) return launch_qscan_tiled_sd<4>(ctx, a, m, grid);
  if (sd == 8) return launch_qscan_tiled_sd<8>(ctx, a, m, grid);
  if (sd == 16) return launch_qscan_tiled_sd<16>(ctx, a, m, grid);
  return false;
}

template <int SD, int MU, int NT>
static void launch_qbound_tiled(lance_hip_ctx *ctx, const QboundArgs &a, unsigned grid) {
  const size_t lds = (size_t)4 * QB_BINS * 4 + 8 * 4;
  hipLaunchKernelGGL((ivfpq_qbound_tiled_kernel<SD, MU, NT>), dim3(grid), dim3(QT_BS), lds, ctx->stream, a);
}

template <int SD>
static bool launch_qbound_tiled_sd(lance_hip_ctx *ctx, const QboundArgs &a, int m, unsigned grid) {
  if (m == 48) { launch_qbound_tiled<SD, 3, 1>(ctx, a, grid); return true; }
  if (m == 64) { launch_qbound_tiled<SD, 4, LH_QT_NT64>(ctx, a, grid); return true; }
  if (m == 96) { launch_qbound_tiled<SD, 6, LH_QT_NT96>(ctx, a, grid); return true; }
  return false;
}

bool qbound_tiled_launch(lance_hip_ctx *ctx, const QboundArgs &a, int m, int sd, unsigned grid) {
  if (sd == 4) return launch_qbound_tiled_sd<4>(ctx, a, m, grid);
  if (sd == 8) return launch_qbound_tiled_sd<8>(ctx, a, m, grid);
  if (sd == 16) return launch_qbound_tiled_sd<16>(ctx, a, m, grid);
  return false;
}

int qscan_classb_to_rescan(lance_hip_ctx *ctx, const uint32_t *tbound, uint32_t nq, uint32_t nprobes, uint32_t *seg_cnt, uint32_t *qovf) {
  const int64_t npairs = (int64_t)nq * nprobes;
  uint32_t *ovf = ctx->scratch_t<uint32_t>("q.ovf", (size_t)npairs + 1);   // the slot qscan_launch zeroed and the scan kernels filled
  if (!ovf) return LANCE_HIP_ENOMEM;
  hipLaunchKernelGGL(q_classb_to_rescan_kernel, dim3((unsigned)cdiv((uint64_t)npairs, 256)), dim3(256), 0, ctx->stream, tbound, npairs, (int)nprobes,
                     seg_cnt, qovf, ovf);
  return LANCE_HIP_OK;
}


// ==== per-query tables + per-row bias (LANCE_HIP_QPT=1; round 3, written against scripts/sim/pqt_filter_spec.py, NOT YET RUN ON
// HARDWARE when the round closed -- DESIGN.md section 8) =====================================================================
// With q~ = q - g, cen~ = cen_p - g (g = the mean centroid: any fixed vector leaves r = q - cen_p unchanged, this one removes the
// data's common offset) the residual table entry splits into
//     ||r_m - c||^2 = ||q~_m - c||^2 + 2 cen~_pm . c + (||cen~_pm||^2 - 2 cen~_pm . q~_m)
// i.e.  dist(q, row) = sum_m A_q[m][code_m] + beta_row + kappa_qp  with a table per QUERY, a constant per stored ROW and a scalar
// per pair.  The filter  dist <= T  becomes  sum_m e_q[m][code_m] <= s_q (T - kappa_qp - beta_row) + slack:  the integer table is
// built once per query (q_pt_table_kernel) and an item LOADS its four queries' tiles instead of computing them.
//   e       = floor(min(A^ s_q, 65535)), A^ the f32 FMA chain, s_q = SE / Theta_q, Theta_q = max over the query's probes of
//             (T - kappa - min beta of the partition): every row that can pass has all its entries below saturation;
//   beta    = f32 of an f64 sum (q_pt_row_beta_kernel, once per index), kappa likewise (q_pt_scale_kernel, per pair);
//   slack   = 2 + u s_q (10 (|T| + |kappa| + max|beta| + |q~|^2) + (SD + M + 6) Theta_q) units, u = 2^-24; a pair whose slack
//             exceeds PT_CAP units is handed to the exact rescan kernel like an overflowed segment;
//   stored sum S' = floor(sum e + s_q (beta + kappa)) ~ s_q dist: S' - 9 <= s_q dist <= S' + M + 10 -- inside the merge kernel's
//             existing cut slack for these shapes (2 M + 8).
constexpr float PT_CAP = 8.0f;

struct PtArgs {
  const uint16_t *tab;     // [nq][M][256] per-query integer tables
  const float *sq;         // [nq] scale (0: the query has no table)
  const float *kap;        // [nq * nprobes] kappa of the pair
  const float *pslack;     // [nq * nprobes] slack of the pair in units
  const float *row_beta;   // [n] per stored row
};

// g[dim] = mean over the centroids (f64 accumulation)
__global__ __launch_bounds__(256) void q_pt_mean_kernel(const float *__restrict__ cent, int nlist, int d, float *__restrict__ g) {
  const int dim = blockIdx.x * 256 + threadIdx.x;
  if (dim >= d) return;
  double acc = 0.0;
  for (int p = 0; p < nlist; ++p) acc += (double)cent[(int64_t)p * d + dim];
  g[dim] = (float)(acc / (double)nlist);
}

__global__ __launch_bounds__(256) void q_pt_translate_kernel(const float *__restrict__ cent, const float *__restrict__ g, int64_t total, int d,
                                                              float *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < total) out[i] = cent[i] - g[i % d];
}

// one workgroup per partition: beta of every stored row (f64 sum, one rounding), the partition's min beta and max |beta|
__global__ __launch_bounds__(256) void q_pt_row_beta_kernel(const float *__restrict__ cen_t, const float *__restrict__ codebook,
                                                             const uint8_t *__restrict__ codes, const uint32_t *__restrict__ part_offsets,
                                                             int d, int m, float *__restrict__ row_beta, float *__restrict__ beta_min,
                                                             float *__restrict__ beta_abs) {
  __shared__ float s_min[4], s_abs[4];
  const int p = blockIdx.x, sd = d / m;
  const uint32_t off = part_offsets[p];
  const int np = (int)(part_offsets[p + 1] - off);
  const float *ct = cen_t + (int64_t)p * d;
  float lmin = INFINITY, labs = 0.0f;
  for (int row = threadIdx.x; row < np; row += 256) {
    const uint8_t *rc = codes + ((int64_t)off + row) * m;
    double acc = 0.0;
    for (int mm = 0; mm < m; ++mm) {
      const float *cw = codebook + ((int64_t)mm * 256 + rc[mm]) * sd;
      for (int u = 0; u < sd; ++u) acc += 2.0 * (double)ct[mm * sd + u] * (double)cw[u];
    }
    const float b = (float)acc;
    row_beta[(int64_t)off + row] = b;
    lmin = fminf(lmin, b); labs = fmaxf(labs, fabsf(b));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { lmin = fminf(lmin, __shfl_xor(lmin, o, 64)); labs = fmaxf(labs, __shfl_xor(labs, o, 64)); }
  if ((threadIdx.x & 63) == 0) { s_min[threadIdx.x >> 6] = lmin; s_abs[threadIdx.x >> 6] = labs; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float mn = fminf(fminf(s_min[0], s_min[1]), fminf(s_min[2], s_min[3]));
    beta_min[p] = np > 0 ? mn : 0.0f;
    beta_abs[p] = fmaxf(fmaxf(s_abs[0], s_abs[1]), fmaxf(s_abs[2], s_abs[3]));
  }
}

// beta of an average code: 2 cen~_p . mu per partition (LANCE_HIP_QPT=2: the pre-scale of the shared tables)
__global__ __launch_bounds__(256) void q_pt_beta_mean_kernel(const float *__restrict__ cen_t, const float *__restrict__ mu, int nlist, int d,
                                                              float *__restrict__ beta_mean) {
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (p >= nlist) return;
  double acc = 0.0;
  for (int dim = lane; dim < d; dim += 64) acc += 2.0 * (double)cen_t[(int64_t)p * d + dim] * (double)mu[dim];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) beta_mean[p] = (float)acc;
}

// one wave per (query, probe) pair: kappa = |cen~_p|^2 - 2 cen~_p . q~ (f64 sum, one rounding); the pair of rank 0 also leaves |q~|^2
__global__ __launch_bounds__(256) void q_pt_kappa_kernel(const float *__restrict__ qs, const float *__restrict__ g, const float *__restrict__ cen_t,
                                                          const uint32_t *__restrict__ probes, int64_t npairs, int nprobes, int d,
                                                          float *__restrict__ kap, double *__restrict__ qn2_out,
                                                          const float *__restrict__ mu = nullptr, double *__restrict__ qmu_out = nullptr) {
  const int64_t pr = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (pr >= npairs) return;
  const int64_t q = pr / nprobes;
  const float *ct = cen_t + (int64_t)probes[pr] * d, *qv = qs + q * d;
  double acc = 0.0, qn2 = 0.0, qmu = 0.0;
#pragma unroll 4
  for (int dim = lane; dim < d; dim += 64) {
    const float v = qv[dim] - g[dim];          // q~ (the same f32 subtraction as the table kernel's)
    const double c = (double)ct[dim];
    acc += c * c - 2.0 * c * (double)v;
    qn2 += (double)v * (double)v;
    if (mu) qmu += (double)v * (double)mu[dim];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { acc += __shfl_xor(acc, o, 64); qn2 += __shfl_xor(qn2, o, 64); qmu += __shfl_xor(qmu, o, 64); }
  if (lane == 0) {
    kap[pr] = (float)acc;
    if (pr % nprobes == 0) { qn2_out[q] = qn2; if (qmu_out) qmu_out[q] = qmu; }
  }
}

// one lane per query: the scale from Theta = max over the probes of (T - kappa - min beta of the partition), the slack of every pair
__global__ __launch_bounds__(256) void q_pt_scale_kernel(const float *__restrict__ beta_min, const float *__restrict__ beta_abs,
                                                          const uint32_t *__restrict__ probes, const uint32_t *__restrict__ tbound, int nq,
                                                          int nprobes, int sd_plus_m, const float *__restrict__ kap, const double *__restrict__ qn2_in,
                                                          float *__restrict__ sq, float *__restrict__ pslack) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= nq) return;
  const uint32_t tb = tbound[q];
  if (tb == 0xFFFFFFFFu) {   // class B: no bound, no table (its segments go to the rescan kernel anyway)
    sq[q] = 0.0f;
    return;
  }
  const double T = (double)key_to_float(tb), qn2 = qn2_in[q];
  double theta = 0.0;
  for (int rank = 0; rank < nprobes; ++rank) {
    const uint32_t p = probes[(int64_t)q * nprobes + rank];
    theta = fmax(theta, T - (double)kap[(int64_t)q * nprobes + rank] - (double)beta_min[p]);
  }
  // theta <= 0 or not finite: no usable scale -- every pair of the query is sent to the rescan kernel (slack = inf)
  const bool ok = theta > 0.0 && theta < 1e300;
  const float s = ok ? (float)((double)QT_SE / theta) : 0.0f;
  sq[q] = s;
  const double u = 5.9604644775390625e-8;   // 2^-24
  for (int rank = 0; rank < nprobes; ++rank) {
    const uint32_t p = probes[(int64_t)q * nprobes + rank];
    const double kf = (double)kap[(int64_t)q * nprobes + rank];
    const double sl = 2.0 + u * (double)s * (10.0 * (fabs(T) + fabs(kf) + (double)beta_abs[p] + qn2) + (double)(sd_plus_m + 6) * theta);
    pslack[(int64_t)q * nprobes + rank] = ok ? (float)sl : INFINITY;
  }
}

// grid (nq, slices of the sub-quantisers), lane = codeword: the query's integer table.  (The first version launched one 256-lane
// workgroup per (sub-quantiser, query) -- 96,000 workgroups of 16 multiply-adds per lane at C3: 0.12 ms per 1000 queries with the
// kappa and scale kernels, a third of the filter scan's time, r04c.)
template <int SD>
__global__ __launch_bounds__(256) void q_pt_table_kernel(const float *__restrict__ qs, const float *__restrict__ g, const float *__restrict__ sq,
                                                          const float *__restrict__ codebook, int d, int m, int mper, uint16_t *__restrict__ tab) {
  const int q = blockIdx.x, c = threadIdx.x;
  const float s = sq[q];
  if (!(s > 0.0f)) return;   // uniform
  const int m0 = blockIdx.y * mper, m1 = min(m, m0 + mper);
#pragma unroll 4
  for (int mm = m0; mm < m1; ++mm) {
    const float *qv = qs + (int64_t)q * d + mm * SD, *gv = g + mm * SD;
    const float *cw = codebook + ((int64_t)mm * 256 + c) * SD;
    float cwv[SD];
#pragma unroll
    for (int u = 0; u < SD / 4; ++u) *reinterpret_cast<f4 *>(&cwv[4 * u]) = *reinterpret_cast<const f4 *>(cw + 4 * u);
    float acc = 0.0f;
#pragma unroll
    for (int u = 0; u < SD; ++u) {
      const float diff = (qv[u] - gv[u]) - cwv[u];      // q~ = q - g rounded once, as in q_pt_kappa_kernel
      acc = fmaf(diff, diff, acc);
    }
    float z = fminf(acc * s, 65535.0f);
    z = z >= 0.0f ? z : 0.0f;                       // NaN -> 0: the row survives and the exact pass decides
    tab[((int64_t)q * m + mm) * 256 + c] = (uint16_t)(uint32_t)z;   // truncation = floor
  }
}

template <int MU, int NT>
__global__ __launch_bounds__(QT_BS) void ivfpq_qscan_tiled_pt_kernel(QscanArgs p, PtArgs t) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int M = MU * 16, MT = M / NT;
  static_assert(M % NT == 0 && MT % 16 == 0, "tile shape");
  __shared__ __attribute__((aligned(16))) uint2 lutq[MT * 256];
  uint32_t *cand = reinterpret_cast<uint32_t *>(smem);                // [4][Q_CAP]
  uint32_t *misc = cand + 4 * Q_CAP;                                  // [0..3] survivor counts, [4..7] 1: the pair goes to the rescan kernel
  float *sc = reinterpret_cast<float *>(misc + 8);                    // [4] s_q
  float *thr = sc + 4;                                                // [4] s_q (T - kappa) + slack  (-1e30: nothing passes)
  float *skap = thr + 4;                                              // [4] s_q kappa
  uint16_t *csum = reinterpret_cast<uint16_t *>(skap + 4);            // [4][Q_CAP] the survivors' sums ~ s_q dist
  const uint32_t item = blockIdx.x;
  if (item >= p.item_start[p.nlist]) return;
  const int4 dsc = p.desc[item];
  const int part_id = dsc.x, i0 = dsc.y, cnt = dsc.z;
  const uint32_t off = p.part_offsets[part_id];
  const int np = (int)(p.part_offsets[part_id + 1] - off);
  if (np == 0) return;   // uniform; seg_cnt stays 0
  uint32_t qj[Q_G], pr[Q_G];
#pragma unroll
  for (int j = 0; j < Q_G; ++j) {
    pr[j] = p.pair_idx[i0 + (j < cnt ? j : 0)];
    qj[j] = pr[j] / (uint32_t)p.nprobes;
  }
  if (threadIdx.x < Q_G) {
    const int j = threadIdx.x;
    misc[j] = 0; misc[4 + j] = 0;
    float s = 0.0f, th = -1e30f, sk = 0.0f;
    if (j < cnt) {
      const float sj = t.sq[qj[j]], sl = t.pslack[pr[j]];
      if (sj > 0.0f && sl <= PT_CAP) {
        const float T = key_to_float(p.tbound[qj[j]]), kp = t.kap[pr[j]];
        s = sj; th = sj * (T - kp) + sl; sk = sj * kp;
      } else {
        misc[4 + j] = 1u;
      }
    }
    sc[j] = s; thr[j] = th; skap[j] = sk;
  }
  __syncthreads();
  const f4 s4 = *reinterpret_cast<const f4 *>(sc), th4 = *reinterpret_cast<const f4 *>(thr), sk4 = *reinterpret_cast<const f4 *>(skap);
  const uint16_t *tq0 = t.tab + (int64_t)qj[0] * M * 256, *tq1 = t.tab + (int64_t)qj[1] * M * 256;
  const uint16_t *tq2 = t.tab + (int64_t)qj[2] * M * 256, *tq3 = t.tab + (int64_t)qj[3] * M * 256;
  const uint8_t *pcodes = p.codes + (int64_t)off * M;
  for (int row0 = 0; row0 < np; row0 += QT_BS * QT_R) {
    uint32_t acc[QT_R][4];
#pragma unroll
    for (int r = 0; r < QT_R; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[r][j] = 0u;
#pragma unroll 1
    for (int tile = 0; tile < NT; ++tile) {
      if (tile > 0 || row0 > 0) __syncthreads();   // every lane is done with the previous tile's table
      // four consecutive codewords of one sub-quantiser per step: one 8-byte load from each query's table (a wave reads 512
      // contiguous bytes per query), a 4 x 4 transpose of the u16 entries in registers (v_perm_b32), two 16-byte LDS stores.
      // (The first version moved one u16 per load: 192 load instructions per lane and row block, and ran no faster than the
      // table build it replaced -- gpurun r04a.)
      constexpr int GROUPS = MT * 64;
#pragma unroll 2
      for (int gi = (int)threadIdx.x; gi < GROUPS; gi += QT_BS) {
        const int ml = gi >> 6, c4 = (gi & 63) * 4, idx = (tile * MT + ml) * 256 + c4;
        const uint2 a0 = *reinterpret_cast<const uint2 *>(tq0 + idx), a1 = *reinterpret_cast<const uint2 *>(tq1 + idx);
        const uint2 a2 = *reinterpret_cast<const uint2 *>(tq2 + idx), a3 = *reinterpret_cast<const uint2 *>(tq3 + idx);
        constexpr uint32_t LO = 0x05040100u, HI = 0x07060302u;   // low / high halves of (second operand, first operand)
        uint4 o01, o23;
        o01.x = __builtin_amdgcn_perm(a1.x, a0.x, LO); o01.y = __builtin_amdgcn_perm(a3.x, a2.x, LO);
        o01.z = __builtin_amdgcn_perm(a1.x, a0.x, HI); o01.w = __builtin_amdgcn_perm(a3.x, a2.x, HI);
        o23.x = __builtin_amdgcn_perm(a1.y, a0.y, LO); o23.y = __builtin_amdgcn_perm(a3.y, a2.y, LO);
        o23.z = __builtin_amdgcn_perm(a1.y, a0.y, HI); o23.w = __builtin_amdgcn_perm(a3.y, a2.y, HI);
        *reinterpret_cast<uint4 *>(&lutq[ml * 256 + c4]) = o01;
        *reinterpret_cast<uint4 *>(&lutq[ml * 256 + c4 + 2]) = o23;
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < QT_R; ++r) {
        const int row = row0 + r * QT_BS + (int)threadIdx.x;
        if (row < np) qt_row_tile<MT>(lutq, pcodes + (int64_t)row * M + tile * MT, acc[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < QT_R; ++r) {
      const int row = row0 + r * QT_BS + (int)threadIdx.x;
      if (row < np) {
        const float beta = t.row_beta[(int64_t)off + row];
        const float a0 = (float)acc[r][0], a1 = (float)acc[r][1], a2 = (float)acc[r][2], a3 = (float)acc[r][3];   // exact: sums < 2^24
        const bool p0 = a0 <= fmaf(-s4.x, beta, th4.x), p1 = a1 <= fmaf(-s4.y, beta, th4.y);
        const bool p2 = a2 <= fmaf(-s4.z, beta, th4.z), p3 = a3 <= fmaf(-s4.w, beta, th4.w);
        if ((p0 | p1 | p2 | p3) && row_allowed(p.allow, off + (uint32_t)row)) {
          const uint32_t pos = off + (uint32_t)row;
          // stored sum ~ s_q dist (what the merge kernel's cut reads): floor(sum e + s_q (beta + kappa)), clamped to the u16 range
          if (p0) { const uint32_t slot = atomicAdd(&misc[0], 1u); if (slot < (uint32_t)Q_CAP) { cand[0 * Q_CAP + slot] = pos; csum[0 * Q_CAP + slot] = (uint16_t)(uint32_t)fminf(fmaxf(a0 + fmaf(s4.x, beta, sk4.x), 0.0f), 65535.0f); } }
          if (p1) { const uint32_t slot = atomicAdd(&misc[1], 1u); if (slot < (uint32_t)Q_CAP) { cand[1 * Q_CAP + slot] = pos; csum[1 * Q_CAP + slot] = (uint16_t)(uint32_t)fminf(fmaxf(a1 + fmaf(s4.y, beta, sk4.y), 0.0f), 65535.0f); } }
          if (p2) { const uint32_t slot = atomicAdd(&misc[2], 1u); if (slot < (uint32_t)Q_CAP) { cand[2 * Q_CAP + slot] = pos; csum[2 * Q_CAP + slot] = (uint16_t)(uint32_t)fminf(fmaxf(a2 + fmaf(s4.z, beta, sk4.z), 0.0f), 65535.0f); } }
          if (p3) { const uint32_t slot = atomicAdd(&misc[3], 1u); if (slot < (uint32_t)Q_CAP) { cand[3 * Q_CAP + slot] = pos; csum[3 * Q_CAP + slot] = (uint16_t)(uint32_t)fminf(fmaxf(a3 + fmaf(s4.w, beta, sk4.w), 0.0f), 65535.0f); } }
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < Q_G; ++j) {
    if (j < cnt) {
      const uint32_t raw = misc[4 + j] ? 0xFFFFFFFFu : misc[j];   // no table / slack over the cap: the rescan kernel does this pair exactly
      const uint32_t n = raw > (uint32_t)Q_CAP ? 0u : raw;
      const int64_t seg = (int64_t)pr[j];
      if (threadIdx.x == 0) {
        p.seg_cnt[seg] = raw;
        if (raw > (uint32_t)Q_CAP) { p.qovf[qj[j]] = 1u; p.ovf[1u + atomicAdd(&p.ovf[0], 1u)] = (uint32_t)seg; }
      }
      for (uint32_t i = threadIdx.x; i < n; i += QT_BS) {
        p.seg_pos[seg * Q_CAP + i] = cand[j * Q_CAP + i];
        p.seg_sum[seg * Q_CAP + i] = csum[j * Q_CAP + i];
      }
    }
  }
}

// ---- host ------------------------------------------------------------------------------------------------------------------
// 0: off; 1: tables after the bound pass (scale from T); 2: tables before the bound pass, shared by both passes
static int qscan_pt_env() {
  static const int mode = [] { const char *e = getenv("LANCE_HIP_QPT"); return e ? (e[0] == '2' ? 2 : (e[0] != '0' ? 1 : 0)) : 2; }();   // unset: mode 2 (r04g: C3 716 k -> 788 k q/s at nprobes 10, 348 k -> 508 k at 50)
  return mode;
}
int qscan_pt_mode(const lance_hip_index *ix) { return qscan_pt_enabled(ix) ? qscan_pt_env() : 0; }

bool qscan_pt_enabled(const lance_hip_index *ix) {
  const bool on = qscan_pt_env() != 0;
  if (!on || !ix || ix->m == 0 || ix->nbits != 8) return false;
  const int m = (int)ix->m, sd = (int)(ix->d / ix->m);
  if (!qscan_tiled_shape(m, sd)) return false;
  if (ix->dtype == LANCE_HIP_F16) return false;      // the reference rounds the residual to f16 there: r is not q - cen any more
  return ix->metric == LANCE_HIP_L2 || ix->metric == LANCE_HIP_COSINE;
}

static int qscan_pt_prepare(lance_hip_ctx *ctx, lance_hip_index *ix) {
  std::lock_guard<std::mutex> lk(ix->lazy_mu);   // the first search of any context builds the constants, the others wait for it
  if (ix->pt) return LANCE_HIP_OK;
  const int d = (int)ix->d, m = (int)ix->m, nlist = (int)ix->nlist;
  auto *pc = new lance_hip_index::PtConst();
  bool ok = hipMalloc(reinterpret_cast<void **>(&pc->g), (size_t)d * 4) == hipSuccess;
  ok = ok && hipMalloc(reinterpret_cast<void **>(&pc->cen_t), (size_t)nlist * d * 4) == hipSuccess;
  ok = ok && hipMalloc(reinterpret_cast<void **>(&pc->row_beta), (size_t)(ix->n ? ix->n : 1) * 4) == hipSuccess;
  ok = ok && hipMalloc(reinterpret_cast<void **>(&pc->beta_min), (size_t)nlist * 4) == hipSuccess;
  ok = ok && hipMalloc(reinterpret_cast<void **>(&pc->beta_abs), (size_t)nlist * 4) == hipSuccess;
  ok = ok && hipMalloc(reinterpret_cast<void **>(&pc->beta_mean), (size_t)nlist * 4) == hipSuccess;
  auto drop = [&]() {
    (void)hipFree(pc->g); (void)hipFree(pc->cen_t); (void)hipFree(pc->row_beta); (void)hipFree(pc->beta_min); (void)hipFree(pc->beta_abs);
    (void)hipFree(pc->beta_mean);
    delete pc;
  };
  if (!ok) { drop(); set_error("per-query tables: out of device memory for the index constants"); return LANCE_HIP_ENOMEM; }
  hipLaunchKernelGGL(q_pt_mean_kernel, dim3((unsigned)cdiv((uint64_t)d, 256)), dim3(256), 0, ctx->stream, ix->centroids, nlist, d, pc->g);
  hipLaunchKernelGGL(q_pt_translate_kernel, dim3((unsigned)cdiv((uint64_t)nlist * d, 256)), dim3(256), 0, ctx->stream, ix->centroids, pc->g,
                     (int64_t)nlist * d, d, pc->cen_t);
  hipLaunchKernelGGL(q_pt_row_beta_kernel, dim3((unsigned)nlist), dim3(256), 0, ctx->stream, pc->cen_t, ix->codebook, ix->codes, ix->part_offsets, d, m,
                     pc->row_beta, pc->beta_min, pc->beta_abs);
  if (ix->cb_mean)
    hipLaunchKernelGGL(q_pt_beta_mean_kernel, dim3((unsigned)cdiv((uint64_t)nlist, 4)), dim3(256), 0, ctx->stream, pc->cen_t, ix->cb_mean, nlist, d, pc->beta_mean);
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {   // other contexts (streams) search the same index
    drop();
    set_error("per-query tables: building the index constants failed");
    return LANCE_HIP_ERUNTIME;
  }
  ix->pt = pc;   // published complete; lance_hip_index's destructor frees it
  return LANCE_HIP_OK;
}


// ==== LANCE_HIP_QPT=2: the per-query tables are built BEFORE the bound pass and serve both passes ==============================
// The bound pass spent a table build per (query, nearest partition) item -- about one per query at C3 (1000 queries over 1024
// partitions), as much arithmetic as the per-query tables of the main pass.  Any scale is sound for the filter (it only sets how
// tight it is), so the scale is fixed before T is known: T_pre = twice the distance of an AVERAGE code of the nearest partition,
//   mean dist = (|q~|^2 - 2 q~.mu + nu) + 2 cen~_p0.mu + kappa_{q,p0}      (mu, nu: the codebook means of the index, cb_mean),
// Theta_pre = max over the probes of (T_pre - kappa - min beta), s_q = SE / Theta_pre.  The bound kernel loads the item's four
// tables, scans the first QT_BS * QT_R rows of the nearest partition and histograms S' = sum e + s (beta + kappa) ~ s dist of the
// rows without a saturated entry (sum e < 65535); with keff rows at or below a bin's upper edge B the keff-th smallest ADC
// distance is at most (B + M + 10 + slack) / s (the same two-sided relation the merge kernel's cut uses).  After the bound pass a
// per-pair kernel computes the slack with the actual T and checks that no passing row can hold a saturated entry
// (s (T - kappa - min beta) <= SE); a pair that fails goes to the exact rescan, as in mode 1.
__global__ __launch_bounds__(256) void q_pt_prescale_kernel(const float *__restrict__ beta_min, const float *__restrict__ beta_abs,
                                                             const float *__restrict__ beta_mean, const uint32_t *__restrict__ probes, int nq,
                                                             int nprobes, int sd_plus_m, const float *__restrict__ nu_ptr, const float *__restrict__ kap,
                                                             const double *__restrict__ qn2_in, const double *__restrict__ qmu_in,
                                                             float *__restrict__ sq, float *__restrict__ theta_out, float *__restrict__ bslack) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= nq) return;
  const uint32_t p0 = probes[(int64_t)q * nprobes];
  const double k0 = (double)kap[(int64_t)q * nprobes], qn2 = qn2_in[q];
  const double abar = qn2 - 2.0 * qmu_in[q] + (double)nu_ptr[0];
  const double tpre = 2.0 * (abar + k0 + (double)beta_mean[p0]);
  double theta = 0.0;
  for (int rank = 0; rank < nprobes; ++rank) {
    const uint32_t p = probes[(int64_t)q * nprobes + rank];
    theta = fmax(theta, tpre - (double)kap[(int64_t)q * nprobes + rank] - (double)beta_min[p]);
  }
  const bool ok = tpre > 0.0 && theta > 0.0 && theta < 1e300;
  const float s = ok ? (float)((double)QT_SE / theta) : 0.0f;
  sq[q] = s;
  theta_out[q] = ok ? (float)theta : 0.0f;
  const double u = 5.9604644775390625e-8;   // 2^-24
  bslack[q] = ok ? (float)(2.0 + u * (double)s * (10.0 * (tpre + fabs(k0) + (double)beta_abs[p0] + qn2) + (double)(sd_plus_m + 6) * theta)) : INFINITY;
}

__global__ __launch_bounds__(256) void q_pt_slack_kernel(const float *__restrict__ beta_min, const float *__restrict__ beta_abs,
                                                          const uint32_t *__restrict__ probes, const uint32_t *__restrict__ tbound, int64_t npairs,
                                                          int nprobes, int sd_plus_m, const float *__restrict__ kap, const double *__restrict__ qn2_in,
                                                          const float *__restrict__ sq, const float *__restrict__ theta_in, float *__restrict__ pslack) {
  const int64_t pr = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (pr >= npairs) return;
  const int64_t q = pr / nprobes;
  const uint32_t tb = tbound[q];
  const float s = sq[q];
  float out = INFINITY;
  if (tb != 0xFFFFFFFFu && s > 0.0f) {
    const double T = (double)key_to_float(tb), kf = (double)kap[pr], theta = (double)theta_in[q];
    const uint32_t p = probes[pr];
    const double u = 5.9604644775390625e-8;
    const double sl = 2.0 + u * (double)s * (10.0 * (fabs(T) + fabs(kf) + (double)beta_abs[p] + qn2_in[q]) + (double)(sd_plus_m + 6) * theta);
    // every row that can pass must have all its entries below saturation: s (T - kappa - min beta) <= SE
    const double thmax = (double)s * (T - kf - (double)beta_min[p]);
    if (thmax <= (double)QT_SE * 1.000001 && sl < 1e30) out = (float)sl;
  }
  pslack[pr] = out;
}

struct PtBoundArgs {
  const uint16_t *tab;
  const float *sq, *kap, *bslack, *row_beta;
  int nprobes;
};

template <int MU, int NT>
__global__ __launch_bounds__(QT_BS) void ivfpq_qbound_tiled_pt_kernel(QboundArgs p, PtBoundArgs t) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int M = MU * 16, MT = M / NT;
  __shared__ __attribute__((aligned(16))) uint2 lutq[MT * 256];
  uint32_t *hist = reinterpret_cast<uint32_t *>(smem);                   // [4][QB_BINS]
  float *sc = reinterpret_cast<float *>(hist + 4 * QB_BINS);              // [4] s_j (0: no table)
  float *skap = sc + 4;                                                   // [4] s_j kappa_j
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t item = blockIdx.x;
  if (item >= p.item_start[p.nlist]) return;
  const int4 dsc = p.desc[item];
  const int part_id = dsc.x, i0 = dsc.y, cnt = dsc.z;
  const uint32_t off = p.part_offsets[part_id];
  const int np = (int)(p.part_offsets[part_id + 1] - off);
  if (np < p.keff) return;   // uniform: fewer rows than k*refine -> no bound from this partition
  uint32_t qj[Q_G];
#pragma unroll
  for (int j = 0; j < Q_G; ++j) qj[j] = p.pair_idx[i0 + (j < cnt ? j : 0)];
  for (int i = threadIdx.x; i < 4 * QB_BINS; i += QT_BS) hist[i] = 0u;
  if (threadIdx.x < Q_G) {
    const int j = threadIdx.x;
    const float s = j < cnt ? t.sq[qj[j]] : 0.0f;
    sc[j] = s;
    skap[j] = s * t.kap[(int64_t)qj[j] * t.nprobes];      // the nearest partition is probe rank 0
  }
  __syncthreads();
  const f4 s4 = *reinterpret_cast<const f4 *>(sc), sk4 = *reinterpret_cast<const f4 *>(skap);
  const uint16_t *tq0 = t.tab + (int64_t)qj[0] * M * 256, *tq1 = t.tab + (int64_t)qj[1] * M * 256;
  const uint16_t *tq2 = t.tab + (int64_t)qj[2] * M * 256, *tq3 = t.tab + (int64_t)qj[3] * M * 256;
  const uint8_t *pcodes = p.codes + (int64_t)off * M;
  const int nrows = min(np, QT_BS * QT_R);
  uint32_t acc[QT_R][4];
#pragma unroll
  for (int r = 0; r < QT_R; ++r)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[r][j] = 0u;
#pragma unroll 1
  for (int tile = 0; tile < NT; ++tile) {
    if (tile > 0) __syncthreads();
    constexpr int GROUPS = MT * 64;
#pragma unroll 2
    for (int gi = (int)threadIdx.x; gi < GROUPS; gi += QT_BS) {
      const int ml = gi >> 6, c4 = (gi & 63) * 4, idx = (tile * MT + ml) * 256 + c4;
      const uint2 a0 = *reinterpret_cast<const uint2 *>(tq0 + idx), a1 = *reinterpret_cast<const uint2 *>(tq1 + idx);
      const uint2 a2 = *reinterpret_cast<const uint2 *>(tq2 + idx), a3 = *reinterpret_cast<const uint2 *>(tq3 + idx);
      constexpr uint32_t LO = 0x05040100u, HI = 0x07060302u;
      uint4 o01, o23;
      o01.x = __builtin_amdgcn_perm(a1.x, a0.x, LO); o01.y = __builtin_amdgcn_perm(a3.x, a2.x, LO);
      o01.z = __builtin_amdgcn_perm(a1.x, a0.x, HI); o01.w = __builtin_amdgcn_perm(a3.x, a2.x, HI);
      o23.x = __builtin_amdgcn_perm(a1.y, a0.y, LO); o23.y = __builtin_amdgcn_perm(a3.y, a2.y, LO);
      o23.z = __builtin_amdgcn_perm(a1.y, a0.y, HI); o23.w = __builtin_amdgcn_perm(a3.y, a2.y, HI);
      *reinterpret_cast<uint4 *>(&lutq[ml * 256 + c4]) = o01;
      *reinterpret_cast<uint4 *>(&lutq[ml * 256 + c4 + 2]) = o23;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < QT_R; ++r) {
      const int row = r * QT_BS + (int)threadIdx.x;
      if (row < nrows) qt_row_tile<MT>(lutq, pcodes + (int64_t)row * M + tile * MT, acc[r]);
    }
  }
#pragma unroll
  for (int r = 0; r < QT_R; ++r) {
    const int row = r * QT_BS + (int)threadIdx.x;
    if (row < nrows && row_allowed(p.allow, off + (uint32_t)row)) {
      const float beta = t.row_beta[(int64_t)off + row];
#pragma unroll
      for (int j = 0; j < Q_G; ++j) {
        if (j < cnt && s4[j] > 0.0f && acc[r][j] < 65535u) {      // sum e < 65535: no entry of the row saturated
          const float S = (float)acc[r][j] + fmaf(s4[j], beta, sk4[j]);
          if (S < 65536.0f) atomicAdd(&hist[j * QB_BINS + (S > 0.0f ? ((uint32_t)S >> QT_BSHIFT) : 0u)], 1u);   // a NaN S is not counted
        }
      }
    }
  }
  __syncthreads();
  if (wave < cnt && sc[wave] > 0.0f) {
    const uint32_t *h = hist + wave * QB_BINS;
    constexpr int PER = QB_BINS / 64;
    uint32_t loc[PER], tot = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) { loc[i] = h[lane * PER + i]; tot += loc[i]; }
    uint32_t incl = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t tt = __shfl_up(incl, o, 64);
      if (lane >= o) incl += tt;
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
        const float B = (float)(((uint32_t)bin + 1u) << QT_BSHIFT);             // every counted row has S' < B
        const float T = (B + (float)(M + 18) + t.bslack[qj[wave]]) / sc[wave] * 1.00001f;     // s dist <= S' + M + 10 (+ the f32 slack, + margin)
        if (T > 0.0f && T < INFINITY) atomicMin(&p.tglobal[qj[wave]], order_key(T));
      }
    }
  }
}

template <int MU, int NT>
static void launch_qscan_tiled_pt(lance_hip_ctx *ctx, const QscanArgs &a, const PtArgs &t, unsigned grid) {
  const size_t lds = (size_t)4 * Q_CAP * 4 + 8 * 4 + 12 * 4 + (size_t)4 * Q_CAP * 2;
  hipLaunchKernelGGL((ivfpq_qscan_tiled_pt_kernel<MU, NT>), dim3(grid), dim3(QT_BS), lds, ctx->stream, a, t);
}

static void pt_launch_tables(lance_hip_ctx *ctx, const lance_hip_index *ix, const float *qs, uint32_t nq, const float *sq, uint16_t *tab) {
  const int d = (int)ix->d, m = (int)ix->m, sd = d / m;
  // enough slices of the sub-quantisers to put about sixteen workgroups on every CU (the kernel is a chain of L2 round trips)
  int msplit = (int)std::min<uint64_t>((uint64_t)m, std::max<uint64_t>(1, cdiv((uint64_t)16 * ctx->num_cus, nq)));
  const int mper = (int)cdiv((uint64_t)m, (uint64_t)msplit);
  msplit = (int)cdiv((uint64_t)m, (uint64_t)mper);
  const dim3 tgrid(nq, (unsigned)msplit);
  if (sd == 4) hipLaunchKernelGGL((q_pt_table_kernel<4>), tgrid, dim3(256), 0, ctx->stream, qs, ix->pt->g, sq, ix->codebook, d, m, mper, tab);
  else if (sd == 8) hipLaunchKernelGGL((q_pt_table_kernel<8>), tgrid, dim3(256), 0, ctx->stream, qs, ix->pt->g, sq, ix->codebook, d, m, mper, tab);
  else hipLaunchKernelGGL((q_pt_table_kernel<16>), tgrid, dim3(256), 0, ctx->stream, qs, ix->pt->g, sq, ix->codebook, d, m, mper, tab);
}

// LANCE_HIP_QPT=2: kappa, pre-scale and tables, then the bound pass on those tables (pair_starts0 / pair_idx0 = the queries grouped by
// their nearest partition, as qbound_launch takes them)
int qbound_pt_launch(lance_hip_ctx *ctx, const lance_hip_index *ix_c, const float *qs, uint32_t nq, uint32_t nprobes, uint32_t keff,
                     const uint32_t *probes, const uint32_t *pair_starts0, const uint32_t *pair_idx0, uint32_t *item_start, int4 *desc,
                     uint32_t max_items, uint32_t *tglobal, const uint32_t *allow) {
  lance_hip_index *ix = const_cast<lance_hip_index *>(ix_c);
  LH_TRY(qscan_pt_prepare(ctx, ix));
  LH_REQUIRE(ix->cb_mean, "per-query tables: the index carries no codebook means");
  const int d = (int)ix->d, m = (int)ix->m, sd = d / m, nlist = (int)ix->nlist;
  double *qn2 = ctx->scratch_t<double>("pt.qn2", nq), *qmu = ctx->scratch_t<double>("pt.qmu", nq);
  float *sq = ctx->scratch_t<float>("pt.sq", nq), *theta = ctx->scratch_t<float>("pt.theta", nq), *bslack = ctx->scratch_t<float>("pt.bslack", nq);
  float *kap = ctx->scratch_t<float>("pt.kap", (size_t)nq * nprobes);
  float *nu_h = nullptr; (void)nu_h;
  uint16_t *tab = ctx->scratch_t<uint16_t>("pt.tab", (size_t)nq * m * 256);
  if (!qn2 || !qmu || !sq || !theta || !bslack || !kap || !tab) return LANCE_HIP_ENOMEM;
  const int64_t npairs = (int64_t)nq * nprobes;
  hipLaunchKernelGGL(q_pt_kappa_kernel, dim3((unsigned)cdiv((uint64_t)npairs, 4)), dim3(256), 0, ctx->stream, qs, ix->pt->g, ix->pt->cen_t, probes,
                     npairs, (int)nprobes, d, kap, qn2, ix->cb_mean, qmu);
  hipLaunchKernelGGL(q_pt_prescale_kernel, dim3((unsigned)cdiv(nq, 256)), dim3(256), 0, ctx->stream, ix->pt->beta_min, ix->pt->beta_abs, ix->pt->beta_mean,
                     probes, (int)nq, (int)nprobes, sd + m, ix->cb_mean + d, kap, qn2, qmu, sq, theta, bslack);
  pt_launch_tables(ctx, ix, qs, nq, sq, tab);
  LH_TRY(qscan_item_tables(ctx, pair_starts0, nlist, Q_G, item_start, desc, max_items));
  QboundArgs a;
  a.rq = nullptr; a.pair_idx = pair_idx0; a.item_start = item_start; a.desc = desc;
  a.centroids = ix->centroids; a.codebook = ix->codebook; a.part_offsets = ix->part_offsets; a.codes = ix->codes;
  a.d = d; a.nlist = nlist; a.keff = (int)keff; a.round_f16 = 0;
  a.tglobal = tglobal; a.allow = allow; a.cb_mean = ix->cb_mean;
  PtBoundArgs t;
  t.tab = tab; t.sq = sq; t.kap = kap; t.bslack = bslack; t.row_beta = ix->pt->row_beta; t.nprobes = (int)nprobes;
  const size_t lds = (size_t)4 * QB_BINS * 4 + 8 * 4;
  if (m == 48) hipLaunchKernelGGL((ivfpq_qbound_tiled_pt_kernel<3, 1>), dim3(max_items), dim3(QT_BS), lds, ctx->stream, a, t);
  else if (m == 64) hipLaunchKernelGGL((ivfpq_qbound_tiled_pt_kernel<4, LH_QT_NT64>), dim3(max_items), dim3(QT_BS), lds, ctx->stream, a, t);
  else if (m == 96) hipLaunchKernelGGL((ivfpq_qbound_tiled_pt_kernel<6, LH_QT_NT96>), dim3(max_items), dim3(QT_BS), lds, ctx->stream, a, t);
  else { set_error("per-query tables: unsupported shape (m=%d)", m); return LANCE_HIP_EINVAL; }
  LH_CHECK_HIP(hipGetLastError());
  return LANCE_HIP_OK;
}

int qscan_pt_launch(lance_hip_ctx *ctx, const lance_hip_index *ix_c, const QscanArgs &a, const float *qs, uint32_t nq, const uint32_t *probes,
                    unsigned grid) {
  lance_hip_index *ix = const_cast<lance_hip_index *>(ix_c);   // the constants are a cache attached to the index
  LH_TRY(qscan_pt_prepare(ctx, ix));
  const int d = (int)ix->d, m = (int)ix->m, sd = d / m, nprobes = a.nprobes;
  double *qn2 = ctx->scratch_t<double>("pt.qn2", nq);
  float *sq = ctx->scratch_t<float>("pt.sq", nq);
  float *kap = ctx->scratch_t<float>("pt.kap", (size_t)nq * nprobes);
  float *pslack = ctx->scratch_t<float>("pt.pslack", (size_t)nq * nprobes);
  uint16_t *tab = ctx->scratch_t<uint16_t>("pt.tab", (size_t)nq * m * 256);
  if (!qn2 || !sq || !kap || !pslack || !tab) return LANCE_HIP_ENOMEM;
  const int64_t npairs = (int64_t)nq * nprobes;
  if (qscan_pt_mode(ix) == 2) {
    // the tables exist since the bound pass (qbound_pt_launch): only the per-pair slack with the actual T is left
    float *theta = ctx->scratch_t<float>("pt.theta", nq);
    if (!theta) return LANCE_HIP_ENOMEM;
    hipLaunchKernelGGL(q_pt_slack_kernel, dim3((unsigned)cdiv((uint64_t)npairs, 256)), dim3(256), 0, ctx->stream, ix->pt->beta_min, ix->pt->beta_abs,
                       probes, a.tbound, npairs, nprobes, sd + m, kap, qn2, sq, theta, pslack);
  } else {
  ScopedTimer tprep(ctx, "q_pt_tables");      // (inside the caller's "ivfpq_scan_c1" timer)
  hipLaunchKernelGGL(q_pt_kappa_kernel, dim3((unsigned)cdiv((uint64_t)npairs, 4)), dim3(256), 0, ctx->stream, qs, ix->pt->g, ix->pt->cen_t, probes,
                     npairs, nprobes, d, kap, qn2);
  hipLaunchKernelGGL(q_pt_scale_kernel, dim3((unsigned)cdiv(nq, 256)), dim3(256), 0, ctx->stream, ix->pt->beta_min, ix->pt->beta_abs, probes, a.tbound,
                     (int)nq, nprobes, sd + m, kap, qn2, sq, pslack);
  ScopedTimer ttab(ctx, "q_pt_table_only");
  pt_launch_tables(ctx, ix, qs, nq, sq, tab);
  }
  PtArgs t;
  t.tab = tab; t.sq = sq; t.kap = kap; t.pslack = pslack; t.row_beta = ix->pt->row_beta;
  if (m == 48) launch_qscan_tiled_pt<3, 1>(ctx, a, t, grid);
  else if (m == 64) launch_qscan_tiled_pt<4, LH_QT_NT64>(ctx, a, t, grid);
  else if (m == 96) launch_qscan_tiled_pt<6, LH_QT_NT96>(ctx, a, t, grid);
  else { set_error("per-query tables: unsupported shape (m=%d)", m); return LANCE_HIP_EINVAL; }
  LH_CHECK_HIP(hipGetLastError());
  return LANCE_HIP_OK;
}

}  // namespace lh
