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

}  // namespace lh
