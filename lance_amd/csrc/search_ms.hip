// search_ms.hip -- the ADC filter scan on the matrix cores (8-bit PQ, L2 / cosine, d = 64 / 128; M = 16 / 32).
//
// Why (round 4, gpurun r04g): the 4-query integer scan (search_q.hip) is the dominant kernel of the C2 step (0.354 of 0.84 ms)
// and sits under two ceilings at once -- random LDS gathers (61 % of its LDS cycles are bank conflicts) and VALU issue (158 M
// wave instructions per launch, 40 % of them the table build per (query, partition) item).  Both come from evaluating
//     dist(q, row) = sum_m |r_m - c_m(code_m)|^2                 (pq/distance.rs:109-144, v2.rs:316-332: r = q - centroid_p)
// one table lookup per (row, sub-quantiser) per group of four queries.  The same number is |r - c^_row|^2 with c^_row the row's
// reconstruction (the concatenated codewords), i.e.
//     dist = |c^_row|^2 - 2 r . c^_row + |r|^2 :
// |c^_row|^2 is a constant of the stored row (4 bytes, computed once per index), |r|^2 a scalar per (query, partition) pair, and
// the cross term is a [rows x d] x [d x queries] product per partition -- matrix-core work.  A workgroup owns 256 rows of one
// partition: each wave gathers the f16 reconstruction of its 64 rows ONCE (one 16-byte codeword fetch per (row, k-slice), i.e. n_p M
// gathers per partition instead of n_p M per four queries) and keeps it in registers as the A operand of
// v_mfma_f32_32x32x16_f16; the partition's queries stream through LDS as B tiles of 32; the accumulator starts at |c^_row|^2 (the
// MFMA's C operand), so D = |c^|^2 - 2 r.c^ comes out of the matrix pipe and the epilogue is ONE compare per (row, query) against
// lim = T + E - |r|^2.
//
// It is a FILTER, exactly like the integer scan it replaces: survivors go to the same per-(query, probe) segments, with an integer
// sum S ~ dist * s for the merge kernel's cut, and ivfpq_qmerge_kernel re-evaluates them in the reference's arithmetic -- ids and
// distances stay bit-equal to the oracle.  Soundness (no row with reference distance <= T is dropped): with u = 2^-11 (binary16),
//   |fl16(a) fl16(b) - a b| <= |a b| (2u + u^2)  ->  |2 r~.c~ - 2 r.c^| <= 2^-9 (1 + 2^-12) |r| |c^|      (Cauchy-Schwarz),
// the f32 accumulation of K <= 128 exact products adds <= K 2^-23 * 2 |r| |c^| (whatever the matrix pipe's internal rounding),
// the f32 evaluations of |c^|^2, |r|^2 and of the reference's own table / sequential sum <= (d + M + 4) 2^-24 of their values
// (together < 2^-13 (|r|^2 + T) with |c^|^2 <= 2 |r|^2 + 2.1 T), and a row that can pass has |c^| <= |r| + sqrt(T)
// (triangle inequality); every SURVIVOR of the test has |c^| <= 1.01 (|r| + sqrt(T + E)) as well, so the same E bounds the error of
// every sum the merge kernel's histogram counts.  ms_prep_kernel evaluates
//   E = 1.05 [2^-9 1.02 |r| (|r| + sqrt T) + 2^-13 (|r|^2 + T)] + E_abs        (E_abs: f16 subnormals flushed, per element 2^-14 / sigma)
// per pair; sigma is a power of two per index that puts the largest codeword component near 2^13 (products and scales by powers
// of two are exact).  A pair whose residual overflows binary16, whose E exceeds 5 % of T, or with anything non-finite is handed
// to the exact rescan (the integer scan's overflow route); so is a segment with more than Q_CAP survivors.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
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

constexpr int MS_RW = 256;          // rows per workgroup (4 waves x 64)
constexpr int MS_SBP = 192;         // pairs per LDS super-block (6 tiles of 32)
constexpr int MS_QCAP = 256;        // survivor queue entries per wave
constexpr float MS_SE = 30000.0f;   // the bound T maps to MS_SE units of the integer sums
constexpr float MS_SLACK_CAP = 1500.0f;   // largest E * s (units) the filter takes: 5 % of T
constexpr int MS_CUT_SHIFT = 6;     // merge histogram: 512 bins of 64 units

// ---- index constants ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ms_codebook_kernel(const float *__restrict__ cb, int64_t nwords, int sd, float scale, _Float16 *__restrict__ cbh,
                                                          float *__restrict__ cbn2) {
  const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (w >= nwords) return;
  float s = 0.0f;
  for (int u = 0; u < sd; ++u) {
    const float v = cb[w * sd + u];
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

// unit_start[p] = exclusive scan of (partition p has class-A pairs ? ceil(n_p / MS_RW) : 0)
__global__ __launch_bounds__(256) void ms_unit_table_kernel(const uint32_t *__restrict__ pair_starts, const uint32_t *__restrict__ part_offsets, int nlist,
                                                            uint32_t *__restrict__ unit_start) {
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t carry_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < nlist; base += 256) {
    const int i = base + threadIdx.x;
    uint32_t v = 0;
    if (i < nlist && pair_starts[i + 1] > pair_starts[i]) v = (part_offsets[i + 1] - part_offsets[i] + MS_RW - 1) / MS_RW;
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
    if (i < nlist) unit_start[i] = carry + woff + incl - v;
    __syncthreads();
    if (threadIdx.x == 255) carry_s = carry + woff + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) unit_start[nlist] = carry_s;
}

// unit -> everything the scan needs to start, in one 32-byte record (the first version searched unit_start by bisection and then
// chased part_offsets / pair_starts: eleven dependent L2 round trips, 6-7 us, in front of ~4 us of arithmetic per unit)
struct MsUnit { uint32_t part, row0, off, np, gs, qp, pad0, pad1; };
__global__ __launch_bounds__(256) void ms_unit_desc_kernel(const uint32_t *__restrict__ unit_start, const uint32_t *__restrict__ pair_starts,
                                                           const uint32_t *__restrict__ part_offsets, int nlist, MsUnit *__restrict__ units) {
  const int part = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (part >= nlist) return;
  const uint32_t u0 = unit_start[part], u1 = unit_start[part + 1];
  MsUnit u;
  u.part = (uint32_t)part; u.off = part_offsets[part]; u.np = part_offsets[part + 1] - u.off;
  u.gs = pair_starts[part]; u.qp = pair_starts[part + 1] - u.gs; u.pad0 = u.pad1 = 0u;
  for (uint32_t c = (uint32_t)lane; c < u1 - u0; c += 64u) { u.row0 = c * (uint32_t)MS_RW; units[u0 + c] = u; }
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
  f4 *prm;                      // [npairs + 32]: {lim sigma^2, s / sigma^2, |r|^2 s, pair (bits)}
  uint32_t *qslack;             // [nq] max over the query's pairs of ceil(E s) (zeroed before the launch)
  uint32_t *seg_cnt, *qovf, *ovf;
  uint32_t nan_slot;            // index into prm of the NaN-limit record (written here, loaded by the scan for padded pair slots)
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
  float n2[MS_PPW], vmax[MS_PPW];
  bool bad[MS_PPW];
#pragma unroll
  for (int t = 0; t < MS_PPW; ++t) { n2[t] = 0.0f; vmax[t] = 0.0f; bad[t] = false; }
  for (int e = lane; e < p.d; e += 64) {
    float v[MS_PPW];
#pragma unroll
    for (int t = 0; t < MS_PPW; ++t) v[t] = qv[t][e] - cv[t][e];      // v2.rs:316-332, the subtraction of the exact path
#pragma unroll
    for (int t = 0; t < MS_PPW; ++t) {
      if (p.round_f16) v[t] = __half2float(__float2half_rn(v[t]));
      n2[t] += v[t] * v[t];
      vmax[t] = fmaxf(vmax[t], fabsf(v[t]));
      bad[t] |= !(fabsf(v[t]) < INFINITY);
      if (g0 + (uint32_t)t < ng) p.rh[(int64_t)(g0 + t) * p.d + e] = (_Float16)(v[t] * p.sigma);
    }
  }
#pragma unroll
  for (int t = 0; t < MS_PPW; ++t) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { n2[t] += __shfl_xor(n2[t], o, 64); vmax[t] = fmaxf(vmax[t], __shfl_xor(vmax[t], o, 64)); }
    bad[t] = __any(bad[t]);
  }
  // lane t finishes pair t
  float n2l = n2[0], vml = vmax[0]; bool badl = bad[0]; uint32_t pairl = pair[0];
#pragma unroll
  for (int t = 1; t < MS_PPW; ++t) if (lane == t) { n2l = n2[t]; vml = vmax[t]; badl = bad[t]; pairl = pair[t]; }
  if (lane >= MS_PPW || g0 + (uint32_t)lane >= ng) return;
  const uint32_t q = pairl / (uint32_t)p.nprobes;
  const float T = key_to_float(p.tbound[q]);     // class A: 0 < T < inf
  const float s = MS_SE / T;
  const float rn = sqrtf(n2l) * 1.000001f, st = sqrtf(T) * 1.000001f;
  const float sqd = sqrtf((float)p.d) + 1.0f;
  const float e_abs = 6.1035156e-5f * sqd * (3.0f * rn + 2.0f * st) / p.sigma + (float)p.d * 3.7252903e-9f / (p.sigma * p.sigma);
  const float E = 1.05f * (1.9921875e-3f * rn * (rn + st) + 1.2207031e-4f * (n2l + T)) + e_abs;     // 2^-9 * 1.02, 2^-13
  const float eu = E * s * 1.1f + 3.0f;
  const float lim = (T * 1.0000077f + E) - n2l;  // T (1 + 2^-17) + E - |r|^2; the cancellation's rounding sits inside the 2^-13 term
  const float sig2 = p.sigma * p.sigma;
  const bool ok = !badl && T > 0.0f && T < INFINITY && s > 0.0f && s < INFINITY && n2l < INFINITY && vml * p.sigma < 60000.0f &&
                  eu <= MS_SLACK_CAP && fabsf(lim * sig2) < INFINITY && n2l * s < 1e30f;
  f4 o;
  if (ok) {
    o.x = lim * sig2; o.y = s / sig2; o.z = n2l * s;
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

// ---- the scan ------------------------------------------------------------------------------------------------------------------
struct MscanArgs {
  const uint32_t *unit_start;   // [nlist+1]
  const uint32_t *pair_starts;  // [nlist+1] (class A)
  const uint32_t *part_offsets;
  const uint8_t *codes;
  const _Float16 *cbh;          // [m][256][sd] = f16(-2 sigma c)
  const float *row_cn2;         // [n] sigma^2 |c^_row|^2
  const _Float16 *rh;           // [pairs][d]
  const f4 *prm;                // [pairs]
  const MsUnit *units;          // [units] (pipelined kernel)
  uint32_t nan_slot;
  int nlist, nprobes, m;
  uint32_t *seg_cnt, *seg_pos;
  uint16_t *seg_sum;
  uint32_t *qovf, *ovf;
  const uint32_t *allow;
};

template <int SD, int KS>
__global__ __launch_bounds__(256, 2) void ivfpq_mscan_kernel(MscanArgs p) {
  constexpr int D = KS * 16;
  constexpr int RS = D + 8;                 // f16 row stride of a staged B tile: 16-byte skew, conflict-free ds_read_b128
  constexpr int M = D / SD;
  __shared__ __attribute__((aligned(16))) _Float16 sB[MS_SBP * RS];
  __shared__ __attribute__((aligned(16))) f4 sP[MS_SBP];
  __shared__ __attribute__((aligned(8))) uint2 sQ[4][MS_QCAP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, g = lane >> 5;
  // consecutive units (the row chunks of one partition share the partition's B tiles) stay on one XCD: blockIdx round-robins the 8 XCDs
  const uint32_t per_xcd = (gridDim.x + 7u) >> 3;
  const uint32_t unit = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  if (unit >= p.unit_start[p.nlist]) return;
  int part = (int)find_partition_dev(p.unit_start, p.nlist, unit);
  while (p.unit_start[part + 1] <= unit) ++part;      // empty ranges share their successor's start
  const uint32_t off = p.part_offsets[part];
  const int np = (int)(p.part_offsets[part + 1] - off);
  const int row0 = (int)(unit - p.unit_start[part]) * MS_RW;
  const uint32_t gs = p.pair_starts[part];
  const int Qp = (int)(p.pair_starts[part + 1] - gs);

  // A: the f16 reconstruction of this wave's 64 rows, in MFMA operand layout (lane (j, g): row j of the 32-block, k-slice g)
  ms_h8 a[2][KS];
  ms_f16v cinit[2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const int rowl = wave * 64 + rb * 32 + j;
    const int rowc = min(row0 + rowl, np - 1);
    const uint8_t *rc = p.codes + ((int64_t)off + rowc) * M;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if constexpr (SD == 8) {
        const int mm = 2 * s + g;
        a[rb][s] = *reinterpret_cast<const ms_h8 *>(p.cbh + ((int64_t)mm * 256 + rc[mm]) * 8);
      } else if constexpr (SD == 4) {
        const int mm = 4 * s + 2 * g;
        const ms_h4 lo = *reinterpret_cast<const ms_h4 *>(p.cbh + ((int64_t)mm * 256 + rc[mm]) * 4);
        const ms_h4 hi = *reinterpret_cast<const ms_h4 *>(p.cbh + ((int64_t)(mm + 1) * 256 + rc[mm + 1]) * 4);
        a[rb][s] = ms_h8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      } else {
        static_assert(SD == 16, "sub-dimension 4 / 8 / 16");
        a[rb][s] = *reinterpret_cast<const ms_h8 *>(p.cbh + ((int64_t)s * 256 + rc[s]) * 16 + g * 8);
      }
    }
    // D[row i][query j]: lane (j, g) holds rows i = (v & 3) + 8 (v >> 2) + 4 g of the 32-block (layout as in mfma_assign.hip)
#pragma unroll
    for (int vq = 0; vq < 4; ++vq)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = row0 + wave * 64 + rb * 32 + 8 * vq + 4 * g + e;
        cinit[rb][vq * 4 + e] = r < np ? p.row_cn2[(int64_t)off + r] : INFINITY;      // a padded row never passes
      }
  }

  uint2 *myq = sQ[wave];
  uint32_t qn = 0;      // wave-uniform
  auto flush = [&]() {
    for (uint32_t e = (uint32_t)lane; e < qn; e += 64u) {
      const uint2 ent = myq[e];
      const uint32_t slot = ent.x >> 8, rowl = ent.x & 255u;
      const f4 P = sP[slot];
      const uint32_t pair = __float_as_uint(P.w);
      const float S = fminf(fmaxf(rintf(__builtin_fmaf(__uint_as_float(ent.y), P.y, P.z)), 0.0f), 65535.0f);
      const uint32_t pos = off + (uint32_t)row0 + rowl;
      if (row_allowed(p.allow, pos)) {
        const uint32_t k = atomicAdd(&p.seg_cnt[pair], 1u);
        if (k < (uint32_t)Q_CAP) {
          p.seg_pos[(int64_t)pair * Q_CAP + k] = pos;
          p.seg_sum[(int64_t)pair * Q_CAP + k] = (uint16_t)S;
        } else if (k == (uint32_t)Q_CAP) {      // the segment lost survivors from here on: exact rescan of this (query, probe)
          p.qovf[pair / (uint32_t)p.nprobes] = 1u;
          p.ovf[1u + atomicAdd(&p.ovf[0], 1u)] = pair;
        }
      }
    }
    qn = 0;
  };

  for (int sb0 = 0; sb0 < Qp; sb0 += MS_SBP) {
    const int nsb = min(MS_SBP, Qp - sb0);
    const int nblk = (nsb + 31) >> 5;
    __syncthreads();      // the previous super-block's tiles and parameters are done with (queues flushed)
    {
      constexpr int CPR = D / 8;      // 16-byte chunks per pair row
      const _Float16 *src = p.rh + ((int64_t)gs + sb0) * D;
      for (int idx = threadIdx.x; idx < nblk * 32 * CPR; idx += 256) {
        const int slot = idx / CPR, c = idx - slot * CPR;
        const uint4 v = *reinterpret_cast<const uint4 *>(src + (int64_t)slot * D + c * 8);      // (rh is padded by 32 rows)
        *reinterpret_cast<uint4 *>(&sB[slot * RS + c * 8]) = v;
      }
      for (int slot = threadIdx.x; slot < nblk * 32; slot += 256) {
        f4 P = {__uint_as_float(0x7FC00000u), 0.0f, 0.0f, 0.0f};      // padding slots read whatever follows in rh: NaN limit, never passes
        if (slot < nsb) P = p.prm[(int64_t)gs + sb0 + slot];
        sP[slot] = P;
      }
    }
    __syncthreads();
    for (int jb = 0; jb < nblk; ++jb) {
      ms_h8 b[KS];
      const _Float16 *br = &sB[(jb * 32 + j) * RS + g * 8];
#pragma unroll
      for (int s = 0; s < KS; ++s) b[s] = *reinterpret_cast<const ms_h8 *>(br + s * 16);
      const float lim = sP[jb * 32 + j].x;
      if (qn + 128u > (uint32_t)MS_QCAP) flush();      // room for this tile's worst plausible burst; checked again per append
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        ms_f16v acc = cinit[rb];
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[rb][s], b[s], acc, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const bool pass = acc[v] <= lim;
          const uint64_t mask = __ballot(pass);
          if (mask) {
            const uint32_t cnt = (uint32_t)__popcll(mask);
            if (qn + cnt > (uint32_t)MS_QCAP) flush();
            if (pass) {
              const uint32_t idx = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
              const uint32_t rowl = (uint32_t)(wave * 64 + rb * 32 + (v & 3) + 8 * (v >> 2) + 4 * g);
              myq[idx] = make_uint2(((uint32_t)(jb * 32 + j) << 8) | rowl, __float_as_uint(acc[v]));
            }
            qn += cnt;
          }
        }
      }
    }
    flush();      // before the parameters of this super-block are overwritten
  }
}

// ---- the scan, pipelined (the default) -----------------------------------------------------------------------------------------------
// First hardware run of the kernel above (gpurun r04h): parity green (62 tests, 209 fuzz cases), 0.343 ms per 10k-query batch at C2 --
// no faster than the integer scan -- because a unit spent 43 us on ~4 us of arithmetic: bisection + pointer chasing in front, then
// per super-block {global -> register -> LDS staging, barrier, MFMAs, flush with two to four DEPENDENT device-scope atomic round trips},
// nothing overlapped, two workgroups per CU.  This version keeps the arithmetic and takes the waiting out:
//   * one 32-byte unit record instead of the search;
//   * the pairs' f16 residuals and parameters arrive by LDS-DMA (global_load_lds_dwordx4) into the OTHER half of a double buffer while
//     the current super-block is computed; the bank-conflict padding of the first version becomes an XOR swizzle of the 16-byte chunk
//     index, applied to the per-lane SOURCE address (the DMA writes lane-linear) and to the ds_read_b128 address alike;
//   * survivors are queued with their integer sum already computed (the parameters are in registers at that point), so the flush
//     needs nothing from LDS but the queue; its atomics are issued right AFTER the super-block's barrier and their dependent stores
//     one super-block later -- both have a whole compute phase to complete before the next barrier's vmcnt(0).
typedef __attribute__((address_space(1))) const void *ms_gptr;
typedef __attribute__((address_space(3))) void *ms_lptr;
constexpr int MS2_SBP = 128;        // pairs per LDS super-block (4 tiles of 32), double buffered
constexpr int MS2_QCAP = 128;       // survivor queue entries per wave (2 per lane pending in registers)

template <int SD, int KS>
__global__ __launch_bounds__(256, 2) void ivfpq_mscan2_kernel(MscanArgs p) {
  constexpr int D = KS * 16;
  constexpr int M = D / SD;
  constexpr int RB = D * 2;                 // bytes of a pair's f16 residual
  constexpr int CPR = RB / 16;              // 16-byte chunks per pair row (16 / 8)
  constexpr int RPK = 256 / RB;             // pair rows per 256 bytes (1 / 2): the swizzle key is (row / RPK) & (CPR - 1)
  constexpr int SPI = 4096 / RB;            // pair slots one 256-lane DMA pass covers (16 / 32)
  constexpr int PE = MS2_QCAP / 64;
  __shared__ __attribute__((aligned(16))) char sB[2][MS2_SBP * RB];
  __shared__ __attribute__((aligned(16))) f4 sP[2][MS2_SBP];
  __shared__ __attribute__((aligned(8))) uint2 sQ[4][MS2_QCAP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, g = lane >> 5;
  const uint32_t per_xcd = (gridDim.x + 7u) >> 3;
  const uint32_t unit = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  if (unit >= p.unit_start[p.nlist]) return;
  const MsUnit U = p.units[unit];
  const uint32_t off = U.off;
  const int np = (int)U.np, row0 = (int)U.row0, Qp = (int)U.qp;
  const uint32_t gs = U.gs;

  // LDS-DMA of super-block `sb` into buffer `buf` (lane-linear destination, swizzled source)
  auto stage = [&](int sb, int buf) {
    const int sb0 = sb * MS2_SBP;
    const int nsb = min(MS2_SBP, Qp - sb0);
    const int nslots = ((nsb + 31) >> 5) << 5;
    const char *src = reinterpret_cast<const char *>(p.rh + ((int64_t)gs + sb0) * D);
#pragma unroll
    for (int it = 0; it < MS2_SBP / SPI; ++it) {
      const int slot = it * SPI + wave * (SPI / 4) + lane / CPR, k = lane % CPR;
      if (it * SPI < nslots && slot < nsb)      // padded slots keep whatever the buffer held: their limit is a NaN
        __builtin_amdgcn_global_load_lds((ms_gptr)(src + (int64_t)slot * RB + ((k ^ ((slot / RPK) & (CPR - 1))) << 4)),
                                         (ms_lptr)(&sB[buf][(it * SPI + wave * (SPI / 4)) * RB]), 16, 0, 0);
    }
    if (wave < MS2_SBP / 64) {
      const int slot = wave * 64 + lane;
      if (slot < nslots) {
        const f4 *ps = slot < nsb ? p.prm + (int64_t)gs + sb0 + slot : p.prm + p.nan_slot;
        __builtin_amdgcn_global_load_lds((ms_gptr)ps, (ms_lptr)(&sP[buf][wave * 64]), 16, 0, 0);
      }
    }
  };
  const int nsbk = (Qp + MS2_SBP - 1) / MS2_SBP;
  stage(0, 0);

  // A: the f16 reconstruction of this wave's 64 rows, in MFMA operand layout (lane (j, g): row j of the 32-block, k-slice g)
  ms_h8 a[2][KS];
  ms_f16v cinit[2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const int rowl = wave * 64 + rb * 32 + j;
    const int rowc = min(row0 + rowl, np - 1);
    uint32_t cw[M / 4];
    {
      const uint4 *rc4 = reinterpret_cast<const uint4 *>(p.codes + ((int64_t)off + rowc) * M);      // rows of 16 / 32 bytes, 16-byte aligned
#pragma unroll
      for (int w = 0; w < M / 16; ++w) { const uint4 t = rc4[w]; cw[4 * w] = t.x; cw[4 * w + 1] = t.y; cw[4 * w + 2] = t.z; cw[4 * w + 3] = t.w; }
    }
    auto code = [&](int mm) -> uint32_t { return (cw[mm >> 2] >> (8 * (mm & 3))) & 255u; };
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if constexpr (SD == 8) {
        // sub-quantiser 2 s + g: both candidates extracted with static shifts, one select
        const uint32_t c0 = code(2 * s), c1 = code(2 * s + 1);
        const int mm = 2 * s + g;
        a[rb][s] = *reinterpret_cast<const ms_h8 *>(p.cbh + ((int64_t)mm * 256 + (g ? c1 : c0)) * 8);
      } else {
        static_assert(SD == 4, "sub-dimension 4 / 8");
        const uint32_t c0 = g ? code(4 * s + 2) : code(4 * s), c1 = g ? code(4 * s + 3) : code(4 * s + 1);
        const int mm = 4 * s + 2 * g;
        const ms_h4 lo = *reinterpret_cast<const ms_h4 *>(p.cbh + ((int64_t)mm * 256 + c0) * 4);
        const ms_h4 hi = *reinterpret_cast<const ms_h4 *>(p.cbh + ((int64_t)(mm + 1) * 256 + c1) * 4);
        a[rb][s] = ms_h8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
    }
    // D[row i][query j]: lane (j, g) holds rows i = (v & 3) + 8 (v >> 2) + 4 g of the 32-block (layout as in mfma_assign.hip)
#pragma unroll
    for (int vq = 0; vq < 4; ++vq)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = row0 + wave * 64 + rb * 32 + 8 * vq + 4 * g + e;
        cinit[rb][vq * 4 + e] = r < np ? p.row_cn2[(int64_t)off + r] : INFINITY;      // a padded row never passes
      }
  }

  uint2 *myq = sQ[wave];
  uint32_t qn = 0;      // wave-uniform: queue entries
  // pending flush: entries whose segment slot has been requested (atomicAdd issued) but not yet used
  uint32_t pd_pair[PE], pd_pos[PE], pd_k[PE], pd_s[PE];
#pragma unroll
  for (int i = 0; i < PE; ++i) { pd_pair[i] = 0xFFFFFFFFu; pd_pos[i] = 0u; pd_k[i] = 0u; pd_s[i] = 0u; }
  auto flush_end = [&]() {
#pragma unroll
    for (int i = 0; i < PE; ++i) {
      if (pd_pair[i] != 0xFFFFFFFFu) {
        const uint32_t k = pd_k[i], pair = pd_pair[i];
        if (k < (uint32_t)Q_CAP) {
          p.seg_pos[(int64_t)pair * Q_CAP + k] = pd_pos[i];
          p.seg_sum[(int64_t)pair * Q_CAP + k] = (uint16_t)pd_s[i];
        } else if (k == (uint32_t)Q_CAP) {      // the segment lost survivors from here on: exact rescan of this (query, probe)
          p.qovf[pair / (uint32_t)p.nprobes] = 1u;
          p.ovf[1u + atomicAdd(&p.ovf[0], 1u)] = pair;
        }
        pd_pair[i] = 0xFFFFFFFFu;
      }
    }
  };
  auto flush_begin = [&]() {
#pragma unroll
    for (int i = 0; i < PE; ++i) {
      const uint32_t e = (uint32_t)lane + 64u * (uint32_t)i;
      if (e < qn) {
        const uint2 ent = myq[e];
        const uint32_t pair = ent.x >> 8, pos = off + (uint32_t)row0 + (ent.x & 255u);
        if (row_allowed(p.allow, pos)) {
          pd_pair[i] = pair; pd_pos[i] = pos; pd_s[i] = ent.y;
          pd_k[i] = atomicAdd(&p.seg_cnt[pair], 1u);
        }
      }
    }
    qn = 0;
  };

  for (int sb = 0; sb < nsbk; ++sb) {
    const int buf = sb & 1;
    __syncthreads();      // vmcnt(0) + barrier: super-block sb has landed for every wave; the other buffer's readers are done
    flush_end();          // stores behind the atomics issued one super-block ago
    flush_begin();        // atomics for the survivors of the previous super-block
    if (sb + 1 < nsbk) stage(sb + 1, buf ^ 1);
    const int nsb = min(MS2_SBP, Qp - sb * MS2_SBP);
    const int nblk = (nsb + 31) >> 5;
    for (int jb = 0; jb < nblk; ++jb) {
      const int slot = jb * 32 + j;
      ms_h8 b[KS];
      const char *br = &sB[buf][slot * RB];
      const int key = (slot / RPK) & (CPR - 1);
#pragma unroll
      for (int s = 0; s < KS; ++s) b[s] = *reinterpret_cast<const ms_h8 *>(br + (((2 * s + g) ^ key) << 4));
      const f4 P = sP[buf][slot];
      const uint32_t pair8 = __float_as_uint(P.w) << 8;
      // room for a tile's usual yield (13 survivors at C2); a burst beyond the queue hands its pairs to the exact rescan below
      if (qn > (uint32_t)(MS2_QCAP - 48)) { flush_end(); flush_begin(); }
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        ms_f16v acc = cinit[rb];
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[rb][s], b[s], acc, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const bool pass = acc[v] <= P.x;
          const uint64_t mask = __ballot(pass);
          if (mask) {
            if (pass) {
              const uint32_t idx = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
              if (idx < (uint32_t)MS2_QCAP) {
                const uint32_t rowl = (uint32_t)(wave * 64 + rb * 32 + (v & 3) + 8 * (v >> 2) + 4 * g);
                const float S = __builtin_amdgcn_fmed3f(rintf(__builtin_fmaf(acc[v], P.y, P.z)), 0.0f, 65535.0f);
                myq[idx] = make_uint2(pair8 | rowl, (uint32_t)S);
              } else {
                // more than a queue's worth of survivors in one tile (>= 4 % of its cells pass: this pair's segment would overflow
                // anyway): the pair is handed to the exact rescan -- the count jumps past Q_CAP, and whoever crosses it lists the pair
                const uint32_t pair = pair8 >> 8;
                const uint32_t k = atomicAdd(&p.seg_cnt[pair], (uint32_t)Q_CAP + 1u);
                if (k <= (uint32_t)Q_CAP) p.ovf[1u + atomicAdd(&p.ovf[0], 1u)] = pair;      // (the rescan kernel flags the query)
              }
            }
            qn = min(qn + (uint32_t)__popcll(mask), (uint32_t)MS2_QCAP);
          }
        }
      }
    }
  }
  flush_end();
  flush_begin();
  flush_end();
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

bool mscan_supported(const lance_hip_index *ix, uint32_t nq, uint32_t nprobes) {
  static const bool off = getenv("LANCE_HIP_NO_MSCAN") != nullptr;
  static const uint32_t minq = getenv("LANCE_HIP_MSCAN_MINQ") ? (uint32_t)std::max(1, atoi(getenv("LANCE_HIP_MSCAN_MINQ"))) : 16u;
  if (off || !ms_shape(ix, nullptr, nullptr) || !qscan_supported(ix, nq, nprobes)) return false;
  if (qscan8_enabled((int)ix->m, (int)(ix->d / ix->m))) return false;
  // a tile is 32 queries of one partition: worth it once the partitions see a couple of tiles' worth of queries on average
  return (uint64_t)nq * nprobes >= (uint64_t)minq * ix->nlist;
}

static int mscan_prepare(lance_hip_ctx *ctx, lance_hip_index *ix) {
  std::lock_guard<std::mutex> lk(ix->lazy_mu);   // the first search of any context builds the constants, the others wait for it
  if (ix->ms) return LANCE_HIP_OK;
  const int d = (int)ix->d, m = (int)ix->m, sd = d / m;
  const int64_t nwords = (int64_t)m * 256;
  std::vector<float> cb((size_t)nwords * sd);
  LH_CHECK_HIP(hipMemcpyAsync(cb.data(), ix->codebook, cb.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
  LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  float cbmax = 0.0f;
  for (float v : cb) cbmax = std::max(cbmax, std::fabs(v));
  auto *mc = new lance_hip_index::MsConst();
  if (!(cbmax > 0.0f) || !std::isfinite(cbmax)) {      // an all-zero codebook: nothing to scale by -- the integer scan serves this index
    mc->usable = false;
    ix->ms = mc;
    return LANCE_HIP_OK;
  }
  int e = 0;
  (void)std::frexp(cbmax, &e);      // cbmax in [2^(e-1), 2^e)
  mc->sigma = std::ldexp(1.0f, 13 - e);      // 2 sigma cbmax in [2^13, 2^14)
  bool ok = hipMalloc(reinterpret_cast<void **>(&mc->cbh), (size_t)nwords * sd * 2) == hipSuccess;
  ok = ok && hipMalloc(reinterpret_cast<void **>(&mc->cbn2), (size_t)nwords * 4) == hipSuccess;
  ok = ok && hipMalloc(reinterpret_cast<void **>(&mc->row_cn2), (size_t)(ix->n ? ix->n : 1) * 4) == hipSuccess;
  auto drop = [&]() { (void)hipFree(mc->cbh); (void)hipFree(mc->cbn2); (void)hipFree(mc->row_cn2); delete mc; };
  if (!ok) { drop(); set_error("matrix-core scan: out of device memory for the index constants"); return LANCE_HIP_ENOMEM; }
  hipLaunchKernelGGL(ms_codebook_kernel, dim3((unsigned)cdiv((uint64_t)nwords, 256)), dim3(256), 0, ctx->stream, ix->codebook, nwords, sd,
                     -2.0f * mc->sigma, reinterpret_cast<_Float16 *>(mc->cbh), mc->cbn2);
  if (ix->n)
    hipLaunchKernelGGL(ms_row_norm_kernel, dim3((unsigned)cdiv(ix->n, 256)), dim3(256), 0, ctx->stream, ix->codes, (int64_t)ix->n, m, mc->cbn2,
                       mc->sigma * mc->sigma, mc->row_cn2);
  uint64_t units = 0;
  for (uint32_t pid = 0; pid < ix->nlist; ++pid) units += cdiv((uint64_t)(ix->part_offsets_h[pid + 1] - ix->part_offsets_h[pid]), MS_RW);
  mc->max_units = (uint32_t)std::min<uint64_t>(units, 0x7FFFFFF0u);
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {   // other contexts (streams) search the same index
    drop();
    set_error("matrix-core scan: building the index constants failed");
    return LANCE_HIP_ERUNTIME;
  }
  mc->usable = true;
  ix->ms = mc;      // published complete; lance_hip_index's destructor frees it
  return LANCE_HIP_OK;
}

// -1: this index cannot take the matrix-core scan (the caller falls back to the integer scan); otherwise a status code.
// Replaces qscan_launch: same outputs (seg_cnt / seg_pos / seg_sum / qovf / ovf), plus qslack for the merge kernel's cut.
int mscan_launch(lance_hip_ctx *ctx, const lance_hip_index *ix_c, const float *qs, uint32_t nq, uint32_t nprobes, const uint32_t *probes,
                 const uint32_t *pair_starts, const uint32_t *pair_idx, const uint32_t *tbound, uint32_t *seg_cnt, uint32_t *seg_pos,
                 uint32_t *qovf, const uint32_t *allow, uint32_t **qslack_out) {
  lance_hip_index *ix = const_cast<lance_hip_index *>(ix_c);   // the constants are a cache attached to the index
  int sd = 0, ks = 0;
  if (!ms_shape(ix, &sd, &ks)) return -1;
  LH_TRY(mscan_prepare(ctx, ix));
  if (!ix->ms->usable || ix->ms->max_units == 0) return -1;
  const int d = (int)ix->d, nlist = (int)ix->nlist;
  const size_t npairs = (size_t)nq * nprobes;
  _Float16 *rh = reinterpret_cast<_Float16 *>(ctx->scratch("ms.rh", (npairs + 32) * (size_t)d * 2));
  f4 *prm = reinterpret_cast<f4 *>(ctx->scratch("ms.prm", (npairs + 32) * 16));
  uint32_t *qslack = ctx->scratch_t<uint32_t>("ms.qslack", nq);
  uint32_t *unit_start = ctx->scratch_t<uint32_t>("ms.unit_start", (size_t)nlist + 1);
  MsUnit *units = reinterpret_cast<MsUnit *>(ctx->scratch("ms.units", ((size_t)ix->ms->max_units + 8) * sizeof(MsUnit)));
  const uint32_t nan_slot = (uint32_t)npairs + 1u;      // inside prm's 32 records of padding
  uint16_t *seg_sum = ctx->scratch_t<uint16_t>("q.seg_sum", npairs * Q_CAP);   // the merge launcher asks for the same slot
  uint32_t *ovf = ctx->scratch_t<uint32_t>("q.ovf", npairs + 1);                // likewise
  if (!rh || !prm || !qslack || !unit_start || !units || !seg_sum || !ovf) return LANCE_HIP_ENOMEM;
  {
    ScopedTimer t(ctx, "q_residual");
    LH_CHECK_HIP(lh::memset_async(seg_cnt, 0, npairs * 4, ctx->stream));
    LH_CHECK_HIP(lh::memset_async(qovf, 0, (size_t)nq * 4, ctx->stream));
    LH_CHECK_HIP(lh::memset_async(qslack, 0, (size_t)nq * 4, ctx->stream));
    LH_CHECK_HIP(lh::memset_async(ovf, 0, 4, ctx->stream));
    MsPrepArgs pa;
    pa.q = qs; pa.centroids = ix->centroids; pa.pair_idx = pair_idx; pa.pair_starts = pair_starts; pa.probes = probes; pa.tbound = tbound;
    pa.d = d; pa.nprobes = (int)nprobes; pa.nlist = nlist; pa.round_f16 = ix->dtype == LANCE_HIP_F16 ? 1 : 0;
    pa.sigma = ix->ms->sigma; pa.rh = rh; pa.prm = prm; pa.qslack = qslack; pa.seg_cnt = seg_cnt; pa.qovf = qovf; pa.ovf = ovf;
    pa.nan_slot = nan_slot;
    hipLaunchKernelGGL(ms_prep_kernel, dim3((unsigned)cdiv(npairs, 4 * MS_PPW)), dim3(256), 0, ctx->stream, pa);
    hipLaunchKernelGGL(ms_unit_table_kernel, dim3(1), dim3(256), 0, ctx->stream, pair_starts, ix->part_offsets, nlist, unit_start);
    hipLaunchKernelGGL(ms_unit_desc_kernel, dim3((unsigned)cdiv((uint64_t)nlist, 4)), dim3(256), 0, ctx->stream, unit_start, pair_starts, ix->part_offsets,
                       nlist, units);
  }
  ScopedTimer t(ctx, "ivfpq_scan_c1");
  MscanArgs a;
  a.unit_start = unit_start; a.pair_starts = pair_starts; a.part_offsets = ix->part_offsets; a.codes = ix->codes;
  a.cbh = reinterpret_cast<const _Float16 *>(ix->ms->cbh); a.row_cn2 = ix->ms->row_cn2; a.rh = rh; a.prm = prm;
  a.nlist = nlist; a.nprobes = (int)nprobes; a.m = (int)ix->m; a.units = units; a.nan_slot = nan_slot;
  a.seg_cnt = seg_cnt; a.seg_pos = seg_pos; a.seg_sum = seg_sum; a.qovf = qovf; a.ovf = ovf; a.allow = allow;
  const unsigned grid = (ix->ms->max_units + 7u) & ~7u;
  ScopedTimer tm(ctx, "ivfpq_mscan");      // the same launch under its own name: tests assert the matrix-core scan was the one taken
  static const bool v1 = getenv("LANCE_HIP_MS_V1") != nullptr;      // the unpipelined first version, kept for A/B
  if (!v1 && sd == 8 && ks == 8) hipLaunchKernelGGL((ivfpq_mscan2_kernel<8, 8>), dim3(grid), dim3(256), 0, ctx->stream, a);
  else if (!v1 && sd == 4 && ks == 8) hipLaunchKernelGGL((ivfpq_mscan2_kernel<4, 8>), dim3(grid), dim3(256), 0, ctx->stream, a);
  else if (!v1 && sd == 4 && ks == 4) hipLaunchKernelGGL((ivfpq_mscan2_kernel<4, 4>), dim3(grid), dim3(256), 0, ctx->stream, a);
  else if (sd == 8 && ks == 8) hipLaunchKernelGGL((ivfpq_mscan_kernel<8, 8>), dim3(grid), dim3(256), 0, ctx->stream, a);
  else if (sd == 4 && ks == 8) hipLaunchKernelGGL((ivfpq_mscan_kernel<4, 8>), dim3(grid), dim3(256), 0, ctx->stream, a);
  else if (sd == 4 && ks == 4) hipLaunchKernelGGL((ivfpq_mscan_kernel<4, 4>), dim3(grid), dim3(256), 0, ctx->stream, a);
  else { set_error("matrix-core scan: unsupported shape (d=%d, sd=%d)", d, sd); return LANCE_HIP_EINVAL; }
  LH_CHECK_HIP(hipGetLastError());
  if (qslack_out) *qslack_out = qslack;
  return LANCE_HIP_OK;
}

void mscan_cut_params(int *cut_shift, uint32_t *cut_slack) { *cut_shift = MS_CUT_SHIFT; *cut_slack = 2u; }

}  // namespace lh
