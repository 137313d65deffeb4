// xform_fused.hip -- the IVF-PQ transform as ONE pass over the rows: coarse assign -> exact re-check -> residual -> PQ encode.
//
//   IvfTransformer chain         ivf.rs:188-236      (PartitionTransformer -> ResidualTransform -> PQTransformer)
//   compute_partitions           kmeans.rs:1187-1246 (argmin over the centroids, KeepFiniteVectors folded in: ivf/transform.rs:75-137)
//   do_compute_residual          residual.rs:58-102  (x - centroid[part], in the column's element type: f16 rounds)
//   ProductQuantizer::transform  pq.rs:116-191       (per sub-vector the L2-nearest codeword, unwrap_or(0))
//   precedent for fusing them    python/python/lance/vector.py:693-714 (one torch pass: assign, residual, encode)
//
// Round 5 ran this as ma_top3_kernel -> ma_finalize_kernel -> pq_mfma_estep_kernel: the rows crossed HBM three times (the third
// time sixteen times over in 32-byte pieces, one per sub-quantiser) and the PQ half spent 4-5 VALU per (row, codeword) pair on
// top of three half-empty K = 8 MFMAs: 2.66 ms per million 128-d rows, 0.025 of the HBM roofline.  Here a
// 256-lane workgroup owns 128 rows and keeps them on chip from the first load to the code bytes:
//
//   1. rows: HBM -> LDS (coalesced, the column's own element type) -> registers: wave w owns rows 32 w .. 32 w + 31, a lane
//      pair (j, g) holds row j as f32 (exact re-check, residual) and as bf16 hi / lo MFMA fragments.  int8 rows are exact in
//      bf16: their lo fragments and the MFMAs that would multiply them do not exist ("terms that are zero").
//   2. coarse sweep, as mfma_assign.hip: centroid planes stream through LDS (double buffered), s(c) = |c|^2 - 2 x.c from
//      v_mfma_f32_32x32x16_bf16 (xh.ch + xl.ch + xh.cl), running four smallest per row, classification against
//      2E = 2^-12 (|x|^2 + max|c|^2): one, two or three candidates, or "recompute" (>= 4 inside the margin, NaN).
//   3. exact re-check IN the kernel: the lane pair evaluates l2_scalar / dot_scalar of its row against each candidate in the
//      reference's order -- lane (j, g) owns lane accumulators 8 g .. 8 g + 7 of the 16 (sums[i] += over the 16-chunks in
//      order), the fold ((0 + s0) + s1) + ... + s15 runs through lane g = 0 and is handed to g = 1 by one shuffle.  Partition id
//      and distance are therefore bit-equal to the exact kernels (argmin_value_float: strictly smaller, then smaller index;
//      non-finite rows: None).  Rows left undecided go to ma_recompute_kernel's list as before.
//   4. residual r = x - c[part] in f32 (f16 columns: rounded to binary16, residual.rs:96), split into bf16 hi / lo planes in
//      LDS (the centroid staging area, dead by now; 16-byte chunks rotated by the row number: conflict-free reads), with an upper
//      bound of every sub-vector's squared norm (bf16, rounded up).
//   5. PQ encode with the roles swapped: wave w now owns sub-quantisers w, w + 4, ... for ALL 128 rows; its 256 codewords sit
//      in registers as MFMA A fragments prepared once per call (xf_pq_prep_kernel), the rows stream from LDS as B.  The
//      whole surrogate comes out of the matrix pipe -- for sub-dimension 8 two K = 16 products per 32 x 32 tile,
//           A1 = [-2 ch | -2 cl]   B1 = [xh | xh]        A2 = [-2 ch | n1 n2 n3 1 0 ..]   B2 = [xl | 1 1 1 R 0 ..]
//      (|c|^2 = n1 + n2 + n3 as three bf16 terms, R = one bf16 >= |r_m|^2 + margin): s'' = |r_m - c|^2 + (R - |r_m|^2) >= 0 with
//      no VALU arithmetic at all, and no K slot multiplies zeros (sub-dimension 4: one product, [xh xh | xl 1 1 1 R]).
//      Epilogue = 3 VALU per (row, codeword): key = (bits & ~63) | slot (v_and_or_b32, slot an inline constant), second
//      smallest by v_med3_u32, smallest by v_min_u32 -- two trackers of 64 slots so the slot number stays an inline constant.
//      The bound.  Per element the three products miss xh.ec + xl.cl + er.c with |ec| <= 2^-18 |c|, |er| <= 2^-18 |r|, |xl| <= 2^-9 |r|,
//      |cl| <= 2^-9 |c| (two round-to-nearest bf16 terms each): 3.01 x 2^-18 |r_k||c_k|, summed (Cauchy-Schwarz) and doubled
//      2^-15.4 |r||c| <= 2^-16.4 (|r|^2 + |c|^2); |c|^2 as computed in f32 and split in three terms 2^-20 |c|^2; the matrix pipe's f32
//      accumulation of at most four K = 16 steps 2^-19 (|r|^2 + |c|^2): every surrogate is within E = 2^-15 (|r_m|^2 + max|c|^2) of
//      |r - c|^2 + (R - |r|^2) -- with a factor two to spare -- and the reference's own f32 distance within 2^-20 of that.  A key sits
//      at most 2^-15 of its value under its surrogate (8 cleared mantissa bits after the codeword number is widened in).  Hence:
//      second key - first key > 2 E + 2^-14 first key  ==>  the first is the reference's argmin (the offset R, one bf16 >= |r_m|^2 +
//      2 E, is a constant of the row and cancels).  Otherwise the (row, sub-quantiser) item goes to the fix kernel's list (exact
//      distances to all 256 codewords).  (pq_mfma.hip's margin, 2^-12, is four times this: at C3's shape -- 16-dimensional sub-vectors,
//      distances concentrate -- it left 4.7 % of the items undecided, gpurun r06n.)  Codes are gathered in LDS and leave as one
//      coalesced store per workgroup.
//
// HBM traffic: the row once, plus M + 8 bytes per row out.  Everything else (centroid planes 2 x 2 B x nlist x d, codeword
// fragments 16 KiB per sub-quantiser, f32 centroids of the candidates) is re-read from L2 by every workgroup.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include <hip/hip_fp16.h>

#include "common.h"
#include "exact.cuh"
#include "kernels.h"
#include "ma_common.cuh"

#pragma clang fp contract(off)

namespace lh {

struct XfArgs {
  const void *x;                 // [n][ldx] rows in the column's element type
  int64_t n, ldx;
  int k;                         // coarse centroids
  const uint16_t *cpl;           // [k rounded up to 64][2 d + 16] bf16 centroid planes (xf_cent_prep_kernel)
  const uint32_t *maxbits;       // [0] max |c|^2
  const float *cent;             // [k][d] f32
  int residual, round_f16, check_finite;
  const uint4 *pqa;              // [m][8 tiles][NMF][64 lanes] codeword fragments (xf_pq_prep_kernel)
  const float *pq_cmax2;         // [m] max |codeword|^2
  uint32_t *part_ids;            // [n]
  float *dists;                  // [n]
  uint8_t *codes;                // [n][m]
  uint32_t *afb_cnt, *afb_rows;  // rows for ma_recompute_kernel
  uint32_t *fb_cnt, *fb_items;   // [m_total], [m_total][n]: undecided rows per sub-quantiser, for xf_fix_kernel
  int m_total = 0;               // sub-quantisers of the whole index (row stride of `codes`)
  int64_t d_total = 0;           // elements per centroid row (xf_tail_kernel: a workgroup sees a 128-column block of it)
  int col_base = 0;              // xf_tail_kernel: first column of this launch's blocks
  // xf_pqtrain_kernel (codebook training E-step): sub-quantiser b's sub-vector of row r at tr_x + b * tr_boff + r * tr_ldx; ids / dists [b][tr_stride]
  const uint8_t *active = nullptr;      // [m_total] 0: the sub-quantiser's problem has converged, nothing of it is touched (ASSIGN: [1], the one problem)
  // MODE 2 (find_partitions over thousands of lists, see coarse_select_kernel in mfma_assign.hip): per (row, group of 16 centroids = one lane's
  // share of a 32-centroid block) the smallest surrogate with the member's register number in its four lowest mantissa bits and the group's
  // second smallest, as 16-byte records {key, key, second, second} of a lane's two groups at gkey[row][2 ng]; e2[row] = 2E; the centroid tiles
  // are split over blockIdx.y
  float *gkey = nullptr, *e2 = nullptr;
  int ng = 0, tiles_per_block = 0;
  const float *bias = nullptr;          // ASSIGN (k-means E-step with a balance factor): argmin over dist + bias[c] (kmeans.rs:317-369); maxbits[1] = max |bias|
  const float *tr_x = nullptr, *tr_cb = nullptr;
  int64_t tr_ldx = 0, tr_boff = 0, tr_stride = 0;
  uint32_t *tr_ids = nullptr;
  float *tr_dists = nullptr;
  unsigned long long *prof = nullptr;   // LANCE_HIP_XF_PROF=1: s_memtime ticks summed over the waves: [0] rows->registers [1] sweep [2] merge + exact re-check
                                        // [3] residual + barrier [4] PQ encode [5] codes out [6] waves [7] undecided PQ items
};

__device__ __forceinline__ uint32_t xf_med3(uint32_t x, uint32_t y, uint32_t z) {
  uint32_t r;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
  return r;
}

// bf16 >= v for v >= 0 (NaN stays NaN): the sub-vector norm kept in 16 bits
__device__ __forceinline__ uint32_t xf_bf16_up(float v) {
  const uint32_t u = __float_as_uint(v);
  if ((u & 0x7FFFFFFFu) >= 0x7F800000u) return u >> 16 | ((u & 0xFFFFu) ? 0x40u : 0u);
  return (u + 0xFFFFu) >> 16;
}

// ---- codeword fragments: one workgroup per sub-quantiser, thread = codeword -------------------------------------------
// SD = 8: per 32-codeword tile two A operands (see the header); SD = 4: one.  Lane (j, g) of the MFMA reads 16 bytes.
// SD = 16: three A operands per tile -- AH = -2 ch (lane half g: dimensions 8 g .. 8 g + 7), AL = -2 cl, AN = [n1 n2 n3 1 0 .. | 0 ..]; the
// kernel runs AH.BH + AL.BH + AH.BL + AN.BN with BH / BL the residual's hi / lo halves and BN = [1 1 1 R 0 .. | 0 ..].
template <int SD> struct XfSd;
template <> struct XfSd<4> { static constexpr int NA = 1; };
template <> struct XfSd<8> { static constexpr int NA = 2; };
template <> struct XfSd<16> { static constexpr int NA = 3; };

template <int SD>
__global__ __launch_bounds__(256) void xf_pq_prep_kernel(const float *__restrict__ codebook, uint4 *__restrict__ pqa, float *__restrict__ cmax2,
                                                         uint32_t *__restrict__ zero_cnt) {
  constexpr int NMF = XfSd<SD>::NA;
  __shared__ uint32_t s_max;
  const int m = blockIdx.x, c = threadIdx.x;
  if (c == 0) { s_max = 0u; if (zero_cnt) zero_cnt[m] = 0u; }      // (the training E-step: one launch fewer per Lloyd iteration)
  __syncthreads();
  const float *cw = codebook + ((int64_t)m * 256 + c) * SD;
  uint32_t hi[SD], lo[SD];
  float s = 0.0f;
#pragma unroll
  for (int e = 0; e < SD; ++e) {
    const float v = cw[e];
    s += v * v;
    const float w = -2.0f * v;
    const uint32_t hb = bf16_rne_bits(w);
    hi[e] = hb & 0xFFFFu;
    lo[e] = bf16_rne_bits(w - bf16_bits_to_float(hb)) & 0xFFFFu;
  }
  const uint32_t n1 = bf16_rne_bits(s) & 0xFFFFu;
  const float r1 = s - bf16_bits_to_float(n1);
  const uint32_t n2 = bf16_rne_bits(r1) & 0xFFFFu;
  const uint32_t n3 = bf16_rne_bits(r1 - bf16_bits_to_float(n2)) & 0xFFFFu;
  if (s == s) atomicMax(&s_max, __float_as_uint(fabsf(s)));
  const int t = c >> 5, j = c & 31;
  uint4 *dst = pqa + ((int64_t)(m * 8 + t) * NMF) * 64;
  if constexpr (SD == 16) {
    dst[j] = make_uint4(hi[0] | hi[1] << 16, hi[2] | hi[3] << 16, hi[4] | hi[5] << 16, hi[6] | hi[7] << 16);
    dst[32 + j] = make_uint4(hi[8] | hi[9] << 16, hi[10] | hi[11] << 16, hi[12] | hi[13] << 16, hi[14] | hi[15] << 16);
    dst[64 + j] = make_uint4(lo[0] | lo[1] << 16, lo[2] | lo[3] << 16, lo[4] | lo[5] << 16, lo[6] | lo[7] << 16);
    dst[96 + j] = make_uint4(lo[8] | lo[9] << 16, lo[10] | lo[11] << 16, lo[12] | lo[13] << 16, lo[14] | lo[15] << 16);
    dst[128 + j] = make_uint4(n1 | n2 << 16, n3 | 0x3F80u << 16, 0u, 0u);
    dst[160 + j] = make_uint4(0u, 0u, 0u, 0u);
  } else if constexpr (SD == 8) {
    const uint4 h = make_uint4(hi[0] | hi[1] << 16, hi[2] | hi[3] << 16, hi[4] | hi[5] << 16, hi[6] | hi[7] << 16);
    const uint4 l = make_uint4(lo[0] | lo[1] << 16, lo[2] | lo[3] << 16, lo[4] | lo[5] << 16, lo[6] | lo[7] << 16);
    dst[j] = h;                                  // A1, g = 0: -2 ch
    dst[32 + j] = l;                             // A1, g = 1: -2 cl
    dst[64 + j] = h;                             // A2, g = 0: -2 ch
    dst[96 + j] = make_uint4(n1 | n2 << 16, n3 | 0x3F80u << 16, 0u, 0u);   // A2, g = 1: |c|^2 in three terms, and a 1 for the row's offset
  } else {
    dst[j] = make_uint4(hi[0] | hi[1] << 16, hi[2] | hi[3] << 16, lo[0] | lo[1] << 16, lo[2] | lo[3] << 16);       // g = 0: [-2 ch | -2 cl]
    dst[32 + j] = make_uint4(hi[0] | hi[1] << 16, hi[2] | hi[3] << 16, n1 | n2 << 16, n3 | 0x3F80u << 16);          // g = 1: [-2 ch | n1 n2 n3 1]
  }
  __syncthreads();
  if (c == 0) cmax2[m] = __uint_as_float(s_max);
}

// ---- coarse centroid planes: per centroid [ -2 c hi (D) | n1 n2 n3 1 0.. (16) | -2 c lo (D) ] bf16 (dot: -c, no norm terms) -------------------
// The extra K = 16 step carries |c|^2 (three bf16 terms) and a 1 that multiplies the row's offset R, so the surrogate leaves the
// matrix pipe complete: s'(c) = R + |c|^2 - 2 x.c.  Rows c >= k (padding to whole 64-centroid tiles) get |c|^2 = +inf: never selected.
template <int METRIC>
__global__ __launch_bounds__(64) void xf_cent_prep_kernel(const float *__restrict__ cent, int k, int d, uint16_t *__restrict__ cpl,
                                                          uint32_t *__restrict__ maxbits, const float *__restrict__ bias = nullptr,
                                                          const uint8_t *__restrict__ active = nullptr) {
  if (active && !active[0]) return;
  const int c = blockIdx.x, w = 2 * d + 16;
  uint16_t *row = cpl + (int64_t)c * w;
  const float scale = METRIC == METRIC_DOT ? -1.0f : -2.0f;
  float s = 0.0f;
  for (int e = threadIdx.x; e < d; e += 64) {
    const float v = c < k ? cent[(int64_t)c * d + e] : 0.0f;
    const float t = scale * v;
    const uint32_t hb = bf16_rne_bits(t);
    row[e] = (uint16_t)hb;
    row[d + 16 + e] = (uint16_t)bf16_rne_bits(t - bf16_bits_to_float(hb));
    s += v * v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (threadIdx.x < 16) {
    uint32_t v = 0u;
    if (threadIdx.x == 3) v = 0x3F80u;                        // 1.0: multiplies the row's offset R
    if (METRIC != METRIC_DOT || c >= k || bias) {
      // (with a bias the norm slot carries |c|^2 + bias[c] -- dot: bias[c] -- so the surrogate still leaves the matrix pipe complete)
      const float n = c < k ? ((METRIC != METRIC_DOT ? s : 0.0f) + (bias ? bias[c] : 0.0f)) : INFINITY;
      const uint32_t n1 = bf16_rne_bits(n) & 0xFFFFu;
      const float r1 = c < k ? n - bf16_bits_to_float(n1) : 0.0f;
      const uint32_t n2 = bf16_rne_bits(r1) & 0xFFFFu;
      const uint32_t n3 = bf16_rne_bits(r1 - bf16_bits_to_float(n2)) & 0xFFFFu;
      if (threadIdx.x == 0) v = n1;
      if (threadIdx.x == 1) v = n2;
      if (threadIdx.x == 2) v = n3;
    }
    row[d + threadIdx.x] = (uint16_t)v;
  }
  if (threadIdx.x == 0 && c < k && s == s) atomicMax(&maxbits[0], __float_as_uint(fabsf(s)));
  if (threadIdx.x == 0 && c < k && bias) atomicMax(&maxbits[1], __float_as_uint(fabsf(bias[c])));      // (a NaN bias: a huge bound, every row recomputed exactly)
}

// LDS (bytes) of one workgroup: the centroid tiles of the sweep and the residual planes of the encode alias each other; the
// workgroup's queues of undecided items -- per sub-quantiser of the block a counter, a base in the global list and XF_QBYTES / M one-byte
// row numbers -- sit behind both
constexpr int XF_QBYTES = 2048;
template <int KS, int SD>
constexpr size_t xf_lds_main() {
  constexpr int D = KS * 16, M = D / SD;
  constexpr size_t a = (size_t)MA_ROWS * (D + 4) * 4;                                            // rows, f32
  constexpr size_t b0 = (size_t)2 * MA_CT * (2 * D + 24) * 2;                                    // two centroid tiles
  constexpr size_t b = a > b0 ? a : b0;
  constexpr size_t c = (size_t)2 * MA_ROWS * D * 2 + (size_t)MA_ROWS * M * 2 + (size_t)MA_ROWS * ((M + 3) / 4) * 4 + MA_ROWS;   // residual planes, norms, codes, flags
  return ((b > c ? b : c) + 15) / 16 * 16;
}
template <int KS, int SD>
constexpr size_t xf_lds_bytes() { return xf_lds_main<KS, SD>() + 256 + (size_t)XF_QBYTES; }

typedef __bf16 xf_bf2 __attribute__((ext_vector_type(2)));
// two floats -> packed bf16 (v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN)
__device__ __forceinline__ uint32_t xf_cvt2(float a, float b) {
  const f2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, xf_bf2));
}
// eight floats -> bf16 hi / lo fragments (x = hi + lo to 2^-17 |x|: the remainder v - hi is exact in f32)
template <bool LO>
__device__ __forceinline__ void xf_split8(const float (&v)[8], uint4 &hi, uint4 &lo) {
  uint32_t h[4], l[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    h[q] = xf_cvt2(v[2 * q], v[2 * q + 1]);
    if constexpr (LO) l[q] = xf_cvt2(v[2 * q] - __uint_as_float(h[q] << 16), v[2 * q + 1] - __uint_as_float(h[q] & 0xFFFF0000u));
  }
  hi = make_uint4(h[0], h[1], h[2], h[3]);
  lo = make_uint4(l[0], l[1], l[2], l[3]);
}

// exact distance of the pair's row to centroid row `y` in the reference's summation order; every lane gets the value
template <int KS, int METRIC>
__device__ __forceinline__ float xf_exact(const float (&xf)[KS][8], const float *__restrict__ y, int j, int g) {
  float sums[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) sums[e] = 0.0f;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const f4 c0 = *reinterpret_cast<const f4 *>(y + s * 16 + g * 8), c1 = *reinterpret_cast<const f4 *>(y + s * 16 + g * 8 + 4);
    const float cv[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if constexpr (METRIC == METRIC_DOT) {
        sums[e] = sums[e] + xf[s][e] * cv[e];
      } else {
        const float diff = xf[s][e] - cv[e];
        sums[e] = sums[e] + diff * diff;
      }
    }
  }
  float t = 0.0f;
#pragma unroll
  for (int e = 0; e < 8; ++e) t = t + sums[e];           // g = 0: ((0 + s0) + s1) + ... + s7
  const float t0 = __shfl(t, j, 64);
  float u = t0;
#pragma unroll
  for (int e = 0; e < 8; ++e) u = u + sums[e];           // g = 1 continues with s8 .. s15
  const float tot = __shfl(u, 32 + j, 64);
  return finish_metric<METRIC>(0.0f + tot);
}

// An undecided (row, sub-quantiser) item goes into the workgroup's LDS queue of that sub-quantiser (written by the ONE wave that owns it in
// the PQ phase: no sub-dword stores of different waves into one dword), flushed with one global atomic per (workgroup, sub-quantiser);
// a full queue (rows full of ties) or a caller outside the PQ phase (direct = true) appends to the global list itself.
template <int QC>
__device__ __forceinline__ void xf_queue_item(const XfArgs &p, uint32_t *q_cnt, uint8_t *q_rows, int m_local, int m_global, uint32_t lrow, int64_t row0, bool direct) {
  uint32_t slot = (uint32_t)QC;
  if (!direct) slot = atomicAdd(&q_cnt[m_local], 1u);
  if (slot < (uint32_t)QC) {
    q_rows[m_local * QC + slot] = (uint8_t)lrow;
  } else {
    const uint32_t s2 = atomicAdd(&p.fb_cnt[m_global], 1u);
    p.fb_items[(int64_t)m_global * p.n + s2] = (uint32_t)(row0 + lrow);
  }
}

// Undecided items: one wave each, exact distances of the (residual) sub-vector to the sub-quantiser's 256 codewords in the reference's
// order (l2_scalar, kmeans.rs:1350-1369), argmin_value_float semantics (first strictly smallest, NaN / +inf never selected, none ->
// code 0: pq.rs:165 unwrap_or(0)) -- pq_mfma_fix_kernel's arithmetic on a flat list (the codewords come from L2: items are ~1 % of the work)
struct XfFixArgs {
  const void *x; int64_t ldx, n;
  const float *cent; const uint32_t *part_ids; int residual, round_f16;
  const float *codebook; int m;
  const uint32_t *cnt, *items;        // [m], [m][n]
  uint8_t *codes;
};
template <int SD, typename TX>
__global__ __launch_bounds__(256) void xf_fix_kernel(XfFixArgs a) {
  __shared__ __attribute__((aligned(16))) float cbf[256 * SD];      // the sub-quantiser's codebook
  const int m = blockIdx.y;
  const uint32_t cnt = a.cnt[m];
  if (blockIdx.x * 4u >= cnt) return;    // uniform: nothing left for this workgroup
  const float *cb = a.codebook + (int64_t)m * 256 * SD;
  for (int i = threadIdx.x; i < 256 * SD / 4; i += 256) reinterpret_cast<f4 *>(cbf)[i] = reinterpret_cast<const f4 *>(cb)[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const uint32_t nwaves = gridDim.x * 4;
  for (uint32_t it = blockIdx.x * 4 + (threadIdx.x >> 6); it < cnt; it += nwaves) {
    const int64_t row = a.items[(int64_t)m * a.n + it];
    RegVec<SD> rv;
#pragma unroll
    for (int i = 0; i < SD / 4; ++i) rv.q[i] = f4{0.0f, 0.0f, 0.0f, 0.0f};
    const uint32_t part = a.residual ? a.part_ids[row] : 0u;
    if (!(a.residual && part == LANCE_HIP_NONE)) {       // no partition: the zero vector
      const TX *src = static_cast<const TX *>(a.x) + row * a.ldx + (int64_t)m * SD;
#pragma unroll
      for (int i = 0; i < SD / 4; ++i) {
        f4 v = load4(src + 4 * i);
        if (a.residual) {
          v = v - *reinterpret_cast<const f4 *>(a.cent + (int64_t)part * a.ldx + (int64_t)m * SD + 4 * i);
          if (a.round_f16) {
            v.x = __half2float(__float2half_rn(v.x)); v.y = __half2float(__float2half_rn(v.y));
            v.z = __half2float(__float2half_rn(v.z)); v.w = __half2float(__float2half_rn(v.w));
          }
        }
        rv.q[i] = v;
      }
    }
    float best = INFINITY;
    uint32_t bi = LANCE_HIP_NONE;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t c = (uint32_t)(u * 64 + lane);
      const float v = dist_exact<SD, METRIC_L2>(rv, &cbf[c * SD]);
      if (v < best) { best = v; bi = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64);
      const uint32_t oi = __shfl_xor(bi, o, 64);
      if (ov < best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) a.codes[row * a.m + m] = bi == LANCE_HIP_NONE ? (uint8_t)0 : (uint8_t)bi;
  }
}

// LDS of the encode half (phases 4 and 5), shared by xf_kernel and xf_tail_kernel
template <int KS, int SD>
struct XfLds {
  static constexpr int D = KS * 16, M = D / SD, MW = (M + 3) / 4;
  __device__ static uint16_t *rh(char *smem) { return reinterpret_cast<uint16_t *>(smem); }                    // [128][D] hi (16-byte chunks rotated by the row)
  __device__ static uint16_t *rl(char *smem) { return rh(smem) + (size_t)MA_ROWS * D; }                        // [128][D] lo
  __device__ static uint16_t *xn2s(char *smem) { return rl(smem) + (size_t)MA_ROWS * D; }                      // [128][M] bf16 upper bounds of |r_m|^2
  // codes: [wave][MW][128] -- a byte is written by the wave that owns the sub-quantiser, its dword neighbours are the adjacent rows
  // of the SAME store instruction (round 6, first version: [row][m] put four waves' bytes into one dword, and about one byte in two
  // million came out stale: concurrent sub-dword LDS stores of different waves to one dword are not safe)
  __device__ static uint8_t *codes(char *smem) { return reinterpret_cast<uint8_t *>(xn2s(smem) + (size_t)MA_ROWS * M); }
  __device__ static uint8_t *rskip(char *smem) { return codes(smem) + (size_t)4 * MW * MA_ROWS; }               // [128] 1: no items of this row from the PQ phase
  static constexpr int QC = XF_QBYTES / M >= 128 ? 128 : XF_QBYTES / M / 4 * 4;           // queue entries per sub-quantiser (dword-aligned lists)
  __device__ static uint32_t *q_cnt(char *smem) { return reinterpret_cast<uint32_t *>(smem + xf_lds_main<KS, SD>()); }            // [M] counts
  __device__ static uint32_t *q_base(char *smem) { return reinterpret_cast<uint32_t *>(smem + xf_lds_main<KS, SD>() + 128); }      // [M] bases in the global lists
  __device__ static uint8_t *q_rows(char *smem) { return reinterpret_cast<uint8_t *>(smem + xf_lds_main<KS, SD>() + 256); }        // [M][QC]
};

// phase 4: this lane's pieces of the row (xf: dimensions 16 s + 8 g .. + 7 of the block) minus the centroid's -> hi / lo planes + norms
template <int KS, int SD>
__device__ __forceinline__ void xf_residual_to_lds(const float (&xf)[KS][8], const float *__restrict__ cb, bool sub, bool zero, int round_f16, char *smem,
                                                   int lrow, int g, bool skip) {
  constexpr int D = KS * 16, M = D / SD, NCH = D / 8;
  uint16_t *rh = XfLds<KS, SD>::rh(smem), *rl = XfLds<KS, SD>::rl(smem), *xn2s = XfLds<KS, SD>::xn2s(smem);
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    float r[8];
    if (sub) {
      const f4 c0 = *reinterpret_cast<const f4 *>(cb + s * 16 + g * 8), c1 = *reinterpret_cast<const f4 *>(cb + s * 16 + g * 8 + 4);
      const float cv[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        r[e] = xf[s][e] - cv[e];
        if (round_f16) r[e] = __half2float(__float2half_rn(r[e]));      // `*v - *cent` in half::f16 (residual.rs:96)
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] = zero ? 0.0f : xf[s][e];
    }
    uint4 h4, l4;
    xf_split8<true>(r, h4, l4);
    const int chunk = 2 * s + g, slot = (chunk + lrow) % NCH;
    *reinterpret_cast<uint4 *>(rh + (size_t)lrow * D + slot * 8) = h4;
    *reinterpret_cast<uint4 *>(rl + (size_t)lrow * D + slot * 8) = l4;
    // (q * 1.0000005: the fused sum may sit an ulp under the true norm)
    if constexpr (SD == 16) {
      float q = 0.0f;
#pragma unroll
      for (int e = 0; e < 8; ++e) q = __builtin_fmaf(r[e], r[e], q);
      q += __shfl_xor(q, 32, 64);
      if (g == 0) xn2s[lrow * M + s] = (uint16_t)xf_bf16_up(q * 1.0000005f);
    } else if constexpr (SD == 8) {
      float q = 0.0f;
#pragma unroll
      for (int e = 0; e < 8; ++e) q = __builtin_fmaf(r[e], r[e], q);
      xn2s[lrow * M + chunk] = (uint16_t)xf_bf16_up(q * 1.0000005f);
    } else {
      float q0 = 0.0f, q1 = 0.0f;
#pragma unroll
      for (int e = 0; e < 4; ++e) { q0 = __builtin_fmaf(r[e], r[e], q0); q1 = __builtin_fmaf(r[4 + e], r[4 + e], q1); }
      *reinterpret_cast<uint32_t *>(xn2s + lrow * M + 2 * chunk) = (xf_bf16_up(q0 * 1.0000005f) & 0xFFFFu) | (xf_bf16_up(q1 * 1.0000005f) << 16);
    }
  }
  if (g == 0) XfLds<KS, SD>::rskip(smem)[lrow] = (uint8_t)(skip ? 1 : 0);
}

// phase 5 (+ the undecided items and the code bytes on their way out).  m0: the block's first sub-quantiser in the whole index.
template <int KS, int SD, bool PROF>
__device__ __forceinline__ void xf_pq_phase(const XfArgs &p, char *smem, int m0, int64_t row0, uint32_t &pc_und, long long &prof_pq, long long &pprev) {
  constexpr int D = KS * 16, M = D / SD, NCH = D / 8, NA = XfSd<SD>::NA, MW = (M + 3) / 4;
  uint16_t *rh = XfLds<KS, SD>::rh(smem), *rl = XfLds<KS, SD>::rl(smem), *xn2s = XfLds<KS, SD>::xn2s(smem);
  uint8_t *codes_s = XfLds<KS, SD>::codes(smem), *rskip = XfLds<KS, SD>::rskip(smem);
  constexpr int QC = XfLds<KS, SD>::QC;
  uint32_t *q_cnt = XfLds<KS, SD>::q_cnt(smem), *q_base = XfLds<KS, SD>::q_base(smem);
  uint8_t *q_rows = XfLds<KS, SD>::q_rows(smem);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, g = lane >> 5;
  const bf16x8 ones = {(short)0x3F80, (short)0x3F80, (short)0x3F80, 0, 0, 0, 0, 0};
  const bf16x8 zeros8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 1
  for (int m = wave, mi = 0; m < M; m += 4, ++mi) {
    if (p.active && !p.active[m0 + m]) continue;      // (training: a converged problem; wave-uniform)
    bf16x8 af[8][NA];
    {
      const uint4 *src = p.pqa + (int64_t)(m0 + m) * 8 * NA * 64 + lane;
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int f = 0; f < NA; ++f) af[t][f] = __builtin_bit_cast(bf16x8, src[(t * NA + f) * 64]);
    }
    const float cmax2 = p.pq_cmax2[m0 + m];
#pragma unroll 1
    for (int rg = 0; rg < MA_ROWS / 32; ++rg) {
      const int lrow = rg * 32 + j;
      bf16x8 b1, b2, b3 = zeros8;
      if constexpr (SD == 16) {
        const int slot = (2 * m + g + lrow) % NCH;
        b1 = *reinterpret_cast<const bf16x8 *>(rh + (size_t)lrow * D + slot * 8);       // BH: dimensions 8 g .. 8 g + 7 of the sub-vector
        b2 = *reinterpret_cast<const bf16x8 *>(rl + (size_t)lrow * D + slot * 8);       // BL
        b3 = g == 0 ? ones : zeros8;                                                    // BN (+ R below)
      } else if constexpr (SD == 8) {
        const int slot = (m + lrow) % NCH;
        b1 = *reinterpret_cast<const bf16x8 *>(rh + (size_t)lrow * D + slot * 8);
        const bf16x8 lo8 = *reinterpret_cast<const bf16x8 *>(rl + (size_t)lrow * D + slot * 8);
        b2 = g == 0 ? lo8 : ones;
      } else {
        const int slot = ((m >> 1) + lrow) % NCH;
        const uint2 h4 = *reinterpret_cast<const uint2 *>(rh + (size_t)lrow * D + slot * 8 + (m & 1) * 4);
        const uint2 l4 = *reinterpret_cast<const uint2 *>(rl + (size_t)lrow * D + slot * 8 + (m & 1) * 4);
        const uint4 bb = g == 0 ? make_uint4(h4.x, h4.y, h4.x, h4.y) : make_uint4(l4.x, l4.y, 0x3F803F80u, 0x00003F80u);
        b1 = __builtin_bit_cast(bf16x8, bb);
        b2 = b1;
      }
      const float xq = __uint_as_float((uint32_t)xn2s[lrow * M + m] << 16);
      const float margin = 0.00006103515625f * (xq + cmax2);     // 2 E, E = 2^-15 (|r_m|^2 + max|c|^2): see the bound in the header
      // the row's offset R = bf16 >= |r_m|^2 + margin rides in the fourth slot of the "ones" vector (the codeword side holds a 1 there):
      // every surrogate is >= 0 (float order == unsigned order of the bits) and the accumulator starts from the inline constant 0
      const uint32_t rb = xf_bf16_up(xq + margin);
      if constexpr (SD == 16) {
        if (g == 0) b3[3] = (short)rb;
      } else if constexpr (SD == 8) {
        if (g == 1) b2[3] = (short)rb;
      } else {
        if (g == 1) { b1[7] = (short)rb; b2 = b1; }
      }
      const f32x16 c0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      uint32_t a1 = 0xFFFFFFFFu, a2 = 0xFFFFFFFFu, q1 = 0xFFFFFFFFu, q2 = 0xFFFFFFFFu;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t][0], b1, c0, 0, 0, 0);
        if constexpr (SD == 8) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t][1], b2, acc, 0, 0, 0);
        if constexpr (SD == 16) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t][1], b1, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t][0], b2, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t][2], b3, acc, 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const uint32_t key = (__float_as_uint(acc[v]) & 0xFFFFFFC0u) | (uint32_t)((t & 3) * 16 + v);
          if (t < 4) { a2 = xf_med3(a1, a2, key); a1 = min(a1, key); }
          else { q2 = xf_med3(q1, q2, key); q1 = min(q1, key); }
        }
      }
      // slot -> codeword: tile (slot >> 4) (+ 4 for the second tracker), register v = slot & 15 holds codeword 8 (v >> 2) + 4 g + (v & 3)
      auto widen = [&](uint32_t key, uint32_t half) {
        const uint32_t sl = key & 63u, v = sl & 15u;
        const uint32_t cw = ((sl >> 4) + 4u * half) * 32u + 8u * (v >> 2) + 4u * (uint32_t)g + (v & 3u);
        return (key & 0xFFFFFF00u) | cw;
      };
      const uint32_t ka1 = widen(a1, 0), ka2 = widen(a2, 0), kb1 = widen(q1, 1), kb2 = widen(q2, 1);
      uint32_t m1 = min(ka1, kb1), m2 = min(max(ka1, kb1), min(ka2, kb2));
      {   // the partner lane holds the other 128 codewords of the row
        const uint32_t p1 = __shfl_xor(m1, 32, 64), p2 = __shfl_xor(m2, 32, 64);
        const uint32_t lo = min(m1, p1), hi2 = max(m1, p1);
        m2 = min(hi2, min(m2, p2));
        m1 = lo;
      }
      if (g == 0) {
        const float s1 = __uint_as_float(m1 & 0xFFFFFF00u), s2 = __uint_as_float(m2 & 0xFFFFFF00u);
        // margin > 2^-100: the relative bounds above assume no product, sum or cleared mantissa bit sits in the denormal range
        // (16 + 16 slots x 2^-126, 255 ulps of a denormal key); columns scaled like 1e-20 take the exact kernel (NaN anywhere: false)
        // + 2^-14 s1: the winner's key sits up to 2^-15 of its value under its surrogate (8 cleared mantissa bits)
        // s1 < 2^120: the winner's exact distance (<= s1 + E) is then finite -- the training E-step writes it without a second look
        const bool decided = (margin < INFINITY) && (margin > 7.888609052210118e-31f) && (s2 - s1 > margin + 0.00006103515625f * s1) && (s1 < 1.329228e36f);
        codes_s[(wave * MW + mi) * MA_ROWS + lrow] = (uint8_t)(m1 & 0xFFu);      // (an undecided item's byte is rewritten by the fix kernel)
        if (!decided && !rskip[lrow]) {
          if constexpr (PROF) ++pc_und;
          xf_queue_item<QC>(p, q_cnt, q_rows, m, m0 + m, (uint32_t)lrow, row0, false);
        }
      }
    }
  }
  if constexpr (PROF) { const long long t = clock64(); prof_pq += t - pprev; pprev = t; }
  __syncthreads();
  // ---- the workgroup's undecided items: one atomic per sub-quantiser that has any ----
  {
    if (threadIdx.x < M) {
      const uint32_t c = min(q_cnt[threadIdx.x], (uint32_t)QC);
      q_base[threadIdx.x] = c ? atomicAdd(&p.fb_cnt[m0 + threadIdx.x], c) : 0u;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < M * QC; i += 256) {
      const int m = i / QC, e = i - m * QC;
      if ((uint32_t)e < min(q_cnt[m], (uint32_t)QC)) p.fb_items[(int64_t)(m0 + m) * p.n + q_base[m] + e] = (uint32_t)(row0 + q_rows[i]);
    }
  }
  // ---- codes: [wave][mi][row] in LDS -> [row][m0 + m] in HBM ----
  if (p.codes) {
    const int64_t nrows = p.n - row0 < MA_ROWS ? p.n - row0 : MA_ROWS;
    if (p.m_total == M) {      // the block is the whole code row: one coalesced store per workgroup
      const int nbytes = (int)nrows * M;
      uint8_t *dst = p.codes + row0 * M;
      const int nw = nbytes / 4;
      for (int i = threadIdx.x; i < nw; i += 256) {
        uint32_t w = 0u;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int o = 4 * i + b, r = o / M, m = o - r * M;
          w |= (uint32_t)codes_s[((m & 3) * MW + (m >> 2)) * MA_ROWS + r] << (8 * b);
        }
        reinterpret_cast<uint32_t *>(dst)[i] = w;
      }
      for (int o = nw * 4 + threadIdx.x; o < nbytes; o += 256) {
        const int r = o / M, m = o - r * M;
        dst[o] = codes_s[((m & 3) * MW + (m >> 2)) * MA_ROWS + r];
      }
    } else {                   // a column block of a long row: M bytes at [row][m0 ..]
      for (int o = threadIdx.x; o < (int)nrows * M; o += 256) {
        const int r = o / M, m = o - r * M;
        p.codes[(row0 + r) * p.m_total + m0 + m] = codes_s[((m & 3) * MW + (m >> 2)) * MA_ROWS + r];
      }
    }
  }
}

// ---- long rows (d > 128: C3's 1536-dimensional embeddings): residual + PQ encode of one 128-column block per workgroup ------------------
// The coarse quantiser of such rows is the K-tiled matrix-core kernel of mfma_assign.hip; this kernel is the second half of the
// transform.  Round 5 ran it as residual_kernel (6 GB read, 6 GB written at C3) + the exact pairwise kernel over 96 sub-quantisers
// (24.7 ms of the 56 ms transform, profiles/r06k_c3_xform_kernel_stats.csv): the undecided-row lists of pq_mfma.hip did not fit its
// 256 MB budget at M = 96.  Here a workgroup takes 128 rows x 128 columns (8 sub-quantisers of 16): the piece of the row and of its
// centroid are read once, the residual never exists in HBM, and phases 4 / 5 of xf_kernel do the rest.
template <int KS, int SD>
__global__ __launch_bounds__(256, 2) void xf_tail_kernel(XfArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int D = KS * 16, XS = D + 4;
  float *xs = reinterpret_cast<float *>(smem);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, g = lane >> 5;
  const int64_t row0 = (int64_t)blockIdx.x * MA_ROWS;
  const int col0 = p.col_base + (int)blockIdx.y * D;
  uint32_t *q_cnt = XfLds<KS, SD>::q_cnt(smem);
  if (threadIdx.x < XfLds<KS, SD>::M) q_cnt[threadIdx.x] = 0u;
  const int64_t row = row0 + wave * 32 + j;
  const bool valid = row < p.n;
  constexpr int NLD = MA_ROWS * (D / 4) / 256;
  {
    f4 stage[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int idx = threadIdx.x + 256 * u, r = idx / (D / 4), c4 = idx - r * (D / 4);
      stage[u] = f4{0.0f, 0.0f, 0.0f, 0.0f};
      if (row0 + r < p.n) stage[u] = *reinterpret_cast<const f4 *>(static_cast<const float *>(p.x) + (row0 + r) * p.ldx + col0 + 4 * c4);
    }
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int idx = threadIdx.x + 256 * u, r = idx / (D / 4), c4 = idx - r * (D / 4);
      *reinterpret_cast<f4 *>(&xs[r * XS + 4 * c4]) = stage[u];
    }
  }
  const uint32_t best = valid ? p.part_ids[row] : LANCE_HIP_NONE;
  __syncthreads();
  float xf[KS][8];
  {
    const float *xr = xs + (wave * 32 + j) * XS + g * 8;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const f4 a = *reinterpret_cast<const f4 *>(xr + s * 16), b = *reinterpret_cast<const f4 *>(xr + s * 16 + 4);
      xf[s][0] = a.x; xf[s][1] = a.y; xf[s][2] = a.z; xf[s][3] = a.w; xf[s][4] = b.x; xf[s][5] = b.y; xf[s][6] = b.z; xf[s][7] = b.w;
    }
  }
  __syncthreads();      // xs is dead: the residual planes take its place
  {
    const bool sub = p.residual && best != LANCE_HIP_NONE;
    const bool zero = !valid || (p.residual && best == LANCE_HIP_NONE);
    xf_residual_to_lds<KS, SD>(xf, p.cent + (int64_t)(sub ? best : 0u) * p.d_total + col0, sub, zero, p.round_f16, smem, wave * 32 + j, g, !valid);
  }
  __syncthreads();
  uint32_t und = 0;
  long long t0 = 0, t1 = 0;
  xf_pq_phase<KS, SD, false>(p, smem, col0 / SD, row0, und, t0, t1);
}

// ---- the codebook training's E-step (pq/builder.rs:89-157 -> kmeans.rs:317-369 per sub-quantiser) on the same machinery --------------------
// Round 5's pq_mfma_estep_kernel (pq_mfma.hip) is the encode half of the round-5 transform: software bf16 splits, three half-empty K = 8
// products and 4-5 VALU per (row, codeword) pair -- 73 us per Lloyd iteration at C2 (16 x 65,536 x 256 x 8), fifty times per build.  Here a
// workgroup takes 128 rows x 128 columns of the residual matrix (the sub-quantisers' slices [b][rows][sd] or the row-major matrix: any
// (row stride, problem offset)), phases 4 / 5 of xf_kernel pick every item's codeword, and the epilogue evaluates the winner's distance
// exactly (l2_scalar order: the loss is a sum of these) -- ids and distances bit-equal to pairwise_kernel's.  Undecided items go to the
// lists of pq_mfma_fix_kernel, which overwrites what the epilogue wrote for them.
template <int SD>
__global__ __launch_bounds__(256, 2) void xf_pqtrain_kernel(XfArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KS = 8, D = KS * 16, XS = D + 4, M = D / SD, MW = (M + 3) / 4, PPS = MA_ROWS * (SD / 4);      // 16-byte pieces per sub-quantiser slice of the tile
  float *xs = reinterpret_cast<float *>(smem);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, g = lane >> 5;
  const int64_t row0 = (int64_t)blockIdx.x * MA_ROWS;
  const int m0 = (int)blockIdx.y * M;
  uint32_t *q_cnt = XfLds<KS, SD>::q_cnt(smem);
  if (threadIdx.x < M) q_cnt[threadIdx.x] = 0u;
  const bool valid = row0 + wave * 32 + j < p.n;
  constexpr int NLD = MA_ROWS * (D / 4) / 256;
  {
    f4 stage[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int idx = threadIdx.x + 256 * u, mm = idx / PPS, rem = idx - mm * PPS, r = rem / (SD / 4), e4 = rem - r * (SD / 4);
      stage[u] = f4{0.0f, 0.0f, 0.0f, 0.0f};
      if (row0 + r < p.n) stage[u] = *reinterpret_cast<const f4 *>(p.tr_x + (int64_t)(m0 + mm) * p.tr_boff + (row0 + r) * p.tr_ldx + 4 * e4);
    }
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int idx = threadIdx.x + 256 * u, mm = idx / PPS, rem = idx - mm * PPS, r = rem / (SD / 4), e4 = rem - r * (SD / 4);
      *reinterpret_cast<f4 *>(&xs[r * XS + mm * SD + 4 * e4]) = stage[u];
    }
  }
  __syncthreads();
  float xf[KS][8];
  {
    const float *xr = xs + (wave * 32 + j) * XS + g * 8;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const f4 a = *reinterpret_cast<const f4 *>(xr + s * 16), b = *reinterpret_cast<const f4 *>(xr + s * 16 + 4);
      xf[s][0] = a.x; xf[s][1] = a.y; xf[s][2] = a.z; xf[s][3] = a.w; xf[s][4] = b.x; xf[s][5] = b.y; xf[s][6] = b.z; xf[s][7] = b.w;
    }
  }
  __syncthreads();      // xs is dead: the bf16 planes take its place
  xf_residual_to_lds<KS, SD>(xf, nullptr, false, !valid, 0, smem, wave * 32 + j, g, !valid);
  __syncthreads();
  uint32_t und = 0;
  long long t0 = 0, t1 = 0;
  xf_pq_phase<KS, SD, false>(p, smem, m0, row0, und, t0, t1);
  // ---- the winners' exact distances: consecutive threads = consecutive rows of one sub-quantiser ----
  const uint8_t *codes_s = XfLds<KS, SD>::codes(smem);
  const int nrows = (int)(p.n - row0 < MA_ROWS ? p.n - row0 : MA_ROWS);
#pragma unroll 2
  for (int i = threadIdx.x; i < MA_ROWS * M; i += 256) {
    const int mm = i / MA_ROWS, r = i - mm * MA_ROWS, mg = m0 + mm;
    if (r >= nrows || (p.active && !p.active[mg])) continue;
    const uint32_t c = codes_s[((mm & 3) * MW + (mm >> 2)) * MA_ROWS + r];
    const float *src = p.tr_x + (int64_t)mg * p.tr_boff + (row0 + r) * p.tr_ldx;
    RegVec<SD> rv;
#pragma unroll
    for (int q = 0; q < SD / 4; ++q) rv.q[q] = *reinterpret_cast<const f4 *>(src + 4 * q);
    const float v = dist_exact<SD, METRIC_L2>(rv, p.tr_cb + ((int64_t)mg * 256 + c) * SD);
    p.tr_ids[(int64_t)mg * p.tr_stride + row0 + r] = c;
    p.tr_dists[(int64_t)mg * p.tr_stride + row0 + r] = v;
  }
}

// ASSIGN = true: phases 1-3 only (rows -> sweep -> exact re-check), with the k-means bias and the problem's `active` flag: the E-step
// of the IVF training and every other f32 assign call of d <= 128 (launch_xform_assign).  One kernel instead of round 5's
// ma_top3_kernel + ma_finalize_kernel (the rows read twice, the candidates through HBM).
// MODE 2: phases 1-2 only, the sweep's epilogue keeps per-group keys instead of the row's four smallest (XfArgs::gkey).
template <int KS, int SD, int METRIC, typename TX, bool PROF = false, int MODE = 0>
__global__ __launch_bounds__(256, 2) void xf_kernel(XfArgs p) {
  constexpr bool ASSIGN = MODE == 1, GROUPS = MODE == 2;
  if constexpr (ASSIGN) { if (p.active && !p.active[0]) return; }
  long long pa[6] = {0, 0, 0, 0, 0, 0}, pprev = 0;      // PROF: s_memtime ticks per phase, summed over this workgroup's row tiles
  uint32_t pc_und = 0;
  if constexpr (PROF) pprev = clock64();
  auto mark = [&](int i) {
    if constexpr (PROF) { const long long t = clock64(); pa[i] += t - pprev; pprev = t; }
  };
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int D = KS * 16, M = D / SD;
  constexpr int XS = D + 4;          // f32 row stride of the x staging tile
  constexpr int CW = 2 * D + 16;     // bf16 elements per centroid in global memory: hi (D) | norm step (16) | lo (D)
  constexpr int CS = CW + 8;         // ... and per LDS row (16-byte skew: conflict-free ds_read_b128)
  constexpr bool XL = !std::is_same<TX, int8_t>::value;       // int8 rows are exact in bf16
  static_assert(SD == 4 || SD == 8 || SD == 16, "sub-dimension 4, 8 or 16");
  float *xs = reinterpret_cast<float *>(smem);                       // phase 1: [128][XS] f32
  uint16_t *cbuf = reinterpret_cast<uint16_t *>(smem);               // phase 2: [2][MA_CT][CS]; phases 4 / 5: XfLds
  uint32_t *q_cnt = XfLds<KS, SD>::q_cnt(smem);                      // undecided items of the workgroup, per sub-quantiser
  uint8_t *q_rows = XfLds<KS, SD>::q_rows(smem);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, g = lane >> 5;
  const int64_t row0 = (int64_t)blockIdx.x * MA_ROWS;

  // ---- 1. rows -> LDS -> registers --------------------------------------------------------------------------------------
  // Coalesced 16-byte loads, ALL of a thread's loads in flight before the first LDS store (one HBM round trip: the loop form of
  // round 5 -- load, LDS store, next load -- paid sixteen dependent ones, 110k of a wave's 335k cycles; lanes fetching their own
  // 16-byte pieces straight from HBM measured 54k: four times the line requests).  (Persistent workgroups that request the next
  // tile's rows before the PQ phase were tried -- gpurun r06i: the 64 staging registers spill, 1.03 -> 1.36 ms.)
  if constexpr (MODE == 0) { if (threadIdx.x < M) q_cnt[threadIdx.x] = 0u; }
  const int64_t row = row0 + wave * 32 + j;
  const bool valid = row < p.n;
  constexpr int NLD = MA_ROWS * (D / 4) / 256;        // 4-element pieces per thread (16 at D = 128)
  {
    f4 stage[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int idx = threadIdx.x + 256 * u, r = idx / (D / 4), c4 = idx - r * (D / 4);
      stage[u] = f4{0.0f, 0.0f, 0.0f, 0.0f};
      if (row0 + r < p.n) stage[u] = load4(static_cast<const TX *>(p.x) + (row0 + r) * p.ldx + 4 * c4);
    }
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int idx = threadIdx.x + 256 * u, r = idx / (D / 4), c4 = idx - r * (D / 4);
      *reinterpret_cast<f4 *>(&xs[r * XS + 4 * c4]) = stage[u];
    }
  }
  __syncthreads();
  float xf[KS][8];
  bf16x8 xh[KS], xl[KS];
  float xn2 = 0.0f;
  {
    const float *xr = xs + (wave * 32 + j) * XS + g * 8;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const f4 a = *reinterpret_cast<const f4 *>(xr + s * 16), b = *reinterpret_cast<const f4 *>(xr + s * 16 + 4);
      xf[s][0] = a.x; xf[s][1] = a.y; xf[s][2] = a.z; xf[s][3] = a.w; xf[s][4] = b.x; xf[s][5] = b.y; xf[s][6] = b.z; xf[s][7] = b.w;
      uint4 h4, l4;
      xf_split8<XL>(xf[s], h4, l4);
      xh[s] = __builtin_bit_cast(bf16x8, h4);
      if constexpr (XL) xl[s] = __builtin_bit_cast(bf16x8, l4);
#pragma unroll
      for (int e = 0; e < 8; ++e) xn2 = __builtin_fmaf(xf[s][e], xf[s][e], xn2);
    }
  }
  xn2 += __shfl_xor(xn2, 32, 64);          // (a non-finite element makes this inf / NaN: the row is then recomputed exactly, below)
  const float cmax2c = __uint_as_float(p.maxbits[0]);
  const float bmax = (ASSIGN && p.bias) ? __uint_as_float(p.maxbits[1]) : 0.0f;      // max |bias|: in E (the biased sum's own roundings) and in R (surrogates stay >= 0)
  // (dot: the reference's distance is the f32 value of 1 - x.c -- two centroids whose products differ by less than an ulp of 1 TIE there and the first
  // wins, whatever the surrogate says about x.c itself; 2^-22 of absolute margin sends such rows to the exact re-check.  Rows and centroids of
  // norm^2 below 2^-10 only: tests/fuzz_dot_flow.py case 25 -- 9 of 26,482 f16 rows of magnitude 1e-3 went to the other of two tied lists)
  const float E2 = 2.0f * 0.0001220703125f * (xn2 + cmax2c + bmax) + (METRIC == METRIC_DOT ? 4.7683716e-7f : 0.0f);   // 2E, E = 2^-13 (|x|^2 + max|c|^2 + max|bias|)
  // the row's offset R (one bf16, rounded up): L2  s' = R + |c|^2 - 2 x.c = |x - c|^2 + (R - |x|^2) >= 0;  dot  s' = R - x.c >= 0
  bf16x8 bx = {0, 0, 0, 0, 0, 0, 0, 0};
  if (g == 0) {
    const float R = (METRIC == METRIC_DOT ? 0.5f * (xn2 + cmax2c) + E2 : xn2 + E2) + bmax;
    bx[0] = (short)0x3F80; bx[1] = (short)0x3F80; bx[2] = (short)0x3F80; bx[3] = (short)xf_bf16_up(R);
  }
  __syncthreads();   // xs is dead: the same LDS now holds centroid tiles
  mark(0);

  // ---- 2. coarse sweep -------------------------------------------------------------------------------------------------
  const int ntiles = (p.k + MA_CT - 1) / MA_CT;
  constexpr int NCHK = MA_CT * CW / 8;        // 16-byte chunks per tile (contiguous in global memory)
  constexpr int CH = (NCHK + 255) / 256;      // per thread (9 at D = 128)
  uint4 pf[CH];
  auto tile_fetch = [&](int t) {
    const uint4 *src = reinterpret_cast<const uint4 *>(p.cpl) + (int64_t)t * NCHK;
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int ch = threadIdx.x + 256 * u;
      pf[u] = make_uint4(0, 0, 0, 0);
      if (ch < NCHK) pf[u] = src[ch];
    }
  };
  auto tile_store = [&](int buf) {
    uint16_t *dst = cbuf + (size_t)buf * MA_CT * CS;
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int ch = threadIdx.x + 256 * u;
      const int cr = ch / (CW / 8), cc = ch - cr * (CW / 8);
      if (ch < NCHK) *reinterpret_cast<uint4 *>(dst + cr * CS + cc * 8) = pf[u];
    }
  };
  Top4 tp{INFINITY, INFINITY, INFINITY, INFINITY, LANCE_HIP_NONE, LANCE_HIP_NONE, LANCE_HIP_NONE};
  int t_first = 0, t_end = ntiles;
  if constexpr (GROUPS) { t_first = (int)blockIdx.y * p.tiles_per_block; t_end = min(ntiles, t_first + p.tiles_per_block); }
  tile_fetch(t_first);
  tile_store(t_first & 1);
  __syncthreads();
  for (int t = t_first; t < t_end; ++t) {
    const int buf = t & 1;
    if (t + 1 < t_end) tile_fetch(t + 1);
    const uint16_t *tl = cbuf + (size_t)buf * MA_CT * CS;
    const uint16_t *r0 = tl + j * CS + g * 8, *r1 = tl + (32 + j) * CS + g * 8;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // the norm / offset step first: its C operand is the inline constant 0
    f32x16 acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(r0 + D), bx, zero, 0, 0, 0);
    f32x16 acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(r1 + D), bx, zero, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const bf16x8 ah0 = *reinterpret_cast<const bf16x8 *>(r0 + s * 16);
      const bf16x8 al0 = *reinterpret_cast<const bf16x8 *>(r0 + D + 16 + s * 16);
      const bf16x8 ah1 = *reinterpret_cast<const bf16x8 *>(r1 + s * 16);
      const bf16x8 al1 = *reinterpret_cast<const bf16x8 *>(r1 + D + 16 + s * 16);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, xh[s], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, xh[s], acc1, 0, 0, 0);
      if constexpr (XL) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, xl[s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, xl[s], acc1, 0, 0, 0);
      }
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al0, xh[s], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al1, xh[s], acc1, 0, 0, 0);
    }
    // D[centroid i][row j]: lane (j, g) holds centroids i = (v & 3) + 8 (v >> 2) + 4 g of each 32-block.  Surrogates are >= 0 (or
    // NaN), so their bit patterns order like the values: the tile's minimum by v_min3_u32, and only when it undercuts the row's
    // running fourth-smallest (with k = 4096 about one tile in twenty) the tile's own four smallest -- value bits with the
    // register number in the five lowest mantissa bits (2^-18 relative, inside E), one v_and_or_b32 + three v_med3_u32 + one
    // v_min_u32 per centroid -- are folded into the running four.
    if constexpr (GROUPS) {
      // surrogates are >= 0: their bit patterns order like the values.  A padding centroid's +inf (and any NaN) is clamped to a huge finite
      // key: the select kernel compares the records as floats
      uint32_t k1[2], k2[2];
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        uint32_t a1 = 0xFFFFFFFFu, a2 = 0xFFFFFFFFu;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const uint32_t key = (__float_as_uint(blk ? acc1[v] : acc0[v]) & 0xFFFFFFF0u) | (uint32_t)v;
          a2 = xf_med3(a1, a2, key);
          a1 = min(a1, key);
        }
        k1[blk] = min(a1, 0x7F000000u); k2[blk] = min(a2, 0x7F000000u);
      }
      if (valid)
        *reinterpret_cast<f4 *>(p.gkey + row * 2 * p.ng + (int64_t)(t * 2 + g) * 4) =
            f4{__uint_as_float(k1[0]), __uint_as_float(k1[1]), __uint_as_float(k2[0]), __uint_as_float(k2[1])};
    }
    uint32_t bm = 0xFFFFFFFFu;
    if constexpr (!GROUPS) {
      bm = min(__float_as_uint(acc0[0]), __float_as_uint(acc1[0]));
#pragma unroll
      for (int v = 1; v < 16; ++v) bm = min(bm, min(__float_as_uint(acc0[v]), __float_as_uint(acc1[v])));
    }
    if (!GROUPS && bm < __float_as_uint(tp.m4)) {
      uint32_t t1 = 0xFFFFFFFFu, t2 = 0xFFFFFFFFu, t3 = 0xFFFFFFFFu, t4 = 0xFFFFFFFFu;
#pragma unroll
      for (int li = 0; li < 32; ++li) {
        const uint32_t key = (__float_as_uint(li < 16 ? acc0[li] : acc1[li - 16]) & 0xFFFFFFE0u) | (uint32_t)li;
        t4 = xf_med3(t3, t4, key);
        t3 = xf_med3(t2, t3, key);
        t2 = xf_med3(t1, t2, key);
        t1 = min(t1, key);
      }
      const uint32_t c0 = (uint32_t)(t * MA_CT + 4 * g);
      auto fold = [&](uint32_t key) {
        const uint32_t li = key & 31u, v = li & 15u;
        top4_insert(tp, __uint_as_float(key & 0xFFFFFFE0u), c0 + (li >> 4) * 32u + 8u * (v >> 2) + (v & 3u));
      };
      fold(t1); fold(t2); fold(t3); fold(t4);
    }
    if (t + 1 < t_end) tile_store(buf ^ 1);
    __syncthreads();
  }
  mark(1);
  if constexpr (GROUPS) {
    if (g == 0 && valid && blockIdx.y == 0) p.e2[row] = E2;
    return;
  }
  // the two lanes of a row hold disjoint centroid subsets: merge the partner's four
  {
    const float pm1 = __shfl_xor(tp.m1, 32, 64), pm2 = __shfl_xor(tp.m2, 32, 64), pm3 = __shfl_xor(tp.m3, 32, 64), pm4 = __shfl_xor(tp.m4, 32, 64);
    const uint32_t pi1 = __shfl_xor(tp.i1, 32, 64), pi2 = __shfl_xor(tp.i2, 32, 64), pi3 = __shfl_xor(tp.i3, 32, 64);
    top4_insert(tp, pm1, pi1);
    top4_insert(tp, pm2, pi2);
    top4_insert(tp, pm3, pi3);
    top4_insert(tp, pm4, LANCE_HIP_NONE);   // >= the three just inserted: can only land in the fourth (index-free) slot
  }
  int cl = 3;                             // number of exact candidates - 1; 3 = recompute against every centroid
  if (tp.m2 - tp.m1 > E2) cl = 0;
  else if (tp.m3 - tp.m1 > E2) cl = 1;
  else if (tp.m4 - tp.m1 > E2) cl = 2;
  // NaN / overflow anywhere, a padding centroid among the candidates (fewer than four real ones in reach: never with k >= 32 finite
  // centroids, but +inf surrogates must not reach the exact step), products in the denormal range (E2 <= 2^-100): recompute exactly
  if (tp.i1 == LANCE_HIP_NONE || !(E2 < INFINITY) || !(E2 > 7.888609052210118e-31f) || !(tp.m1 < INFINITY)) cl = 3;
  // lane g = 0 decides for the pair (equal surrogates may sit in a different order in the partner's registers)
  cl = __shfl(cl, j, 64);
  const uint32_t cand[3] = {(uint32_t)__shfl((int)tp.i1, j, 64), (uint32_t)__shfl((int)tp.i2, j, 64), (uint32_t)__shfl((int)tp.i3, j, 64)};

  // ---- 3. exact re-check (argmin_value_float over the candidates) ------------------------------------------------------
  uint32_t best = LANCE_HIP_NONE;
  float bestv = INFINITY;
  {
    float bestb = INFINITY;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const bool need = valid && cl < 3 && t <= cl && cand[t] != LANCE_HIP_NONE && cand[t] < (uint32_t)p.k;
      if (!__any(need)) continue;                                   // wave-uniform: rounds 2 and 3 are rare
      const uint32_t c = need ? cand[t] : 0u;
      const float v = xf_exact<KS, METRIC>(xf, p.cent + (int64_t)c * D, j, g);
      float vb = v;
      if constexpr (ASSIGN) { if (p.bias) vb = v + p.bias[c]; }      // argmin_value_float over the biased values, the plain distance is what is stored
      if (need && (vb < bestb || (vb == bestb && best != LANCE_HIP_NONE && c < best))) { bestb = vb; bestv = v; best = c; }
    }
  }
  const bool queued = valid && cl == 3;        // ma_recompute_kernel answers the row; its PQ items go to the fix lists
  if (best == LANCE_HIP_NONE) bestv = INFINITY;
  if (g == 0 && valid) {
    if (queued) {
      const uint32_t slot = atomicAdd(p.afb_cnt, 1u);
      p.afb_rows[slot] = (uint32_t)row;
      if constexpr (!ASSIGN) for (int m = 0; m < M; ++m) xf_queue_item<XfLds<KS, SD>::QC>(p, q_cnt, q_rows, m, m, (uint32_t)(wave * 32 + j), row0, true);      // (rare: straight to the global lists)
    } else {
      p.part_ids[row] = best;
      p.dists[row] = bestv;
    }
  }

  mark(2);
  if constexpr (ASSIGN) return;
  // ---- 4. residual -> bf16 planes in LDS (the centroid tiles are dead: the sweep's last barrier is behind every wave) ----
  {
    const bool sub = p.residual && best != LANCE_HIP_NONE;
    const bool zero = !valid || (p.residual && best == LANCE_HIP_NONE);     // rows without a partition encode the zero vector
    xf_residual_to_lds<KS, SD>(xf, p.cent + (int64_t)(sub ? best : 0u) * D, sub, zero, p.round_f16, smem, wave * 32 + j, g, !valid || queued);
  }
  __syncthreads();
  mark(3);
  // ---- 5. PQ encode: this wave's sub-quantisers against all 128 rows; undecided items; codes out ----
  xf_pq_phase<KS, SD, PROF>(p, smem, 0, row0, pc_und, pa[4], pprev);
  mark(5);
  if constexpr (PROF) {
    uint32_t und = pc_und;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) und += __shfl_xor(und, o, 64);
    if (lane == 0 && p.prof) {
      for (int i = 0; i < 6; ++i) atomicAdd(&p.prof[i], (unsigned long long)pa[i]);
      atomicAdd(&p.prof[6], 1ull);
      atomicAdd(&p.prof[7], (unsigned long long)und);
    }
  }
}

bool xform_fused_supported(int dtype, int metric, int d, int m, int nbits, int64_t n, int nlist, const void *x, const float *cent,
                           const float *codebook, uint8_t *codes, bool lanes32) {
  static const bool off = getenv("LANCE_HIP_NO_MFMA") != nullptr || getenv("LANCE_HIP_NO_XFORM_FUSED") != nullptr ||
                          getenv("LANCE_HIP_NO_MFMA_PQ") != nullptr || getenv("LANCE_HIP_NO_MFMA_ENCODE") != nullptr;
  if (off || lanes32 || nbits != 8 || m <= 0 || d % m != 0) return false;
  if (metric != METRIC_L2 && metric != METRIC_DOT) return false;
  if (d % 16 != 0 || d < 16 || d > 128) return false;
  const int sd = d / m;
  if (sd != 4 && sd != 8 && sd != 16) return false;
  if (n < 2048 || nlist < 32) return false;
  const size_t es = dtype == LANCE_HIP_F16 ? 2 : (dtype == LANCE_HIP_I8 ? 1 : 4);
  if (reinterpret_cast<uintptr_t>(x) % (4 * es)) return false;
  if ((reinterpret_cast<uintptr_t>(cent) & 15) || (reinterpret_cast<uintptr_t>(codebook) & 15) || (reinterpret_cast<uintptr_t>(codes) & 3)) return false;
  return true;
}

static unsigned xf_grid(lance_hip_ctx *, int64_t n) { return (unsigned)cdiv((uint64_t)n, MA_ROWS); }      // one workgroup per 128 rows

template <int KS, int SD, int METRIC, typename TX>
static void xf_launch_one(lance_hip_ctx *ctx, const XfArgs &a) {
  constexpr size_t lds = xf_lds_bytes<KS, SD>();
  hipLaunchKernelGGL((xf_kernel<KS, SD, METRIC, TX>), dim3(xf_grid(ctx, a.n)), dim3(256), lds, ctx->stream, a);
}
template <int KS, int SD>
static void xf_launch_ks(lance_hip_ctx *ctx, const XfArgs &a, int metric, int dtype) {
  if (metric == METRIC_DOT) {
    if (dtype == LANCE_HIP_F16) xf_launch_one<KS, SD, METRIC_DOT, __half>(ctx, a);
    else if (dtype == LANCE_HIP_I8) xf_launch_one<KS, SD, METRIC_DOT, int8_t>(ctx, a);
    else xf_launch_one<KS, SD, METRIC_DOT, float>(ctx, a);
  } else {
    if (dtype == LANCE_HIP_F16) xf_launch_one<KS, SD, METRIC_L2, __half>(ctx, a);
    else if (dtype == LANCE_HIP_I8) xf_launch_one<KS, SD, METRIC_L2, int8_t>(ctx, a);
    else xf_launch_one<KS, SD, METRIC_L2, float>(ctx, a);
  }
}
template <int SD>
static int xf_launch_sd(lance_hip_ctx *ctx, const XfArgs &a, int d, int metric, int dtype) {
  switch (d / 16) {
    case 1: xf_launch_ks<1, SD>(ctx, a, metric, dtype); break;
    case 2: xf_launch_ks<2, SD>(ctx, a, metric, dtype); break;
    case 3: xf_launch_ks<3, SD>(ctx, a, metric, dtype); break;
    case 4: xf_launch_ks<4, SD>(ctx, a, metric, dtype); break;
    case 5: xf_launch_ks<5, SD>(ctx, a, metric, dtype); break;
    case 6: xf_launch_ks<6, SD>(ctx, a, metric, dtype); break;
    case 7: xf_launch_ks<7, SD>(ctx, a, metric, dtype); break;
    case 8: xf_launch_ks<8, SD>(ctx, a, metric, dtype); break;
    default: return LANCE_HIP_EINVAL;
  }
  return LANCE_HIP_OK;
}

static void xf_pq_prep_launch(lance_hip_ctx *ctx, int sd, int m, const float *codebook, uint4 *pqa, float *cmax2, uint32_t *zero_cnt = nullptr) {
  if (sd == 16) hipLaunchKernelGGL(xf_pq_prep_kernel<16>, dim3((unsigned)m), dim3(256), 0, ctx->stream, codebook, pqa, cmax2, zero_cnt);
  else if (sd == 8) hipLaunchKernelGGL(xf_pq_prep_kernel<8>, dim3((unsigned)m), dim3(256), 0, ctx->stream, codebook, pqa, cmax2, zero_cnt);
  else hipLaunchKernelGGL(xf_pq_prep_kernel<4>, dim3((unsigned)m), dim3(256), 0, ctx->stream, codebook, pqa, cmax2, zero_cnt);
}

template <int SD>
static void xf_fix_launch_sd(lance_hip_ctx *ctx, int dtype, dim3 fgrid, const XfFixArgs &fa) {
  if (dtype == LANCE_HIP_F16) hipLaunchKernelGGL((xf_fix_kernel<SD, __half>), fgrid, dim3(256), 0, ctx->stream, fa);
  else if (dtype == LANCE_HIP_I8) hipLaunchKernelGGL((xf_fix_kernel<SD, int8_t>), fgrid, dim3(256), 0, ctx->stream, fa);
  else hipLaunchKernelGGL((xf_fix_kernel<SD, float>), fgrid, dim3(256), 0, ctx->stream, fa);
}
static void xf_fix_launch(lance_hip_ctx *ctx, int sd, int dtype, int64_t rows, const XfFixArgs &fa) {
  // grid (blocks, sub-quantisers); a workgroup whose first wave has no item returns before staging the codebook
  const dim3 fgrid((unsigned)std::min<uint64_t>(std::max<uint64_t>(1, cdiv((uint64_t)rows, 256)), 64), (unsigned)fa.m);
  if (sd == 16) xf_fix_launch_sd<16>(ctx, dtype, fgrid, fa);
  else if (sd == 8) xf_fix_launch_sd<8>(ctx, dtype, fgrid, fa);
  else xf_fix_launch_sd<4>(ctx, dtype, fgrid, fa);
}

// x: [n][d] rows in the column's element type; metric: METRIC_L2 (residual encoded) or METRIC_DOT (the row itself encoded);
// round_f16: the residual is rounded to binary16 (f16 columns, also when they arrive as a normalised f32 copy: cosine);
// part_ids / dists [n] and codes [n][m] are filled exactly as launch_assign + the fused encode fill them.
int launch_xform_fused(lance_hip_ctx *ctx, int dtype, int metric, const void *x, int64_t n, int d, const float *cent, int nlist,
                       const float *codebook, int m, uint32_t *part_ids, float *dists, uint8_t *codes, bool round_f16) {
  if (n == 0) return LANCE_HIP_OK;
  const int sd = d / m, nmf = sd == 16 ? 3 : (sd == 8 ? 2 : 1);
  const int kpad = (nlist + MA_CT - 1) / MA_CT * MA_CT;
  uint16_t *cpl = ctx->scratch_t<uint16_t>("xf.cpl", (size_t)kpad * (2 * d + 16));
  uint32_t *maxbits = ctx->scratch_t<uint32_t>("ma.maxbits", 4);      // [0] max |c|^2, [2] rows left to the recompute kernel
  if (!cpl || !maxbits) return LANCE_HIP_ENOMEM;
  LH_CHECK_HIP(lh::memset_async(maxbits, 0, 16, ctx->stream));
  if (metric == METRIC_DOT) hipLaunchKernelGGL(xf_cent_prep_kernel<METRIC_DOT>, dim3((unsigned)kpad), dim3(64), 0, ctx->stream, cent, nlist, d, cpl, maxbits);
  else hipLaunchKernelGGL(xf_cent_prep_kernel<METRIC_L2>, dim3((unsigned)kpad), dim3(64), 0, ctx->stream, cent, nlist, d, cpl, maxbits);
  uint4 *pqa = ctx->scratch_t<uint4>("xf.pqa", (size_t)m * 8 * nmf * 64);
  float *cmax2 = ctx->scratch_t<float>("xf.cmax2", (size_t)m);
  // rows go through in chunks: the undecided-item list is [chunk * m] words of scratch (256 MB at most), whatever n is; row numbers take 25 bits
  const int64_t chunk = std::max<int64_t>(MA_ROWS, std::min<int64_t>(std::min<int64_t>(n, (int64_t)1 << 24), ((int64_t)64 << 20) / m / MA_ROWS * MA_ROWS));
  uint32_t *fb_cnt = ctx->scratch_t<uint32_t>("xf.fb_cnt", (size_t)m);
  uint32_t *fb_items = ctx->scratch_t<uint32_t>("xf.fb_items", (size_t)chunk * m);
  uint32_t *afb_rows = ctx->scratch_t<uint32_t>("ma.fb_rows", (size_t)chunk);
  if (!pqa || !cmax2 || !fb_cnt || !fb_items || !afb_rows) return LANCE_HIP_ENOMEM;
  ScopedTimer t(ctx, "xform_fused");
  xf_pq_prep_launch(ctx, sd, m, codebook, pqa, cmax2);
  const size_t es = dtype == LANCE_HIP_F16 ? 2 : (dtype == LANCE_HIP_I8 ? 1 : 4);
  for (int64_t r0 = 0; r0 < n; r0 += chunk) {
    const int64_t rows = std::min<int64_t>(chunk, n - r0);
    XfArgs a;
    a.x = static_cast<const char *>(x) + (size_t)r0 * d * es; a.n = rows; a.ldx = d; a.k = nlist;
    a.cpl = cpl; a.maxbits = maxbits; a.cent = cent;
    a.residual = metric == METRIC_L2 ? 1 : 0; a.round_f16 = round_f16 ? 1 : 0; a.check_finite = 1;
    a.pqa = pqa; a.pq_cmax2 = cmax2;
    a.part_ids = part_ids + r0; a.dists = dists + r0; a.codes = codes + r0 * m; a.m_total = m; a.d_total = d;
    a.afb_cnt = maxbits + 2; a.afb_rows = afb_rows; a.fb_cnt = fb_cnt; a.fb_items = fb_items;
    LH_CHECK_HIP(lh::memset_multi(ctx->stream, {{fb_cnt, 0, (size_t)m * 4}, {maxbits + 2, 0, 4}}));
    {
      ScopedTimer t1(ctx, "xf_main");
      static const bool prof = getenv("LANCE_HIP_XF_PROF") != nullptr;      // s_memtime phase stamps (d = 128, sub-dimension 8, f32, L2), printed per launch
      if (prof && sd == 8 && d == 128 && metric == METRIC_L2 && dtype == LANCE_HIP_F32) {
        a.prof = ctx->scratch_t<unsigned long long>("xf.prof", 8);
        if (!a.prof) return LANCE_HIP_ENOMEM;
        LH_CHECK_HIP(lh::memset_async(a.prof, 0, 64, ctx->stream));
        constexpr size_t lds = xf_lds_bytes<8, 8>();
        hipLaunchKernelGGL((xf_kernel<8, 8, METRIC_L2, float, true>), dim3(xf_grid(ctx, a.n)), dim3(256), lds, ctx->stream, a);
        unsigned long long h[8];
        LH_CHECK_HIP(hipMemcpyAsync(h, a.prof, 64, hipMemcpyDeviceToHost, ctx->stream));
        LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        if (h[6])
          fprintf(stderr, "[xf prof] waves=%llu | s_memtime ticks per wave: rows->regs %.0f | sweep %.0f | merge+exact %.0f | residual+barrier %.0f | pq %.0f | codes out %.0f | "
                  "undecided pq items %llu\n", h[6], (double)h[0] / h[6], (double)h[1] / h[6], (double)h[2] / h[6], (double)h[3] / h[6], (double)h[4] / h[6],
                  (double)h[5] / h[6], h[7]);
      } else if (sd == 16) LH_TRY(xf_launch_sd<16>(ctx, a, d, metric, dtype));
      else if (sd == 8) LH_TRY(xf_launch_sd<8>(ctx, a, d, metric, dtype));
      else LH_TRY(xf_launch_sd<4>(ctx, a, d, metric, dtype));
    }
    ScopedTimer t2(ctx, "xf_fix");          // the two exact clean-up kernels
    // rows the coarse surrogate left undecided: exact distances to every centroid (writes part_ids / dists) ...
    MaArgs ma{};
    ma.x = a.x; ma.n = rows; ma.ldx = d; ma.d = d; ma.k = nlist; ma.cent = cent; ma.bias = nullptr;
    ma.ids = a.part_ids; ma.dists = a.dists; ma.check_finite = 1; ma.fb_cnt = a.afb_cnt; ma.fb_rows = afb_rows; ma.active = nullptr;
    LH_TRY(ma_recompute_launch(ctx, ma, metric, dtype));
    // ... then the undecided (row, sub-quantiser) items, which read the partition ids
    XfFixArgs fa;
    fa.x = a.x; fa.ldx = d; fa.n = rows; fa.cent = cent; fa.part_ids = a.part_ids; fa.residual = a.residual; fa.round_f16 = a.round_f16;
    fa.codebook = codebook; fa.m = m; fa.cnt = fb_cnt; fa.items = fb_items; fa.codes = a.codes;
    xf_fix_launch(ctx, sd, dtype, rows, fa);
  }
  LH_CHECK_HIP(hipGetLastError());
  return LANCE_HIP_OK;
}

// ---- long rows: the second half of the transform (residual + PQ encode) after the K-tiled coarse quantiser -------------------------------
bool xform_tail_supported(int d, int m, int nbits, int64_t n, const float *x, const float *cent, const float *codebook) {
  static const bool off = getenv("LANCE_HIP_NO_MFMA") != nullptr || getenv("LANCE_HIP_NO_XFORM_FUSED") != nullptr ||
                          getenv("LANCE_HIP_NO_MFMA_PQ") != nullptr || getenv("LANCE_HIP_NO_MFMA_ENCODE") != nullptr;
  if (off || nbits != 8 || m <= 0 || m > 127 || d % m != 0 || d <= 128 || n < 2048) return false;
  const int sd = d / m;
  if (sd != 4 && sd != 8 && sd != 16) return false;
  if (d % 16 != 0 || (d % 128 != 0 && sd != 16)) return false;      // a last block of fewer than 128 columns is instantiated for sub-dimension 16 only
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(cent) | reinterpret_cast<uintptr_t>(codebook)) & 15) return false;
  return true;
}

template <int KS, int SD>
static void xf_tail_launch_one(lance_hip_ctx *ctx, const XfArgs &a, unsigned blocks_y) {
  constexpr size_t lds = xf_lds_bytes<KS, SD>();
  hipLaunchKernelGGL((xf_tail_kernel<KS, SD>), dim3((unsigned)cdiv((uint64_t)a.n, MA_ROWS), blocks_y), dim3(256), lds, ctx->stream, a);
}

// x: [n][d] f32 rows (already normalised for cosine); part_ids: their partitions (LANCE_HIP_NONE: the zero vector is encoded when `residual`)
int launch_xform_tail(lance_hip_ctx *ctx, const float *x, int64_t n, int d, const float *cent, const uint32_t *part_ids, int residual, bool round_f16,
                      const float *codebook, int m, uint8_t *codes) {
  if (n == 0) return LANCE_HIP_OK;
  const int sd = d / m, na = sd == 16 ? 3 : (sd == 8 ? 2 : 1);
  uint4 *pqa = ctx->scratch_t<uint4>("xf.pqa", (size_t)m * 8 * na * 64);
  float *cmax2 = ctx->scratch_t<float>("xf.cmax2", (size_t)m);
  const int64_t chunk = std::max<int64_t>(MA_ROWS, std::min<int64_t>(std::min<int64_t>(n, (int64_t)1 << 24), ((int64_t)64 << 20) / m / MA_ROWS * MA_ROWS));
  uint32_t *fb_cnt = ctx->scratch_t<uint32_t>("xf.fb_cnt", (size_t)m);
  uint32_t *fb_items = ctx->scratch_t<uint32_t>("xf.fb_items", (size_t)chunk * m);
  if (!pqa || !cmax2 || !fb_cnt || !fb_items) return LANCE_HIP_ENOMEM;
  ScopedTimer t(ctx, "xform_tail");
  xf_pq_prep_launch(ctx, sd, m, codebook, pqa, cmax2);
  for (int64_t r0 = 0; r0 < n; r0 += chunk) {
    const int64_t rows = std::min<int64_t>(chunk, n - r0);
    XfArgs a{};
    a.x = x + (size_t)r0 * d; a.n = rows; a.ldx = d; a.cent = cent; a.d_total = d;
    a.residual = residual; a.round_f16 = round_f16 ? 1 : 0;
    a.pqa = pqa; a.pq_cmax2 = cmax2; a.part_ids = const_cast<uint32_t *>(part_ids) + r0; a.codes = codes + r0 * m; a.m_total = m;
    a.fb_cnt = fb_cnt; a.fb_items = fb_items;
    LH_CHECK_HIP(lh::memset_async(fb_cnt, 0, (size_t)m * 4, ctx->stream));
    const int full = d / 128, rem = d % 128;
    if (full) {
      a.col_base = 0;
      if (sd == 16) xf_tail_launch_one<8, 16>(ctx, a, (unsigned)full);
      else if (sd == 8) xf_tail_launch_one<8, 8>(ctx, a, (unsigned)full);
      else xf_tail_launch_one<8, 4>(ctx, a, (unsigned)full);
    }
    if (rem) {      // (sub-dimension 16 only: xform_tail_supported)
      a.col_base = full * 128;
      switch (rem / 16) {
        case 1: xf_tail_launch_one<1, 16>(ctx, a, 1); break;
        case 2: xf_tail_launch_one<2, 16>(ctx, a, 1); break;
        case 3: xf_tail_launch_one<3, 16>(ctx, a, 1); break;
        case 4: xf_tail_launch_one<4, 16>(ctx, a, 1); break;
        case 5: xf_tail_launch_one<5, 16>(ctx, a, 1); break;
        case 6: xf_tail_launch_one<6, 16>(ctx, a, 1); break;
        default: xf_tail_launch_one<7, 16>(ctx, a, 1); break;
      }
    }
    XfFixArgs fa;
    fa.x = a.x; fa.ldx = d; fa.n = rows; fa.cent = cent; fa.part_ids = a.part_ids; fa.residual = residual; fa.round_f16 = a.round_f16;
    fa.codebook = codebook; fa.m = m; fa.cnt = fb_cnt; fa.items = fb_items; fa.codes = a.codes;
    xf_fix_launch(ctx, sd, LANCE_HIP_F32, rows, fa);
    static const bool prof = getenv("LANCE_HIP_XF_PROF") != nullptr;      // how many (row, sub-quantiser) items the surrogate left undecided
    if (prof) {
      std::vector<uint32_t> h((size_t)m);
      LH_CHECK_HIP(hipMemcpyAsync(h.data(), fb_cnt, (size_t)m * 4, hipMemcpyDeviceToHost, ctx->stream));
      LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
      uint64_t tot = 0, mx = 0;
      for (uint32_t v : h) { tot += v; mx = std::max<uint64_t>(mx, v); }
      fprintf(stderr, "[xf tail prof] rows %lld x %d sub-quantisers: %llu undecided items (%.2f %%), largest list %llu\n", (long long)rows, m,
              (unsigned long long)tot, 100.0 * (double)tot / ((double)rows * m), (unsigned long long)mx);
    }
  }
  LH_CHECK_HIP(hipGetLastError());
  return LANCE_HIP_OK;
}

// ---- PQ codebook training: the E-step of all sub-quantisers in one launch (called by launch_pq_mfma) ------------------------------------
bool xform_pqtrain_supported(const PairwiseArgs &p, int sd, int batches) {
  static const bool off = getenv("LANCE_HIP_NO_XFORM_FUSED") != nullptr || getenv("LANCE_HIP_NO_XF_TRAIN") != nullptr;
  if (off || (sd != 4 && sd != 8 && sd != 16) || batches <= 0 || (batches * sd) % 128 != 0) return false;
  if (p.k != 256 || !p.ids || !p.dists || p.codes) return false;
  if (p.cent_batch_stride != (int64_t)256 * sd) return false;                 // codebook [b][256][sd], contiguous
  if ((p.ldx & 3) || (p.x_batch_off & 3) || !p.x_aligned || !p.cent_aligned) return false;
  return true;
}

// fb_cnt / fb_items: the undecided-item lists [batches], [batches][n] the caller hands to pq_mfma_fix_kernel afterwards (zeroed here)
int launch_xform_pqtrain(lance_hip_ctx *ctx, const PairwiseArgs &p, int sd, int batches, uint32_t *fb_cnt, uint32_t *fb_items) {
  const int na = sd == 16 ? 3 : (sd == 8 ? 2 : 1);
  uint4 *pqa = ctx->scratch_t<uint4>("xf.pqa", (size_t)batches * 8 * na * 64);
  float *cmax2 = ctx->scratch_t<float>("xf.cmax2", (size_t)batches);
  if (!pqa || !cmax2) return LANCE_HIP_ENOMEM;
  ScopedTimer t(ctx, "xf_pqtrain");
  xf_pq_prep_launch(ctx, sd, batches, p.cent, pqa, cmax2, fb_cnt);
  XfArgs a{};
  a.n = p.n; a.pqa = pqa; a.pq_cmax2 = cmax2; a.m_total = batches; a.fb_cnt = fb_cnt; a.fb_items = fb_items; a.codes = nullptr;
  a.active = p.active; a.tr_x = p.x; a.tr_cb = p.cent; a.tr_ldx = p.ldx; a.tr_boff = p.x_batch_off; a.tr_stride = p.out_batch_stride;
  a.tr_ids = p.ids; a.tr_dists = p.dists;
  const dim3 grid((unsigned)cdiv((uint64_t)p.n, MA_ROWS), (unsigned)(batches * sd / 128));
  if (sd == 16) hipLaunchKernelGGL(xf_pqtrain_kernel<16>, grid, dim3(256), (xf_lds_bytes<8, 16>()), ctx->stream, a);
  else if (sd == 8) hipLaunchKernelGGL(xf_pqtrain_kernel<8>, grid, dim3(256), (xf_lds_bytes<8, 8>()), ctx->stream, a);
  else hipLaunchKernelGGL(xf_pqtrain_kernel<4>, grid, dim3(256), (xf_lds_bytes<8, 4>()), ctx->stream, a);
  LH_CHECK_HIP(hipGetLastError());
  return LANCE_HIP_OK;
}

// ---- assign (k-means E-step, lance_hip_assign, hierarchical splits) of f32 rows, d <= 128: phases 1-3 of the transform kernel ----------
bool xform_assign_supported(const PairwiseArgs &p, int d, int metric, int batches) {
  static const bool off = getenv("LANCE_HIP_NO_MFMA") != nullptr || getenv("LANCE_HIP_NO_XFORM_FUSED") != nullptr || getenv("LANCE_HIP_NO_XF_TRAIN") != nullptr;
  if (off || batches != 1 || p.codes || p.matrix || p.lanes32 || !p.ids || !p.dists) return false;
  if (metric != METRIC_L2 && metric != METRIC_DOT) return false;
  if (d % 16 != 0 || d < 16 || d > 128 || p.k < 32 || p.n < 2048 || p.n >= (1ll << 32)) return false;
  if (!p.x || (p.x_native && p.x_dtype != LANCE_HIP_F32) || !p.x_aligned || !p.cent_aligned) return false;
  return true;
}

template <int KS>
static void xf_assign_launch_ks(lance_hip_ctx *ctx, const XfArgs &a, int metric) {
  constexpr size_t lds = xf_lds_bytes<KS, 8>();
  const dim3 grid(xf_grid(ctx, a.n));
  if (metric == METRIC_DOT) hipLaunchKernelGGL((xf_kernel<KS, 8, METRIC_DOT, float, false, 1>), grid, dim3(256), lds, ctx->stream, a);
  else hipLaunchKernelGGL((xf_kernel<KS, 8, METRIC_L2, float, false, 1>), grid, dim3(256), lds, ctx->stream, a);
}

int launch_xform_assign(lance_hip_ctx *ctx, const PairwiseArgs &p, int d, int metric) {
  const int kpad = (p.k + MA_CT - 1) / MA_CT * MA_CT;
  uint16_t *cpl = ctx->scratch_t<uint16_t>("xf.cpl", (size_t)kpad * (2 * d + 16));
  uint32_t *maxbits = ctx->scratch_t<uint32_t>("ma.maxbits", 4);      // [0] max |c|^2, [1] max |bias|, [2] rows left to the recompute kernel
  uint32_t *afb_rows = ctx->scratch_t<uint32_t>("ma.fb_rows", (size_t)p.n);
  if (!cpl || !maxbits || !afb_rows) return LANCE_HIP_ENOMEM;
  LH_CHECK_HIP(lh::memset_async(maxbits, 0, 16, ctx->stream));
  if (metric == METRIC_DOT) hipLaunchKernelGGL(xf_cent_prep_kernel<METRIC_DOT>, dim3((unsigned)kpad), dim3(64), 0, ctx->stream, p.cent, p.k, d, cpl, maxbits, p.bias, p.active);
  else hipLaunchKernelGGL(xf_cent_prep_kernel<METRIC_L2>, dim3((unsigned)kpad), dim3(64), 0, ctx->stream, p.cent, p.k, d, cpl, maxbits, p.bias, p.active);
  XfArgs a{};
  a.x = p.x; a.n = p.n; a.ldx = p.ldx; a.k = p.k; a.cpl = cpl; a.maxbits = maxbits; a.cent = p.cent; a.check_finite = p.check_finite ? 1 : 0;
  a.part_ids = p.ids; a.dists = p.dists; a.afb_cnt = maxbits + 2; a.afb_rows = afb_rows; a.bias = p.bias; a.active = p.active; a.d_total = d;
  {
    ScopedTimer t(ctx, "ma_sweep");      // (the stage names of the two-kernel route it replaces: bench.py reads them)
    switch (d / 16) {
      case 1: xf_assign_launch_ks<1>(ctx, a, metric); break;
      case 2: xf_assign_launch_ks<2>(ctx, a, metric); break;
      case 3: xf_assign_launch_ks<3>(ctx, a, metric); break;
      case 4: xf_assign_launch_ks<4>(ctx, a, metric); break;
      case 5: xf_assign_launch_ks<5>(ctx, a, metric); break;
      case 6: xf_assign_launch_ks<6>(ctx, a, metric); break;
      case 7: xf_assign_launch_ks<7>(ctx, a, metric); break;
      default: xf_assign_launch_ks<8>(ctx, a, metric); break;
    }
  }
  ScopedTimer t2(ctx, "ma_recheck");      // rows the surrogate left undecided (>= 4 candidates inside the margin, non-finite rows): exact distances to every centroid
  ctx->count_stage("xf_assign");
  MaArgs ma{};
  ma.x = p.x; ma.n = p.n; ma.ldx = p.ldx; ma.d = d; ma.k = p.k; ma.cent = p.cent; ma.bias = p.bias;
  ma.ids = p.ids; ma.dists = p.dists; ma.check_finite = p.check_finite ? 1 : 0; ma.fb_cnt = a.afb_cnt; ma.fb_rows = afb_rows; ma.active = p.active;
  LH_TRY(ma_recompute_launch(ctx, ma, metric, LANCE_HIP_F32));
  LH_CHECK_HIP(hipGetLastError());
  return LANCE_HIP_OK;
}

// ---- find_partitions over thousands of lists: the sweep with per-group keys (the select kernel is coarse_select_kernel<.., GROUPS>) -------
template <int KS>
static void xf_groups_launch_ks(lance_hip_ctx *ctx, const XfArgs &a, int metric, dim3 grid) {
  constexpr size_t lds = xf_lds_bytes<KS, 8>();
  if (metric == METRIC_DOT) hipLaunchKernelGGL((xf_kernel<KS, 8, METRIC_DOT, float, false, 2>), grid, dim3(256), lds, ctx->stream, a);
  else hipLaunchKernelGGL((xf_kernel<KS, 8, METRIC_L2, float, false, 2>), grid, dim3(256), lds, ctx->stream, a);
}

// q: [nq][d] f32, d a multiple of 16 and <= 128; maxbits: four zeroed words ([0] receives max |c|^2); gkey: [nq][2 ng] floats, ng = 4 * ceil(nlist / 64)
size_t xform_coarse_planes_elems(uint32_t nlist, int d) { return (size_t)((nlist + MA_CT - 1) / MA_CT * MA_CT) * (size_t)(2 * d + 16); }

int xform_coarse_planes(lance_hip_ctx *ctx, int metric, const float *cent, uint32_t nlist, int d, uint16_t *cpl, uint32_t *maxbits) {
  const int kpad = (int)((nlist + MA_CT - 1) / MA_CT * MA_CT);
  if (metric == METRIC_DOT) hipLaunchKernelGGL(xf_cent_prep_kernel<METRIC_DOT>, dim3((unsigned)kpad), dim3(64), 0, ctx->stream, cent, (int)nlist, d, cpl, maxbits, (const float *)nullptr, (const uint8_t *)nullptr);
  else hipLaunchKernelGGL(xf_cent_prep_kernel<METRIC_L2>, dim3((unsigned)kpad), dim3(64), 0, ctx->stream, cent, (int)nlist, d, cpl, maxbits, (const float *)nullptr, (const uint8_t *)nullptr);
  LH_CHECK_HIP(hipGetLastError());
  return LANCE_HIP_OK;
}

int launch_xform_sweep_groups(lance_hip_ctx *ctx, int metric, const float *q, uint32_t nq, int d, const float *cent, uint32_t nlist, uint32_t *maxbits,
                              float *gkey, int ng, float *e2, const uint16_t *cpl_ready, const uint32_t *maxbits_ready) {
  const int kpad = (int)((nlist + MA_CT - 1) / MA_CT * MA_CT);
  const uint16_t *cpl = cpl_ready;
  const uint32_t *mb = maxbits_ready;
  if (!cpl_ready || !maxbits_ready) {
    uint16_t *own = ctx->scratch_t<uint16_t>("cq.cpl", xform_coarse_planes_elems(nlist, d));
    if (!own) return LANCE_HIP_ENOMEM;
    LH_TRY(xform_coarse_planes(ctx, metric, cent, nlist, d, own, maxbits));
    cpl = own; mb = maxbits;
  }
  XfArgs a{};
  a.x = q; a.n = nq; a.ldx = d; a.k = (int)nlist; a.cpl = cpl; a.maxbits = mb; a.cent = cent; a.d_total = d;
  a.gkey = gkey; a.ng = ng; a.e2 = e2;
  const unsigned rblocks = (unsigned)cdiv(nq, MA_ROWS);
  const int ntiles = kpad / MA_CT;
  // enough slices of the centroid range to put about two workgroups on every CU
  int slices = (int)std::min<uint64_t>((uint64_t)ntiles, std::max<uint64_t>(1, ((uint64_t)2 * ctx->num_cus) / rblocks));      // (rounded DOWN: 553 workgroups on 512 slots are two rounds, the second one almost empty -- 1.42 ms instead of 0.8 for 10,000 x 65,536, gpurun r06zf)
  a.tiles_per_block = (int)cdiv((uint64_t)ntiles, (uint64_t)slices);
  slices = (int)cdiv((uint64_t)ntiles, (uint64_t)a.tiles_per_block);
  const dim3 grid(rblocks, (unsigned)slices);
  switch (d / 16) {
    case 1: xf_groups_launch_ks<1>(ctx, a, metric, grid); break;
    case 2: xf_groups_launch_ks<2>(ctx, a, metric, grid); break;
    case 3: xf_groups_launch_ks<3>(ctx, a, metric, grid); break;
    case 4: xf_groups_launch_ks<4>(ctx, a, metric, grid); break;
    case 5: xf_groups_launch_ks<5>(ctx, a, metric, grid); break;
    case 6: xf_groups_launch_ks<6>(ctx, a, metric, grid); break;
    case 7: xf_groups_launch_ks<7>(ctx, a, metric, grid); break;
    default: xf_groups_launch_ks<8>(ctx, a, metric, grid); break;
  }
  LH_CHECK_HIP(hipGetLastError());
  return LANCE_HIP_OK;
}

}  // namespace lh
