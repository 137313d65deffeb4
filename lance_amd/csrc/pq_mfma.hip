// pq_mfma.hip -- the PQ sub-quantiser argmin (k-means E-step of the codebook training; pq/builder.rs:89-157 -> kmeans.rs:317-369) on
// the matrix cores: bf16x3 MFMA surrogate distances pick each row's candidate codeword, exact arithmetic decides.
//
// Why: pairwise_kernel<8, L2, 256> -- 16 sub-quantisers x 65,536 rows x 256 codewords x 8 dimensions, exact
// l2_scalar arithmetic, 24 VALU per (row, codeword) -- was the largest line of the build's kernel trace (50 x 155 us).  The
// round-2 MFMA attempt lost to it because its epilogue kept a running top-4 with ~12 VALU per pair.  This one spends 4:
//   s'' = |c|^2 - 2 x.c + (|x|^2 + margin) >= 0      packed: one v_pk_add_f32 + one v_pk_fma_f32 per TWO pairs
//   key = (bits(s'') & ~0xFF) | codeword              one v_and_or_b32: the index rides in the low mantissa bits, so that
//   m1 = min(m1, key);  m2 = med3(m1, m2, key)        two integer ops keep the two smallest keys WITH their indices
// x.c = xh.ch + xh.cl + xl.ch from v_mfma_f32_32x32x8bf16_1k / 32x32x16_bf16 (two-term bf16 split of both operands, f32
// accumulation: |error| <= 2^-14 |x||c|, as in mfma_assign.hip); clearing 8 mantissa bits costs < 2^-15 relative.  With
// margin = 2^-12 (|x|^2 + max|c|^2) every codeword whose exact distance could be the minimum has a key within `margin` of the
// smallest key.  If the SECOND smallest key is farther than that, the smallest is the reference's argmin and its exact
// distance (l2_scalar order, kmeans.rs:1350-1369) is computed once.  Otherwise (1-2 % of the rows: near ties, duplicate
// codewords, non-finite rows) the (row, sub-quantiser) item goes to a list and pq_mfma_fix_kernel gives it a wave: exact
// distances to all 256 codewords, argmin_value_float semantics (first strictly smallest, NaN / +inf never selected, all-NaN
// -> None; kernels.rs:79-111).  Ids, distances and codes are therefore bit-equal to pairwise_kernel's.
// Operand roles as in mfma_assign.hip: A = 32 codewords (rows of D), B = 32 data rows (columns of D): one data row per lane pair.
#include <algorithm>
#include <cstdlib>

#include <hip/hip_fp16.h>

#include "common.h"
#include "exact.cuh"
#include "kernels.h"

#pragma clang fp contract(off)

namespace lh {

typedef short pq_bf16x4 __attribute__((ext_vector_type(4)));
typedef short pq_bf16x8 __attribute__((ext_vector_type(8)));
typedef float pq_f32x16 __attribute__((ext_vector_type(16)));

#ifndef LH_PQM_RG
#define LH_PQM_RG 8
#endif
constexpr int PQM_RG = LH_PQM_RG;         // 32-row groups per wave: a workgroup covers 4 waves x 8 x 32 = 1024 rows (the codebook's LDS set-up is paid per workgroup: 4 -> 8 groups took 0.28 ms off the C2 build, gpurun r05l)
constexpr int PQM_WG_ROWS = 4 * PQM_RG * 32;

__device__ __forceinline__ uint32_t pqm_bf16_rne(float x) {
  const uint32_t u = __float_as_uint(x);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (u >> 16) | 0x40u;
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float pqm_bf16_f32(uint32_t b) { return __uint_as_float(b << 16); }
__device__ __forceinline__ uint32_t pqm_med3(uint32_t x, uint32_t y, uint32_t z) {   // median of three (the compiler emits min + max otherwise)
  uint32_t r;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
  return r;
}

__device__ __forceinline__ f4 pqm_load4(const float *p) { return *reinterpret_cast<const f4 *>(p); }
__device__ __forceinline__ f4 pqm_load4(const __half *p) {
  const uint2 u = *reinterpret_cast<const uint2 *>(p);
  const __half2 a = *reinterpret_cast<const __half2 *>(&u.x), b = *reinterpret_cast<const __half2 *>(&u.y);
  return f4{__low2float(a), __high2float(a), __low2float(b), __high2float(b)};
}
__device__ __forceinline__ f4 pqm_load4(const int8_t *p) {
  const uint32_t u = *reinterpret_cast<const uint32_t *>(p);
  return f4{(float)(int8_t)(u & 255u), (float)(int8_t)((u >> 8) & 255u), (float)(int8_t)((u >> 16) & 255u), (float)(int8_t)(u >> 24)};
}

// this lane's EL elements of (row, sub-quantiser b): training mode reads the f32 residual matrix, encode mode builds the residual
template <int EL, bool ENC, typename TX>
__device__ __forceinline__ void pqm_load_row(const PqmArgs &a, int64_t row, int b, int col0, float (&xv)[EL]) {
  const PairwiseArgs &p = a.p;
  if constexpr (!ENC) {
    const float *src = p.x + (int64_t)b * p.x_batch_off + row * p.ldx + col0;
#pragma unroll
    for (int e4 = 0; e4 < EL / 4; ++e4) {
      const f4 v = pqm_load4(src + 4 * e4);
      xv[4 * e4 + 0] = v.x; xv[4 * e4 + 1] = v.y; xv[4 * e4 + 2] = v.z; xv[4 * e4 + 3] = v.w;
    }
  } else {
    const uint32_t part = a.rcent ? a.rpart[row] : 0u;
    if (a.rcent && part == LANCE_HIP_NONE) return;       // no partition: the zero vector (xv is zero-initialised by the caller)
    const TX *src = static_cast<const TX *>(a.xn) + row * p.ldx + (int64_t)b * p.x_batch_off + col0;
#pragma unroll
    for (int e4 = 0; e4 < EL / 4; ++e4) {
      f4 v = pqm_load4(src + 4 * e4);
      if (a.rcent) {
        v = v - pqm_load4(a.rcent + (int64_t)part * p.ldx + (int64_t)b * p.x_batch_off + col0 + 4 * e4);
        if (a.round_f16) {
          v.x = __half2float(__float2half_rn(v.x)); v.y = __half2float(__float2half_rn(v.y));
          v.z = __half2float(__float2half_rn(v.z)); v.w = __half2float(__float2half_rn(v.w));
        }
      }
      xv[4 * e4 + 0] = v.x; xv[4 * e4 + 1] = v.y; xv[4 * e4 + 2] = v.z; xv[4 * e4 + 3] = v.w;
    }
  }
}

template <int EL> struct PqmFrag;                       // EL bf16 elements per lane and MFMA operand
template <> struct PqmFrag<4> { typedef pq_bf16x4 type; };
template <> struct PqmFrag<8> { typedef pq_bf16x8 type; };

template <int EL>
__device__ __forceinline__ pq_f32x16 pqm_mfma(typename PqmFrag<EL>::type a, typename PqmFrag<EL>::type b, pq_f32x16 c) {
  if constexpr (EL == 4) return __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a, b, c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// SD = sub-vector length (4, 8, 16).  MFMA K = 8 for SD <= 8 (SD = 4: the upper half of K is zero), 16 for SD = 16; a lane
// holds EL = K / 2 consecutive elements: lane (j, g) the dimensions [g * EL, (g + 1) * EL).
template <int SD, bool ENC = false, typename TX = float>
__global__ __launch_bounds__(256) void pq_mfma_estep_kernel(PqmArgs a) {
  const PairwiseArgs &p = a.p;
  constexpr int KK = SD == 16 ? 16 : 8, EL = KK / 2;
  typedef typename PqmFrag<EL>::type frag_t;
  __shared__ __attribute__((aligned(16))) float cbf[256 * SD];      // the sub-quantiser's codebook, f32 (exact re-check)
  __shared__ __attribute__((aligned(16))) uint16_t chi[256 * KK], clo[256 * KK];
  __shared__ __attribute__((aligned(16))) float cn[256];
  __shared__ uint32_t s_cmax;
  const int b = blockIdx.y;
  if (p.active && !p.active[b]) return;
  const float *cb = p.cent + (int64_t)b * p.cent_batch_stride;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, g = lane >> 5;
  if (threadIdx.x == 0) s_cmax = 0u;
  __syncthreads();
  {   // codeword c = threadIdx.x: f32 copy, bf16 planes (zero-padded to KK), squared norm
    const int c = threadIdx.x;
    float s = 0.0f;
#pragma unroll
    for (int e = 0; e < KK; ++e) {
      const float v = e < SD ? cb[(int64_t)c * SD + e] : 0.0f;
      if (e < SD) cbf[c * SD + e] = v;
      const uint32_t hb = pqm_bf16_rne(v);
      chi[c * KK + e] = (uint16_t)hb;
      clo[c * KK + e] = (uint16_t)pqm_bf16_rne(v - pqm_bf16_f32(hb));
      s += v * v;
    }
    cn[c] = s;
    if (s == s) atomicMax(&s_cmax, __float_as_uint(fabsf(s)));   // non-negative floats order like their bit patterns
  }
  __syncthreads();
  const float cmax2 = __uint_as_float(s_cmax);
  // A fragments of all 8 codeword tiles stay in registers for the whole kernel
  frag_t ah[8], al[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    ah[t] = *reinterpret_cast<const frag_t *>(&chi[(t * 32 + j) * KK + g * EL]);
    al[t] = *reinterpret_cast<const frag_t *>(&clo[(t * 32 + j) * KK + g * EL]);
  }
  const uint32_t gidx = (uint32_t)(4 * g);
#pragma unroll 1
  for (int rg = 0; rg < PQM_RG; ++rg) {
    const int64_t row = (int64_t)blockIdx.x * PQM_WG_ROWS + (int64_t)(rg * 4 + wave) * 32 + j;
    if ((int64_t)blockIdx.x * PQM_WG_ROWS + (int64_t)(rg * 4 + wave) * 32 >= p.n) break;   // wave-uniform
    const bool valid = row < p.n;
    // this lane's EL elements of the row (zero beyond SD or beyond n)
    float xv[EL];
#pragma unroll
    for (int e = 0; e < EL; ++e) xv[e] = 0.0f;
    if (valid && g * EL < SD) pqm_load_row<EL, ENC, TX>(a, row, b, g * EL, xv);
    frag_t xh, xl;
    float xn2 = 0.0f;
#pragma unroll
    for (int e = 0; e < EL; ++e) {
      const uint32_t hb = pqm_bf16_rne(xv[e]);
      xh[e] = (short)hb; xl[e] = (short)pqm_bf16_rne(xv[e] - pqm_bf16_f32(hb));
      xn2 += xv[e] * xv[e];
    }
    xn2 += __shfl_xor(xn2, 32, 64);
    const float margin = 0.000244140625f * (xn2 + cmax2);      // 2^-12 (|x|^2 + max|c|^2)
    const float rowc = xn2 + margin;                            // keeps every surrogate >= 0: float order == unsigned order of the bits
    const f2 rowc2 = {rowc, rowc};
    const f2 m2x = {-2.0f, -2.0f};
    uint32_t m1 = 0xFFFFFFFFu, m2 = 0xFFFFFFFFu;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      // (the zero C operand is an inline constant of the first MFMA: no sixteen v_mov per tile)
      pq_f32x16 acc = pqm_mfma<EL>(ah[t], xh, pq_f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f});
      acc = pqm_mfma<EL>(ah[t], xl, acc);
      acc = pqm_mfma<EL>(al[t], xh, acc);
      // D[codeword i][row j]: lane (j, g) holds codewords i = 8 vq + 4 g + e of the tile
#pragma unroll
      for (int vq = 0; vq < 4; ++vq) {
        const f4 cn4 = *reinterpret_cast<const f4 *>(&cn[t * 32 + 8 * vq + 4 * g]);
        const f2 s01 = __builtin_elementwise_fma(m2x, f2{acc[vq * 4 + 0], acc[vq * 4 + 1]}, f2{cn4.x, cn4.y} + rowc2);
        const f2 s23 = __builtin_elementwise_fma(m2x, f2{acc[vq * 4 + 2], acc[vq * 4 + 3]}, f2{cn4.z, cn4.w} + rowc2);
        const float sv[4] = {s01.x, s01.y, s23.x, s23.y};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t key = (__float_as_uint(sv[e]) & 0xFFFFFF00u) | ((uint32_t)(t * 32 + 8 * vq + e) + gidx);
          m2 = pqm_med3(m1, m2, key);                   // second smallest of {m1 <= m2, key}
          m1 = min(m1, key);
        }
      }
    }
    {   // the partner lane holds the other 128 codewords of the row
      const uint32_t p1 = __shfl_xor(m1, 32, 64), p2 = __shfl_xor(m2, 32, 64);
      const uint32_t lo = min(m1, p1), hi = max(m1, p1);
      m2 = min(hi, min(m2, p2));
      m1 = lo;
    }
    // the full row in lane g == 0 (exact re-check): own elements + the partner's
    RegVec<SD> rv;
    if constexpr (SD == 4) {
      rv.q[0] = f4{xv[0], xv[1], xv[2], xv[3]};
    } else {
      float ov[EL];
#pragma unroll
      for (int e = 0; e < EL; ++e) ov[e] = __shfl_xor(xv[e], 32, 64);
#pragma unroll
      for (int e4 = 0; e4 < EL / 4; ++e4) {
        rv.q[e4] = f4{xv[4 * e4], xv[4 * e4 + 1], xv[4 * e4 + 2], xv[4 * e4 + 3]};
        rv.q[EL / 4 + e4] = f4{ov[4 * e4], ov[4 * e4 + 1], ov[4 * e4 + 2], ov[4 * e4 + 3]};
      }
    }
    if (g == 0 && valid) {
      const float s1 = __uint_as_float(m1 & 0xFFFFFF00u), s2 = __uint_as_float(m2 & 0xFFFFFF00u);
      const bool decided = (margin < INFINITY) && (margin > 7.888609052210118e-31f) && (s2 - s1 > margin);     // NaN anywhere: false; margin <= 2^-100: products / cleared key bits in the denormal range
      if (decided) {
        const uint32_t c = m1 & 0xFFu;
        const float v = dist_exact<SD, METRIC_L2>(rv, &cbf[c * SD]);
        // argmin_value_float: a NaN / +inf distance is never selected (only possible with non-finite codewords: then undecided)
        if (v < INFINITY) {
          if (p.ids) p.ids[(int64_t)b * p.out_batch_stride + row] = c;
          if (p.dists) p.dists[(int64_t)b * p.out_batch_stride + row] = v;
          if (p.codes) p.codes[row * p.codes_ld + b] = (uint8_t)c;
        } else {
          const uint32_t slot = atomicAdd(&a.fb_cnt[b], 1u);
          a.fb_items[(int64_t)b * p.n + slot] = (uint32_t)row;
        }
      } else {
        const uint32_t slot = atomicAdd(&a.fb_cnt[b], 1u);
        a.fb_items[(int64_t)b * p.n + slot] = (uint32_t)row;
      }
    }
  }
}

// undecided rows of sub-quantiser b = blockIdx.y: one wave each, exact distances to its 256 codewords (staged in LDS)
template <int SD, bool ENC = false, typename TX = float>
__global__ __launch_bounds__(256) void pq_mfma_fix_kernel(PqmArgs a) {
  const PairwiseArgs &p = a.p;
  __shared__ __attribute__((aligned(16))) float cbf[256 * SD];
  const int b = blockIdx.y;
  if (p.active && !p.active[b]) return;
  const uint32_t cnt = a.fb_cnt[b];
  if (blockIdx.x * 4u >= cnt) return;    // uniform: nothing left for this workgroup
  const float *cb = p.cent + (int64_t)b * p.cent_batch_stride;
  for (int i = threadIdx.x; i < 256 * SD / 4; i += 256) reinterpret_cast<f4 *>(cbf)[i] = reinterpret_cast<const f4 *>(cb)[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const uint32_t nwaves = gridDim.x * 4, w0 = blockIdx.x * 4 + (threadIdx.x >> 6);
  for (uint32_t it = w0; it < cnt; it += nwaves) {
    const int64_t row = a.fb_items[(int64_t)b * p.n + it];
    RegVec<SD> rv;
    {
      float xv[SD];
#pragma unroll
      for (int e = 0; e < SD; ++e) xv[e] = 0.0f;
      pqm_load_row<SD, ENC, TX>(a, row, b, 0, xv);
#pragma unroll
      for (int i = 0; i < SD / 4; ++i) rv.q[i] = f4{xv[4 * i], xv[4 * i + 1], xv[4 * i + 2], xv[4 * i + 3]};
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
    if (lane == 0) {
      if (bi == LANCE_HIP_NONE) best = INFINITY;
      if (p.ids) p.ids[(int64_t)b * p.out_batch_stride + row] = bi;
      if (p.dists) p.dists[(int64_t)b * p.out_batch_stride + row] = best;
      if (p.codes) p.codes[row * p.codes_ld + b] = bi == LANCE_HIP_NONE ? (uint8_t)0 : (uint8_t)bi;
    }
  }
}

bool pq_mfma_supported(const PairwiseArgs &p, int d, int metric, int batches) {
  static const bool off = getenv("LANCE_HIP_NO_MFMA") != nullptr || getenv("LANCE_HIP_NO_MFMA_PQ") != nullptr;
  if (off || metric != METRIC_L2 || p.matrix || p.bias || p.check_finite || p.lanes32) return false;
  if (d != 4 && d != 8 && d != 16) return false;
  if (p.k != 256 || p.n < 2048 || batches < 1) return false;
  if (!p.x || !p.x_aligned || !p.cent_aligned) return false;
  if ((uint64_t)p.n >= (1ull << 32)) return false;                          // row numbers in 32 bits
  // the undecided-row list is n * batches words of persistent scratch and this launcher does not chunk the rows (the encode
  // launcher does): beyond 256 MB the exact kernels take the call (an IVF assign of 10^8 short rows against 256 centroids)
  if ((uint64_t)p.n * (uint64_t)batches * 4 > (256ull << 20)) return false;
  return true;
}

int launch_pq_mfma(lance_hip_ctx *ctx, const PairwiseArgs &p, int d, int batches) {
  PqmArgs a;
  a.p = p; a.batches = batches;
  a.fb_cnt = ctx->scratch_t<uint32_t>("pqm.fb_cnt", (size_t)batches);
  a.fb_items = ctx->scratch_t<uint32_t>("pqm.fb_items", (size_t)p.n * batches);
  if (!a.fb_cnt || !a.fb_items) return LANCE_HIP_ENOMEM;
  const dim3 grid((unsigned)cdiv((uint64_t)p.n, PQM_WG_ROWS), (unsigned)batches);
  // fix kernel: grid (blocks, batches); a workgroup whose first wave has no item returns before staging the codebook
  const dim3 fix_grid((unsigned)std::min<uint64_t>(std::max<uint64_t>(1, cdiv((uint64_t)p.n, 256)), 64), (unsigned)batches);
  ScopedTimer t(ctx, "pq_mfma_estep");
  if (xform_pqtrain_supported(p, d, batches)) {
    // round 6: the transform's PQ phase picks the codewords (xf_pqtrain_kernel: two full K = 16 products per tile, 3 VALU per pair, codeword
    // fragments prepared once per iteration -- which also zeroes the lists' counters); the undecided items come here as before
    LH_TRY(launch_xform_pqtrain(ctx, p, d, batches, a.fb_cnt, a.fb_items));
    if (d == 4) hipLaunchKernelGGL(pq_mfma_fix_kernel<4>, fix_grid, dim3(256), 0, ctx->stream, a);
    else if (d == 8) hipLaunchKernelGGL(pq_mfma_fix_kernel<8>, fix_grid, dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL(pq_mfma_fix_kernel<16>, fix_grid, dim3(256), 0, ctx->stream, a);
    LH_CHECK_HIP(hipGetLastError());
    return LANCE_HIP_OK;
  }
  LH_CHECK_HIP(lh::memset_async(a.fb_cnt, 0, (size_t)batches * 4, ctx->stream));
  switch (d) {
    case 4:
      hipLaunchKernelGGL(pq_mfma_estep_kernel<4>, grid, dim3(256), 0, ctx->stream, a);
      hipLaunchKernelGGL(pq_mfma_fix_kernel<4>, fix_grid, dim3(256), 0, ctx->stream, a);
      break;
    case 8:
      hipLaunchKernelGGL(pq_mfma_estep_kernel<8>, grid, dim3(256), 0, ctx->stream, a);
      hipLaunchKernelGGL(pq_mfma_fix_kernel<8>, fix_grid, dim3(256), 0, ctx->stream, a);
      break;
    default:
      hipLaunchKernelGGL(pq_mfma_estep_kernel<16>, grid, dim3(256), 0, ctx->stream, a);
      hipLaunchKernelGGL(pq_mfma_fix_kernel<16>, fix_grid, dim3(256), 0, ctx->stream, a);
      break;
  }
  LH_CHECK_HIP(hipGetLastError());
  return LANCE_HIP_OK;
}

// ---- fused residual + PQ encode on the same kernels (IvfTransformer chain: ivf.rs:188-236; residual.rs:58-102, pq.rs:116-191) ----
bool pq_mfma_encode_supported(int dtype, int d, int m, int nbits, const void *x, const float *cent, const float *codebook, int64_t n) {
  static const bool off = getenv("LANCE_HIP_NO_MFMA") != nullptr || getenv("LANCE_HIP_NO_MFMA_PQ") != nullptr || getenv("LANCE_HIP_NO_MFMA_ENCODE") != nullptr;
  if (off || nbits != 8 || m <= 0 || d % m != 0) return false;
  const int sd = d / m;
  if (sd != 4 && sd != 8 && sd != 16) return false;
  if (n < 2048) return false;
  const size_t es = dtype == LANCE_HIP_F16 ? 2 : (dtype == LANCE_HIP_I8 ? 1 : 4);
  if (reinterpret_cast<uintptr_t>(x) % (4 * es) || (d % 4)) return false;
  if ((cent && (reinterpret_cast<uintptr_t>(cent) & 15)) || (reinterpret_cast<uintptr_t>(codebook) & 15)) return false;
  return true;
}

template <int SD, typename TX>
static void launch_pqm_encode_tx(lance_hip_ctx *ctx, const PqmArgs &a, dim3 grid, dim3 fix_grid) {
  hipLaunchKernelGGL((pq_mfma_estep_kernel<SD, true, TX>), grid, dim3(256), 0, ctx->stream, a);
  hipLaunchKernelGGL((pq_mfma_fix_kernel<SD, true, TX>), fix_grid, dim3(256), 0, ctx->stream, a);
}

template <int SD>
static void launch_pqm_encode_sd(lance_hip_ctx *ctx, int dtype, const PqmArgs &a, dim3 grid, dim3 fix_grid) {
  if (dtype == LANCE_HIP_F16) launch_pqm_encode_tx<SD, __half>(ctx, a, grid, fix_grid);
  else if (dtype == LANCE_HIP_I8) launch_pqm_encode_tx<SD, int8_t>(ctx, a, grid, fix_grid);
  else launch_pqm_encode_tx<SD, float>(ctx, a, grid, fix_grid);
}

// x: [n][d] rows in the column's element type; cent / part_ids: the residual (NULL / ignored when residual == 0); codes: [n][m]
int launch_pq_mfma_encode(lance_hip_ctx *ctx, int dtype, const void *x, int64_t n, int d, const float *cent, const uint32_t *part_ids,
                          int residual, const float *codebook, int m, uint8_t *codes) {
  if (n == 0) return LANCE_HIP_OK;
  const int sd = d / m;
  // rows go through in chunks: the undecided-row lists are [m][chunk] words of scratch (256 MB at most), whatever n is
  const int64_t chunk = std::max<int64_t>(PQM_WG_ROWS, std::min<int64_t>(n, ((int64_t)64 << 20) / m / PQM_WG_ROWS * PQM_WG_ROWS));
  uint32_t *fb_cnt = ctx->scratch_t<uint32_t>("pqm.fb_cnt", (size_t)m);
  uint32_t *fb_items = ctx->scratch_t<uint32_t>("pqm.fb_items", (size_t)chunk * m);
  if (!fb_cnt || !fb_items) return LANCE_HIP_ENOMEM;
  const size_t es = dtype == LANCE_HIP_F16 ? 2 : (dtype == LANCE_HIP_I8 ? 1 : 4);
  ScopedTimer t(ctx, "encode_fused");
  for (int64_t r0 = 0; r0 < n; r0 += chunk) {
    const int64_t rows = std::min<int64_t>(chunk, n - r0);
    PqmArgs a;
    a.p.n = rows; a.p.ldx = d; a.p.x_batch_off = sd;
    a.p.cent = codebook; a.p.k = 256; a.p.cent_batch_stride = (int64_t)256 * sd;
    a.p.codes = codes + r0 * m; a.p.codes_ld = m;
    a.xn = static_cast<const char *>(x) + (size_t)r0 * d * es;
    a.rcent = residual ? cent : nullptr; a.rpart = part_ids ? part_ids + r0 : nullptr; a.round_f16 = dtype == LANCE_HIP_F16 ? 1 : 0;
    a.batches = m; a.fb_cnt = fb_cnt; a.fb_items = fb_items;
    LH_CHECK_HIP(lh::memset_async(fb_cnt, 0, (size_t)m * 4, ctx->stream));
    const dim3 grid((unsigned)cdiv((uint64_t)rows, PQM_WG_ROWS), (unsigned)m);
    const dim3 fix_grid((unsigned)std::min<uint64_t>(std::max<uint64_t>(1, cdiv((uint64_t)rows, 256)), 64), (unsigned)m);
    if (sd == 4) launch_pqm_encode_sd<4>(ctx, dtype, a, grid, fix_grid);
    else if (sd == 8) launch_pqm_encode_sd<8>(ctx, dtype, a, grid, fix_grid);
    else launch_pqm_encode_sd<16>(ctx, dtype, a, grid, fix_grid);
  }
  LH_CHECK_HIP(hipGetLastError());
  return LANCE_HIP_OK;
}

}  // namespace lh
