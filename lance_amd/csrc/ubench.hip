// ubench.hip -- in-process ceilings for bench.py's roofline: what the LDS random-gather pattern of the ADC scan, a
// device copy and a dependent-free VALU stream reach on THIS box, measured with HIP events on the context's stream.
// The figures are denominators ("peak") for roofline.frac; nothing on the product path depends on them.
#include <vector>

#include "common.h"

namespace lh {

typedef float f2u __attribute__((ext_vector_type(2)));
typedef float f4u __attribute__((ext_vector_type(4)));

// PQ-LUT access pattern: 16 tables x 256 entries of W floats; every lane gathers table[m][code] for 16 code bytes of
// its row (codes random).  512-lane workgroups, LDS footprint = the LUT only -> as many workgroups per CU as fit.
template <int W>
__global__ __launch_bounds__(512) void ub_lds_gather_kernel(const uint8_t *__restrict__ codes, float *__restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lut[];
  for (int i = threadIdx.x; i < 16 * 256 * W; i += 512) lut[i] = (float)(i % 97);
  __syncthreads();
  const uint4 *c4 = reinterpret_cast<const uint4 *>(codes) + ((size_t)blockIdx.x * 512 + threadIdx.x) * 4;
  float acc[W];
#pragma unroll
  for (int w = 0; w < W; ++w) acc[w] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint4 cw = c4[u];
      const uint32_t cws[4] = {cw.x + it, cw.y + it, cw.z + it, cw.w + it};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int mm = e * 4 + b;
          const uint32_t c = (cws[e] >> (8 * b)) & 255u;
          if constexpr (W == 1) acc[0] += lut[mm * 256 + c];
          if constexpr (W == 2) { const f2u v = *reinterpret_cast<const f2u *>(&lut[(mm * 256 + c) * 2]); acc[0] += v.x; acc[1] += v.y; }
          if constexpr (W == 4) { const f4u v = *reinterpret_cast<const f4u *>(&lut[(mm * 256 + c) * 4]); acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w; }
        }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < W; ++w) s += acc[w];
  out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
}

// Code-major, m-staggered table: entry (c, m) of 8 bytes (4 x u16) at byte c * 8M + 8m; lane l visits the
// sub-quantisers in the order m = (l + s) mod M (integer sums are order-free).  Bank pair of a ds_read_b64 = (c * M + m) mod 32:
// for M = 32 every lane of a 32-lane group has its own pair whatever the codes are (conflict-free by construction), for M = 16
// lanes l and l + 16 share a pair when their codes have the same parity.  Price, modelled here exactly as a scan kernel would
// pay it: the row's code bytes are rotated once per row by (l mod M) bytes (v_alignbyte + two / three levels of v_cndmask)
// so that step s reads a compile-time byte position, and the lane-dependent part of the address, 8 * ((l + s) mod M), sits in
// M registers computed once per kernel; per lookup: byte -> c << log2(8M) (SDWA shift), OR with the step's offset, ds_read_b64,
// two v_add_u32.
template <int M>
__global__ __launch_bounds__(512) void ub_lds_stagger_kernel(const uint8_t *__restrict__ codes, uint32_t *__restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(16))) uint2 slut[];   // [256][M]
  for (int i = threadIdx.x; i < 256 * M; i += 512) slut[i] = make_uint2((uint32_t)(i % 97), (uint32_t)(i % 89));
  __syncthreads();
  constexpr int W = M / 4;                      // code dwords per row
  constexpr int SH = M == 16 ? 7 : 8;           // log2(8 * M)
  const uint32_t r = threadIdx.x & (M - 1);
  uint32_t off[M];
#pragma unroll
  for (int s = 0; s < M; ++s) off[s] = ((r + s) & (M - 1)) * 8u;
  const uint32_t *cw = reinterpret_cast<const uint32_t *>(codes) + ((size_t)blockIdx.x * 512 + threadIdx.x) * 16;
  uint32_t a0 = 0, a1 = 0;
  const char *base = reinterpret_cast<const char *>(slut);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16 / W; ++u) {
      uint32_t b[W], t[W];
#pragma unroll
      for (int i = 0; i < W; ++i) b[i] = cw[u * W + i] + it;
      // bytes rotated left by r: first the in-dword part, then whole dwords by r >> 2
#pragma unroll
      for (int i = 0; i < W; ++i) t[i] = __builtin_amdgcn_alignbyte(b[(i + 1) % W], b[i], r & 3u);
#pragma unroll
      for (int lvl = 0; (1 << lvl) < W; ++lvl) {
        const bool on = ((r >> 2) >> lvl) & 1u;
#pragma unroll
        for (int i = 0; i < W; ++i) b[i] = on ? t[(i + (1 << lvl)) % W] : t[i];
#pragma unroll
        for (int i = 0; i < W; ++i) t[i] = b[i];
      }
#pragma unroll
      for (int s = 0; s < M; ++s) {
        const uint32_t c = (t[s >> 2] >> (8 * (s & 3))) & 255u;
        const uint2 v = *reinterpret_cast<const uint2 *>(base + ((c << SH) | off[s]));
        a0 += v.x; a1 += v.y;
      }
    }
  }
  out[(size_t)blockIdx.x * 512 + threadIdx.x] = a0 ^ a1;
}

// the same loop on today's [m][code] layout with the same integer accumulate (reference point for the staggered variant)
__global__ __launch_bounds__(512) void ub_lds_u16x4_kernel(const uint8_t *__restrict__ codes, uint32_t *__restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(16))) uint2 slut[];   // [16][256]
  for (int i = threadIdx.x; i < 256 * 16; i += 512) slut[i] = make_uint2((uint32_t)(i % 97), (uint32_t)(i % 89));
  __syncthreads();
  const uint4 *c4 = reinterpret_cast<const uint4 *>(codes) + ((size_t)blockIdx.x * 512 + threadIdx.x) * 4;
  uint32_t a0 = 0, a1 = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint4 cw = c4[u];
      const uint32_t cws[4] = {cw.x + it, cw.y + it, cw.z + it, cw.w + it};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const uint2 v = slut[(e * 4 + b) * 256 + ((cws[e] >> (8 * b)) & 255u)];
          a0 += v.x; a1 += v.y;
        }
    }
  }
  out[(size_t)blockIdx.x * 512 + threadIdx.x] = a0 ^ a1;
}

// conflict-free ds_read_b64 stream (lane l reads entry l of a rotating window): the LDS pipe's own ceiling on this box
__global__ __launch_bounds__(512) void ub_lds_linear_kernel(uint32_t *__restrict__ out, int iters) {
  __shared__ __attribute__((aligned(16))) uint2 slut[4096];
  for (int i = threadIdx.x; i < 4096; i += 512) slut[i] = make_uint2((uint32_t)i, (uint32_t)(i * 3));
  __syncthreads();
  uint32_t a0 = 0, a1 = 0;
  uint32_t idx = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 64; ++s) {
      const uint2 v = slut[(idx + s * 64) & 4095u];
      a0 += v.x; a1 += v.y;
    }
    idx += a0 & 64u;   // data-dependent so that the loads cannot be hoisted
  }
  out[(size_t)blockIdx.x * 512 + threadIdx.x] = a0 ^ a1;
}

__global__ __launch_bounds__(256) void ub_copy_kernel(const f4u *__restrict__ src, f4u *__restrict__ dst, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

// independent f32 adds (8 accumulators per lane): wave-instructions issued per second
template <int PACKED>
__global__ __launch_bounds__(256) void ub_valu_kernel(float *__restrict__ out, int iters, float seed) {
  f2u a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i].x = seed + i + threadIdx.x; a[i].y = seed - i; }
  const f2u inc = {seed, seed * 0.5f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if constexpr (PACKED) a[i] = a[i] + inc;     // v_pk_add_f32
        else a[i].x = a[i].x + inc.x;                // v_add_f32
      }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

// dense MFMA issue rate: four independent 32x32x16 accumulator chains per wave, operands in registers (bench.py: roofline peak_measured)
typedef _Float16 ub_f16x8 __attribute__((ext_vector_type(8)));
typedef short ub_bf16x8 __attribute__((ext_vector_type(8)));
typedef float ub_f32x16 __attribute__((ext_vector_type(16)));
template <int BF16>
__global__ __launch_bounds__(256) void ub_mfma_kernel(float *__restrict__ out, int iters, float seed) {
  ub_f32x16 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[c][v] = 0.0f;
  ub_f16x8 ah, bh;
  ub_bf16x8 ab, bb;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    ah[e] = (_Float16)(seed * 0.001f * (float)((threadIdx.x + e) & 7)); bh[e] = (_Float16)(seed * 0.002f * (float)((threadIdx.x * 3 + e) & 7));
    ab[e] = (short)(0x3C00 + ((threadIdx.x + e) & 7)); bb[e] = (short)(0x3B80 + ((threadIdx.x * 3 + e) & 7));
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if constexpr (BF16) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[c], 0, 0, 0);
        else acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[c], 0, 0, 0);
      }
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int v = 0; v < 16; ++v) s += acc[c][v];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

}  // namespace lh

using namespace lh;

extern "C" {

// what: 0..2 = LDS random gather of 4 / 8 / 16-byte entries (result: gathers per second, one gather = one lane's read);
//       3 = device copy (bytes read + written per second); 4 / 5 = f32 VALU wave-instructions per second (v_add_f32 /
//       v_pk_add_f32, 64 lanes each); 6 / 7 = code-major m-staggered u16x4 table, M = 16 / 32 (lane-gathers per second, all
//       address work included); 8 = today's [m][code] u16x4 table with the same integer accumulate; 9 = conflict-free ds_read_b64;
//       10 / 11 = dense v_mfma_f32_32x32x16_f16 / _bf16 rate, flop per second (the roofline's measured MFMA peak)
int lance_hip_ubench(lance_hip_ctx *ctx, int what, double *result) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(ctx && result, "ubench: NULL argument");
  LH_REQUIRE(what >= 0 && what <= 11, "ubench: unknown measurement %d", what);
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  hipEvent_t e0, e1;
  LH_CHECK_HIP(hipEventCreate(&e0));
  LH_CHECK_HIP(hipEventCreate(&e1));
  float best = 1e30f;
  double work = 0.0;
  int rc = LANCE_HIP_OK;
  if (what <= 2) {
    const int W = 1 << what, blocks = ctx->num_cus * 12, iters = 100;
    const size_t nbytes = (size_t)blocks * 512 * 64;
    uint8_t *codes = ctx->scratch_t<uint8_t>("ubench.codes", nbytes);
    float *out = ctx->scratch_t<float>("ubench.out", (size_t)blocks * 512);
    if (!codes || !out) return LANCE_HIP_ENOMEM;
    std::vector<uint8_t> h(nbytes);
    uint32_t s = 12345u;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (uint8_t)(s >> 24); }
    LH_CHECK_HIP(hipMemcpyAsync(codes, h.data(), nbytes, hipMemcpyHostToDevice, ctx->stream));
    LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    const size_t lds = (size_t)16 * 256 * 4 * W;
    for (int rep = 0; rep < 4; ++rep) {
      LH_CHECK_HIP(hipEventRecord(e0, ctx->stream));
      if (W == 1) hipLaunchKernelGGL(ub_lds_gather_kernel<1>, dim3(blocks), dim3(512), lds, ctx->stream, codes, out, iters);
      if (W == 2) hipLaunchKernelGGL(ub_lds_gather_kernel<2>, dim3(blocks), dim3(512), lds, ctx->stream, codes, out, iters);
      if (W == 4) hipLaunchKernelGGL(ub_lds_gather_kernel<4>, dim3(blocks), dim3(512), lds, ctx->stream, codes, out, iters);
      LH_CHECK_HIP(hipEventRecord(e1, ctx->stream));
      LH_CHECK_HIP(hipEventSynchronize(e1));
      float ms = 0.f;
      LH_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
      if (rep) best = std::min(best, ms);
    }
    work = (double)blocks * 512 * iters * 64;
  } else if (what >= 6 && what <= 9) {
    // 6 / 7: staggered code-major table, M = 16 / 32; 8: today's layout with the integer accumulate; 9: conflict-free ds_read_b64
    const int blocks = ctx->num_cus * 12, iters = 100;
    const size_t nbytes = (size_t)blocks * 512 * 64;
    uint8_t *codes = ctx->scratch_t<uint8_t>("ubench.codes", nbytes);
    uint32_t *out = ctx->scratch_t<uint32_t>("ubench.out", (size_t)blocks * 512);
    if (!codes || !out) return LANCE_HIP_ENOMEM;
    std::vector<uint8_t> h(nbytes);
    uint32_t s = 12345u;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (uint8_t)(s >> 24); }
    LH_CHECK_HIP(hipMemcpyAsync(codes, h.data(), nbytes, hipMemcpyHostToDevice, ctx->stream));
    LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    for (int rep = 0; rep < 4; ++rep) {
      LH_CHECK_HIP(hipEventRecord(e0, ctx->stream));
      if (what == 6) hipLaunchKernelGGL(ub_lds_stagger_kernel<16>, dim3(blocks), dim3(512), (size_t)256 * 16 * 8, ctx->stream, codes, out, iters);
      if (what == 7) hipLaunchKernelGGL(ub_lds_stagger_kernel<32>, dim3(blocks), dim3(512), (size_t)256 * 32 * 8, ctx->stream, codes, out, iters);
      if (what == 8) hipLaunchKernelGGL(ub_lds_u16x4_kernel, dim3(blocks), dim3(512), (size_t)256 * 16 * 8, ctx->stream, codes, out, iters);
      if (what == 9) hipLaunchKernelGGL(ub_lds_linear_kernel, dim3(blocks), dim3(512), 0, ctx->stream, out, iters);
      LH_CHECK_HIP(hipEventRecord(e1, ctx->stream));
      LH_CHECK_HIP(hipEventSynchronize(e1));
      float ms = 0.f;
      LH_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
      if (rep) best = std::min(best, ms);
    }
    work = (double)blocks * 512 * iters * 64;   // lane-gathers: 64 per lane per iteration in every variant
  } else if (what >= 10) {
    // 10 / 11: v_mfma_f32_32x32x16_f16 / _bf16, flop per second (2 x 32 x 32 x 16 per instruction), two waves per SIMD
    const int blocks = ctx->num_cus * 2, iters = 4000;
    float *out = ctx->scratch_t<float>("ubench.out", (size_t)blocks * 512);
    if (!out) return LANCE_HIP_ENOMEM;
    for (int rep = 0; rep < 4; ++rep) {
      LH_CHECK_HIP(hipEventRecord(e0, ctx->stream));
      if (what == 10) hipLaunchKernelGGL(ub_mfma_kernel<0>, dim3(blocks), dim3(256), 0, ctx->stream, out, iters, 1.0f);
      else hipLaunchKernelGGL(ub_mfma_kernel<1>, dim3(blocks), dim3(256), 0, ctx->stream, out, iters, 1.0f);
      LH_CHECK_HIP(hipEventRecord(e1, ctx->stream));
      LH_CHECK_HIP(hipEventSynchronize(e1));
      float ms = 0.f;
      LH_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
      if (rep) best = std::min(best, ms);
    }
    work = (double)blocks * 4 * iters * 16 * 32768.0;   // 4 waves x 16 MFMAs per iteration x 2 * 32 * 32 * 16 flop
  } else if (what == 3) {
    const size_t n4 = (size_t)64 << 20;   // 1 GiB each way
    f4u *src = ctx->scratch_t<f4u>("ubench.src", n4);
    f4u *dst = ctx->scratch_t<f4u>("ubench.dst", n4);
    if (!src || !dst) return LANCE_HIP_ENOMEM;
    LH_CHECK_HIP(hipMemsetAsync(src, 1, n4 * 16, ctx->stream));
    for (int rep = 0; rep < 4; ++rep) {
      LH_CHECK_HIP(hipEventRecord(e0, ctx->stream));
      hipLaunchKernelGGL(ub_copy_kernel, dim3(ctx->num_cus * 16), dim3(256), 0, ctx->stream, src, dst, n4);
      LH_CHECK_HIP(hipEventRecord(e1, ctx->stream));
      LH_CHECK_HIP(hipEventSynchronize(e1));
      float ms = 0.f;
      LH_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
      if (rep) best = std::min(best, ms);
    }
    work = (double)n4 * 32;
  } else {
    const int blocks = ctx->num_cus * 8, iters = 2000;
    float *out = ctx->scratch_t<float>("ubench.out", (size_t)blocks * 512);
    if (!out) return LANCE_HIP_ENOMEM;
    for (int rep = 0; rep < 4; ++rep) {
      LH_CHECK_HIP(hipEventRecord(e0, ctx->stream));
      if (what == 4) hipLaunchKernelGGL(ub_valu_kernel<0>, dim3(blocks), dim3(256), 0, ctx->stream, out, iters, 1.0f);
      else hipLaunchKernelGGL(ub_valu_kernel<1>, dim3(blocks), dim3(256), 0, ctx->stream, out, iters, 1.0f);
      LH_CHECK_HIP(hipEventRecord(e1, ctx->stream));
      LH_CHECK_HIP(hipEventSynchronize(e1));
      float ms = 0.f;
      LH_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
      if (rep) best = std::min(best, ms);
    }
    work = (double)blocks * 4 * iters * 64;   // wave-instructions: 4 waves x 64 adds per iteration
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  LH_CHECK_HIP(hipGetLastError());
  *result = work / ((double)best * 1e-3);
  return rc;
}

}  // extern "C"
