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

}  // namespace lh

using namespace lh;

extern "C" {

// what: 0..2 = LDS random gather of 4 / 8 / 16-byte entries (result: gathers per second, one gather = one lane's read);
//       3 = device copy (bytes read + written per second); 4 / 5 = f32 VALU wave-instructions per second (v_add_f32 /
//       v_pk_add_f32, 64 lanes each)
int lance_hip_ubench(lance_hip_ctx *ctx, int what, double *result) {
  LH_REQUIRE(ctx && result, "ubench: NULL argument");
  LH_REQUIRE(what >= 0 && what <= 5, "ubench: unknown measurement %d", what);
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
