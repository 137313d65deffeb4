// LDS random-gather micro-benchmark: lookups/clk/CU for 4-, 8- and 16-byte entries (PQ LUT access pattern).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int W>  // W = floats per entry (1,2,4)
__global__ __launch_bounds__(256) void gather(const uint8_t *codes, float *out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lut[];
  for (int i = threadIdx.x; i < 16 * 256 * W; i += 256) lut[i] = (float)(i % 97);
  __syncthreads();
  const uint4 *c4 = reinterpret_cast<const uint4 *>(codes) + (blockIdx.x * 256 + threadIdx.x) * 4;
  float acc[W];
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
          if constexpr (W == 2) { f2 v = *reinterpret_cast<const f2 *>(&lut[(mm * 256 + c) * 2]); acc[0] += v.x; acc[1] += v.y; }
          if constexpr (W == 4) { f4 v = *reinterpret_cast<const f4 *>(&lut[(mm * 256 + c) * 4]); acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w; }
        }
    }
  }
  float s = 0.f;
  for (int w = 0; w < W; ++w) s += acc[w];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  const int blocks = 256 * 4, iters = 200;
  std::vector<uint8_t> h((size_t)blocks * 256 * 64);
  for (auto &v : h) v = rand() & 255;
  uint8_t *d; float *o;
  hipMalloc(&d, h.size()); hipMalloc(&o, blocks * 256 * 4);
  hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int W : {1, 2, 4}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(a);
      if (W == 1) hipLaunchKernelGGL(gather<1>, dim3(blocks), dim3(256), 16 * 256 * 4 * 1, 0, d, o, iters);
      if (W == 2) hipLaunchKernelGGL(gather<2>, dim3(blocks), dim3(256), 16 * 256 * 4 * 2, 0, d, o, iters);
      if (W == 4) hipLaunchKernelGGL(gather<4>, dim3(blocks), dim3(256), 16 * 256 * 4 * 4, 0, d, o, iters);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      double lookups = (double)blocks * 256 * iters * 64;   // row-lookups (each returns W values)
      if (rep) printf("W=%d: %.3f ms, %.2f G row-lookups/s, %.2f G values/s, per CU per clk@2.4GHz: %.2f row-lookups (%.2f values)\n", W, ms,
                      lookups / ms / 1e6, lookups * W / ms / 1e6, lookups / (ms * 1e-3) / 256 / 2.4e9, lookups * W / (ms * 1e-3) / 256 / 2.4e9);
    }
  }
  return 0;
}
