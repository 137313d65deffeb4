// ADC scan inner loop, two queries per gather (the main pass of search_pm.hip), with the three ways of turning a PQ code
// into an LDS address.  Question ("next on performance"): the loop issues 2 address VALU ops per gather
// (v_bfe_u32 + v_lshl_add_u32); does removing one of them -- codes stored as 16-bit pre-scaled offsets -- move the loop
// from VALU/LDS co-limited to LDS-limited?
//   A  u8 codes, LUT in dynamic LDS (base in an SGPR): today's kernel
//   B  u8 codes, LUT in STATIC LDS (base folds into the ds_read offset immediate)
//   C  u16 codes pre-scaled by 8 (byte offset of the float2 entry), LUT in static LDS: one v_and / v_lshrrev per gather
// Sums are sequential in m (the parity-relevant order), one float2 per (row, sub-vector): lanes own rows.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scan_addr.hip -o scan_addr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int M = 16;

template <int MODE>
__global__ __launch_bounds__(256) void scan(const uint8_t *__restrict__ codes8, const uint16_t *__restrict__ codes16, int rows_per_thread,
                                            float *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float dyn[];
  __shared__ __attribute__((aligned(16))) float stat[MODE == 0 ? 1 : M * 256 * 2];
  float *lut = MODE == 0 ? dyn : stat;
  for (int i = threadIdx.x; i < M * 256 * 2; i += 256) lut[i] = (float)((i * 2654435761u) >> 20) * 0.001f;
  __syncthreads();
  const char *lutb = reinterpret_cast<const char *>(lut);
  f2 best = {3.0e38f, 3.0e38f};
  const int64_t row0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * rows_per_thread;
  for (int r = 0; r < rows_per_thread; ++r) {
    f2 acc = {0.0f, 0.0f};
    if constexpr (MODE <= 1) {
      const uint4 cw = *reinterpret_cast<const uint4 *>(codes8 + (row0 + r) * M);
      const uint32_t w[4] = {cw.x, cw.y, cw.z, cw.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const uint32_t c = (w[e] >> (8 * b)) & 255u;
          acc += *reinterpret_cast<const f2 *>(lutb + ((e * 4 + b) * 256 + c) * 8);
        }
    } else {
      const uint4 lo = *reinterpret_cast<const uint4 *>(codes16 + (row0 + r) * M);
      const uint4 hi = *reinterpret_cast<const uint4 *>(codes16 + (row0 + r) * M + 8);
      const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        acc += *reinterpret_cast<const f2 *>(lutb + (2 * e) * 2048 + (w[e] & 0xffffu));
        acc += *reinterpret_cast<const f2 *>(lutb + (2 * e + 1) * 2048 + (w[e] >> 16));
      }
    }
    best.x = fminf(best.x, acc.x);
    best.y = fminf(best.y, acc.y);
  }
  out[(int64_t)blockIdx.x * 256 + threadIdx.x] = best.x + best.y;
}

int main() {
  const int blocks = 256 * 8, rpt = 64;
  const size_t rows = (size_t)blocks * 256 * rpt;
  std::vector<uint8_t> h8(rows * M);
  std::vector<uint16_t> h16(rows * M);
  for (size_t i = 0; i < h8.size(); ++i) { h8[i] = rand() & 255; h16[i] = (uint16_t)(h8[i] * 8); }
  uint8_t *d8; uint16_t *d16; float *o;
  hipMalloc(&d8, h8.size()); hipMalloc(&d16, h16.size() * 2); hipMalloc(&o, (size_t)blocks * 256 * 4);
  hipMemcpy(d8, h8.data(), h8.size(), hipMemcpyHostToDevice);
  hipMemcpy(d16, h16.data(), h16.size() * 2, hipMemcpyHostToDevice);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  std::vector<float> ref, got((size_t)blocks * 256);
  const char *names[3] = {"A u8 codes, dynamic-LDS LUT", "B u8 codes, static-LDS LUT ", "C u16 pre-scaled, static   "};
  for (int mode = 0; mode < 3; ++mode) {
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(a);
      if (mode == 0) hipLaunchKernelGGL(scan<0>, dim3(blocks), dim3(256), M * 256 * 8, 0, d8, d16, rpt, o);
      if (mode == 1) hipLaunchKernelGGL(scan<1>, dim3(blocks), dim3(256), 0, 0, d8, d16, rpt, o);
      if (mode == 2) hipLaunchKernelGGL(scan<2>, dim3(blocks), dim3(256), 0, 0, d8, d16, rpt, o);
      hipEventRecord(b); hipEventSynchronize(b);
      hipEventElapsedTime(&ms, a, b);
    }
    hipMemcpy(got.data(), o, got.size() * 4, hipMemcpyDeviceToHost);
    if (mode == 0) ref = got;
    bool same = true;
    for (size_t i = 0; i < got.size(); ++i) same &= got[i] == ref[i];
    const double pairs = (double)rows * 2;
    printf("%s: %.3f ms  %.2f G pair-distances/s  %.2f per clk per CU @2.4 GHz  code bytes %.0f MB  %s\n", names[mode], ms, pairs / ms / 1e6,
           pairs / (ms * 1e-3) / 256 / 2.4e9, (mode == 2 ? 2.0 : 1.0) * rows * M / 1e6, same ? "same result" : "RESULT DIFFERS");
  }
  return 0;
}
