// dtype.hip -- element-type plumbing: f16 buffers are widened (exactly) to f32 scratch before
// the f32 kernels run and narrowed on the way out.  See f16.h for where rounding is applied.
#include <hip/hip_fp16.h>

#include <algorithm>
#include "common.h"
#include "kernels.h"

namespace lh {

__global__ __launch_bounds__(256) void widen_f16_kernel(const __half *__restrict__ in, float *__restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = __half2float(in[i]);
}
__global__ __launch_bounds__(256) void narrow_f16_kernel(const float *__restrict__ in, __half *__restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = __float2half_rn(in[i]);
}

__global__ __launch_bounds__(256) void widen_i8_kernel(const int8_t *__restrict__ in, float *__restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = (float)in[i];
}

int check_dtype(int dtype, const char *what) {
  LH_REQUIRE(dtype == LANCE_HIP_F32 || dtype == LANCE_HIP_F16 || dtype == LANCE_HIP_I8,
             "%s: unsupported element type %d (f32 = 0, f16 = 1, i8 = 2)", what, dtype);
  return LANCE_HIP_OK;
}

// element type of the model operands (centroids, codebook, outputs) that go with data of type `dtype`
int model_dtype(int dtype) { return dtype == LANCE_HIP_I8 ? LANCE_HIP_F32 : dtype; }

static void launch_widen(lance_hip_ctx *ctx, int dtype, const void *p, size_t count, float *dst) {
  const unsigned grid = (unsigned)std::min<uint64_t>(cdiv(count, 256), 65536);
  if (dtype == LANCE_HIP_I8)
    hipLaunchKernelGGL(widen_i8_kernel, dim3(grid), dim3(256), 0, ctx->stream, static_cast<const int8_t *>(p), dst, (int64_t)count);
  else
    hipLaunchKernelGGL(widen_f16_kernel, dim3(grid), dim3(256), 0, ctx->stream, static_cast<const __half *>(p), dst, (int64_t)count);
}

int as_f32(lance_hip_ctx *ctx, int dtype, const void *p, size_t count, const char *slot, const float **out) {
  if (dtype == LANCE_HIP_F32 || p == nullptr) {
    *out = static_cast<const float *>(p);
    return LANCE_HIP_OK;
  }
  float *w = ctx->scratch_t<float>(slot, count ? count : 1);
  if (!w) return LANCE_HIP_ENOMEM;
  if (count) {
    launch_widen(ctx, dtype, p, count, w);
    LH_CHECK_HIP(hipGetLastError());
  }
  *out = w;
  return LANCE_HIP_OK;
}

int widen_into(lance_hip_ctx *ctx, int dtype, const void *p, size_t count, float *dst) {
  if (count == 0) return LANCE_HIP_OK;
  if (dtype == LANCE_HIP_F32) {
    LH_CHECK_HIP(hipMemcpyAsync(dst, p, count * 4, hipMemcpyDefault, ctx->stream));
  } else {
    launch_widen(ctx, dtype, p, count, dst);
    LH_CHECK_HIP(hipGetLastError());
  }
  return LANCE_HIP_OK;
}

int from_f32(lance_hip_ctx *ctx, int dtype, const float *src, void *dst, size_t count) {
  if (count == 0 || static_cast<const void *>(src) == dst) return LANCE_HIP_OK;
  LH_REQUIRE(dtype != LANCE_HIP_I8, "internal: int8 is never an output element type");
  if (dtype == LANCE_HIP_F32) {
    LH_CHECK_HIP(hipMemcpyAsync(dst, src, count * 4, hipMemcpyDeviceToDevice, ctx->stream));
  } else {
    const unsigned grid = (unsigned)std::min<uint64_t>(cdiv(count, 256), 65536);
    hipLaunchKernelGGL(narrow_f16_kernel, dim3(grid), dim3(256), 0, ctx->stream, src, static_cast<__half *>(dst), (int64_t)count);
    LH_CHECK_HIP(hipGetLastError());
  }
  return LANCE_HIP_OK;
}

__global__ __launch_bounds__(256) void fill_words_kernel(uint32_t *__restrict__ p, uint32_t v, size_t words) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < words) p[i] = v;
}
__global__ __launch_bounds__(256) void fill_bytes_kernel(uint8_t *__restrict__ p, uint8_t v, size_t bytes) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < bytes) p[i] = v;
}

hipError_t memset_async(void *ptr, int value, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return hipSuccess;
  const uint32_t b = (uint32_t)value & 255u;
  if ((reinterpret_cast<uintptr_t>(ptr) & 3) == 0 && (bytes & 3) == 0) {
    const size_t words = bytes / 4;
    hipLaunchKernelGGL(fill_words_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, stream, static_cast<uint32_t *>(ptr), b * 0x01010101u, words);
  } else {
    hipLaunchKernelGGL(fill_bytes_kernel, dim3((unsigned)((bytes + 255) / 256)), dim3(256), 0, stream, static_cast<uint8_t *>(ptr), (uint8_t)b, bytes);
  }
  return hipGetLastError();
}

// Several word-aligned fills in ONE launch (the search pipeline cleared four small arrays in a row with four kernels: each a node of the
// captured graph and ~8 us of the stream's timeline -- rocprofv3 counted ten fill launches per 10,000-query batch, gpurun r05i).
struct FillList { uint32_t *p[6]; uint32_t v[6]; size_t words[6]; };
__global__ __launch_bounds__(256) void fill_multi_kernel(FillList f) {
  const int k = blockIdx.y;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < f.words[k]) f.p[k][i] = f.v[k];
}
hipError_t memset_multi(hipStream_t stream, std::initializer_list<FillSpec> fills) {
  FillList f{};
  int n = 0;
  size_t maxw = 0;
  for (const FillSpec &s : fills) {
    if (s.bytes == 0) continue;
    if (n == 6 || (reinterpret_cast<uintptr_t>(s.ptr) & 3) != 0 || (s.bytes & 3) != 0) {      // not this kernel's shape: one by one
      for (const FillSpec &t : fills) { const hipError_t e = memset_async(t.ptr, t.value, t.bytes, stream); if (e != hipSuccess) return e; }
      return hipSuccess;
    }
    f.p[n] = static_cast<uint32_t *>(s.ptr); f.v[n] = ((uint32_t)s.value & 255u) * 0x01010101u; f.words[n] = s.bytes / 4;
    maxw = std::max(maxw, f.words[n]);
    ++n;
  }
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(fill_multi_kernel, dim3((unsigned)((maxw + 255) / 256), (unsigned)n), dim3(256), 0, stream, f);
  return hipGetLastError();
}

}  // namespace lh
