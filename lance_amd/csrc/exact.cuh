// exact.cuh -- device-side distance arithmetic in the reference's summation order.
//
// These restate, for the GPU, the exact f32 operation order of
//   l2_scalar<T,f32,16>   rust/lance-linalg/src/distance/l2.rs:57-91
//   dot_scalar<T,f32,16>  rust/lance-linalg/src/distance/dot.rs:30-58
// so that every distance (and therefore every argmin id, PQ code, LUT entry and
// returned row id) is bit-identical to the reference CPU path:
//   remainder (len % 16) summed sequentially first into s; 16 lane accumulators
//   sums[i] += (x-y)^2 over the full 16-chunks; result = s + (((0+sums[0])+sums[1])+...).
// No FMA anywhere: the TU is compiled with -ffp-contract=off and the pragma below.
//
// Layout trick: lanes are processed 4 at a time (one float4 of the B operand per
// 16-chunk), which lets the compiler use packed v_pk_add_f32 / v_pk_mul_f32 -- the
// element-wise IEEE result is unchanged, only two lanes share an instruction.
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace lh {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

enum { METRIC_L2 = 0, METRIC_COSINE = 1, METRIC_DOT = 2 };

// f32::total_cmp order as an unsigned key (lance-index graph.rs:66-82 OrderedFloat).
__device__ __forceinline__ uint32_t order_key(float f) {
  uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_to_float(uint32_t k) {
  uint32_t b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(b);
}

// Per-lane operand held in VGPRs as float4 quads (so that packed f32 ops see naturally
// paired registers).  D need not be a multiple of 4: the last quad is zero-padded and
// never read beyond D.
template <int D>
struct RegVec {
  static constexpr int Q = (D + 3) / 4;
  f4 q[Q];
  __device__ __forceinline__ float get(int i) const { return q[i >> 2][i & 3]; }
};

// Distance between a per-lane vector `a` (registers, compile-time D) and `b`
// (any address space; wave-uniform when it points to an LDS tile -> broadcast reads).
// DOT returns the raw dot product (callers apply 1 - dot, dot.rs:68-70).
// BNEG: `b` holds the NEGATED operand (tiles are negated once when staged into LDS), so
// x - y is evaluated as x + (-y): bit-identical in IEEE-754, and it lets the backend emit
// v_pk_add_f32 instead of two scalar v_sub_f32 plus register shuffles.
// LANES = 16 for f32 / widened-f16 L2 and f32 dot; 32 for dot products of f16 columns (dot_scalar::<f16, f32, 32>,
// dot.rs:91-102,138-161): same shape, twice as many lane accumulators per chunk.
template <int D, int METRIC, bool BNEG = false, int LANES = 16>
__device__ __forceinline__ float dist_exact(const RegVec<D> &a, const float *__restrict__ b) {
  static_assert(LANES == 16 || LANES == 32, "lane accumulators: 16 or 32");
  constexpr int FULL = D / LANES * LANES;
  float s = 0.0f;
  if constexpr (FULL != D) {
    float acc = 0.0f;
    if constexpr ((D - FULL) % 4 == 0) {
      // whole quads: packed sub/mul, sequential adds (order = element order)
#pragma unroll
      for (int i = FULL; i < D; i += 4) {
        const f4 bv = *reinterpret_cast<const f4 *>(b + i);
        f4 t;
        if constexpr (METRIC == METRIC_DOT) {
          t = a.q[i >> 2] * bv;
        } else {
          const f4 diff = BNEG ? a.q[i >> 2] + bv : a.q[i >> 2] - bv;
          t = diff * diff;
        }
        acc = acc + t.x;
        acc = acc + t.y;
        acc = acc + t.z;
        acc = acc + t.w;
      }
    } else {
#pragma unroll
      for (int i = FULL; i < D; ++i) {
        if constexpr (METRIC == METRIC_DOT) {
          acc = acc + a.get(i) * b[i];
        } else {
          float diff = BNEG ? a.get(i) + b[i] : a.get(i) - b[i];
          acc = acc + diff * diff;
        }
      }
    }
    s = acc;
  }
  float tot = 0.0f;
  if constexpr (FULL > 0) {
    constexpr int NJ = FULL / LANES;
#pragma unroll
    for (int ig = 0; ig < LANES / 4; ++ig) {
      // issue all of this lane-group's tile reads first (one LDS latency per group, not per read)
      f4 bv[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) bv[j] = *reinterpret_cast<const f4 *>(b + LANES * j + 4 * ig);
      f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const f4 av = a.q[(LANES / 4) * j + ig];
        if constexpr (METRIC == METRIC_DOT) {
          acc += av * bv[j];
        } else {
          const f4 diff = BNEG ? av + bv[j] : av - bv[j];
          acc += diff * diff;
        }
      }
      tot = tot + acc.x;
      tot = tot + acc.y;
      tot = tot + acc.z;
      tot = tot + acc.w;
    }
  }
  return s + tot;
}

__device__ __forceinline__ float ld_elem(const float *p, int i) { return p[i]; }
__device__ __forceinline__ float ld_elem(const __half *p, int i) { return __half2float(p[i]); }  // `as_()` widening, exact
__device__ __forceinline__ float ld_elem(const int8_t *p, int i) { return (float)p[i]; }          // Int8 -> f32 (l2.rs:253-260)

// four consecutive elements of a row in the column's own element type, widened exactly (l2.rs:128-159 widens f16 per element,
// kmeans.rs:1216-1224 / l2.rs:253-260 convert Int8 columns to f32): no f32 copy of the column is ever materialised
__device__ __forceinline__ f4 load4(const float *p) { return *reinterpret_cast<const f4 *>(p); }
__device__ __forceinline__ f4 load4(const __half *p) {
  const uint2 u = *reinterpret_cast<const uint2 *>(p);
  const __half2 a = *reinterpret_cast<const __half2 *>(&u.x), b = *reinterpret_cast<const __half2 *>(&u.y);
  return f4{__low2float(a), __high2float(a), __low2float(b), __high2float(b)};
}
__device__ __forceinline__ f4 load4(const int8_t *p) {
  const uint32_t u = *reinterpret_cast<const uint32_t *>(p);
  return f4{(float)(int8_t)(u & 255u), (float)(int8_t)((u >> 8) & 255u), (float)(int8_t)((u >> 16) & 255u), (float)(int8_t)(u >> 24)};
}

// Runtime-d version (both operands through pointers); same order.  Used by the
// generic-dimension fallbacks and by small host-order helpers.  TB = float or __half
// (f16 elements are widened one by one, l2.rs:128-159).
template <int METRIC, typename TB = float, int LANES = 16>
__device__ __forceinline__ float dist_exact_rt(const float *__restrict__ a, const TB *__restrict__ bp, int d) {
  struct BView { const TB *p; __device__ __forceinline__ float operator[](int i) const { return ld_elem(p, i); } };
  const BView b{bp};
  const int full = d / LANES * LANES;
  float s = 0.0f;
  if (full != d) {
    float acc = 0.0f;
    for (int i = full; i < d; ++i) {
      if constexpr (METRIC == METRIC_DOT) {
        acc = acc + a[i] * b[i];
      } else {
        float diff = a[i] - b[i];
        acc = acc + diff * diff;
      }
    }
    s = acc;
  }
  float sums[LANES];
#pragma unroll
  for (int i = 0; i < LANES; ++i) sums[i] = 0.0f;
  for (int c = 0; c < full; c += LANES) {
#pragma unroll
    for (int i = 0; i < LANES; ++i) {
      if constexpr (METRIC == METRIC_DOT) {
        sums[i] += a[c + i] * b[c + i];
      } else {
        float diff = a[c + i] - b[c + i];
        sums[i] += diff * diff;
      }
    }
  }
  float tot = 0.0f;
#pragma unroll
  for (int i = 0; i < LANES; ++i) tot = tot + sums[i];
  return s + tot;
}

// ---- a3: flat cosine (cosine.rs:127-231, norm_l2.rs:106-129) ---------------------------
// The reference's f32 cosine uses explicit SIMD types: on the default x86_64 build
// (target-cpu=haswell) f32x16 is two __m256, multiply_add is vfmadd (simd/f32.rs:776-785) and
// reduce_sum is the permute/hadd tree ((a0+a4)+(a2+a6))+((a1+a5)+(a3+a7)) (simd/f32.rs:203-218,
// 625-644).  The fused multiply-adds are written with __fmaf_rn here on purpose.
template <typename TB = float, int LANES = 16>
__device__ __forceinline__ float norm_l2_rt(const TB *__restrict__ vp, int d) {
  struct VView { const TB *p; __device__ __forceinline__ float operator[](int i) const { return ld_elem(p, i); } };
  const VView v{vp};
  const int full = d / LANES * LANES;
  float s = 0.0f;
  if (full != d) {
    float acc = 0.0f;
    for (int i = full; i < d; ++i) acc = acc + v[i] * v[i];
    s = acc;
  }
  float sums[LANES];
#pragma unroll
  for (int i = 0; i < LANES; ++i) sums[i] = 0.0f;
  for (int c = 0; c < full; c += LANES)
#pragma unroll
    for (int i = 0; i < LANES; ++i) sums[i] += v[c + i] * v[c + i];
  float tot = 0.0f;
#pragma unroll
  for (int i = 0; i < LANES; ++i) tot = tot + sums[i];
  return sqrtf(s + tot);
}

// Cosine distance of f16 columns: the trait default `cosine_scalar` (cosine.rs:171-179, the arm `impl Cosine for f16` takes
// without the fp16kernels feature): xy = dot(x, y), y_sq = dot(y, y), both dot_scalar::<f16, f32, 32>;
// 1 - xy / (x_norm * sqrt(y_sq)) with x_norm = norm_l2_impl::<f16, f32, 32>(x) (norm_l2.rs:60-85).  Operands arrive widened.
template <typename TB = float>
__device__ __forceinline__ float cosine_scalar32_rt(const float *__restrict__ x, float x_norm, const TB *__restrict__ yp, int d) {
  struct YView { const TB *p; __device__ __forceinline__ float operator[](int i) const { return ld_elem(p, i); } };
  const YView y{yp};
  const int full = d / 32 * 32;
  float sxy = 0.0f, syy = 0.0f;
  if (full != d) {
    float axy = 0.0f, ayy = 0.0f;
    for (int i = full; i < d; ++i) { const float yv = y[i]; axy = axy + x[i] * yv; ayy = ayy + yv * yv; }
    sxy = axy; syy = ayy;
  }
  float pxy[32], pyy[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) { pxy[i] = 0.0f; pyy[i] = 0.0f; }
  for (int c = 0; c < full; c += 32)
#pragma unroll
    for (int i = 0; i < 32; ++i) { const float yv = y[c + i]; pxy[i] += x[c + i] * yv; pyy[i] += yv * yv; }
  float txy = 0.0f, tyy = 0.0f;
#pragma unroll
  for (int i = 0; i < 32; ++i) { txy = txy + pxy[i]; tyy = tyy + pyy[i]; }
  const float xy = sxy + txy, y_sq = syy + tyy;
  return 1.0f - xy / (x_norm * sqrtf(y_sq));
}

__device__ __forceinline__ float reduce8_tree(const float (&a)[8]) {
  const float s0 = a[0] + a[4], s1 = a[1] + a[5], s2 = a[2] + a[6], s3 = a[3] + a[7];
  return (s0 + s2) + (s1 + s3);
}

template <typename TB = float>
__device__ __forceinline__ float cosine_exact_rt(const float *__restrict__ x, float x_norm, const TB *__restrict__ yp, int d) {
  struct YView { const TB *p; __device__ __forceinline__ float operator[](int i) const { return ld_elem(p, i); } };
  const YView y{yp};
  if (d == 8 || d == 16) {  // cosine_once
    float t[8], u[8];
    if (d == 16) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        t[i] = x[i] * y[i] + x[i + 8] * y[i + 8];
        u[i] = y[i] * y[i] + y[i + 8] * y[i + 8];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) { t[i] = x[i] * y[i]; u[i] = y[i] * y[i]; }
    }
    return 1.0f - reduce8_tree(t) / x_norm / sqrtf(reduce8_tree(u));
  }
  const int unrolled = d / 16 * 16, aligned = d / 8 * 8;
  float xy16[16], yn16[16], xy8[8], yn8[8];
#pragma unroll
  for (int i = 0; i < 16; ++i) { xy16[i] = 0.0f; yn16[i] = 0.0f; }
#pragma unroll
  for (int i = 0; i < 8; ++i) { xy8[i] = 0.0f; yn8[i] = 0.0f; }
  for (int c = 0; c < unrolled; c += 16)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      xy16[i] = __fmaf_rn(x[c + i], y[c + i], xy16[i]);
      yn16[i] = __fmaf_rn(y[c + i], y[c + i], yn16[i]);
    }
  for (int c = unrolled; c < aligned; c += 8)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      xy8[i] = __fmaf_rn(x[c + i], y[c + i], xy8[i]);
      yn8[i] = __fmaf_rn(y[c + i], y[c + i], yn8[i]);
    }
  float t16[8], u16[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { t16[i] = xy16[i] + xy16[i + 8]; u16[i] = yn16[i] + yn16[i + 8]; }
  const float nrest = norm_l2_rt<TB>(yp + aligned, d - aligned);
  const float y_norm = reduce8_tree(u16) + reduce8_tree(yn8) + nrest * nrest;
  const float xy = reduce8_tree(t16) + reduce8_tree(xy8) + dist_exact_rt<METRIC_DOT, TB>(x + aligned, yp + aligned, d - aligned);
  return 1.0f - xy / x_norm / sqrtf(y_norm);
}

// metric value as the reference scans see it: L2 -> squared L2; DOT -> 1 - dot.
template <int METRIC>
__device__ __forceinline__ float finish_metric(float raw) {
  if constexpr (METRIC == METRIC_DOT) return 1.0f - raw;
  return raw;
}

}  // namespace lh
