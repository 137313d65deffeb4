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
template <int D, int METRIC, bool BNEG = false>
__device__ __forceinline__ float dist_exact(const RegVec<D> &a, const float *__restrict__ b) {
  constexpr int FULL = D / 16 * 16;
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
    constexpr int NJ = FULL / 16;
#pragma unroll
    for (int ig = 0; ig < 4; ++ig) {
      // issue all of this lane-group's tile reads first (one LDS latency per group, not per read)
      f4 bv[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) bv[j] = *reinterpret_cast<const f4 *>(b + 16 * j + 4 * ig);
      f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const f4 av = a.q[4 * j + ig];
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

// Runtime-d version (both operands through pointers); same order.  Used by the
// generic-dimension fallbacks and by small host-order helpers.
template <int METRIC>
__device__ __forceinline__ float dist_exact_rt(const float *__restrict__ a, const float *__restrict__ b, int d) {
  const int full = d / 16 * 16;
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
  float sums[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) sums[i] = 0.0f;
  for (int c = 0; c < full; c += 16) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
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
  for (int i = 0; i < 16; ++i) tot = tot + sums[i];
  return s + tot;
}

// metric value as the reference scans see it: L2 -> squared L2; DOT -> 1 - dot.
template <int METRIC>
__device__ __forceinline__ float finish_metric(float raw) {
  if constexpr (METRIC == METRIC_DOT) return 1.0f - raw;
  return raw;
}

}  // namespace lh
