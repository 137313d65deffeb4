// f16.h -- binary16 helpers.  The reference's f16 element type is half::f16 (half 2.7.1); on
// x86_64 its arithmetic is "convert to f32, operate, round back" (binary16/arch.rs fallbacks).
// The engine therefore carries f16 data in f32 containers (exact widening) and applies
// round_f16() at every point where the reference's value type is f16: M-step sums and scaling
// (kmeans.rs:380,405-418), split_clusters (:174-207), residuals (residual.rs:96, v2.rs:326).
#pragma once
#include <cstdint>
#include <cstring>

namespace lh {

inline float h2f_host(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1fu, man = h & 0x3ffu, bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else {
      int e = -1;
      do { man <<= 1; e++; } while ((man & 0x400u) == 0);
      man &= 0x3ffu;
      bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
    }
  } else if (exp == 31) {
    bits = sign | 0x7f800000u | (man << 13);
  } else {
    bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
  }
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

inline uint16_t f2h_host(float f) {  // round to nearest even
  uint32_t x;
  memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  const uint32_t exp = (x >> 23) & 0xffu;
  uint32_t man = x & 0x7fffffu;
  if (exp == 255) return (uint16_t)(sign | 0x7c00u | (man ? (0x200u | (man >> 13)) : 0));
  const int32_t e = (int32_t)exp - 127 + 15;
  if (e >= 31) return (uint16_t)(sign | 0x7c00u);
  if (e <= 0) {
    if (e < -10) return (uint16_t)sign;
    man |= 0x800000u;
    const uint32_t shift = (uint32_t)(14 - e);
    uint32_t hm = man >> shift;
    const uint32_t rem = man & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (hm & 1))) hm++;
    return (uint16_t)(sign | hm);
  }
  const uint32_t hm = man >> 13, rem = man & 0x1fffu;
  uint16_t h = (uint16_t)(sign | ((uint32_t)e << 10) | hm);
  if (rem > 0x1000u || (rem == 0x1000u && (hm & 1))) h++;
  return h;
}

inline float round_f16_host(float x) { return h2f_host(f2h_host(x)); }

}  // namespace lh
