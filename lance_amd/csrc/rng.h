// rng.h -- the engine's seeded RNG (xoshiro256++ seeded through splitmix64) and the
// reservoir used for k-means initialisation.  The reference seeds SmallRng from the OS
// (kmeans.rs:181,646), so its stream is unpinned; this specification is shared with the
// test oracle so that seeded runs can be compared bit for bit.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define LH_HD __host__ __device__
#else
#define LH_HD
#endif

namespace lh {

struct Rng {
  uint64_t s[4];
  LH_HD static uint64_t splitmix(uint64_t &x) {
    uint64_t z = (x += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
  }
  LH_HD void seed(uint64_t v) {
    for (int i = 0; i < 4; ++i) s[i] = splitmix(v);
  }
  LH_HD static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
  LH_HD uint64_t next() {
    const uint64_t result = rotl(s[0] + s[3], 23) + s[0];
    const uint64_t t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t; s[3] = rotl(s[3], 45);
    return result;
  }
  // uniform in [0,1): top 24 bits
  LH_HD float next_f32() { return (float)(next() >> 40) * (1.0f / 16777216.0f); }
  // uniform integer in [0, n] inclusive, rejection on the masked top bits
  LH_HD uint64_t upto(uint64_t n) {
    const uint64_t range = n + 1;
    uint64_t mask = range - 1;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4;
    mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
    for (;;) {
      const uint64_t v = next() & mask;
      if (v < range) return v;
    }
  }
};

// (0..n).choose_multiple(rng, k): first k, then index i replaces slot j ~ U[0,i] if j < k
// (shape of rand's reservoir; kmeans_random_init, kmeans.rs:149-170)
inline void kmeans_init_indices(uint64_t n, uint32_t k, uint64_t seed, uint64_t *out) {
  Rng r;
  r.seed(seed);
  for (uint64_t i = 0; i < k; ++i) out[i] = i;
  for (uint64_t i = k; i < n; ++i) {
    const uint64_t j = r.upto(i);
    if (j < k) out[j] = i;
  }
}

}  // namespace lh
