/*
 * lance_oracle.c -- CPU restatement of the reference's IVF-PQ hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / the timed CPU baseline.  The product path
 * (lance_amd/, liblance_hip.so) never links or calls it.
 *
 * Every function restates, operation for operation, the arithmetic of the cited
 * reference source (paths relative to /root/reference).  Compile with
 *   gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp
 * (-ffp-contract=off matters: the reference is plain Rust, which never contracts
 * a*b+c into an fma; gcc would with -march=native).
 *
 * Parity pins: tests/test_oracle_golden.py transcribes the reference's own
 * known-answer tests (l2.rs:281-375,432-447; dot.rs; kernels.rs:278-300;
 * kmeans.rs:1398-1486; pq.rs:580-665; pq/distance.rs:337-364; pq/utils.rs:84-99;
 * simd/dist_table.rs:178-217) and checks orc_sum_4bit_dist_table against the reference's own C kernel
 * (rust/lance-linalg/src/simd/dist_table.c compiled from where it lies into oracle/_ref/ by oracle/Makefile), and the
 * f16 distances against simd/f16.c built the same way (bit-equal where f32 arithmetic is exact: the C file is -ffast-math).
 *
 * Pinned on outputs of the reference itself (tests/test_index_files.py, fixtures in tests/golden/ref_index.npz archived
 * from the reference's test_data/ by tests/golden/make_ref_index_fixtures.py -- index directories written by Lance
 * 0.21.0 / 0.27.1 together with the data they were built from): orc residual + PQ encode reproduce the stored PQ
 * codes byte for byte, the f64 sum of orc assign distances equals the recorded k-means loss to the bit, and orc k-means
 * over the rows the reference trained a single IVF centroid on reproduces that centroid bit for bit (M-step order);
 * on Lance 0.8.14's IVF4/PQ16 indices (d = 128) orc assign + encode reproduce the stored partition and all 16 code
 * bytes of each of 3000 rows.
 *
 * Not pinned by any reference test (reference is OS-seeded, kmeans.rs:181,646):
 * the RNG stream used for k-means initialisation and empty-cluster splitting.
 * The oracle and the product share the RNG specified below (xoshiro256++ seeded
 * by splitmix64); given the same seed both must produce bit-identical centroids.
 *
 * Third-party algorithms restated here (not vendored in /root/reference):
 *   - Rust std 1.90.0 (rust-toolchain.toml) alloc::collections::BinaryHeap
 *     push / pop (sift_up, sift_down_to_bottom) -- used by FlatIndex::search
 *     (rust/lance-index/src/vector/flat/index.rs:94-126); determines which of
 *     several equal-distance rows survive in a full heap.
 *   - rand 0.9 IteratorRandom::choose_multiple (reservoir) shape for
 *     kmeans_random_init (kmeans.rs:149-170); stream itself is unpinned.
 *   - half 2.7.1 (Cargo.lock) binary16 arithmetic, the non-intrinsic x86_64 arms: every operator, Float::powi and
 *     Float::sqrt convert to f32, operate, and round the result to binary16 (round-to-nearest-even); `impl Sum for f16`
 *     adds the widened terms in f32 and rounds once.  Used by the Float16Type k-means M-step (kmeans.rs:380,405-418),
 *     the f16 residual (residual.rs:96) and normalize_fsl::<Float16Type> (kernels.rs:141-186, orc_normalize_h).
 *     The f16 distance arms themselves (dot_scalar::<f16, f32, 32>, norm_l2_impl::<f16, f32, 32>, cosine_scalar) are in the
 *     reference tree and are pinned on its own simd/f16.c and on its property tests (tests/test_oracle_golden.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_L2 0
#define ORC_COSINE 1
#define ORC_DOT 2
/* The same metrics on a Float16 column (values carried in f32 containers).  half::f16 has its own arms of the distance traits
 * (without the optional fp16kernels feature): Dot = dot_scalar::<f16, f32, 32> (dot.rs:91-102,138-161), Normalize =
 * norm_l2_impl::<f16, f32, 32> (norm_l2.rs:60-85), Cosine = the trait default cosine_scalar (cosine.rs:36-45,171-179);
 * normalize_fsl::<Float16Type> runs in half-precision arithmetic (kernels.rs:141-186).  L2 stays l2_scalar::<f16, f32, 16>. */
#define ORC_DOT_H 3
#define ORC_COSINE_H 4
static inline int orc_is_dot(int m) { return m == ORC_DOT || m == ORC_DOT_H; }
static inline int orc_is_cos(int m) { return m == ORC_COSINE || m == ORC_COSINE_H; }
static inline int orc_metric_h(int metric, int f16) {
  if (f16 && metric == ORC_DOT) return ORC_DOT_H;
  if (f16 && metric == ORC_COSINE) return ORC_COSINE_H;
  return metric;
}
#define ORC_NONE 0xFFFFFFFFu

/* ------------------------------------------------------------------------- */
/* f16 <-> f32 (IEEE binary16, the `half` crate's f16::to_f32 is exact).       */
static inline float orc_h2f(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1fu;
  uint32_t man = h & 0x3ffu;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else { /* subnormal */
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

/* round-to-nearest-even f32 -> f16 (half crate f16::from_f32). */
static inline uint16_t orc_f2h(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t exp = (x >> 23) & 0xffu;
  uint32_t man = x & 0x7fffffu;
  if (exp == 255) return (uint16_t)(sign | 0x7c00u | (man ? (0x200u | (man >> 13)) : 0));
  int32_t e = (int32_t)exp - 127 + 15;
  if (e >= 31) return (uint16_t)(sign | 0x7c00u);
  if (e <= 0) {
    if (e < -10) return (uint16_t)sign;
    man |= 0x800000u;
    uint32_t shift = (uint32_t)(14 - e);
    uint32_t hm = man >> shift;
    uint32_t rem = man & ((1u << shift) - 1);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (hm & 1))) hm++;
    return (uint16_t)(sign | hm);
  }
  uint32_t hm = man >> 13;
  uint32_t rem = man & 0x1fffu;
  uint16_t h = (uint16_t)(sign | ((uint32_t)e << 10) | hm);
  if (rem > 0x1000u || (rem == 0x1000u && (hm & 1))) h++;
  return h;
}

/* f16 arithmetic as the `half` crate 2.7.1 does it on x86_64: operate in f32, round the result
 * to binary16 (binary16/arch.rs fallbacks).  f16 data is carried in f32 containers holding
 * f16-representable values; ORC_RH marks every point where the reference rounds to f16. */
static inline float orc_rh(float x) { return orc_h2f(orc_f2h(x)); }
#define ORC_RH(f16, x) ((f16) ? orc_rh(x) : (x))
float orc_round_f16(float x) { return orc_rh(x); }
uint16_t orc_f32_to_f16(float f) { return orc_f2h(f); }
float orc_f16_to_f32(uint16_t h) { return orc_h2f(h); }

/* ------------------------------------------------------------------------- */
/* a1: l2_scalar<T,f32,16>  rust/lance-linalg/src/distance/l2.rs:57-91,161-168
 * remainder (len % 16) summed sequentially first; 16 lane accumulators over full
 * chunks; result = s + (((0 + sums[0]) + sums[1]) + ... + sums[15]).           */
float orc_l2_f32(const float *x, const float *y, size_t d) {
  const size_t LANES = 16;
  size_t full = d / LANES * LANES;
  float s = 0.0f;
  if (full != d) {
    float acc = 0.0f; /* iter().sum::<f32>() */
    for (size_t i = full; i < d; i++) {
      float diff = x[i] - y[i];
      acc = acc + diff * diff;
    }
    s = acc;
  }
  float sums[16];
  for (size_t i = 0; i < LANES; i++) sums[i] = 0.0f;
  for (size_t c = 0; c < full; c += LANES)
    for (size_t i = 0; i < LANES; i++) {
      float diff = x[c + i] - y[c + i];
      sums[i] += diff * diff;
    }
  float tot = 0.0f;
  for (size_t i = 0; i < LANES; i++) tot = tot + sums[i];
  return s + tot;
}

/* f16 data: each element widened to f32 (`as_()`), l2_scalar<f16,f32,16>
 * l2.rs:128-159 (the non-fp16kernels fallback arm).                           */
float orc_l2_f16(const uint16_t *x, const uint16_t *y, size_t d) {
  const size_t LANES = 16;
  size_t full = d / LANES * LANES;
  float s = 0.0f;
  if (full != d) {
    float acc = 0.0f;
    for (size_t i = full; i < d; i++) {
      float diff = orc_h2f(x[i]) - orc_h2f(y[i]);
      acc = acc + diff * diff;
    }
    s = acc;
  }
  float sums[16];
  for (size_t i = 0; i < LANES; i++) sums[i] = 0.0f;
  for (size_t c = 0; c < full; c += LANES)
    for (size_t i = 0; i < LANES; i++) {
      float diff = orc_h2f(x[c + i]) - orc_h2f(y[c + i]);
      sums[i] += diff * diff;
    }
  float tot = 0.0f;
  for (size_t i = 0; i < LANES; i++) tot = tot + sums[i];
  return s + tot;
}

/* l2_distance_uint_scalar  l2.rs:44-49 */
float orc_l2_u8(const uint8_t *x, const uint8_t *y, size_t d) {
  uint32_t acc = 0;
  for (size_t i = 0; i < d; i++) {
    uint32_t a = x[i] > y[i] ? (uint32_t)(x[i] - y[i]) : (uint32_t)(y[i] - x[i]);
    acc += a * a;
  }
  return (float)acc;
}

/* a2: dot_scalar<f32,f32,16>  dot.rs:30-58; dot_distance = 1 - dot  dot.rs:68-70 */
float orc_dot_f32(const float *x, const float *y, size_t d) {
  const size_t LANES = 16;
  size_t full = d / LANES * LANES;
  float s = 0.0f;
  if (full != d) {
    float acc = 0.0f;
    for (size_t i = full; i < d; i++) acc = acc + x[i] * y[i];
    s = acc;
  }
  float sums[16];
  for (size_t i = 0; i < LANES; i++) sums[i] = 0.0f;
  for (size_t c = 0; c < full; c += LANES)
    for (size_t i = 0; i < LANES; i++) sums[i] += x[c + i] * y[c + i];
  float tot = 0.0f;
  for (size_t i = 0; i < LANES; i++) tot = tot + sums[i];
  return s + tot;
}

/* dot_scalar<f16,f32,32>  dot.rs:138-161 (32 lanes for f16) */
float orc_dot_f16(const uint16_t *x, const uint16_t *y, size_t d) {
  const size_t LANES = 32;
  size_t full = d / LANES * LANES;
  float s = 0.0f;
  if (full != d) {
    float acc = 0.0f;
    for (size_t i = full; i < d; i++) acc = acc + orc_h2f(x[i]) * orc_h2f(y[i]);
    s = acc;
  }
  float sums[32];
  for (size_t i = 0; i < LANES; i++) sums[i] = 0.0f;
  for (size_t c = 0; c < full; c += LANES)
    for (size_t i = 0; i < LANES; i++) sums[i] += orc_h2f(x[c + i]) * orc_h2f(y[c + i]);
  float tot = 0.0f;
  for (size_t i = 0; i < LANES; i++) tot = tot + sums[i];
  return s + tot;
}

float orc_dot_distance_f32(const float *x, const float *y, size_t d) {
  return 1.0f - orc_dot_f32(x, y, d);
}

/* norm_l2_impl<f32,f32,16>  norm_l2.rs:106-129 */
float orc_norm_l2_f32(const float *x, size_t d) {
  const size_t LANES = 16;
  size_t full = d / LANES * LANES;
  float s = 0.0f;
  if (full != d) {
    float acc = 0.0f;
    for (size_t i = full; i < d; i++) acc = acc + x[i] * x[i];
    s = acc;
  }
  float sums[16];
  for (size_t i = 0; i < LANES; i++) sums[i] = 0.0f;
  for (size_t c = 0; c < full; c += LANES)
    for (size_t i = 0; i < LANES; i++) sums[i] += x[c + i] * x[c + i];
  float tot = 0.0f;
  for (size_t i = 0; i < LANES; i++) tot = tot + sums[i];
  return sqrtf(s + tot);
}

/* f32x8::reduce_sum, x86_64 AVX2 arm  simd/f32.rs:203-218:
 * ((a0+a4)+(a2+a6)) + ((a1+a5)+(a3+a7))                                        */
static inline float orc_reduce8(const float *a) {
  float s0 = a[0] + a[4], s1 = a[1] + a[5], s2 = a[2] + a[6], s3 = a[3] + a[7];
  return (s0 + s2) + (s1 + s3);
}

/* a3: Cosine for f32::cosine_fast  cosine.rs:143-175 on the default x86_64 build
 * (target-cpu=haswell: f32x16 = 2 x __m256, multiply_add = vfmadd, reduce_sum
 * simd/f32.rs:625-644; .cargo/config.toml:12-13).  x_norm = norm_l2(x).
 * Used only by the flat cosine scan (un-indexed KNN / refine of a cosine index). */
float orc_cosine_f32(const float *x, float x_norm, const float *y, size_t d) {
  /* cosine_batch (cosine.rs:210-231): dimension 8 / 16 take cosine_once (plain products,
   * SIMD tree reduce, cosine.rs:127-140); everything else cosine_fast. */
  if (d == 8 || d == 16) {
    float xy[16], y2[16], t[8], u[8];
    for (size_t i = 0; i < d; i++) { xy[i] = x[i] * y[i]; y2[i] = y[i] * y[i]; }
    if (d == 16) {
      for (int i = 0; i < 8; i++) { t[i] = xy[i] + xy[i + 8]; u[i] = y2[i] + y2[i + 8]; }
    } else {
      for (int i = 0; i < 8; i++) { t[i] = xy[i]; u[i] = y2[i]; }
    }
    return 1.0f - orc_reduce8(t) / x_norm / sqrtf(orc_reduce8(u));
  }
  size_t unrolled = d / 16 * 16, aligned = d / 8 * 8;
  float xy16[16], yn16[16], xy8[8], yn8[8];
  for (int i = 0; i < 16; i++) { xy16[i] = 0.0f; yn16[i] = 0.0f; }
  for (int i = 0; i < 8; i++) { xy8[i] = 0.0f; yn8[i] = 0.0f; }
  for (size_t c = 0; c < unrolled; c += 16)
    for (int i = 0; i < 16; i++) {
      xy16[i] = fmaf(x[c + i], y[c + i], xy16[i]);
      yn16[i] = fmaf(y[c + i], y[c + i], yn16[i]);
    }
  for (size_t c = unrolled; c < aligned; c += 8)
    for (int i = 0; i < 8; i++) {
      xy8[i] = fmaf(x[c + i], y[c + i], xy8[i]);
      yn8[i] = fmaf(y[c + i], y[c + i], yn8[i]);
    }
  float t16[8], u16[8];
  for (int i = 0; i < 8; i++) { t16[i] = xy16[i] + xy16[i + 8]; u16[i] = yn16[i] + yn16[i + 8]; }
  float nrest = orc_norm_l2_f32(y + aligned, d - aligned);
  float y_norm = orc_reduce8(u16) + orc_reduce8(yn8) + nrest * nrest;
  float xy = orc_reduce8(t16) + orc_reduce8(xy8) + orc_dot_f32(x + aligned, y + aligned, d - aligned);
  return 1.0f - xy / x_norm / sqrtf(y_norm);
}

/* a4: normalize  kernels.rs:141-146 -- l2_norm = sqrt(sequential sum of x^2 in T) */
void orc_normalize_f32(const float *x, size_t n, size_t d, float *out) {
#pragma omp parallel for schedule(static) if (n * d >= 65536)
  for (size_t r = 0; r < n; r++) {
    const float *v = x + r * d;
    float acc = 0.0f;
    for (size_t i = 0; i < d; i++) acc = acc + v[i] * v[i];
    float norm = sqrtf(acc);
    for (size_t i = 0; i < d; i++) out[r * d + i] = v[i] / norm;
  }
}

/* dot_scalar::<f16, f32, 32> on widened values (dot.rs:30-58 with LANES = 32) */
float orc_dot32_f32(const float *x, const float *y, size_t d) {
  const size_t LANES = 32;
  size_t full = d / LANES * LANES;
  float s = 0.0f;
  if (full != d) {
    float acc = 0.0f;
    for (size_t i = full; i < d; i++) acc = acc + x[i] * y[i];
    s = acc;
  }
  float sums[32];
  for (size_t i = 0; i < LANES; i++) sums[i] = 0.0f;
  for (size_t c = 0; c < full; c += LANES)
    for (size_t i = 0; i < LANES; i++) sums[i] += x[c + i] * y[c + i];
  float tot = 0.0f;
  for (size_t i = 0; i < LANES; i++) tot = tot + sums[i];
  return s + tot;
}

/* norm_l2_impl::<f16, f32, 32>  norm_l2.rs:106-129 */
float orc_norm_l2_32_f32(const float *x, size_t d) {
  const size_t LANES = 32;
  size_t full = d / LANES * LANES;
  float s = 0.0f;
  if (full != d) {
    float acc = 0.0f;
    for (size_t i = full; i < d; i++) acc = acc + x[i] * x[i];
    s = acc;
  }
  float sums[32];
  for (size_t i = 0; i < LANES; i++) sums[i] = 0.0f;
  for (size_t c = 0; c < full; c += LANES)
    for (size_t i = 0; i < LANES; i++) sums[i] += x[c + i] * x[c + i];
  float tot = 0.0f;
  for (size_t i = 0; i < LANES; i++) tot = tot + sums[i];
  return sqrtf(s + tot);
}

/* cosine_scalar  cosine.rs:171-179: y_sq = dot(y, y); xy = dot(x, y); 1 - xy / (x_norm * sqrt(y_sq)) -- f16's cosine_fast */
float orc_cosine_scalar32_f32(const float *x, float x_norm, const float *y, size_t d) {
  float y_sq = orc_dot32_f32(y, y, d);
  float xy = orc_dot32_f32(x, y, d);
  return 1.0f - xy / (x_norm * sqrtf(y_sq));
}

/* normalize::<f16>  kernels.rs:141-146 in the `half` crate's arithmetic (2.7.1: every operator is the f32 operation rounded
 * to binary16; Float::powi / sqrt likewise; `impl Sum for f16` adds the widened terms in f32 and rounds once). */
void orc_normalize_h(const float *x, size_t n, size_t d, float *out) {
#pragma omp parallel for schedule(static) if (n * d >= 65536)
  for (size_t r = 0; r < n; r++) {
    const float *v = x + r * d;
    float acc = 0.0f;
    for (size_t i = 0; i < d; i++) acc = acc + orc_rh(v[i] * v[i]);
    float norm = orc_rh(sqrtf(orc_rh(acc)));
    for (size_t i = 0; i < d; i++) out[r * d + i] = orc_rh(v[i] / norm);
  }
}

/* utils.rs:263-286 is_finite: 1 if every element finite */
void orc_is_finite_f32(const float *x, size_t n, size_t d, uint8_t *out) {
  for (size_t r = 0; r < n; r++) {
    uint8_t ok = 1;
    for (size_t i = 0; i < d; i++)
      if (!isfinite(x[r * d + i])) { ok = 0; break; }
    out[r] = ok;
  }
}

static inline float orc_dist(int metric, const float *x, const float *y, size_t d) {
  if (metric == ORC_DOT_H) return 1.0f - orc_dot32_f32(x, y, d);
  return metric == ORC_DOT ? orc_dot_distance_f32(x, y, d) : orc_l2_f32(x, y, d);
}

/* DistanceType::func / arrow_batch_func  distance.rs:56-75 for the flat scan */
void orc_distance_batch_f32(int metric, const float *q, const float *x, size_t n, size_t d,
                            float *out) {
  float qn = metric == ORC_COSINE ? orc_norm_l2_f32(q, d) : (metric == ORC_COSINE_H ? orc_norm_l2_32_f32(q, d) : 0.0f);
  /* small batches (one partition of an IVF_FLAT test) stay on the calling thread: a fork/join per call costs more than the
   * loop, and on a CPU-quota'd container with many visible cores it costs milliseconds */
#pragma omp parallel for schedule(static) if (n * d >= 65536)
  for (size_t r = 0; r < n; r++)
    out[r] = metric == ORC_COSINE ? orc_cosine_f32(q, qn, x + r * d, d)
           : metric == ORC_COSINE_H ? orc_cosine_scalar32_f32(q, qn, x + r * d, d) : orc_dist(metric, q, x + r * d, d);
}

/* ------------------------------------------------------------------------- */
/* a5: argmin_value_float / argmin_value_float_with_bias  kernels.rs:79-111
 * strict '<' against +inf start: NaN and +inf are never selected; all such -> None.
 * Returns 1 if found.                                                          */
static inline int orc_argmin_row(int metric, const float *v, const float *cent, size_t k,
                                 size_t d, const float *bias, uint32_t *id, float *dist) {
  int found = 0;
  uint32_t min_idx = 0;
  float min_value = INFINITY, min_orig = INFINITY;
  if (bias == NULL) {
    for (size_t c = 0; c < k; c++) {
      float value = orc_dist(metric, v, cent + c * d, d);
      if (value < min_value) { min_value = value; min_idx = (uint32_t)c; found = 1; }
    }
    min_orig = min_value;
  } else {
    for (size_t c = 0; c < k; c++) {
      float value = orc_dist(metric, v, cent + c * d, d);
      float vb = value + bias[c];
      if (vb < min_value) { min_value = vb; min_orig = value; min_idx = (uint32_t)c; found = 1; }
    }
  }
  *id = found ? min_idx : ORC_NONE;
  *dist = min_orig;
  return found;
}

/* Distances of ONE vector to kp vectors stored transposed ([d][kp], kp a multiple of 16), SIMD lanes across the kp vectors: every
 * distance goes through exactly the operation sequence of orc_l2_f32 / orc_dot_f32 (tail elements first, 16 lane accumulators,
 * lanes added in order, tail + total; dot: 1 - that) -- sixteen of them advance together, the values are bit-identical to the
 * one-at-a-time functions (tests/test_oracle_golden.py pins that).  Used where the restatement's time goes: one table per
 * (query, probed partition) in the search, the E-step / assign of the builds.                                                  */
static void orc_dist_many_T(int is_dot, const float *x, size_t d, const float *T, size_t kp, float *out) {
  enum { VW = 16, LANES = 16 };
  const size_t full = d / LANES * LANES;
  for (size_t c0 = 0; c0 < kp; c0 += VW) {
    float s[VW], tot[VW], sums[LANES][VW];
    for (int v = 0; v < VW; v++) s[v] = 0.0f;
    if (full != d) {
      float acc[VW];
      for (int v = 0; v < VW; v++) acc[v] = 0.0f;
      for (size_t i = full; i < d; i++) {
        const float xi = x[i];
        const float *row = T + i * kp + c0;
        if (is_dot) {
#pragma omp simd
          for (int v = 0; v < VW; v++) acc[v] = acc[v] + xi * row[v];
        } else {
#pragma omp simd
          for (int v = 0; v < VW; v++) { const float diff = xi - row[v]; acc[v] = acc[v] + diff * diff; }
        }
      }
      for (int v = 0; v < VW; v++) s[v] = acc[v];
    }
    for (int i = 0; i < LANES; i++)
      for (int v = 0; v < VW; v++) sums[i][v] = 0.0f;
    for (size_t c = 0; c < full; c += LANES)
      for (int i = 0; i < LANES; i++) {
        const float xi = x[c + i];
        const float *row = T + (c + i) * kp + c0;
        if (is_dot) {
#pragma omp simd
          for (int v = 0; v < VW; v++) sums[i][v] += xi * row[v];
        } else {
#pragma omp simd
          for (int v = 0; v < VW; v++) { const float diff = xi - row[v]; sums[i][v] += diff * diff; }
        }
      }
    for (int v = 0; v < VW; v++) tot[v] = 0.0f;
    for (int i = 0; i < LANES; i++) {
#pragma omp simd
      for (int v = 0; v < VW; v++) tot[v] = tot[v] + sums[i][v];
    }
    if (is_dot) { for (int v = 0; v < VW; v++) out[c0 + v] = 1.0f - (s[v] + tot[v]); }
    else { for (int v = 0; v < VW; v++) out[c0 + v] = s[v] + tot[v]; }
  }
}

/* a6/a9: compute_membership_and_dist  kmeans.rs:317-369 ; compute_partition :1350
 * L2 / dot with enough work: the centroids transposed once ([d][k rounded up to 16], zero padding), every row's k distances from
 * orc_dist_many_T, then the same strict-'<' argmin over them (with the bias, if any) as orc_argmin_row.                          */
void orc_assign_f32(int metric, const float *x, size_t n, size_t d, const float *cent, size_t k,
                    const float *bias, uint32_t *ids, float *dists) {
  float *cT = NULL;
  const size_t kp = (k + 15) / 16 * 16;
  if ((metric == ORC_L2 || metric == ORC_DOT) && k >= 16 && n * k * d >= (size_t)1 << 20) {
    cT = (float *)calloc(d * kp, sizeof(float));
    if (cT)
      for (size_t c = 0; c < k; c++)
        for (size_t i = 0; i < d; i++) cT[i * kp + c] = cent[c * d + i];
  }
#pragma omp parallel
  {
    float *dv = cT ? (float *)malloc(kp * sizeof(float)) : NULL;
#pragma omp for schedule(static)
    for (size_t r = 0; r < n; r++) {
      uint32_t id; float dist;
      if (dv) {
        orc_dist_many_T(metric == ORC_DOT, x + r * d, d, cT, kp, dv);
        int found = 0; uint32_t min_idx = 0; float min_value = INFINITY, min_orig = INFINITY;
        for (size_t c = 0; c < k; c++) {
          const float value = dv[c], vb = bias ? value + bias[c] : value;
          if (vb < min_value) { min_value = vb; min_orig = value; min_idx = (uint32_t)c; found = 1; }
        }
        id = found ? min_idx : ORC_NONE; dist = min_orig;
      } else {
        orc_argmin_row(metric, x + r * d, cent, k, d, bias, &id, &dist);
      }
      ids[r] = id;
      if (dists) dists[r] = dist;
    }
    free(dv);
  }
  free(cT);
}

/* f16 data + f16 centroids (KMeansAlgoFloat<Float16Type>) */
void orc_assign_f16(int metric, const uint16_t *x, size_t n, size_t d, const uint16_t *cent,
                    size_t k, uint32_t *ids, float *dists) {
  /* L2: orc_l2_f16 is orc_l2_f32 on the widened elements (same 16 lanes, same order) -- widen once and take the f32 route, whose
   * SIMD-across-centroids form is bit-identical to the one-pair-at-a-time form (tests/test_oracle_golden.py) */
  if (!orc_is_dot(metric) && !orc_is_cos(metric) && k >= 16 && n * k * d >= (size_t)1 << 20) {
    float *xf = (float *)malloc(n * d * sizeof(float)), *cf = (float *)malloc(k * d * sizeof(float));
    if (xf && cf) {
#pragma omp parallel for schedule(static)
      for (size_t i = 0; i < n * d; i++) xf[i] = orc_h2f(x[i]);
      for (size_t i = 0; i < k * d; i++) cf[i] = orc_h2f(cent[i]);
      orc_assign_f32(ORC_L2, xf, n, d, cf, k, NULL, ids, dists);
      free(xf); free(cf);
      return;
    }
    free(xf); free(cf);
  }
#pragma omp parallel for schedule(static)
  for (size_t r = 0; r < n; r++) {
    int found = 0; uint32_t mi = 0; float mv = INFINITY;
    for (size_t c = 0; c < k; c++) {
      float value = orc_is_dot(metric) ? 1.0f - orc_dot_f16(x + r * d, cent + c * d, d)
                                      : orc_l2_f16(x + r * d, cent + c * d, d);
      if (value < mv) { mv = value; mi = (uint32_t)c; found = 1; }
    }
    ids[r] = found ? mi : ORC_NONE;
    if (dists) dists[r] = mv;
  }
}

/* ------------------------------------------------------------------------- */
/* RNG shared by oracle and product: xoshiro256++ seeded with splitmix64.
 * (The reference seeds SmallRng from the OS: kmeans.rs:181,646 -- unpinned.)   */
typedef struct { uint64_t s[4]; } orc_rng;
static inline uint64_t orc_splitmix(uint64_t *x) {
  uint64_t z = (*x += 0x9e3779b97f4a7c15ULL);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  return z ^ (z >> 31);
}
static inline void orc_rng_seed(orc_rng *r, uint64_t seed) {
  for (int i = 0; i < 4; i++) r->s[i] = orc_splitmix(&seed);
}
static inline uint64_t orc_rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static inline uint64_t orc_rng_next(orc_rng *r) {
  uint64_t *s = r->s;
  uint64_t result = orc_rotl(s[0] + s[3], 23) + s[0];
  uint64_t t = s[1] << 17;
  s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
  s[2] ^= t; s[3] = orc_rotl(s[3], 45);
  return result;
}
/* uniform f32 in [0,1): top 24 bits */
static inline float orc_rng_f32(orc_rng *r) {
  return (float)(orc_rng_next(r) >> 40) * (1.0f / 16777216.0f);
}
/* uniform integer in [0, n] inclusive (n < 2^63), by rejection on the top bits */
static inline uint64_t orc_rng_upto(orc_rng *r, uint64_t n) {
  uint64_t range = n + 1;
  uint64_t mask = range - 1;
  mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4;
  mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
  for (;;) {
    uint64_t v = orc_rng_next(r) & mask;
    if (v < range) return v;
  }
}

/* kmeans_random_init  kmeans.rs:149-170: (0..n).choose_multiple(rng, k) --
 * reservoir: first k indices, then for i >= k draw j in [0,i], replace if j < k. */
void orc_kmeans_init_indices(uint64_t n, uint32_t k, uint64_t seed, uint64_t *out) {
  orc_rng r; orc_rng_seed(&r, seed);
  for (uint64_t i = 0; i < k; i++) out[i] = i;
  for (uint64_t i = k; i < n; i++) {
    uint64_t j = orc_rng_upto(&r, i);
    if (j < k) out[j] = i;
  }
}

/* split_clusters  kmeans.rs:174-207 (f32) */
static void orc_split_clusters(size_t n, uint64_t *cnts, size_t k, float *centroids, size_t dim,
                               orc_rng *rng, int f16) {
  const float eps = 1.0f / 1024.0f;
  for (size_t i = 0; i < k; i++) {
    if (cnts[i] == 0) {
      /* the reference's rejection loop never ends when no cluster has >= 2 members (every p <= 0, e.g. all
       * distances NaN after an f16 overflow): stop splitting there instead of hanging -- the only deviation */
      int splittable = 0;
      for (size_t c = 0; c < k; c++) if (cnts[c] >= 2) { splittable = 1; break; }
      if (!splittable) return;
      size_t j = 0;
      for (;;) {
        float p = ((float)cnts[j] - 1.0f) / (float)(n - k);
        if (orc_rng_f32(rng) < p) break;
        j += 1;
        j %= k;
      }
      cnts[i] = cnts[j] / 2;
      cnts[j] -= cnts[i];
      for (size_t t = 0; t < dim; t++) {
        if (t % 2 == 0) {
          centroids[i * dim + t] = ORC_RH(f16, centroids[j * dim + t] * (1.0f + eps));
          centroids[j * dim + t] = ORC_RH(f16, centroids[j * dim + t] * (1.0f - eps));
        } else {
          centroids[i * dim + t] = ORC_RH(f16, centroids[j * dim + t] * (1.0f - eps));
          centroids[j * dim + t] = ORC_RH(f16, centroids[j * dim + t] * (1.0f + eps));
        }
      }
    }
  }
}

/* a7/a8: KMeans::train_kmeans  kmeans.rs:610-719 with
 *   compute_membership_and_loss :250-281, compute_cluster_sizes :210-232,
 *   compute_balance_loss :234-237, to_kmeans :371-446, split_clusters :174-207.
 * `balance_factor` is the already-scaled value (train_kmeans :1344 divides by n).
 * The caller applies the k*512 / sample_rate*k row caps (:623-627, :1328-1340).
 * init_centroids NULL -> random init from `seed`; split RNG = seed ^ 0x5bd1e995.
 * Returns the number of iterations executed.                                   */
int orc_kmeans_train_x(int metric, const float *x, size_t n, size_t d, size_t k,
                       uint32_t max_iters, double tol, float balance_factor,
                       const float *init_centroids, uint64_t seed, float *centroids_out,
                       double *loss_out, uint64_t *sizes_out, int f16);
int orc_kmeans_train_f32(int metric, const float *x, size_t n, size_t d, size_t k,
                         uint32_t max_iters, double tol, float balance_factor,
                         const float *init_centroids, uint64_t seed, float *centroids_out,
                         double *loss_out, uint64_t *sizes_out) {
  return orc_kmeans_train_x(metric, x, n, d, k, max_iters, tol, balance_factor, init_centroids, seed, centroids_out,
                            loss_out, sizes_out, 0);
}
/* f16 != 0: T = half::f16 (KMeansAlgoFloat<Float16Type>): x / centroids are f16 values in f32
 * containers; the M-step accumulates, scales and splits in f16 arithmetic (kmeans.rs:380,405-418). */
int orc_kmeans_train_x(int metric, const float *x, size_t n, size_t d, size_t k,
                       uint32_t max_iters, double tol, float balance_factor,
                       const float *init_centroids, uint64_t seed, float *centroids_out,
                       double *loss_out, uint64_t *sizes_out, int f16) {
  metric = orc_metric_h(metric, f16);     /* Float16Type + dot: compute_partitions calls f16's 32-lane dot */
  float *cent = centroids_out;
  if (init_centroids) {
    memcpy(cent, init_centroids, k * d * sizeof(float));
  } else {
    uint64_t *idx = (uint64_t *)malloc(k * sizeof(uint64_t));
    orc_kmeans_init_indices(n, (uint32_t)k, seed, idx);
    for (size_t c = 0; c < k; c++) memcpy(cent + c * d, x + idx[c] * d, d * sizeof(float));
    free(idx);
  }
  orc_rng split_rng; orc_rng_seed(&split_rng, seed ^ 0x5bd1e995ULL);

  uint32_t *membership = (uint32_t *)malloc(n * sizeof(uint32_t));
  float *dists = (float *)malloc(n * sizeof(float));
  uint64_t *cluster_sizes = (uint64_t *)calloc(k, sizeof(uint64_t));
  float *bias = (float *)malloc(k * sizeof(float));
  float *radius = (float *)malloc(k * sizeof(float));
  double *losses = (double *)malloc(k * sizeof(double));
  float *newc = (float *)malloc(k * d * sizeof(float));
  float adjusted_balance_factor = FLT_MAX;
  double loss = DBL_MAX, last_loss = DBL_MAX;
  int iters = 0;

  for (uint32_t it = 1; it <= max_iters; it++) {
    iters = (int)it;
    float bf = adjusted_balance_factor < balance_factor ? adjusted_balance_factor : balance_factor;
    /* f32::min: if either is NaN returns the other; not reachable with finite inputs */
    for (size_t c = 0; c < k; c++) bias[c] = bf * (float)cluster_sizes[c];
    orc_assign_f32(metric, x, n, d, cent, k, bias, membership, dists);

    for (size_t c = 0; c < k; c++) { radius[c] = 0.0f; losses[c] = 0.0; }
    for (size_t r = 0; r < n; r++) {
      if (membership[r] != ORC_NONE) {
        size_t c = membership[r];
        radius[c] = fmaxf(radius[c], dists[r]); /* f32::max */
        losses[c] += (double)dists[r];
      }
    }
    /* compute_cluster_sizes */
    for (size_t c = 0; c < k; c++) cluster_sizes[c] = 0;
    size_t max_cluster_id = 0; uint64_t max_cluster_size = 0;
    for (size_t r = 0; r < n; r++) {
      if (membership[r] != ORC_NONE) {
        size_t c = membership[r];
        cluster_sizes[c] += 1;
        if (cluster_sizes[c] > max_cluster_size) { max_cluster_size = cluster_sizes[c]; max_cluster_id = c; }
      }
    }
    adjusted_balance_factor =
        (radius[max_cluster_id] - (float)losses[max_cluster_id] / (float)cluster_sizes[max_cluster_id]) /
        (float)n;
    /* compute_balance_loss: usize sums, then as f32 */
    uint64_t size_loss_u = 0;
    for (size_t c = 0; c < k; c++) size_loss_u += cluster_sizes[c] * cluster_sizes[c];
    float size_loss = (float)size_loss_u;
    float balance_loss = bf * (size_loss - (float)((uint64_t)n * (uint64_t)n) / (float)k);
    double lsum = 0.0;
    for (size_t c = 0; c < k; c++) lsum = lsum + losses[c];
    last_loss = lsum + (double)balance_loss;

    /* to_kmeans: per-centroid sequential sum in row order, in T; then *= 1/cnt */
    memset(newc, 0, k * d * sizeof(float));
    for (size_t r = 0; r < n; r++) {
      if (membership[r] != ORC_NONE) {
        float *c = newc + (size_t)membership[r] * d;
        const float *v = x + r * d;
        for (size_t t = 0; t < d; t++) c[t] = ORC_RH(f16, c[t] + v[t]);
      }
    }
    for (size_t c = 0; c < k; c++) {
      if (cluster_sizes[c] > 0) {
        /* T::one() / T::from_usize(cnt) */
        float norm = ORC_RH(f16, 1.0f / ORC_RH(f16, (float)cluster_sizes[c]));
        for (size_t t = 0; t < d; t++) newc[c * d + t] = ORC_RH(f16, newc[c * d + t] * norm);
      }
    }
    orc_split_clusters(n, cluster_sizes, k, newc, d, &split_rng, f16);
    memcpy(cent, newc, k * d * sizeof(float));

    if (fabs(loss - last_loss) < tol * last_loss) break;
    loss = last_loss;
  }
  if (loss_out) *loss_out = last_loss;
  if (sizes_out) for (size_t c = 0; c < k; c++) sizes_out[c] = cluster_sizes[c];
  free(membership); free(dists); free(cluster_sizes); free(bias); free(radius); free(losses); free(newc);
  return iters;
}


/* ------------------------------------------------------------------------- */
/* a8: KMeans::train_hierarchical_kmeans  kmeans.rs:746-1003 (k > 256, hierarchical_k > 1;
 * dispatcher new_with_params :1008-1073).  Start with hierarchical_k clusters, then repeatedly
 * pop the largest non-finalized cluster from a std BinaryHeap (ordered by (not finalized, size))
 * and split it with a sub-k-means; final centroids ordered by creation id; loss = 0.
 * Each inner train_kmeans applies the k*512 row cap (:623-627).  Seeds: run r uses seed + r.   */
typedef struct { size_t id; uint32_t *idx; size_t n; float *centroid; int finalized; } orc_cluster;
static inline int orc_cluster_le(const orc_cluster *a, const orc_cluster *b) {
  /* a <= b under Ord: non-finalized > finalized, then size */
  int ka = a->finalized ? 0 : 1, kb = b->finalized ? 0 : 1;
  if (ka != kb) return ka < kb;
  return a->n <= b->n;
}
typedef struct { orc_cluster *d; size_t len; } orc_cheap;
static void orc_cheap_sift_up(orc_cheap *h, size_t start, size_t pos) {
  orc_cluster e = h->d[pos];
  while (pos > start) {
    size_t parent = (pos - 1) / 2;
    if (orc_cluster_le(&e, &h->d[parent])) break;
    h->d[pos] = h->d[parent];
    pos = parent;
  }
  h->d[pos] = e;
}
static void orc_cheap_push(orc_cheap *h, orc_cluster c) { h->d[h->len] = c; orc_cheap_sift_up(h, 0, h->len); h->len++; }
static orc_cluster orc_cheap_pop(orc_cheap *h) {
  orc_cluster item = h->d[--h->len];
  if (h->len > 0) {
    orc_cluster t = h->d[0]; h->d[0] = item; item = t;
    size_t end = h->len, pos = 0, child = 1;
    orc_cluster e = h->d[0];
    while (end >= 2 && child <= end - 2) {
      if (orc_cluster_le(&h->d[child], &h->d[child + 1])) child += 1;
      h->d[pos] = h->d[child]; pos = child; child = 2 * pos + 1;
    }
    if (child == end - 1) { h->d[pos] = h->d[child]; pos = child; }
    h->d[pos] = e;
    orc_cheap_sift_up(h, 0, pos);
  }
  return item;
}

static int orc_kmeans_capped(int metric, const float *x, size_t n, size_t d, size_t k, uint32_t max_iters, double tol,
                             float bf, uint64_t seed, float *cent, int f16) {
  size_t rows = n >= k * 512 ? k * 512 : n;
  double loss;
  return orc_kmeans_train_x(metric, x, rows, d, k, max_iters, tol, bf, NULL, seed, cent, &loss, NULL, f16);
}
static int orc_cluster_by_id(const void *a, const void *b) {
  size_t ia = ((const orc_cluster *)a)->id, ib = ((const orc_cluster *)b)->id;
  return ia < ib ? -1 : (ia > ib ? 1 : 0);
}

/* One split of the hierarchical trainer on its own (kmeans.rs:866-905: the k-means over a cluster's rows, then compute_partitions of
 * those rows) -- the unit of work the multi-GPU trainer hands to a rank (lance_amd/dist.py train_kmeans_hierarchical_sharded); the same
 * two calls as inside orc_kmeans_train_hierarchical_x below.  rows == NULL: all n rows (the first level).  TEST INFRASTRUCTURE. */
void orc_kmeans_split_x(int metric, const float *x, size_t n, size_t d, const uint32_t *rows, size_t n_rows, size_t k, uint32_t max_iters,
                        double tol, float balance_factor_scaled, uint64_t seed, float *cent_out, uint32_t *mem_out, int f16) {
  const int amet = orc_metric_h(metric, f16);
  const float *sub = x;
  float *tmp = NULL;
  if (rows) {
    tmp = (float *)malloc(n_rows * d * sizeof(float));
    for (size_t r = 0; r < n_rows; r++) memcpy(tmp + r * d, x + (size_t)rows[r] * d, d * sizeof(float));
    sub = tmp;
  } else {
    n_rows = n;
  }
  orc_kmeans_capped(metric, sub, n_rows, d, k, max_iters, tol, balance_factor_scaled, seed, cent_out, f16);
  orc_assign_f32(amet, sub, n_rows, d, cent_out, k, NULL, mem_out, NULL);
  free(tmp);
}

/* returns the number of clusters produced (== target_k unless splitting stalls) */
/* f16 != 0: train_hierarchical_kmeans::<Float16Type, KMeansAlgoFloat<Float16Type>> (kmeans.rs:1030-1033): every inner k-means
 * runs the f16 M-step, the membership pass is compute_partitions on the f16 values (widened per element; dot: 32 lanes)       */
size_t orc_kmeans_train_hierarchical_x(int metric, const float *x, size_t n, size_t d, size_t target_k,
                                       uint32_t max_iters, double tol, float balance_factor_scaled,
                                       size_t hierarchical_k, uint64_t seed, float *centroids_out, int f16);
size_t orc_kmeans_train_hierarchical_f32(int metric, const float *x, size_t n, size_t d, size_t target_k,
                                         uint32_t max_iters, double tol, float balance_factor_scaled,
                                         size_t hierarchical_k, uint64_t seed, float *centroids_out) {
  return orc_kmeans_train_hierarchical_x(metric, x, n, d, target_k, max_iters, tol, balance_factor_scaled, hierarchical_k, seed,
                                         centroids_out, 0);
}
size_t orc_kmeans_train_hierarchical_x(int metric, const float *x, size_t n, size_t d, size_t target_k,
                                       uint32_t max_iters, double tol, float balance_factor_scaled,
                                       size_t hierarchical_k, uint64_t seed, float *centroids_out, int f16) {
  const int amet = orc_metric_h(metric, f16);
  uint64_t run = 0;
  size_t initial_k = hierarchical_k < target_k ? hierarchical_k : target_k;
  if (initial_k > n) initial_k = n;
  float *c0 = (float *)malloc(initial_k * d * sizeof(float));
  orc_kmeans_capped(metric, x, n, d, initial_k, max_iters, tol, balance_factor_scaled, seed + run++, c0, f16);
  uint32_t *mem = (uint32_t *)malloc(n * sizeof(uint32_t));
  orc_assign_f32(amet, x, n, d, c0, initial_k, NULL, mem, NULL);
  orc_cheap heap; heap.d = (orc_cluster *)malloc((target_k + hierarchical_k + 2) * sizeof(orc_cluster)); heap.len = 0;
  size_t next_id = 0;
  for (size_t i = 0; i < initial_k; i++) {
    size_t cnt = 0;
    for (size_t r = 0; r < n; r++) cnt += mem[r] == (uint32_t)i;
    if (!cnt) continue;
    orc_cluster c; c.id = next_id++; c.n = cnt; c.finalized = 0;
    c.idx = (uint32_t *)malloc(cnt * sizeof(uint32_t));
    size_t w = 0;
    for (size_t r = 0; r < n; r++) if (mem[r] == (uint32_t)i) c.idx[w++] = (uint32_t)r;
    c.centroid = (float *)malloc(d * sizeof(float));
    memcpy(c.centroid, c0 + i * d, d * sizeof(float));
    orc_cheap_push(&heap, c);
  }
  free(c0); free(mem);
  while (heap.len < target_k) {
    if (heap.len == 0) break;
    orc_cluster big = orc_cheap_pop(&heap);
    if (big.finalized || big.n <= 1) { orc_cheap_push(&heap, big); break; }
    size_t remaining_k = target_k - heap.len;
    size_t cluster_k;
    if (big.n <= hierarchical_k) {
      cluster_k = 2; if (remaining_k < cluster_k) cluster_k = remaining_k; if (big.n < cluster_k) cluster_k = big.n;
    } else {
      cluster_k = big.n / hierarchical_k;
      if (remaining_k < cluster_k) cluster_k = remaining_k;
      if (hierarchical_k < cluster_k) cluster_k = hierarchical_k;
      if (cluster_k < 2) cluster_k = 2;
    }
    float *sub = (float *)malloc(big.n * d * sizeof(float));
    for (size_t r = 0; r < big.n; r++) memcpy(sub + r * d, x + (size_t)big.idx[r] * d, d * sizeof(float));
    float *sc = (float *)malloc(cluster_k * d * sizeof(float));
    orc_kmeans_capped(metric, sub, big.n, d, cluster_k, max_iters, tol, balance_factor_scaled, seed + run++, sc, f16);
    uint32_t *sm = (uint32_t *)malloc(big.n * sizeof(uint32_t));
    orc_assign_f32(amet, sub, big.n, d, sc, cluster_k, NULL, sm, NULL);
    int all_same = 1, have_first = 0; uint32_t first = 0;
    for (size_t r = 0; r < big.n; r++) {
      if (sm[r] == ORC_NONE) continue;
      if (have_first) { if (sm[r] != first) all_same = 0; } else { first = sm[r]; have_first = 1; }
    }
    if (all_same) {
      big.finalized = 1;
      orc_cheap_push(&heap, big);
    } else {
      for (size_t i = 0; i < cluster_k; i++) {
        size_t cnt = 0;
        for (size_t r = 0; r < big.n; r++) cnt += sm[r] == (uint32_t)i;
        if (!cnt) continue;
        orc_cluster c; c.id = next_id++; c.n = cnt; c.finalized = 0;
        c.idx = (uint32_t *)malloc(cnt * sizeof(uint32_t));
        size_t w = 0;
        for (size_t r = 0; r < big.n; r++) if (sm[r] == (uint32_t)i) c.idx[w++] = big.idx[r];
        c.centroid = (float *)malloc(d * sizeof(float));
        memcpy(c.centroid, sc + i * d, d * sizeof(float));
        orc_cheap_push(&heap, c);
      }
      free(big.idx); free(big.centroid);
    }
    free(sub); free(sc); free(sm);
  }
  /* sort by id, emit */
  size_t outn = heap.len;
  qsort(heap.d, outn, sizeof(orc_cluster), orc_cluster_by_id);     /* ids are distinct: any sort gives the same order */
  for (size_t i = 0; i < outn; i++) {
    memcpy(centroids_out + i * d, heap.d[i].centroid, d * sizeof(float));
    free(heap.d[i].idx); free(heap.d[i].centroid);
  }
  free(heap.d);
  return outn;
}

/* ------------------------------------------------------------------------- */
/* a10: do_compute_residual  residual.rs:58-102 */
void orc_residual_x(const float *x, size_t n, size_t d, const float *cent,
                    const uint32_t *part_ids, float *out, int f16) {
#pragma omp parallel for schedule(static)
  for (size_t r = 0; r < n; r++) {
    const float *c = cent + (size_t)part_ids[r] * d;
    for (size_t t = 0; t < d; t++) out[r * d + t] = ORC_RH(f16, x[r * d + t] - c[t]);
  }
}
void orc_residual_f32(const float *x, size_t n, size_t d, const float *cent,
                      const uint32_t *part_ids, float *out) {
  orc_residual_x(x, n, d, cent, part_ids, out, 0);
}

/* divide_to_subvectors  pq/utils.rs:14-49: sub-matrix m = columns [m*sd,(m+1)*sd) */
void orc_divide_to_subvectors_f32(const float *x, size_t n, size_t d, size_t m_count, float *out) {
  size_t sd = d / m_count;
  for (size_t m = 0; m < m_count; m++)
    for (size_t r = 0; r < n; r++)
      memcpy(out + (m * n + r) * sd, x + r * d + m * sd, sd * sizeof(float));
}

/* a11: PQBuildParams::build_from_fsl  pq/builder.rs:89-157: M sequential k-means
 * (k=2^nbits, L2, balance 0) on sub-vector matrices; codebook laid out [M][k][sd].
 * Sub-quantiser m uses seed + m.  Inner train_kmeans applies the sample_rate*k
 * slice (kmeans.rs:1328-1340) and the k*512 cap (:623-627).                    */
void orc_pq_train_x(const float *resid, size_t n, size_t d, size_t m_count, uint32_t nbits,
                    uint32_t max_iters, size_t sample_rate, uint64_t seed, float *codebook_out,
                    int *iters_out, int f16);
void orc_pq_train_f32(const float *resid, size_t n, size_t d, size_t m_count, uint32_t nbits,
                      uint32_t max_iters, size_t sample_rate, uint64_t seed, float *codebook_out,
                      int *iters_out) {
  orc_pq_train_x(resid, n, d, m_count, nbits, max_iters, sample_rate, seed, codebook_out, iters_out, 0);
}
void orc_pq_train_x(const float *resid, size_t n, size_t d, size_t m_count, uint32_t nbits,
                    uint32_t max_iters, size_t sample_rate, uint64_t seed, float *codebook_out,
                    int *iters_out, int f16) {
  size_t sd = d / m_count, kc = (size_t)1 << nbits;
  size_t rows = n;
  if (rows > sample_rate * kc) rows = sample_rate * kc;
  if (rows >= kc * 512) rows = kc * 512;
  float *sub = (float *)malloc(rows * sd * sizeof(float));
  for (size_t m = 0; m < m_count; m++) {
    for (size_t r = 0; r < rows; r++) memcpy(sub + r * sd, resid + r * d + m * sd, sd * sizeof(float));
    double loss;
    int it = orc_kmeans_train_x(ORC_L2, sub, rows, sd, kc, max_iters, 1e-4, 0.0f, NULL, seed + m,
                                codebook_out + m * kc * sd, &loss, NULL, f16);
    if (iters_out) iters_out[m] = it;
  }
  free(sub);
}

void orc_transpose_codebook_f32(const float *codebook, size_t d, size_t m_count, float *cbT);
/* a12: ProductQuantizer::transform_impl<8>  pq.rs:116-191: per sub-vector argmin
 * (compute_partition, no bias), unwrap_or(0).  codes row-major [n][M].        */
void orc_pq_encode_f32(int metric, const float *x, size_t n, size_t d, const float *codebook,
                       size_t m_count, uint32_t nbits, uint8_t *codes) {
  size_t sd = d / m_count, kc = (size_t)1 << nbits;
  /* 8-bit codes under L2 / dot: the 256 distances of a sub-vector from orc_dist_many_T (bit-identical), same strict-'<' argmin */
  float *cbT = NULL;
  if (nbits == 8 && (metric == ORC_L2 || metric == ORC_DOT) && n * d >= (size_t)1 << 12) {
    cbT = (float *)malloc(m_count * 256 * sd * sizeof(float));
    if (cbT) orc_transpose_codebook_f32(codebook, d, m_count, cbT);
  }
#pragma omp parallel for schedule(static)
  for (size_t r = 0; r < n; r++) {
    float dv[256];
    for (size_t m = 0; m < m_count; m++) {
      if (cbT) {
        orc_dist_many_T(metric == ORC_DOT, x + r * d + m * sd, sd, cbT + m * sd * 256, 256, dv);
        int found = 0; uint32_t mi = 0; float mv = INFINITY;
        for (size_t c = 0; c < 256; c++)
          if (dv[c] < mv) { mv = dv[c]; mi = (uint32_t)c; found = 1; }
        codes[r * m_count + m] = found ? (uint8_t)mi : 0;
        continue;
      }
      uint32_t id; float dist;
      int found = orc_argmin_row(metric, x + r * d + m * sd, codebook + m * kc * sd, kc, sd, NULL, &id, &dist);
      codes[r * m_count + m] = found ? (uint8_t)id : 0;
    }
  }
  free(cbT);
}

/* 4-bit packing  pq.rs:168-172: (v[1] << 4) | v[0] */
void orc_pq_encode4_f32(int metric, const float *x, size_t n, size_t d, const float *codebook,
                        size_t m_count, uint8_t *codes) {
  size_t sd = d / m_count, kc = 16;
#pragma omp parallel for schedule(static)
  for (size_t r = 0; r < n; r++) {
    for (size_t m = 0; m < m_count; m += 2) {
      uint32_t id0, id1; float dist;
      int f0 = orc_argmin_row(metric, x + r * d + m * sd, codebook + m * kc * sd, kc, sd, NULL, &id0, &dist);
      int f1 = orc_argmin_row(metric, x + r * d + (m + 1) * sd, codebook + (m + 1) * kc * sd, kc, sd, NULL, &id1, &dist);
      uint8_t c0 = f0 ? (uint8_t)id0 : 0, c1 = f1 ? (uint8_t)id1 : 0;
      codes[r * (m_count / 2) + m / 2] = (uint8_t)((c1 << 4) | c0);
    }
  }
}

/* a13: transpose  pq/storage.rs:430-449: [n][M] -> [M][n] */
void orc_transpose_u8(const uint8_t *in, size_t n, size_t m_count, uint8_t *out) {
  for (size_t r = 0; r < n; r++)
    for (size_t m = 0; m < m_count; m++) out[m * n + r] = in[r * m_count + m];
}

/* a16: build_distance_table_l2 / _dot  pq/distance.rs:24-92 -> [M][2^nbits] f32 */
void orc_build_lut_f32(int metric, const float *q, size_t d, const float *codebook, size_t m_count,
                       uint32_t nbits, float *lut) {
  size_t sd = d / m_count, kc = (size_t)1 << nbits;
  for (size_t m = 0; m < m_count; m++)
    for (size_t c = 0; c < kc; c++)
      lut[m * kc + c] = orc_dist(metric, q + m * sd, codebook + (m * kc + c) * sd, sd);
}

/* The same table, SIMD lanes across CODEWORDS (8-bit codes, L2 / dot): every entry goes through exactly the operation sequence of
 * orc_l2_f32 / orc_dot_f32 above (tail elements first, 16 lane accumulators, lanes added in order, tail + total) -- only sixteen
 * entries advance together, so the values are bit-identical to orc_build_lut_f32 (tests/test_oracle_golden.py pins that).
 * The search restatement builds one table per (query, probed partition): this is where a fuzz case's oracle time went.
 * cbT = the codebook transposed to [M][sd][256] (orc_transpose_codebook_f32, once per search call).                          */
void orc_transpose_codebook_f32(const float *codebook, size_t d, size_t m_count, float *cbT) {
  const size_t sd = d / m_count, kc = 256;
  for (size_t m = 0; m < m_count; m++)
    for (size_t c = 0; c < kc; c++)
      for (size_t i = 0; i < sd; i++) cbT[(m * sd + i) * kc + c] = codebook[(m * kc + c) * sd + i];
}

void orc_build_lut_T_f32(int metric, const float *q, size_t d, const float *cbT, size_t m_count, float *lut) {
  const size_t sd = d / m_count, kc = 256;
  for (size_t m = 0; m < m_count; m++) orc_dist_many_T(metric == ORC_DOT, q + m * sd, sd, cbT + m * sd * kc, kc, lut + m * kc);
}

/* a17: compute_pq_distance (8-bit)  pq/distance.rs:109-144 over TRANSPOSED codes;
 * dist[j] = ((0 + LUT[0][c0j]) + LUT[1][c1j]) + ... ; dot post-bias -(M-1)
 * pq/storage.rs:949-957.                                                        */
void orc_pq_scan_f32(int metric, const float *lut, size_t m_count, const uint8_t *codes_t,
                     size_t n_p, float *dists) {
  for (size_t j = 0; j < n_p; j++) dists[j] = 0.0f;
  for (size_t m = 0; m < m_count; m++) {
    const float *t = lut + m * 256;
    const uint8_t *c = codes_t + m * n_p;
    for (size_t j = 0; j < n_p; j++) dists[j] += t[c[j]];
  }
  if (orc_is_dot(metric)) {
    float diff = (float)m_count - 1.0f;
    for (size_t j = 0; j < n_p; j++) dists[j] = dists[j] - diff;
  }
}

/* row-major variant used by pq/distance.rs:337-364 test (must equal transposed exactly) */
void orc_pq_scan_rowmajor_f32(const float *lut, size_t m_count, const uint8_t *codes, size_t n_p,
                              float *dists) {
  for (size_t j = 0; j < n_p; j++) {
    float s = 0.0f;
    for (size_t m = 0; m < m_count; m++) s += lut[m * 256 + codes[j * m_count + m]];
    dists[j] = s;
  }
}


/* ------------------------------------------------------------------------- */
static inline uint32_t orc_key(float f);
/* a18: 4-bit PQ.  compute_pq_distance_4bit  pq/distance.rs:147-242 with
 * compute_pq_distance_4bit_flat :246-268 and quantize_distance_table :275-284.
 * code: transposed [M/2][n] bytes, byte b of a row = sub-vector 2b (low nibble) | 2b+1 (high).
 * The first flat_num = max(200, min(k_hint, n)) rows and the n%16 tail are exact f32 sums; the
 * rest are saturating u8 sums of the quantised table (u8x16 `+=` is _mm_adds_epu8,
 * simd/u8.rs:303-309), de-quantised as q*range + qmin (no fma).                              */
static void orc_pq4_flat(const float *lut, size_t n, const uint8_t *code, size_t m_count, size_t offset,
                         size_t length, float *dists) {
  for (size_t b = 0; b < m_count / 2; b++) {
    const uint8_t *vi = code + b * n + offset;
    const float *t0 = lut + (b * 2) * 16, *t1 = lut + (b * 2 + 1) * 16;
    for (size_t i = 0; i < length; i++) {
      dists[offset + i] += t0[vi[i] & 0xF];
      dists[offset + i] += t1[vi[i] >> 4];
    }
  }
}
void orc_pq_scan4_f32(int metric, const float *lut, size_t m_count, const uint8_t *code, size_t n,
                      size_t k_hint, float *dists) {
  if (n == 0) return;
  for (size_t j = 0; j < n; j++) dists[j] = 0.0f;
  if (k_hint > n) k_hint = n;
  size_t flat_num = k_hint > 200 ? k_hint : 200;
  if (flat_num > n) flat_num = n;
  orc_pq4_flat(lut, n, code, m_count, 0, flat_num, dists);
  float qmax = dists[0];
  for (size_t j = 1; j < flat_num; j++)
    if (orc_key(dists[j]) >= orc_key(qmax)) qmax = dists[j]; /* max_by(total_cmp): last maximum */
  float qmin = INFINITY;
  for (size_t i = 0; i < m_count * 16; i++) qmin = fminf(qmin, lut[i]); /* f32::min */
  float factor = 255.0f / (qmax - qmin);
  uint8_t *qt = (uint8_t *)malloc(m_count * 16);
  for (size_t i = 0; i < m_count * 16; i++) {
    float v = roundf((lut[i] - qmin) * factor); /* f32::round: half away from zero; `as u8` saturates, NaN -> 0 */
    qt[i] = v != v ? 0 : (v <= 0.0f ? 0 : (v >= 255.0f ? 255 : (uint8_t)v));
  }
  size_t remainder = n % 16;
  float range = (qmax - qmin) / 255.0f;
  for (size_t j = flat_num; j < n - remainder; j++) {
    unsigned acc = 0;
    for (size_t b = 0; b < m_count / 2; b++) {
      uint8_t c = code[b * n + j];
      acc += qt[(b * 2) * 16 + (c & 0xF)]; if (acc > 255) acc = 255;
      acc += qt[(b * 2 + 1) * 16 + (c >> 4)]; if (acc > 255) acc = 255;
    }
    dists[j] = (float)acc * range + qmin;
  }
  if (remainder > 0) {
    size_t offset = n - remainder > flat_num ? n - remainder : flat_num;
    orc_pq4_flat(lut, n, code, m_count, offset, n - offset, dists);
  }
  free(qt);
  if (orc_is_dot(metric)) {
    float diff = (float)m_count - 1.0f;
    for (size_t j = 0; j < n; j++) dists[j] = dists[j] - diff;
  }
}

/* sum_4bit_dist_table_scalar  lance-linalg/src/simd/dist_table.rs:65-91 (PERM0 layout, u16
 * saturating sums; RabitQ fast-scan).  Pinned against the reference's own C kernel
 * (simd/dist_table.c built into oracle/_ref) and its known answer dists[1] == 38 (:178-217). */
static const size_t ORC_PERM0[16] = {0, 8, 1, 9, 2, 10, 3, 11, 4, 12, 5, 13, 6, 14, 7, 15};
void orc_sum_4bit_dist_table(size_t n, size_t code_len, const uint8_t *codes, const uint8_t *dist_table,
                             uint16_t *dists) {
  for (size_t vb = 0; vb < n / 32; vb++) {
    const uint8_t *blocks = codes + vb * 32 * code_len;
    for (size_t sv = 0; sv < code_len; sv++) {
      const uint8_t *block = blocks + sv * 32;
      const uint8_t *cur = dist_table + sv * 2 * 16, *nxt = dist_table + (sv * 2 + 1) * 16;
      for (size_t j = 0; j < 16; j++) {
        size_t lo_id = vb * 32 + ORC_PERM0[j], hi_id = lo_id + 16;
        unsigned a = dists[lo_id];
        a += cur[block[j] & 0x0F]; if (a > 65535) a = 65535;
        a += nxt[block[j + 16] & 0x0F]; if (a > 65535) a = 65535;
        dists[lo_id] = (uint16_t)a;
        unsigned b = dists[hi_id];
        b += cur[block[j] >> 4]; if (b > 65535) b = 65535;
        b += nxt[block[j + 16] >> 4]; if (b > 65535) b = 65535;
        dists[hi_id] = (uint16_t)b;
      }
    }
  }
}

/* ------------------------------------------------------------------------- */
/* f32::total_cmp key (graph.rs:66-82 OrderedFloat): monotone u32 */
static inline uint32_t orc_key(float f) {
  uint32_t b; memcpy(&b, &f, 4);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

/* Rust std 1.90 BinaryHeap<OrderedNode> (max-heap on dist only). */
typedef struct { uint32_t key; float dist; uint64_t id; } orc_node;
typedef struct { orc_node *data; size_t len; } orc_heap;

static void orc_heap_sift_up(orc_heap *h, size_t start, size_t pos) {
  orc_node elt = h->data[pos];
  while (pos > start) {
    size_t parent = (pos - 1) / 2;
    if (elt.key <= h->data[parent].key) break;
    h->data[pos] = h->data[parent];
    pos = parent;
  }
  h->data[pos] = elt;
}
static void orc_heap_push(orc_heap *h, orc_node n) {
  size_t old_len = h->len;
  h->data[h->len++] = n;
  orc_heap_sift_up(h, 0, old_len);
}
static void orc_heap_sift_down_to_bottom(orc_heap *h, size_t pos) {
  size_t end = h->len, start = pos;
  orc_node elt = h->data[pos];
  size_t child = 2 * pos + 1;
  size_t lim = end >= 2 ? end - 2 : 0; /* end.saturating_sub(2) */
  while (child <= lim && end >= 2) {
    if (h->data[child].key <= h->data[child + 1].key) child += 1;
    h->data[pos] = h->data[child];
    pos = child;
    child = 2 * pos + 1;
  }
  if (child == end - 1) {
    h->data[pos] = h->data[child];
    pos = child;
  }
  h->data[pos] = elt;
  orc_heap_sift_up(h, start, pos);
}
static orc_node orc_heap_pop(orc_heap *h) {
  orc_node item = h->data[--h->len];
  if (h->len > 0) {
    orc_node t = h->data[0]; h->data[0] = item; item = t;
    orc_heap_sift_down_to_bottom(h, 0);
  }
  return item;
}

/* a19: FlatIndex::search  flat/index.rs:82-177 (no prefilter): push while < k,
 * else replace the root only if root.dist > dist (total order).  Optional
 * [lower, upper) range.  Output = heap vector order (unsorted).  Returns count. */
size_t orc_heap_topk(const float *dists, const uint64_t *row_ids, size_t n, size_t k,
                     int has_range, float lower, float upper, uint64_t *out_ids, float *out_dists) {
  if (k == 0) return 0;
  orc_heap h; h.data = (orc_node *)malloc((k + 1) * sizeof(orc_node)); h.len = 0;
  uint32_t lo = orc_key(lower), hi = orc_key(upper);
  for (size_t j = 0; j < n; j++) {
    orc_node nd; nd.dist = dists[j]; nd.key = orc_key(dists[j]); nd.id = row_ids[j];
    if (has_range && (nd.key < lo || nd.key >= hi)) continue;
    if (h.len < k) {
      orc_heap_push(&h, nd);
    } else if (h.data[0].key > nd.key) {
      orc_heap_pop(&h);
      orc_heap_push(&h, nd);
    }
  }
  size_t cnt = h.len;
  for (size_t i = 0; i < cnt; i++) { out_ids[i] = h.data[i].id; out_dists[i] = h.data[i].dist; }
  free(h.data);
  return cnt;
}

/* a21: SortExec([_distance asc, _rowid asc]).fetch(k)  scanner.rs:3440-3468 */
typedef struct { uint32_t key; float dist; uint64_t id; } orc_pair;
static int orc_pair_cmp(const void *a, const void *b) {
  const orc_pair *x = (const orc_pair *)a, *y = (const orc_pair *)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  if (x->id != y->id) return x->id < y->id ? -1 : 1;
  return 0;
}
size_t orc_sort_fetch(uint64_t *ids, float *dists, size_t n, size_t k) {
  orc_pair *p = (orc_pair *)malloc((n ? n : 1) * sizeof(orc_pair));
  for (size_t i = 0; i < n; i++) { p[i].key = orc_key(dists[i]); p[i].dist = dists[i]; p[i].id = ids[i]; }
  qsort(p, n, sizeof(orc_pair), orc_pair_cmp);
  size_t out = n < k ? n : k;
  for (size_t i = 0; i < out; i++) { ids[i] = p[i].id; dists[i] = p[i].dist; }
  free(p);
  return out;
}

/* a14: kmeans_find_partitions  kmeans.rs:1134-1158: all distances, then
 * sort_to_indices(limit=nprobes) ascending.  arrow's partial sort is unstable, so
 * the order of EQUAL distances is unspecified in the reference; we fix (dist, id). */
void orc_find_partitions_f32(int metric, const float *q, size_t nq, size_t d, const float *cent,
                             size_t nlist, size_t nprobes, uint32_t *part_ids, float *dists) {
  if (nprobes > nlist) nprobes = nlist;
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < nq; i++) {
    orc_pair *p = (orc_pair *)malloc(nlist * sizeof(orc_pair));
    for (size_t c = 0; c < nlist; c++) {
      float v = orc_dist(metric, q + i * d, cent + c * d, d);
      p[c].key = orc_key(v); p[c].dist = v; p[c].id = c;
    }
    qsort(p, nlist, sizeof(orc_pair), orc_pair_cmp);
    for (size_t j = 0; j < nprobes; j++) {
      part_ids[i * nprobes + j] = (uint32_t)p[j].id;
      if (dists) dists[i * nprobes + j] = p[j].dist;
    }
    free(p);
  }
}

/* a20+a21: flat KNN  flat.rs:95-148 + scanner.rs:3386-3406: all distances, then
 * sort by (dist, rowid) and fetch k.  Implemented as a bounded selection with the
 * same total order (result identical to a full sort + fetch).                   */
void orc_flat_knn_f32(int metric, const float *x, const uint64_t *row_ids, size_t n, size_t d,
                      const float *q, size_t nq, size_t k, uint64_t *out_ids, float *out_dists) {
#pragma omp parallel for schedule(dynamic, 1)
  for (size_t i = 0; i < nq; i++) {
    /* keep the k best in a small sorted array (k is small) */
    orc_pair *best = (orc_pair *)malloc((k + 1) * sizeof(orc_pair));
    size_t cnt = 0;
    float qn = metric == ORC_COSINE ? orc_norm_l2_f32(q + i * d, d) : (metric == ORC_COSINE_H ? orc_norm_l2_32_f32(q + i * d, d) : 0.0f);
    for (size_t r = 0; r < n; r++) {
      float v = metric == ORC_COSINE ? orc_cosine_f32(q + i * d, qn, x + r * d, d)
              : metric == ORC_COSINE_H ? orc_cosine_scalar32_f32(q + i * d, qn, x + r * d, d)
                                     : orc_dist(metric, q + i * d, x + r * d, d);
      orc_pair e; e.key = orc_key(v); e.dist = v; e.id = row_ids ? row_ids[r] : (uint64_t)r;
      if (cnt == k && orc_pair_cmp(&e, &best[k - 1]) >= 0) continue;
      size_t pos = cnt < k ? cnt : k - 1;
      while (pos > 0 && orc_pair_cmp(&e, &best[pos - 1]) < 0) { best[pos] = best[pos - 1]; pos--; }
      best[pos] = e;
      if (cnt < k) cnt++;
    }
    for (size_t j = 0; j < k; j++) {
      out_ids[i * k + j] = j < cnt ? best[j].id : UINT64_MAX;
      out_dists[i * k + j] = j < cnt ? best[j].dist : INFINITY;
    }
    free(best);
  }
}

/* ------------------------------------------------------------------------- */
/* Full IVF_PQ query (3.2 of SURVEY): per query
 *   [cosine: normalise q  knn.rs:495-498]
 *   find_partitions (a14) -> first nprobes partitions
 *   per partition: residual q - centroid[p] (v2.rs:316-332, L2/cosine only),
 *     LUT (a16), ADC scan over transposed codes (a17), heap top-(k*refine) (a19)
 *   SortExec (dist,rowid) fetch k*refine (a21)
 *   [refine: exact distance on raw vectors of the candidates, sort, fetch k
 *            scanner.rs:2884-2904,3336-3412]
 * Index layout: part_offsets[nlist+1]; codes_t = per-partition transposed [M][n_p]
 * blocks concatenated (block p starts at part_offsets[p]*M); row_ids[N] in
 * partition order.  raw/raw_pos: raw vectors [*][d] and, for slot i, raw row of
 * row_ids[i] given through rowid_to_raw (NULL => rowid is the raw row index).   */
void orc_ivfpq_search_x(int metric, const float *centroids, size_t nlist, size_t d,
                        const float *codebook, size_t m_count, const uint32_t *part_offsets,
                        const uint8_t *codes_t, const uint64_t *row_ids, const float *queries,
                        size_t nq, size_t k, size_t nprobes, size_t refine, const float *raw,
                        uint64_t *out_ids, float *out_dists, int f16);
void orc_ivfpq_search_x2(int metric, const float *centroids, size_t nlist, size_t d,
                         const float *codebook, size_t m_count, uint32_t nbits, const uint32_t *part_offsets,
                         const uint8_t *codes_t, const uint64_t *row_ids, const float *queries,
                         size_t nq, size_t k, size_t nprobes, size_t refine, const float *raw,
                         uint64_t *out_ids, float *out_dists, int f16);
void orc_ivfpq_search_f32(int metric, const float *centroids, size_t nlist, size_t d,
                          const float *codebook, size_t m_count, const uint32_t *part_offsets,
                          const uint8_t *codes_t, const uint64_t *row_ids, const float *queries,
                          size_t nq, size_t k, size_t nprobes, size_t refine, const float *raw,
                          uint64_t *out_ids, float *out_dists) {
  orc_ivfpq_search_x(metric, centroids, nlist, d, codebook, m_count, part_offsets, codes_t, row_ids, queries, nq, k,
                     nprobes, refine, raw, out_ids, out_dists, 0);
}
/* f16 != 0: the query, centroids, codebook and raw vectors are f16 (in f32 containers); the
 * residual query is an f16 subtraction (arrow `sub` on Float16, v2.rs:326). */
void orc_ivfpq_search_x(int metric, const float *centroids, size_t nlist, size_t d,
                        const float *codebook, size_t m_count, const uint32_t *part_offsets,
                        const uint8_t *codes_t, const uint64_t *row_ids, const float *queries,
                        size_t nq, size_t k, size_t nprobes, size_t refine, const float *raw,
                        uint64_t *out_ids, float *out_dists, int f16) {
  orc_ivfpq_search_x2(metric, centroids, nlist, d, codebook, m_count, 8, part_offsets, codes_t, row_ids, queries, nq, k,
                      nprobes, refine, raw, out_ids, out_dists, f16);
}
/* nbits = 4: codes_t blocks are [M/2][n_p] packed bytes and distance_all takes the 4-bit path with
 * k_hint = k*refine (flat/index.rs:94 `dist_calc.distance_all(k)`). */
/* DistCalculator::distance(id) of PQDistCalculator (pq/storage.rs:893-919), the per-row form the PREFILTER branch of
 * FlatIndex::search uses (flat/index.rs:129-165): 8-bit = Iterator::sum over the M table entries in order (the same
 * association as distance_all); 4-bit = per code byte (table[2i][lo] + table[2i+1][hi]), those pair sums added in
 * order, on the UNQUANTISED f32 table (distance_all quantises it); dot: minus (M-1).                                */
static float orc_pq_distance_one(int metric, const float *lut, size_t m_count, uint32_t nbits, const uint8_t *codes_t,
                                 size_t n_p, size_t j) {
  float s = -0.0f;   /* f32::sum folds from -0.0 (core::iter::traits::accum, Rust >= 1.81) */
  if (nbits == 4) {
    for (size_t i = 0; i < m_count / 2; i++) {
      const uint8_t c = codes_t[i * n_p + j];
      s += lut[(2 * i) * 16 + (c & 0x0F)] + lut[(2 * i + 1) * 16 + (c >> 4)];
    }
  } else {
    for (size_t m = 0; m < m_count; m++) s += lut[m * 256 + codes_t[m * n_p + j]];
  }
  if (orc_is_dot(metric)) s = s - ((float)m_count - 1.0f);
  return s;
}

static void orc_ivfpq_search_impl(int metric, const float *centroids, size_t nlist, size_t d,
                         const float *codebook, size_t m_count, uint32_t nbits, const uint32_t *part_offsets,
                         const uint8_t *codes_t, const uint64_t *row_ids, const float *queries,
                         size_t nq, size_t k, size_t nprobes, size_t refine, const float *raw,
                         uint64_t *out_ids, float *out_dists, int f16, const uint8_t *allow, size_t n_allow,
                         int has_range, float lower, float upper) {
  const size_t mbytes = nbits == 4 ? m_count / 2 : m_count;
  if (nprobes > nlist) nprobes = nlist;
  metric = orc_metric_h(metric, f16);     /* Float16 column: half::f16's dot / cosine / normalize */
  int scan_metric = orc_is_cos(metric) ? ORC_L2 : metric;
  size_t keff = k * (refine ? refine : 1);
  size_t max_np = 0;
  for (size_t p = 0; p < nlist; p++) {
    size_t np_ = part_offsets[p + 1] - part_offsets[p];
    if (np_ > max_np) max_np = np_;
  }
  /* 8-bit codes under L2 / dot: tables built sixteen codewords at a time from a transposed codebook (bit-identical values) */
  float *cbT = NULL;
  if (nbits == 8 && (scan_metric == ORC_L2 || scan_metric == ORC_DOT)) {
    cbT = (float *)malloc(m_count * 256 * (d / m_count) * sizeof(float));
    if (cbT) orc_transpose_codebook_f32(codebook, d, m_count, cbT);
  }
#pragma omp parallel for schedule(dynamic, 1)
  for (size_t i = 0; i < nq; i++) {
    float *q = (float *)malloc(d * sizeof(float));
    float *qr = (float *)malloc(d * sizeof(float));
    float *lut = (float *)malloc(m_count * 256 * sizeof(float));
    float *pd = (float *)malloc((max_np ? max_np : 1) * sizeof(float));
    uint64_t *pid = (uint64_t *)malloc((max_np ? max_np : 1) * sizeof(uint64_t));
    uint32_t *parts = (uint32_t *)malloc(nprobes * sizeof(uint32_t));
    uint64_t *cand_ids = (uint64_t *)malloc((nprobes * keff + 1) * sizeof(uint64_t));
    float *cand_d = (float *)malloc((nprobes * keff + 1) * sizeof(float));
    size_t ncand = 0;
    if (metric == ORC_COSINE) orc_normalize_f32(queries + i * d, 1, d, q);
    else if (metric == ORC_COSINE_H) orc_normalize_h(queries + i * d, 1, d, q);
    else memcpy(q, queries + i * d, d * sizeof(float));
    /* single-query find_partitions (serial inside the parallel region) */
    {
      orc_pair *p = (orc_pair *)malloc(nlist * sizeof(orc_pair));
      for (size_t c = 0; c < nlist; c++) {
        float v = orc_dist(scan_metric, q, centroids + c * d, d);
        p[c].key = orc_key(v); p[c].dist = v; p[c].id = c;
      }
      qsort(p, nlist, sizeof(orc_pair), orc_pair_cmp);
      for (size_t j = 0; j < nprobes; j++) parts[j] = (uint32_t)p[j].id;
      free(p);
    }
    for (size_t j = 0; j < nprobes; j++) {
      size_t p = parts[j];
      size_t off = part_offsets[p], np_ = part_offsets[p + 1] - off;
      if (np_ == 0) continue;
      if (scan_metric == ORC_L2) {
        for (size_t t = 0; t < d; t++) qr[t] = ORC_RH(f16, q[t] - centroids[p * d + t]);
      } else {
        memcpy(qr, q, d * sizeof(float));
      }
      if (cbT) orc_build_lut_T_f32(scan_metric, qr, d, cbT, m_count, lut);
      else orc_build_lut_f32(scan_metric, qr, d, codebook, m_count, nbits, lut);
      if (allow) {
        /* prefilter branch: only selected rows, in storage order, each through distance(id); the heap sees the same
         * (row id, distance) sequence as FlatIndex::search's loop */
        size_t na = 0;
        for (size_t j = 0; j < np_; j++) {
          const uint64_t rid = row_ids[off + j];
          if (rid >= n_allow || !allow[rid]) continue;
          pd[na] = orc_pq_distance_one(scan_metric, lut, m_count, nbits, codes_t + off * mbytes, np_, j);
          pid[na++] = rid;
        }
        ncand += orc_heap_topk(pd, pid, na, keff, has_range, lower, upper, cand_ids + ncand, cand_d + ncand);
        continue;
      }
      if (nbits == 4) orc_pq_scan4_f32(scan_metric, lut, m_count, codes_t + off * mbytes, np_, keff, pd);
      else orc_pq_scan_f32(scan_metric, lut, m_count, codes_t + off * mbytes, np_, pd);
      ncand += orc_heap_topk(pd, row_ids + off, np_, keff, has_range, lower, upper, cand_ids + ncand, cand_d + ncand);
    }
    size_t got = orc_sort_fetch(cand_ids, cand_d, ncand, keff);
    if (refine && raw) {
      /* flat_knn on the taken rows with the index's metric and the ORIGINAL query
       * (scanner.rs:2884-2904): KNNVectorDistanceExec + SortExec(dist,rowid).fetch(k) */
      const float *qo = queries + i * d;
      float qn = metric == ORC_COSINE ? orc_norm_l2_f32(qo, d) : (metric == ORC_COSINE_H ? orc_norm_l2_32_f32(qo, d) : 0.0f);
      size_t kept = 0;
      for (size_t c = 0; c < got; c++) {
        const float *rv = raw + cand_ids[c] * d;
        const float ex = metric == ORC_COSINE ? orc_cosine_f32(qo, qn, rv, d)
                       : metric == ORC_COSINE_H ? orc_cosine_scalar32_f32(qo, qn, rv, d) : orc_dist(metric, qo, rv, d);
        /* a distance range is applied to the exact distances too: LanceFilterExec(dist >= lower AND dist < upper)
         * between KNNVectorDistanceExec and the final SortExec (scanner.rs:3334-3377) */
        if (has_range && !(ex >= lower && ex < upper)) continue;
        cand_ids[kept] = cand_ids[c];
        cand_d[kept++] = ex;
      }
      got = orc_sort_fetch(cand_ids, cand_d, kept, k);
    } else if (got > k) {
      got = k;
    }
    for (size_t j = 0; j < k; j++) {
      out_ids[i * k + j] = j < got ? cand_ids[j] : UINT64_MAX;
      out_dists[i * k + j] = j < got ? cand_d[j] : INFINITY;
    }
    free(q); free(qr); free(lut); free(pd); free(pid); free(parts); free(cand_ids); free(cand_d);
  }
  free(cbT);
}

void orc_ivfpq_search_x2(int metric, const float *centroids, size_t nlist, size_t d,
                         const float *codebook, size_t m_count, uint32_t nbits, const uint32_t *part_offsets,
                         const uint8_t *codes_t, const uint64_t *row_ids, const float *queries,
                         size_t nq, size_t k, size_t nprobes, size_t refine, const float *raw,
                         uint64_t *out_ids, float *out_dists, int f16) {
  orc_ivfpq_search_impl(metric, centroids, nlist, d, codebook, m_count, nbits, part_offsets, codes_t, row_ids, queries, nq, k,
                        nprobes, refine, raw, out_ids, out_dists, f16, NULL, 0, 0, 0.0f, 0.0f);
}

/* The same search under a row-id prefilter (scanner prefilter=True -> PreFilter::mask, flat/index.rs:129-165):
 * allow[row id] != 0 selects a row; ids >= n_allow are not selected. */
void orc_ivfpq_search_filtered(int metric, const float *centroids, size_t nlist, size_t d,
                               const float *codebook, size_t m_count, uint32_t nbits, const uint32_t *part_offsets,
                               const uint8_t *codes_t, const uint64_t *row_ids, const float *queries,
                               size_t nq, size_t k, size_t nprobes, size_t refine, const float *raw,
                               uint64_t *out_ids, float *out_dists, int f16, const uint8_t *allow, size_t n_allow) {
  orc_ivfpq_search_impl(metric, centroids, nlist, d, codebook, m_count, nbits, part_offsets, codes_t, row_ids, queries, nq, k,
                        nprobes, refine, raw, out_ids, out_dists, f16, allow, n_allow, 0, 0.0f, 0.0f);
}

/* Distance-range query: Query::lower_bound / upper_bound reach FlatIndex::search (flat/index.rs:98-113, 131-146): in each
 * probed partition a row enters the heap only if lower <= dist < upper (total order on f32), with or without a
 * prefilter.  allow may be NULL.  With refine the exact distances are filtered by the same range before the final fetch
 * (scanner.rs:3334-3377). */
void orc_ivfpq_search_range(int metric, const float *centroids, size_t nlist, size_t d,
                            const float *codebook, size_t m_count, uint32_t nbits, const uint32_t *part_offsets,
                            const uint8_t *codes_t, const uint64_t *row_ids, const float *queries,
                            size_t nq, size_t k, size_t nprobes, uint64_t *out_ids, float *out_dists, int f16,
                            const uint8_t *allow, size_t n_allow, float lower, float upper, size_t refine, const float *raw) {
  orc_ivfpq_search_impl(metric, centroids, nlist, d, codebook, m_count, nbits, part_offsets, codes_t, row_ids, queries, nq, k,
                        nprobes, refine, raw, out_ids, out_dists, f16, allow, n_allow, 1, lower, upper);
}

/* Index build glue (builder.rs:555-846 in canonical, stable row order):
 * counting sort of rows by partition id -> part_offsets, slot->row permutation.
 * Rows with part id NONE (all-NaN, kmeans.rs:1447-1486) are dropped.           */
size_t orc_partition_layout(const uint32_t *part_ids, size_t n, size_t nlist, uint32_t *part_offsets,
                            uint32_t *perm) {
  memset(part_offsets, 0, (nlist + 1) * sizeof(uint32_t));
  for (size_t r = 0; r < n; r++)
    if (part_ids[r] != ORC_NONE) part_offsets[part_ids[r] + 1]++;
  for (size_t p = 0; p < nlist; p++) part_offsets[p + 1] += part_offsets[p];
  uint32_t *cur = (uint32_t *)malloc((nlist + 1) * sizeof(uint32_t));
  memcpy(cur, part_offsets, (nlist + 1) * sizeof(uint32_t));
  for (size_t r = 0; r < n; r++)
    if (part_ids[r] != ORC_NONE) perm[cur[part_ids[r]]++] = (uint32_t)r;
  size_t total = part_offsets[nlist];
  free(cur);
  return total;
}

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void orc_set_threads(int t) {
#ifdef _OPENMP
  omp_set_num_threads(t);
#else
  (void)t;
#endif
}
