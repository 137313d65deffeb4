// search.hip -- the batched ANN query on the device.
//
//   find_partitions         ivf/storage.rs:107-119 -> kmeans_find_partitions kmeans.rs:1134-1158
//   preprocess_query        lance/src/index/vector/ivf/v2.rs:316-332   (q - centroid[p])
//   build_distance_table    pq/distance.rs:24-92                         (LUT [m][256] f32)
//   compute_pq_distance     pq/distance.rs:109-144, pq/storage.rs:921-960 (sequential-m ADC)
//   FlatIndex::search       flat/index.rs:82-177                         (per-partition top-k)
//   SortExec(dist,rowid)    lance/src/dataset/scanner.rs:3440-3468       (global merge)
//   refine                  scanner.rs:2884-2904,3336-3412
//
// Scan kernel design (query-major): one workgroup per (query, probe-split).  For each
// probed partition it builds the residual LUT in LDS (exact l2_scalar order per entry),
// then every lane streams rows of the partition: one 16-byte load of a row's PQ codes
// (row-major codes, coalesced), m LDS gathers summed in m order (bit-equal to the
// reference's transposed ADC), and a compare against the query's running threshold T.
// Rows with key <= T are appended to an LDS candidate buffer; when the buffer fills, T is
// tightened to a k-th-smallest bound computed from per-lane minima (wave bitonic sort +
// rank merge), which keeps every row with key <= T.  The scan is LDS-gather-bound, the
// codes stream from L2/Infinity Cache (16 MB for SIFT-1M), HBM only sees them once.
//
// Exactness: the final candidate list holds EVERY scanned row with dist <= T_final, so
// sorting it by (dist, row id) reproduces the reference's global SortExec.  The one case
// that depends on the reference's heap internals -- a single partition holding more than
// k rows tied at the boundary distance -- is detected and flagged.
#include <algorithm>
#include <cstdlib>
#include <string>
#include <vector>

#include "common.h"
#include "exact.cuh"
#include "index.h"
#include "kernels.h"
#include "search_common.cuh"

#pragma clang fp contract(off)

namespace lh {

#ifndef LH_SCAN_CAP
#define LH_SCAN_CAP 1024
#endif
#ifndef LH_SCAN_ROUND
#define LH_SCAN_ROUND 512
#endif
#ifndef LH_LUT_BATCH
#define LH_LUT_BATCH 4
#endif
constexpr int SCAN_CAP = LH_SCAN_CAP;      // LDS candidate buffer entries
constexpr int SCAN_ROUND = LH_SCAN_ROUND;  // rows per round

// ------------------------------------------------------------------------------------
// bitonic sort of P (power of two) 64-bit keys in LDS with 256 threads
__device__ __forceinline__ void bitonic_sort_u64(uint64_t *a, int P) {
  for (int k2 = 2; k2 <= P; k2 <<= 1) {
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < P / 2; i += 256) {
        const int ix = 2 * j * (i / j) + (i % j);
        const int px = ix + j;
        const bool up = (ix & k2) == 0;
        const uint64_t x = a[ix], y = a[px];
        if ((x > y) == up) { a[ix] = y; a[px] = x; }
      }
      __syncthreads();
    }
  }
}

// a14: per query, sort (dist, id) and keep the first nprobes
__global__ __launch_bounds__(256) void select_probes_kernel(const float *__restrict__ matrix, int nlist, int P, int nprobes,
                                                            uint32_t *__restrict__ part_ids, float *__restrict__ dists) {
  extern __shared__ __attribute__((aligned(16))) uint64_t skeys[];
  const int q = blockIdx.x;
  const float *row = matrix + (int64_t)q * nlist;
  for (int i = threadIdx.x; i < P; i += 256)
    skeys[i] = i < nlist ? ((uint64_t)order_key(row[i]) << 32) | (uint32_t)i : ~0ull;
  __syncthreads();
  bitonic_sort_u64(skeys, P);
  for (int i = threadIdx.x; i < nprobes; i += 256) {
    const uint64_t e = skeys[i];
    part_ids[(int64_t)q * nprobes + i] = (uint32_t)e;
    if (dists) dists[(int64_t)q * nprobes + i] = key_to_float((uint32_t)(e >> 32));
  }
}

// nlist <= 256 (C2): one WAVE per query, the 256 (dist, id) keys live 4 per lane and the whole bitonic network runs on
// registers and shuffles -- no LDS, no barriers (the workgroup-wide LDS version spends its time in 36 __syncthreads).
// Element e of the network sits in register e / 64 of lane e % 64.
__global__ __launch_bounds__(256) void select_probes_wave_kernel(const float *__restrict__ matrix, int nlist, int nprobes, int nq,
                                                                 uint32_t *__restrict__ part_ids, float *__restrict__ dists) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= nq) return;
  const float *row = matrix + (int64_t)q * nlist;
  unsigned long long v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = j * 64 + lane;
    v[j] = e < nlist ? ((unsigned long long)order_key(row[e]) << 32) | (uint32_t)e : ~0ull;
  }
#pragma unroll
  for (int k2 = 2; k2 <= 256; k2 <<= 1) {
#pragma unroll
    for (int dd = k2 >> 1; dd > 0; dd >>= 1) {
      if (dd >= 64) {
        const int jd = dd >> 6;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if ((j & jd) == 0) {
            const bool up = ((j * 64) & k2) == 0;          // lane bits never reach k2 >= 128
            const unsigned long long a = v[j], b = v[j | jd];
            if ((a > b) == up) { v[j] = b; v[j | jd] = a; }
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int e = j * 64 + lane;
          const unsigned long long o = __shfl_xor(v[j], dd, 64);
          const bool up = (e & k2) == 0, lower = (lane & dd) == 0;
          v[j] = (lower == up) ? (v[j] < o ? v[j] : o) : (v[j] > o ? v[j] : o);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = j * 64 + lane;
    if (e < nprobes) {
      part_ids[(int64_t)q * nprobes + e] = (uint32_t)v[j];
      if (dists) dists[(int64_t)q * nprobes + e] = key_to_float((uint32_t)(v[j] >> 32));
    }
  }
}

// nlist > 8192: the keys no longer fit an LDS sort.  Pass 1: every lane scans its share of the row and keeps its smallest
// (dist, id) key; the nprobes-th smallest of the 256 lane minima is an upper bound T of the answer's last key.  Pass 2:
// keys <= T go to an LDS list (first-come, 4096 entries); if more exist the list's own nprobes-th smallest replaces T
// (it removes at least 4096 - nprobes of them) and the pass repeats.  The list is sorted and the first nprobes emitted.
constexpr int SELBIG_CAP = 4096;
__global__ __launch_bounds__(256) void select_probes_big_kernel(const float *__restrict__ matrix, int nlist, int nprobes,
                                                                uint32_t *__restrict__ part_ids, float *__restrict__ dists) {
  __shared__ uint64_t list[SELBIG_CAP];
  __shared__ uint64_t lmin[256];
  __shared__ uint32_t s_cnt;
  const int q = blockIdx.x;
  const float *row = matrix + (int64_t)q * nlist;
  uint64_t mn = ~0ull;
  for (int i = threadIdx.x; i < nlist; i += 256) {
    const uint64_t v = ((uint64_t)order_key(row[i]) << 32) | (uint32_t)i;
    mn = v < mn ? v : mn;
  }
  lmin[threadIdx.x] = mn;
  __syncthreads();
  bitonic_sort_u64(lmin, 256);
  // more probes than lanes: no bound from the lane minima -- the first round collects the first 4096 keys it meets and takes
  // its bound from those (each later round drops at least 4096 - nprobes keys: nprobes <= 2048 converges within the 64 rounds)
  uint64_t T = nprobes <= 256 ? lmin[nprobes - 1] : ~0ull;
  for (int round = 0; round < 64; ++round) {
    __syncthreads();
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < nlist; i += 256) {
      const uint64_t v = ((uint64_t)order_key(row[i]) << 32) | (uint32_t)i;
      if (v <= T) {
        const uint32_t slot = atomicAdd(&s_cnt, 1u);
        if (slot < (uint32_t)SELBIG_CAP) list[slot] = v;
      }
    }
    __syncthreads();
    const int c = (int)min(s_cnt, (uint32_t)SELBIG_CAP);
    const bool over = s_cnt > (uint32_t)SELBIG_CAP;
    int P = 64;
    while (P < c) P <<= 1;
    for (int i = c + threadIdx.x; i < P; i += 256) list[i] = ~0ull;
    __syncthreads();
    bitonic_sort_u64(list, P);
    if (!over) break;
    T = list[nprobes - 1];          // strictly smaller than before: at least CAP - nprobes collected keys exceed it
  }
  for (int i = threadIdx.x; i < nprobes; i += 256) {
    const uint64_t e = list[i];
    part_ids[(int64_t)q * nprobes + i] = (uint32_t)e;
    if (dists) dists[(int64_t)q * nprobes + i] = key_to_float((uint32_t)(e >> 32));
  }
}

static void launch_select_probes(lance_hip_ctx *ctx, const float *matrix, int nlist, int nprobes, int nq, uint32_t *part_ids, float *dists) {
  if (nlist > 8192) {
    hipLaunchKernelGGL(select_probes_big_kernel, dim3(nq), dim3(256), 0, ctx->stream, matrix, nlist, nprobes, part_ids, dists);
  } else if (nlist <= 256) {
    hipLaunchKernelGGL(select_probes_wave_kernel, dim3((unsigned)cdiv(nq, 4)), dim3(256), 0, ctx->stream, matrix, nlist, nprobes, nq, part_ids, dists);
  } else {
    const int P = next_pow2(nlist);
    hipLaunchKernelGGL(select_probes_kernel, dim3(nq), dim3(256), (size_t)P * 8, ctx->stream, matrix, nlist, P, nprobes, part_ids, dists);
  }
}

// ------------------------------------------------------------------------------------
struct ScanArgs {
  const float *q;  // [nq][d], normalised for cosine
  const uint32_t *probes;  // [nq][nprobes]
  const float *centroids;
  const float *codebook;
  const uint32_t *part_offsets;
  const uint8_t *codes;
  int d, m, sd, nprobes, nsplit, keff;
  int nbits;     // 8, or 4 (a18)
  int residual;  // 1: q - centroid (L2 / cosine), 0: dot
  int has_range;
  uint32_t lo_key, hi_key;
  uint32_t *out_keys;  // [nq*nsplit][SCAN_LCAP]
  uint32_t *out_pos;
  uint32_t *out_cnt;  // [nq*nsplit]
  uint32_t *flags;    // [nq]
  int round_f16;      // index element type is f16: the residual query is an f16 subtraction (v2.rs:326)
#ifdef LH_TIMING_EXPERIMENTS      // builds with -DLH_TIMING_EXPERIMENTS only (scripts/build_variant.sh): results WRONG, timing only
  int ablate;         // LANCE_HIP_ABLATE: 1 = LUT once per query, 2 = no candidate appends, 4 = skip scan
#define SCAN_ABLATE(p) ((p).ablate)
#else
#define SCAN_ABLATE(p) 0
#endif
  const uint32_t *allow;   // prefilter bitmap over storage positions or NULL
  int lanes32 = 0;         // f16 column under dot with sub-vectors of more than 16 elements: 32-lane table entries (lut_entry_rt)
};

// one LUT entry at a run-time sub-dimension.  lanes32: an f16 column under dot -- the table entries are dot products of f16
// sub-vectors, dot_scalar::<f16, f32, 32> (dot.rs:91-102,138-161): 32 lane accumulators, which differs from the 16-lane form once
// a sub-vector has more than 16 elements
template <int METRIC>
__device__ __forceinline__ float lut_entry_rt(const float *__restrict__ r, const float *__restrict__ cw, int sd, int lanes32) {
  if constexpr (METRIC == METRIC_DOT) {
    if (lanes32) return finish_metric<METRIC>(dist_exact_rt<METRIC, float, 32>(r, cw, sd));
  }
  return finish_metric<METRIC>(dist_exact_rt<METRIC>(r, cw, sd));
}

struct ScanShared {
  float *r;
  float *lut;
  uint32_t *ckey;
  uint32_t *cpos;
  uint32_t *sorted;  // 256
  uint32_t *misc;    // [0]=count [1]=T [2]=Tnew [3]=flags
};

// k-th smallest (0-based rank kk) of the 256 per-thread values `v`; result broadcast via misc[2]
__device__ __forceinline__ void kth_smallest_256(uint32_t v, int kk, const ScanShared &s) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // wave-level bitonic sort (ascending by lane)
#pragma unroll
  for (int k2 = 2; k2 <= 64; k2 <<= 1) {
#pragma unroll
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      const uint32_t o = __shfl_xor(v, j, 64);
      const bool up = (lane & k2) == 0;
      const bool lower = (lane & j) == 0;
      v = (lower == up) ? min(v, o) : max(v, o);
    }
  }
  s.sorted[threadIdx.x] = v;
  __syncthreads();
  // global rank with ties ordered by (run, position)
  int rank = lane;
  for (int w = 0; w < 4; ++w) {
    if (w == wave) continue;
    const uint32_t *run = s.sorted + w * 64;
    int lo = 0, hi = 64;  // first index with run[i] > v (w < wave) or run[i] >= v (w > wave)
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      const bool before = w < wave ? run[mid] <= v : run[mid] < v;
      if (before) lo = mid + 1; else hi = mid;
    }
    rank += lo;
  }
  if (rank == kk) s.misc[2] = v;
  __syncthreads();
}

// Tighten the running threshold: T <- upper bound of the keff-th smallest key in the buffer
// (exact when the buffer holds <= 256 entries), then drop entries with key > T.
__device__ __forceinline__ void tighten(const ScanShared &s, int keff) {
  __syncthreads();
  const int c = min((int)s.misc[0], SCAN_CAP);
  if (c < keff) return;  // uniform
  uint32_t ek[SCAN_CAP / 256], ep[SCAN_CAP / 256];
  uint32_t mymin = 0xFFFFFFFFu;
#pragma unroll
  for (int j = 0; j < SCAN_CAP / 256; ++j) {
    const int i = threadIdx.x + 256 * j;
    ek[j] = 0xFFFFFFFFu; ep[j] = 0;
    if (i < c) { ek[j] = s.ckey[i]; ep[j] = s.cpos[i]; mymin = min(mymin, ek[j]); }
  }
  kth_smallest_256(mymin, keff - 1, s);
  const uint32_t tnew = s.misc[2];
  __syncthreads();
  if (threadIdx.x == 0) { s.misc[0] = 0; s.misc[1] = tnew; }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < SCAN_CAP / 256; ++j) {
    const int i = threadIdx.x + 256 * j;
    if (i < c && ek[j] <= tnew) {
      const uint32_t slot = atomicAdd(&s.misc[0], 1u);
      s.ckey[slot] = ek[j]; s.cpos[slot] = ep[j];
    }
  }
  __syncthreads();
}

template <int SD, int METRIC, int MU>
__global__ __launch_bounds__(256) void ivfpq_scan_kernel(ScanArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  ScanShared s;
  const int dpad = (p.d + 3) & ~3;
  s.r = reinterpret_cast<float *>(smem);
  s.lut = s.r + dpad;
  s.ckey = reinterpret_cast<uint32_t *>(s.lut + p.m * 256);
  s.cpos = s.ckey + SCAN_CAP;
  s.sorted = s.cpos + SCAN_CAP;
  s.misc = s.sorted + 256;

  const int qi = blockIdx.x / p.nsplit, sp = blockIdx.x % p.nsplit;
  const float *qv = p.q + (int64_t)qi * p.d;
  if (threadIdx.x == 0) { s.misc[0] = 0; s.misc[1] = 0xFFFFFFFFu; s.misc[3] = 0; }
  __syncthreads();
  const int m = MU > 0 ? MU * 16 : p.m;
  const int sd = SD > 0 ? SD : p.sd;

  for (int pi = sp; pi < p.nprobes; pi += p.nsplit) {
    const uint32_t part = p.probes[(int64_t)qi * p.nprobes + pi];
    const uint32_t off = p.part_offsets[part];
    const int np = (int)(p.part_offsets[part + 1] - off);
    if (np == 0) continue;
    __syncthreads();  // previous partition's LUT readers are done
    // v2.rs:316-332 residual query
    for (int t = threadIdx.x; t < p.d; t += 256) {
      float rv = p.residual ? qv[t] - p.centroids[(int64_t)part * p.d + t] : qv[t];
      if (p.round_f16 && p.residual) rv = __half2float(__float2half_rn(rv));
      s.r[t] = rv;
    }
    __syncthreads();
    // pq/distance.rs:24-92: LUT[mm][c] = dist(q_sub[mm], codebook[mm][c]) in l2_scalar / dot_scalar order
    if (!((SCAN_ABLATE(p) & 1) && pi > sp)) {
      if constexpr (SD > 0 && SD % 4 == 0) {
        // 4 entries per step: all codebook loads of the step are issued before the arithmetic,
        // so one L2 round trip covers 4 entries (the codebook is 128 KiB, L2-resident).
        constexpr int Q = SD / 4;
        constexpr int LB = LH_LUT_BATCH;
        for (int i0 = 0; i0 < m; i0 += LB) {
          f4 cbv[LB][Q];
#pragma unroll
          for (int u = 0; u < LB; ++u) {
            const int mm = min(i0 + u, m - 1);
            const f4 *src = reinterpret_cast<const f4 *>(p.codebook + ((int64_t)mm * 256 + threadIdx.x) * SD);
#pragma unroll
            for (int i = 0; i < Q; ++i) cbv[u][i] = src[i];
          }
#pragma unroll
          for (int u = 0; u < LB; ++u) {
            const int mm = i0 + u;
            if (mm < m) {
              RegVec<SD> a;
#pragma unroll
              for (int i = 0; i < Q; ++i) a.q[i] = *reinterpret_cast<const f4 *>(&s.r[mm * SD + 4 * i]);
              const float v = dist_exact<SD, METRIC>(a, reinterpret_cast<const float *>(&cbv[u][0]));
              s.lut[mm * 256 + threadIdx.x] = finish_metric<METRIC>(v);
            }
          }
        }
      } else {
        for (int idx = threadIdx.x; idx < m * 256; idx += 256) {
          const int mm = idx >> 8;
          s.lut[idx] = lut_entry_rt<METRIC>(&s.r[mm * sd], p.codebook + (int64_t)idx * sd, sd, p.lanes32);
        }
      }
    }
    __syncthreads();

    const uint8_t *pcodes = p.codes + (int64_t)off * m;
    for (int base = 0; base < ((SCAN_ABLATE(p) & 4) ? 0 : np); base += SCAN_ROUND) {
      // The decision must be the same in every lane, and a lane that runs ahead appends (atomicAdd on misc[0]) as soon as it is
      // past this point: read the count, THEN a barrier, then decide.  (Round 1-2 read it without the barrier: a wave that saw
      // the count just over the limit entered tighten()'s barriers while the others were in the scan round -- rows lost or
      // distances from a half-written state for ~1 query in 1000 of a large batch; found by tests/fuzz_parity.py in round 3.)
      const bool need_tighten = (int)s.misc[0] > SCAN_CAP - SCAN_ROUND;
      __syncthreads();
      if (need_tighten) tighten(s, p.keff);
      const uint32_t T = s.misc[1];
#pragma unroll
      for (int u = 0; u < SCAN_ROUND / 256; ++u) {
        const int row = base + u * 256 + threadIdx.x;
        if (row < np) {
          float dist = 0.0f;  // pq/distance.rs:128-141: distances start at 0.0, += table[code] for m = 0..M-1
          if constexpr (MU > 0) {
#pragma unroll
            for (int w = 0; w < MU; ++w) {
              const uint4 cw = *reinterpret_cast<const uint4 *>(pcodes + (int64_t)row * (MU * 16) + w * 16);
              const uint32_t cws[4] = {cw.x, cw.y, cw.z, cw.w};
#pragma unroll
              for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) dist += s.lut[(w * 16 + e * 4 + bb) * 256 + ((cws[e] >> (8 * bb)) & 255u)];
            }
          } else {
            const uint8_t *rc = pcodes + (int64_t)row * m;
            for (int mm = 0; mm < m; ++mm) dist += s.lut[mm * 256 + rc[mm]];
          }
          if constexpr (METRIC == METRIC_DOT) dist = dist - ((float)m - 1.0f);  // pq/storage.rs:949-957
          const uint32_t key = order_key(dist);
          const bool in_range = !p.has_range || (key >= p.lo_key && key < p.hi_key);  // flat/index.rs:98-105
          if (in_range && key <= T && !((SCAN_ABLATE(p) & 2) && base > 0) && row_allowed(p.allow, off + (uint32_t)row)) {
            const uint32_t slot = atomicAdd(&s.misc[0], 1u);
            if (slot < SCAN_CAP) { s.ckey[slot] = key; s.cpos[slot] = off + (uint32_t)row; }
            else s.misc[3] = FLAG_OVERFLOW;
          }
        }
      }
      __syncthreads();
    }
  }
  // final: shrink to <= LCAP entries (exact threshold once <= 256 entries remain)
  __syncthreads();
  for (int iter = 0; iter < 8 && (int)s.misc[0] > SCAN_LCAP; ++iter) tighten(s, p.keff);
  __syncthreads();
  int c = min((int)s.misc[0], SCAN_CAP);
  uint32_t fl = s.misc[3];
  if (c > SCAN_LCAP) { c = SCAN_LCAP; fl |= FLAG_OVERFLOW; }
  const int64_t ob = (int64_t)blockIdx.x * SCAN_LCAP;
  for (int i = threadIdx.x; i < c; i += 256) { p.out_keys[ob + i] = s.ckey[i]; p.out_pos[ob + i] = s.cpos[i]; }
  if (threadIdx.x == 0) {
    p.out_cnt[blockIdx.x] = (uint32_t)c;
    if (fl) atomicOr(&p.flags[qi], fl);
  }
}


// ------------------------------------------------------------------------------------
// a18: 4-bit PQ (compute_pq_distance_4bit, pq/distance.rs:147-284).  Per (query, partition):
// the first flat_num = max(200, min(k_hint, n_p)) rows and the n_p % 16 tail get exact f32 LUT
// sums; every other row gets the saturating u8 sum of the table quantised to [qmin, qmax]
// (qmax = largest exact distance of the flat rows) and is de-quantised as q*range + qmin.
// Codes are row-major [n_p][M/2] here (low nibble = sub-vector 2b, high = 2b+1).
struct Pq4Shared {
  float *fd;        // exact distances of the flat rows
  uint8_t *qt;      // quantised table [M][16]
  uint32_t *sc;     // [0] = qmax key, [1] = qmin key, [2] = flat_num
};

template <int METRIC>
__device__ __forceinline__ float pq4_exact_row(const float *lut, const uint8_t *rc, int mb) {
  float dist = 0.0f;
  for (int b = 0; b < mb; ++b) {
    const uint32_t c = rc[b];
    dist += lut[(2 * b) * 16 + (c & 15u)];
    dist += lut[(2 * b + 1) * 16 + (c >> 4)];
  }
  return dist;
}

// Under a prefilter FlatIndex::search scores every selected row with PQDistCalculator::distance(id) (flat/index.rs:129-165,
// pq/storage.rs:893-921): the UNQUANTISED table, one term per code byte = table[2i][low nibble] + table[2i+1][high nibble],
// terms folded by f32::sum (from -0.0) -- neither the quantised fast-scan nor the order of pq4_exact_row.
template <int METRIC>
__device__ __forceinline__ float pq4_masked_row(const float *lut, const uint8_t *rc, int m) {
  float dist = -0.0f;
  for (int b = 0; b < m / 2; ++b) {
    const uint32_t c = rc[b];
    dist = dist + (lut[(2 * b) * 16 + (c & 15u)] + lut[(2 * b + 1) * 16 + (c >> 4)]);
  }
  if constexpr (METRIC == METRIC_DOT) dist = dist - ((float)m - 1.0f);
  return dist;
}

template <int METRIC, int BS>
__device__ __forceinline__ void pq4_prelude(const float *lut, int m, const uint8_t *pcodes, int np, int keff, const Pq4Shared &q4) {
  const int mb = m / 2;
  const int flat_num = min(max(200, min(keff, np)), np);
  if (threadIdx.x == 0) { q4.sc[0] = 0u; q4.sc[1] = order_key(INFINITY); q4.sc[2] = (uint32_t)flat_num; }
  __syncthreads();
  for (int row = threadIdx.x; row < flat_num; row += BS) {
    const float dv = pq4_exact_row<METRIC>(lut, pcodes + (int64_t)row * mb, mb);
    q4.fd[row] = dv;
    atomicMax(&q4.sc[0], order_key(dv));  // max_by(total_cmp)
  }
  for (int idx = threadIdx.x; idx < m * 16; idx += BS) {
    const float v = lut[idx];
    if (v == v) atomicMin(&q4.sc[1], order_key(v));  // fold(INFINITY, f32::min): NaN is ignored
  }
  __syncthreads();
  const float qmax = key_to_float(q4.sc[0]), qmin = key_to_float(q4.sc[1]);
  const float factor = 255.0f / (qmax - qmin);
  for (int idx = threadIdx.x; idx < m * 16; idx += BS) {
    const float v = roundf((lut[idx] - qmin) * factor);  // f32::round (half away from zero); `as u8` saturates, NaN -> 0
    q4.qt[idx] = v != v ? (uint8_t)0 : (v <= 0.0f ? (uint8_t)0 : (v >= 255.0f ? (uint8_t)255 : (uint8_t)v));
  }
  __syncthreads();
}

template <int METRIC>
__device__ __forceinline__ float pq4_row_distance(const float *lut, int m, const uint8_t *pcodes, int np, int row, const Pq4Shared &q4) {
  const int mb = m / 2;
  const int flat_num = (int)q4.sc[2];
  const int rem = np % 16;
  const uint8_t *rc = pcodes + (int64_t)row * mb;
  float dist;
  if (row < flat_num) {
    dist = q4.fd[row];
  } else if (row < np - rem) {
    uint32_t acc = 0;
    for (int b = 0; b < mb; ++b) {
      const uint32_t c = rc[b];
      acc = min(255u, acc + q4.qt[(2 * b) * 16 + (c & 15u)]);      // _mm_adds_epu8
      acc = min(255u, acc + q4.qt[(2 * b + 1) * 16 + (c >> 4)]);
    }
    const float qmax = key_to_float(q4.sc[0]), qmin = key_to_float(q4.sc[1]);
    const float range = (qmax - qmin) / 255.0f;
    dist = (float)acc * range + qmin;
  } else {
    dist = pq4_exact_row<METRIC>(lut, rc, mb);
  }
  if constexpr (METRIC == METRIC_DOT) dist = dist - ((float)m - 1.0f);
  return dist;
}

template <int METRIC>
__global__ __launch_bounds__(256) void ivfpq_scan4_kernel(ScanArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  ScanShared s;
  Pq4Shared q4;
  const int dpad = (p.d + 3) & ~3;
  const int m = p.m, sd = p.sd, mb = p.m / 2;
  s.r = reinterpret_cast<float *>(smem);
  s.lut = s.r + dpad;
  s.ckey = reinterpret_cast<uint32_t *>(s.lut + m * 16);
  s.cpos = s.ckey + SCAN_CAP;
  s.sorted = s.cpos + SCAN_CAP;
  s.misc = s.sorted + 256;
  q4.sc = s.misc + 8;
  q4.fd = reinterpret_cast<float *>(q4.sc + 4);
  q4.qt = reinterpret_cast<uint8_t *>(q4.fd + 256);

  const int qi = blockIdx.x / p.nsplit, sp = blockIdx.x % p.nsplit;
  const float *qv = p.q + (int64_t)qi * p.d;
  if (threadIdx.x == 0) { s.misc[0] = 0; s.misc[1] = 0xFFFFFFFFu; s.misc[3] = 0; }
  __syncthreads();
  for (int pi = sp; pi < p.nprobes; pi += p.nsplit) {
    const uint32_t part = p.probes[(int64_t)qi * p.nprobes + pi];
    const uint32_t off = p.part_offsets[part];
    const int np = (int)(p.part_offsets[part + 1] - off);
    if (np == 0) continue;
    __syncthreads();
    for (int t = threadIdx.x; t < p.d; t += 256) {
      float rv = p.residual ? qv[t] - p.centroids[(int64_t)part * p.d + t] : qv[t];
      if (p.round_f16 && p.residual) rv = __half2float(__float2half_rn(rv));
      s.r[t] = rv;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < m * 16; idx += 256)
      s.lut[idx] = lut_entry_rt<METRIC>(&s.r[(idx >> 4) * sd], p.codebook + (int64_t)idx * sd, sd, p.lanes32);
    __syncthreads();
    const uint8_t *pcodes = p.codes + (int64_t)off * mb;
    if (!p.allow) pq4_prelude<METRIC, 256>(s.lut, m, pcodes, np, p.keff, q4);   // uniform branch
    for (int base = 0; base < np; base += SCAN_ROUND) {
      const bool need_tighten = (int)s.misc[0] > SCAN_CAP - SCAN_ROUND;   // read, barrier, decide (see ivfpq_scan_kernel)
      __syncthreads();
      if (need_tighten) tighten(s, p.keff);
      const uint32_t T = s.misc[1];
      for (int u = 0; u < SCAN_ROUND / 256; ++u) {
        const int row = base + u * 256 + threadIdx.x;
        if (row < np && row_allowed(p.allow, off + (uint32_t)row)) {
          const uint32_t key = order_key(p.allow ? pq4_masked_row<METRIC>(s.lut, pcodes + (int64_t)row * mb, m)
                                                 : pq4_row_distance<METRIC>(s.lut, m, pcodes, np, row, q4));
          const bool in_range = !p.has_range || (key >= p.lo_key && key < p.hi_key);
          if (in_range && key <= T) {
            const uint32_t slot = atomicAdd(&s.misc[0], 1u);
            if (slot < SCAN_CAP) { s.ckey[slot] = key; s.cpos[slot] = off + (uint32_t)row; }
            else s.misc[3] = FLAG_OVERFLOW;
          }
        }
      }
      __syncthreads();
    }
  }
  __syncthreads();
  for (int iter = 0; iter < 8 && (int)s.misc[0] > SCAN_LCAP; ++iter) tighten(s, p.keff);
  __syncthreads();
  int c = min((int)s.misc[0], SCAN_CAP);
  uint32_t fl = s.misc[3];
  if (c > SCAN_LCAP) { c = SCAN_LCAP; fl |= FLAG_OVERFLOW; }
  const int64_t ob = (int64_t)blockIdx.x * SCAN_LCAP;
  for (int i = threadIdx.x; i < c; i += 256) { p.out_keys[ob + i] = s.ckey[i]; p.out_pos[ob + i] = s.cpos[i]; }
  if (threadIdx.x == 0) {
    p.out_cnt[blockIdx.x] = (uint32_t)c;
    if (fl) atomicOr(&p.flags[qi], fl);
  }
}

// ------------------------------------------------------------------------------------
struct MergeArgs {
  const uint32_t *keys, *pos, *cnt;  // per (query, split)
  const uint64_t *row_ids;
  const uint32_t *part_offsets;
  int nlist, nsplit, keff, k, P;
  int refine;             // 1: write candidates for the refine kernel
  uint64_t *out_ids;      // [nq][k]     (refine == 0)
  float *out_dists;
  uint64_t *cand_rid;     // [nq][keff]  (refine == 1)
  uint32_t *cand_cnt;     // [nq]
  uint32_t *flags;
};

__global__ __launch_bounds__(256) void ivfpq_merge_kernel(MergeArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint64_t *rid = reinterpret_cast<uint64_t *>(smem);
  uint32_t *key = reinterpret_cast<uint32_t *>(rid + p.P);
  uint32_t *pos = key + p.P;
  __shared__ int s_total, s_amb;
  const int q = blockIdx.x;
  if (threadIdx.x == 0) { s_total = 0; s_amb = 0; }
  for (int i = threadIdx.x; i < p.P; i += 256) { key[i] = 0xFFFFFFFFu; rid[i] = ~0ull; pos[i] = 0; }
  __syncthreads();
  for (int sp = 0; sp < p.nsplit; ++sp) {
    const int blk = q * p.nsplit + sp;
    const int c = (int)p.cnt[blk];
    const int base = s_total;
    __syncthreads();
    for (int i = threadIdx.x; i < c; i += 256) {
      const uint32_t ps = p.pos[(int64_t)blk * SCAN_LCAP + i];
      key[base + i] = p.keys[(int64_t)blk * SCAN_LCAP + i];
      pos[base + i] = ps;
      rid[base + i] = p.row_ids[ps];
    }
    if (threadIdx.x == 0) s_total = base + c;
    __syncthreads();
  }
  const int total = s_total;
  // sort only the occupied power-of-two prefix (the tail is already padding)
  int Pq = 64;
  while (Pq < total) Pq <<= 1;
  bitonic_sort_kr(key, rid, pos, Pq);
  const int got = min(total, p.keff);
  // tie check: more rows at the boundary distance than fit?
  if (total > p.keff && key[p.keff] == key[p.keff - 1]) {
    const uint32_t tf = key[p.keff - 1];
    // L = entries with key <= tf (sorted prefix)
    int lo = p.keff, hi = total;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (key[mid] <= tf) lo = mid + 1; else hi = mid; }
    const int L = lo;
    for (int i = threadIdx.x; i < L; i += 256) {
      const uint32_t pi = find_partition_dev(p.part_offsets, p.nlist, pos[i]);
      const uint32_t a = p.part_offsets[pi], b = p.part_offsets[pi + 1];
      int same = 0;
      for (int j = 0; j < L; ++j) same += (pos[j] >= a && pos[j] < b) ? 1 : 0;
      if (same > p.keff) s_amb = 1;
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_amb) atomicOr(&p.flags[q], FLAG_AMBIGUOUS);
  }
  if (p.refine) {
    for (int i = threadIdx.x; i < p.keff; i += 256) p.cand_rid[(int64_t)q * p.keff + i] = i < got ? rid[i] : ~0ull;
    if (threadIdx.x == 0) p.cand_cnt[q] = (uint32_t)got;
  } else {
    for (int i = threadIdx.x; i < p.k; i += 256) {
      p.out_ids[(int64_t)q * p.k + i] = i < got ? rid[i] : ~0ull;
      p.out_dists[(int64_t)q * p.k + i] = i < got ? key_to_float(key[i]) : INFINITY;
    }
  }
}

// refine: exact distance of the original query to the raw vectors of the candidates, then
// (dist, rowid) order, fetch k  (scanner.rs:2884-2904 take + flat_knn :3336-3412)
// H32: the column is f16 -- its dot products and norms are the 32-lane dot_scalar / norm_l2_impl (dot.rs:91-102, norm_l2.rs:60-85)
// and its cosine distance is the trait default cosine_scalar (cosine.rs:171-179), not f32's cosine_fast.
// Final selection of the refine kernels: key / rid [0, P) hold the candidates' (exact distance key, row id), invalid slots (key
// 0xFFFFFFFF).  The reference sorts by (dist, rowid) and takes k; only the k best are needed, so up to 256 slots are RANKED instead
// of sorted: slot t counts the slots that precede it in (key, rid, slot) order -- P broadcast LDS reads, one barrier, against the 28
// barrier-separated passes of a 128-entry bitonic sort (a third of the kernel's life once the row reads stopped being its bound) --
// and writes itself to out[rank] when rank < min(c, k).  Same order as the sort (the slot index only separates identical pairs, which
// are interchangeable).  More than 256 slots: the sort.
__device__ __forceinline__ void refine_emit_topk(uint32_t *key, uint64_t *rid, uint32_t *pos, int P, int c, int k, int qi,
                                                 uint64_t *__restrict__ out_ids, float *__restrict__ out_dists) {
  const int got = min(c, k);
  if (P <= 256) {
    const int t = threadIdx.x;
    if (t < P) {
      const uint32_t mk = key[t];
      const uint64_t mr = rid[t];
      int rank = 0;
      for (int j0 = 0; j0 < P; j0 += 4) {
        const uint4 kj = *reinterpret_cast<const uint4 *>(&key[j0]);
        const uint32_t kk[4] = {kj.x, kj.y, kj.z, kj.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          bool before = kk[u] < mk;
          if (kk[u] == mk) { const uint64_t rj = rid[j0 + u]; before = rj < mr || (rj == mr && j0 + u < t); }
          rank += before ? 1 : 0;
        }
      }
      if (rank < got) { out_ids[(int64_t)qi * k + rank] = mr; out_dists[(int64_t)qi * k + rank] = key_to_float(mk); }
    }
    for (int i = got + (int)threadIdx.x; i < k; i += 256) { out_ids[(int64_t)qi * k + i] = ~0ull; out_dists[(int64_t)qi * k + i] = INFINITY; }
    return;
  }
  bitonic_sort_kr(key, rid, pos, P);
  for (int i = threadIdx.x; i < k; i += 256) {
    out_ids[(int64_t)qi * k + i] = i < got ? rid[i] : ~0ull;
    out_dists[(int64_t)qi * k + i] = i < got ? key_to_float(key[i]) : INFINITY;
  }
}

// lance_hip_ivfpq_search_candidates: the exact distance of every candidate IN THE CANDIDATE LIST'S ORDER (the list-sharded multi-GPU search
// gathers PQ-ordered candidates and their exact distances from one scan; +inf for an empty slot)
__device__ __forceinline__ void refine_write_cand_exact(float *__restrict__ cand_exact, const uint32_t *key, int keff, int c, int qi) {
  if (!cand_exact) return;      // uniform
  for (int i = threadIdx.x; i < keff; i += 256) cand_exact[(int64_t)qi * keff + i] = (i < c && key[i] != 0xFFFFFFFFu) ? key_to_float(key[i]) : INFINITY;
  __syncthreads();              // the selection below may sort `key` in place
}

// rows long enough for the four-lanes-per-candidate cosine path (no f32x8 / scalar tail: d % 16 == 0; 16-byte aligned rows)
__host__ __device__ __forceinline__ bool refine_wide_rows(int d) { return d >= 256 && (d & 15) == 0; }

template <int METRIC, typename TR, bool H32 = false>
__global__ __launch_bounds__(256) void refine_kernel(const float *__restrict__ q, int d, const TR *__restrict__ raw,
                                                     uint64_t n_raw, const uint64_t *__restrict__ cand_rid,
                                                     const uint32_t *__restrict__ cand_cnt, int keff, int k, int P,
                                                     uint64_t *__restrict__ out_ids, float *__restrict__ out_dists, uint32_t *__restrict__ flags,
                                                     float *__restrict__ cand_exact = nullptr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint64_t *rid = reinterpret_cast<uint64_t *>(smem);
  uint32_t *key = reinterpret_cast<uint32_t *>(rid + P);
  uint32_t *pos = key + P;
  const int qi = blockIdx.x;
  const int c = (int)cand_cnt[qi];
  const float *qv = q + (int64_t)qi * d;
  float qnorm = 0.0f;
  const bool wide = METRIC == METRIC_COSINE && !H32 && sizeof(TR) == 4 && refine_wide_rows(d) && (reinterpret_cast<uintptr_t>(raw) & 15) == 0;
  if constexpr (METRIC == METRIC_COSINE) {
    if (!wide) qnorm = H32 ? norm_l2_rt<float, 32>(qv, d) : norm_l2_rt(qv, d);  // cosine_batch: x_norm = norm_l2(x)
  }
  if constexpr (METRIC == METRIC_COSINE && !H32 && sizeof(TR) == 4) {
    if (wide) {
      // Long f32 rows (C3: d = 1536): FOUR lanes per candidate.  cosine_fast (cosine.rs:143-175) keeps 16 FMA accumulators,
      // accumulator i taking the elements = i (mod 16); lane s of a quad owns accumulators 4s .. 4s+3, i.e. one 16-byte load per
      // 16-element chunk, and the quad reads 64 contiguous bytes -- 64 candidates in flight per workgroup instead of one lane
      // walking a 6 KB row with 4-byte strided loads (0.46 ms per 1000 queries at refine 10).  The f32x16 reduce_sum tree
      // ((a_i + a_{i+8}) -> reduce8_tree) runs across the quad with two xor-shuffles; additions are commutative, so both sides
      // of a shuffle pair hold the same bits.  The d % 16 == 0 rows have neither the f32x8 nor the scalar tail (their +0.0 terms
      // are kept).  Query row staged in LDS once.
      float *qs = reinterpret_cast<float *>(pos + P);
      __shared__ float s_qnorm;
      for (int e = threadIdx.x; e < d; e += 256) qs[e] = qv[e];
      __syncthreads();
      if (threadIdx.x < 64) {   // norm_l2_impl::<f32, f32, 16> (norm_l2.rs:106-129): 16 lane sums (mul, then add), folded 0..15, no remainder here
        const int ln = threadIdx.x & 15;
        float sacc = 0.0f;
        for (int e = ln; e < d; e += 16) sacc += qs[e] * qs[e];
        float tot = 0.0f;
#pragma unroll
        for (int i2 = 0; i2 < 16; ++i2) tot = tot + __shfl(sacc, i2, 16);
        if (threadIdx.x == 0) s_qnorm = sqrtf(0.0f + tot);
      }
      __syncthreads();
      qnorm = s_qnorm;
      const int sub = threadIdx.x & 3, slot = threadIdx.x >> 2;   // 64 candidates per round
      for (int i0 = 0; i0 < P; i0 += 64) {
        const int i = i0 + slot;
        uint64_t r = ~0ull;
        if (i < c) {
          r = cand_rid[(int64_t)qi * keff + i];
          if (r >= n_raw && sub == 0) atomicOr(&flags[qi], FLAG_BADROW);
        }
        const bool ok = i < c && r < n_raw;
        f4 xy = {0.0f, 0.0f, 0.0f, 0.0f}, yn = {0.0f, 0.0f, 0.0f, 0.0f};
        if (ok) {
          const f4 *yp = reinterpret_cast<const f4 *>(reinterpret_cast<const float *>(raw) + r * d) + sub;
          const f4 *xp = reinterpret_cast<const f4 *>(qs) + sub;
#pragma unroll 4
          for (int cch = 0; cch < d / 16; ++cch) {
            const f4 yv = yp[cch * 4], xv = xp[cch * 4];
            xy = __builtin_elementwise_fma(xv, yv, xy);
            yn = __builtin_elementwise_fma(yv, yv, yn);
          }
        }
        // a_i + a_{i+8}: quad lane s <-> s ^ 2; then t_i + t_{i+4}: s <-> s ^ 1; then (s0 + s2) + (s1 + s3) in the lane
        f4 t, u;
#pragma unroll
        for (int e = 0; e < 4; ++e) { t[e] = xy[e] + __shfl_xor(xy[e], 2, 64); u[e] = yn[e] + __shfl_xor(yn[e], 2, 64); }
#pragma unroll
        for (int e = 0; e < 4; ++e) { t[e] = t[e] + __shfl_xor(t[e], 1, 64); u[e] = u[e] + __shfl_xor(u[e], 1, 64); }
        const float xyr = ((t[0] + t[2]) + (t[1] + t[3])) + 0.0f + 0.0f;
        const float ynr = ((u[0] + u[2]) + (u[1] + u[3])) + 0.0f + 0.0f;
        if (sub == 0 && i < P) {
          key[i] = ok ? order_key(1.0f - xyr / qnorm / sqrtf(ynr)) : 0xFFFFFFFFu;
          rid[i] = r; pos[i] = 0;
        }
      }
    } else {
      for (int i = threadIdx.x; i < P; i += 256) {
        uint32_t kk = 0xFFFFFFFFu;
        uint64_t r = ~0ull;
        if (i < c) {
          r = cand_rid[(int64_t)qi * keff + i];
          if (r >= n_raw) atomicOr(&flags[qi], FLAG_BADROW);
          if (r < n_raw) kk = order_key(cosine_exact_rt<TR>(qv, qnorm, raw + r * d, d));
        }
        key[i] = kk; rid[i] = r; pos[i] = 0;
      }
    }
  } else if constexpr (METRIC == METRIC_COSINE) {
    for (int i = threadIdx.x; i < P; i += 256) {
      uint32_t kk = 0xFFFFFFFFu;
      uint64_t r = ~0ull;
      if (i < c) {
        r = cand_rid[(int64_t)qi * keff + i];
        if (r >= n_raw) atomicOr(&flags[qi], FLAG_BADROW);   // stored row id beyond the raw vectors handed to set_raw: reported, not ranked
        if (r < n_raw) kk = order_key(H32 ? cosine_scalar32_rt<TR>(qv, qnorm, raw + r * d, d) : cosine_exact_rt<TR>(qv, qnorm, raw + r * d, d));
      }
      key[i] = kk; rid[i] = r; pos[i] = 0;
    }
  } else {
    // one lane per candidate row.  (Measured alternative: 16 lanes per row = the 16 lane accumulators, coalesced 64-byte
    // reads, lane-order fold through shuffles: bit-equal but 0.21 ms instead of 0.11 per 10k queries -- the serial fold and the
    // eight dependent passes cost more than the strided reads.)
    for (int i = threadIdx.x; i < P; i += 256) {
      uint32_t kk = 0xFFFFFFFFu;
      uint64_t r = ~0ull;
      if (i < c) {
        r = cand_rid[(int64_t)qi * keff + i];
        if (r >= n_raw) atomicOr(&flags[qi], FLAG_BADROW);
        if (r < n_raw) kk = order_key(finish_metric<METRIC>(dist_exact_rt<METRIC, TR, (H32 && METRIC == METRIC_DOT) ? 32 : 16>(qv, raw + r * d, d)));
      }
      key[i] = kk; rid[i] = r; pos[i] = 0;
    }
  }
  __syncthreads();
  refine_write_cand_exact(cand_exact, key, keff, c, qi);
  refine_emit_topk(key, rid, pos, P, c, k, qi, out_ids, out_dists);
}

// refine, f32 rows of d % 16 == 0 elements under L2 / dot: TWO lanes per candidate, the whole row in flight.
// The kernel above walks a row with one lane in a ROLLED loop of 16-element chunks: eight dependent memory round trips of 64 bytes per
// candidate at d = 128 (read off the gfx950 assembly: `global_load_dwordx4 x4 -> s_waitcnt vmcnt(3..0) -> loop`), i.e. the kernel was bound
// by HBM LATENCY x workgroups in flight (0.114 ms per 10,000 x 100 candidates = 512 MB at C2), not by bandwidth.  Here lane h of a pair owns
// the lane accumulators 8h .. 8h+7 of l2_scalar / dot_scalar (l2.rs:57-91, dot.rs:52-89: accumulator i takes the elements = i mod 16), i.e.
// the two 16-byte halves [8h, 8h+8) of every 64-byte chunk, and requests all of them -- up to 8 chunks = 16 loads per lane -- before it
// consumes any: one round trip per 128 elements, 32 bytes contiguous per lane, 64 per pair.  Same arithmetic: per accumulator the chunks
// arrive in ascending order (mul, then add: -ffp-contract=off), the final fold is the reference's ((0 + s0) + s1) + ... + s15, done by lane 0
// of the pair after one xor-shuffle per accumulator; d % 16 == 0, so the scalar tail is the +0.0 the reference adds too.
template <int METRIC>
__global__ __launch_bounds__(256) void refine_pair_kernel(const float *__restrict__ q, int d, const float *__restrict__ raw, uint64_t n_raw,
                                                          const uint64_t *__restrict__ cand_rid, const uint32_t *__restrict__ cand_cnt, int keff,
                                                          int k, int P, uint64_t *__restrict__ out_ids, float *__restrict__ out_dists,
                                                          uint32_t *__restrict__ flags, float *__restrict__ cand_exact = nullptr) {
  static_assert(METRIC == METRIC_L2 || METRIC == METRIC_DOT, "squared L2 / dot");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint64_t *rid = reinterpret_cast<uint64_t *>(smem);
  uint32_t *key = reinterpret_cast<uint32_t *>(rid + P);
  uint32_t *pos = key + P;
  f4 *qs4 = reinterpret_cast<f4 *>(pos + P);      // [d / 4] the query
  const int qi = blockIdx.x;
  const int c = (int)cand_cnt[qi];
  const f4 *qv4 = reinterpret_cast<const f4 *>(q + (int64_t)qi * d);
  for (int e = threadIdx.x; e < d / 4; e += 256) qs4[e] = qv4[e];
  __syncthreads();
  const int h = threadIdx.x & 1, slot = threadIdx.x >> 1;   // 128 candidates per round
  const int nchunk = d >> 4;
  for (int i0 = 0; i0 < P; i0 += 128) {
    const int i = i0 + slot;
    uint64_t r = ~0ull;
    if (i < c) {
      r = cand_rid[(int64_t)qi * keff + i];
      if (r >= n_raw && h == 0) atomicOr(&flags[qi], FLAG_BADROW);   // stored row id beyond the raw vectors handed to set_raw: reported, not ranked
    }
    const bool ok = i < c && r < n_raw;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
    if (ok) {
      const f4 *yp = reinterpret_cast<const f4 *>(raw + r * d) + 2 * h;
      const f4 *xp = qs4 + 2 * h;
      for (int c0 = 0; c0 < nchunk; c0 += 8) {
        f4 y[8][2];
#pragma unroll
        for (int cc = 0; cc < 8; ++cc)
          if (c0 + cc < nchunk) { y[cc][0] = yp[(c0 + cc) * 4]; y[cc][1] = yp[(c0 + cc) * 4 + 1]; }
#pragma unroll
        for (int cc = 0; cc < 8; ++cc)
          if (c0 + cc < nchunk) {
            const f4 x0 = xp[(c0 + cc) * 4], x1 = xp[(c0 + cc) * 4 + 1];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if constexpr (METRIC == METRIC_DOT) {
                acc[e] += x0[e] * y[cc][0][e];
                acc[4 + e] += x1[e] * y[cc][1][e];
              } else {
                const float d0 = x0[e] - y[cc][0][e], d1 = x1[e] - y[cc][1][e];
                acc[e] += d0 * d0;
                acc[4 + e] += d1 * d1;
              }
            }
          }
      }
    }
    float oth[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) oth[e] = __shfl_xor(acc[e], 1, 64);
    if (h == 0 && i < P) {
      float tot = 0.0f;
#pragma unroll
      for (int e = 0; e < 8; ++e) tot = tot + acc[e];
#pragma unroll
      for (int e = 0; e < 8; ++e) tot = tot + oth[e];
      tot = 0.0f + tot;      // `s + tot` of dist_exact_rt with an empty scalar tail
      key[i] = ok ? order_key(finish_metric<METRIC>(tot)) : 0xFFFFFFFFu;
      rid[i] = r; pos[i] = 0;
    }
  }
  __syncthreads();
  refine_write_cand_exact(cand_exact, key, keff, c, qi);
  refine_emit_topk(key, rid, pos, P, c, k, qi, out_ids, out_dists);
}

// The same refine from the index's lossless u8 copy of an integer-valued f32 column (index.h raw_u8): lane h of a pair reads the 8 bytes
// [8h, 8h+8) of every 16-element chunk -- a quarter of the row bytes --, widens them (v_cvt_f32_ubyte0..3: exact) as they are consumed
// and runs the SAME arithmetic on the same f32 values: per accumulator the chunks in ascending order, mul then add, the reference's
// fold.  The candidate ids are requested together with the count (the list is [nq][keff]; slots beyond the count are masked after the
// loads come back), so the kernel's dependent chain is ids -> rows -> rank.
template <int METRIC>
__global__ __launch_bounds__(256) void refine_u8_kernel(const float *__restrict__ q, int d, const uint8_t *__restrict__ raw, uint64_t n_raw,
                                                        const uint64_t *__restrict__ cand_rid, const uint32_t *__restrict__ cand_cnt, int keff,
                                                        int k, int P, uint64_t *__restrict__ out_ids, float *__restrict__ out_dists,
                                                        uint32_t *__restrict__ flags, float *__restrict__ cand_exact = nullptr) {
  static_assert(METRIC == METRIC_L2 || METRIC == METRIC_DOT, "squared L2 / dot");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint64_t *rid = reinterpret_cast<uint64_t *>(smem);
  uint32_t *key = reinterpret_cast<uint32_t *>(rid + P);
  uint32_t *pos = key + P;
  f4 *qs4 = reinterpret_cast<f4 *>(pos + P);      // [d / 4] the query
  const int qi = blockIdx.x;
  const int h = threadIdx.x & 1, slot = threadIdx.x >> 1;   // 128 candidates per round
  uint64_t r0 = ~0ull;
  if (slot < keff) r0 = cand_rid[(int64_t)qi * keff + slot];      // first round's ids: in flight with the count and the query
  const int c = (int)cand_cnt[qi];
  const f4 *qv4 = reinterpret_cast<const f4 *>(q + (int64_t)qi * d);
  for (int e = threadIdx.x; e < d / 4; e += 256) qs4[e] = qv4[e];
  __syncthreads();
  const int nchunk = d >> 4;
  for (int i0 = 0; i0 < P; i0 += 128) {
    const int i = i0 + slot;
    uint64_t r = ~0ull;
    if (i < c) {
      r = i0 == 0 ? r0 : cand_rid[(int64_t)qi * keff + i];
      if (r >= n_raw && h == 0) atomicOr(&flags[qi], FLAG_BADROW);   // stored row id beyond the raw vectors handed to set_raw: reported, not ranked
    }
    const bool ok = i < c && r < n_raw;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
    if (ok) {
      const uint2 *yp = reinterpret_cast<const uint2 *>(raw + r * d) + h;
      const f4 *xp = qs4 + 2 * h;
      for (int c0 = 0; c0 < nchunk; c0 += 8) {
        uint2 yb[8];
#pragma unroll
        for (int cc = 0; cc < 8; ++cc)
          if (c0 + cc < nchunk) yb[cc] = yp[(c0 + cc) * 2];
#pragma unroll
        for (int cc = 0; cc < 8; ++cc)
          if (c0 + cc < nchunk) {
            const f4 x0 = xp[(c0 + cc) * 4], x1 = xp[(c0 + cc) * 4 + 1];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float y0 = (float)((yb[cc].x >> (8 * e)) & 255u), y1 = (float)((yb[cc].y >> (8 * e)) & 255u);
              if constexpr (METRIC == METRIC_DOT) {
                acc[e] += x0[e] * y0;
                acc[4 + e] += x1[e] * y1;
              } else {
                const float d0 = x0[e] - y0, d1 = x1[e] - y1;
                acc[e] += d0 * d0;
                acc[4 + e] += d1 * d1;
              }
            }
          }
      }
    }
    float oth[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) oth[e] = __shfl_xor(acc[e], 1, 64);
    if (h == 0 && i < P) {
      float tot = 0.0f;
#pragma unroll
      for (int e = 0; e < 8; ++e) tot = tot + acc[e];
#pragma unroll
      for (int e = 0; e < 8; ++e) tot = tot + oth[e];
      tot = 0.0f + tot;      // `s + tot` of dist_exact_rt with an empty scalar tail
      key[i] = ok ? order_key(finish_metric<METRIC>(tot)) : 0xFFFFFFFFu;
      rid[i] = r; pos[i] = 0;
    }
  }
  __syncthreads();
  refine_write_cand_exact(cand_exact, key, keff, c, qi);
  refine_emit_topk(key, rid, pos, P, c, k, qi, out_ids, out_dists);
}

// ---- the lossless u8 refine source (index.h: raw_u8) ---------------------------------------------------------------------------------
// One pass over the f32 column: 16 elements per lane -> 16 bytes, and a flag if ANY element is not the widening of its byte (a fraction,
// a value outside [0, 255], NaN).  The flag decides whether the copy is kept.  -0.0 is accepted as the byte 0 (numpy's rint / clip
// pipelines leave it behind): the refine kernel then sees +0.0 where the column holds -0.0, and neither metric can tell -- squared L2
// forms d = x - y, and x - (-0.0) and x - (+0.0) differ at most in the sign of a zero that is squared next; dot forms x * y = -+0.0 and
// adds it to an accumulator that starts at +0.0 and therefore is never -0.0 (+0.0 + -0.0 = +0.0 in round-to-nearest), so the sum's
// bits do not depend on the sign of a zero addend.
__global__ __launch_bounds__(256) void raw_to_u8_kernel(const f4 *__restrict__ x, int64_t n16, uint4 *__restrict__ out, uint32_t *__restrict__ bad) {
  bool ok = true;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
    uint32_t w[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const f4 v = x[i * 4 + p];
      uint32_t pk = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t b = (v[e] >= 0.0f && v[e] < 256.0f) ? (uint32_t)v[e] : 0u;
        ok = ok && (float)b == v[e];
        pk |= b << (8 * e);
      }
      w[p] = pk;
    }
    out[i] = make_uint4(w[0], w[1], w[2], w[3]);
  }
  if (__any(!ok) && (threadIdx.x & 63) == 0) atomicOr(bad, 1u);
}

// Called by a refining search (not while its stream is being captured) and by lance_hip_index_prewarm.  Synchronises the stream once per
// index; afterwards one mutex-protected read.  Returns the u8 column or nullptr (f32 path).
static bool raw_compact_enabled() {
  static const bool off = getenv("LANCE_HIP_NO_RAW_COMPACT") != nullptr;      // A/B switch: always refine from the caller's column
  return !off;
}
// find_partitions over thousands of lists: the centroids' bf16 planes as constants of the index (index.h CqConst).  Built by the first
// uncaptured search of such an index or by lance_hip_index_prewarm; no memory for them, or a capture in progress: the call builds its own.
int coarse_planes_prepare(lance_hip_ctx *ctx, const lance_hip_index *ix_c, int scan_metric) {
  lance_hip_index *ix = const_cast<lance_hip_index *>(ix_c);
  std::lock_guard<std::mutex> lk(ix->lazy_mu);
  if (ix->cq || ctx->capturing) return LANCE_HIP_OK;
  auto *cq = new lance_hip_index::CqConst();
  bool ok = hipMalloc(reinterpret_cast<void **>(&cq->cpl), xform_coarse_planes_elems(ix->nlist, (int)ix->d) * 2) == hipSuccess;
  ok = ok && hipMalloc(reinterpret_cast<void **>(&cq->maxbits), 16) == hipSuccess;
  auto drop = [&]() { if (cq->cpl) (void)hipFree(cq->cpl); if (cq->maxbits) (void)hipFree(cq->maxbits); delete cq; };
  if (!ok) { (void)hipGetLastError(); drop(); return LANCE_HIP_OK; }
  if (lh::memset_async(cq->maxbits, 0, 16, ctx->stream) != hipSuccess ||
      xform_coarse_planes(ctx, scan_metric, ix->centroids, ix->nlist, (int)ix->d, cq->cpl, cq->maxbits) != LANCE_HIP_OK ||
      hipStreamSynchronize(ctx->stream) != hipSuccess) {      // (other contexts search the same index: published complete)
    drop();
    set_error("find_partitions: building the centroid planes of the index failed");
    return LANCE_HIP_ERUNTIME;
  }
  ix->cq = cq;
  return LANCE_HIP_OK;
}

const uint8_t *raw_compact_prepare(lance_hip_ctx *ctx, const lance_hip_index *ix_c) {
  lance_hip_index *ix = const_cast<lance_hip_index *>(ix_c);      // a cache attached to the index, like the scan constants
  if (!raw_compact_enabled() || ix->dtype != LANCE_HIP_F32 || !ix->raw || ix->n_raw == 0 || (ix->d & 15) != 0 ||
      (ix->metric != LANCE_HIP_L2 && ix->metric != LANCE_HIP_DOT) || (reinterpret_cast<uintptr_t>(ix->raw) & 15) != 0)
    return nullptr;
  std::lock_guard<std::mutex> lk(ix->lazy_mu);
  if (ix->raw_compact_state != 0) return ix->raw_compact_state > 0 ? ix->raw_u8 : nullptr;
  if (ctx->capturing) return nullptr;      // this capture keeps the f32 kernel; a later plain call builds the copy
  const uint64_t bytes = ix->n_raw * (uint64_t)ix->d;
  uint8_t *buf = nullptr;
  uint32_t *bad = ctx->scratch_t<uint32_t>("raw.compact_flag", 1);
  if (!bad || hipMalloc(reinterpret_cast<void **>(&buf), bytes) != hipSuccess) {
    (void)hipGetLastError();
    ix->raw_compact_state = -1;
    return nullptr;
  }
  uint32_t bad_h = 1;
  bool ran = lh::memset_async(bad, 0, 4, ctx->stream) == hipSuccess;
  if (ran) {
    const int64_t n16 = (int64_t)(bytes / 16);
    hipLaunchKernelGGL(raw_to_u8_kernel, dim3((unsigned)std::min<uint64_t>(cdiv((uint64_t)n16, 256), 1u << 20)), dim3(256), 0, ctx->stream,
                       static_cast<const f4 *>(ix->raw), n16, reinterpret_cast<uint4 *>(buf), bad);
    ran = hipGetLastError() == hipSuccess && hipMemcpyAsync(&bad_h, bad, 4, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
          hipStreamSynchronize(ctx->stream) == hipSuccess;
  }
  if (!ran || bad_h != 0) {
    (void)hipFree(buf);
    ix->raw_compact_state = -1;
    return nullptr;
  }
  ix->raw_u8 = buf;
  ix->raw_compact_state = 1;
  return buf;
}


// ------------------------------------------------------------------------------------
// Exact fallback: one wave per FLAGGED query re-runs the query the way the reference
// does -- per partition, rows in scan order through a max-heap with the semantics of
// Rust's std BinaryHeap as FlatIndex::search drives it (flat/index.rs:94-126: push while
// len < k, else replace the root only if root.dist > dist) -- so that the rows surviving
// an over-full tie at the boundary are the reference's.  Distances are computed by all 64
// lanes; a ballot pre-filters rows that cannot enter the heap (the root only decreases),
// lane 0 replays the surviving rows in order.  Partition heaps are merged into a running
// (dist, rowid)-sorted top list (the SortExec order).
struct ExactArgs {
  ScanArgs s;
  const uint64_t *row_ids;
  int k, refine;
  uint64_t *out_ids;
  float *out_dists;
  uint64_t *cand_rid;
  uint32_t *cand_cnt;
  uint32_t *n_fallback;  // statistics
};

template <int METRIC, int NBITS>
__global__ __launch_bounds__(64) void ivfpq_exact_kernel(ExactArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const ScanArgs &p = a.s;
  const int qi = blockIdx.x;
  if (!p.flags[qi]) return;
  const int lane = threadIdx.x;
  const int dpad = (p.d + 3) & ~3;
  constexpr int KC = NBITS == 4 ? 16 : 256;
  float *r = reinterpret_cast<float *>(smem);
  float *lut = r + dpad;
  uint64_t *trid = reinterpret_cast<uint64_t *>(lut + p.m * 256);   // sized for 8-bit in both cases
  uint32_t *tkey = reinterpret_cast<uint32_t *>(trid + p.keff);
  uint32_t *hk = tkey + p.keff;
  uint32_t *hp = hk + p.keff + 1;
  uint32_t *skey = hp + p.keff + 1;
  Pq4Shared q4;
  q4.sc = skey + 64;
  q4.fd = reinterpret_cast<float *>(q4.sc + 4);
  q4.qt = reinterpret_cast<uint8_t *>(q4.fd + max(200, p.keff));
  __shared__ int s_hlen, s_tcnt;
  if (lane == 0) { s_hlen = 0; s_tcnt = 0; }
  const float *qv = p.q + (int64_t)qi * p.d;
  const int m = p.m, sd = p.sd;
  __syncthreads();
  for (int pi = 0; pi < p.nprobes; ++pi) {
    const uint32_t part = p.probes[(int64_t)qi * p.nprobes + pi];
    const uint32_t off = p.part_offsets[part];
    const int np = (int)(p.part_offsets[part + 1] - off);
    if (np == 0) continue;
    __syncthreads();
    for (int t = lane; t < p.d; t += 64) {
      float rv = p.residual ? qv[t] - p.centroids[(int64_t)part * p.d + t] : qv[t];
      if (p.round_f16 && p.residual) rv = __half2float(__float2half_rn(rv));
      r[t] = rv;
    }
    __syncthreads();
    for (int idx = lane; idx < m * KC; idx += 64)
      lut[idx] = lut_entry_rt<METRIC>(&r[(idx / KC) * sd], p.codebook + (int64_t)idx * sd, sd, p.lanes32);
    if (lane == 0) s_hlen = 0;
    __syncthreads();
    const uint8_t *pcodes = p.codes + (int64_t)off * (NBITS == 4 ? m / 2 : m);
    if constexpr (NBITS == 4) { if (!p.allow) pq4_prelude<METRIC, 64>(lut, m, pcodes, np, p.keff, q4); }
    for (int base = 0; base < np; base += 64) {
      const int row = base + lane;
      uint32_t key = 0xFFFFFFFFu;
      bool cand = false;
      if (row < np) {
        float dist = 0.0f;
        if constexpr (NBITS == 4) {
          dist = p.allow ? pq4_masked_row<METRIC>(lut, pcodes + (int64_t)row * (m / 2), m) : pq4_row_distance<METRIC>(lut, m, pcodes, np, row, q4);
        } else {
          const uint8_t *rc = pcodes + (int64_t)row * m;
          for (int mm = 0; mm < m; ++mm) dist += lut[mm * 256 + rc[mm]];
          if constexpr (METRIC == METRIC_DOT) dist = dist - ((float)m - 1.0f);
        }
        key = order_key(dist);
        const bool in_range = !p.has_range || (key >= p.lo_key && key < p.hi_key);
        cand = in_range && row_allowed(p.allow, off + (uint32_t)row) && (s_hlen < p.keff || key < hk[0]);
      }
      const uint64_t mask = __ballot(cand);
      skey[lane] = key;
      __syncthreads();
      if (lane == 0 && mask) {
        int hl = s_hlen;
        uint64_t mm = mask;
        while (mm) {
          const int b = __ffsll((long long)mm) - 1;
          mm &= mm - 1;
          const uint32_t kk = skey[b];
          if (hl < p.keff) {
            heap_push(hk, hp, hl, kk, off + (uint32_t)(base + b));
          } else if (hk[0] > kk) {
            heap_pop(hk, hp, hl);
            heap_push(hk, hp, hl, kk, off + (uint32_t)(base + b));
          }
        }
        s_hlen = hl;
      }
      __syncthreads();
    }
    // merge this partition's heap into the running (key, rowid)-sorted top list
    if (lane == 0) {
      int tc = s_tcnt;
      for (int i = 0; i < s_hlen; ++i) {
        const uint32_t kk = hk[i];
        const uint64_t rr = a.row_ids[hp[i]];
        if (tc == p.keff) {
          const uint32_t wk = tkey[tc - 1];
          const uint64_t wr = trid[tc - 1];
          if (!(kk < wk || (kk == wk && rr < wr))) continue;
        }
        int pos = tc < p.keff ? tc : p.keff - 1;
        while (pos > 0) {
          const uint32_t pk = tkey[pos - 1];
          const uint64_t pr = trid[pos - 1];
          if (pk < kk || (pk == kk && pr < rr)) break;
          tkey[pos] = pk; trid[pos] = pr;
          --pos;
        }
        tkey[pos] = kk; trid[pos] = rr;
        if (tc < p.keff) ++tc;
      }
      s_tcnt = tc;
    }
    __syncthreads();
  }
  __syncthreads();
  const int got = s_tcnt;
  if (a.refine) {
    for (int i = lane; i < p.keff; i += 64) a.cand_rid[(int64_t)qi * p.keff + i] = i < got ? trid[i] : ~0ull;
    if (lane == 0) a.cand_cnt[qi] = (uint32_t)got;
  } else {
    for (int i = lane; i < a.k; i += 64) {
      a.out_ids[(int64_t)qi * a.k + i] = i < got ? trid[i] : ~0ull;
      a.out_dists[(int64_t)qi * a.k + i] = i < got ? key_to_float(tkey[i]) : INFINITY;
    }
  }
  if (lane == 0) { p.flags[qi] = 0; atomicAdd(a.n_fallback, 1u); }
}

__global__ void fill_u32_kernel(uint32_t *p, uint32_t v, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = v;
}

// ------------------------------------------------------------------------------------

template <int SD, int METRIC>
static void launch_scan_mu(lance_hip_ctx *ctx, const ScanArgs &a, int grid, size_t lds) {
  const int mu = (a.m % 16 == 0) ? a.m / 16 : 0;
  // (a register-resident-codebook persistent variant was measured 2x slower -- 181 VGPRs leave 2 waves/SIMD and the
  // LDS gathers need occupancy -- and removed)
  switch (mu) {
    case 1: hipLaunchKernelGGL((ivfpq_scan_kernel<SD, METRIC, 1>), dim3(grid), dim3(256), lds, ctx->stream, a); break;
    case 2: hipLaunchKernelGGL((ivfpq_scan_kernel<SD, METRIC, 2>), dim3(grid), dim3(256), lds, ctx->stream, a); break;
    case 4: hipLaunchKernelGGL((ivfpq_scan_kernel<SD, METRIC, 4>), dim3(grid), dim3(256), lds, ctx->stream, a); break;
    case 6: hipLaunchKernelGGL((ivfpq_scan_kernel<SD, METRIC, 6>), dim3(grid), dim3(256), lds, ctx->stream, a); break;
    default: hipLaunchKernelGGL((ivfpq_scan_kernel<SD, METRIC, 0>), dim3(grid), dim3(256), lds, ctx->stream, a); break;
  }
}

template <int METRIC>
static void launch_scan(lance_hip_ctx *ctx, const ScanArgs &a, int grid, size_t lds) {
  const bool cb_aligned = (reinterpret_cast<uintptr_t>(a.codebook) & 15) == 0;
  const bool codes_aligned = (reinterpret_cast<uintptr_t>(a.codes) & 15) == 0;
  if (cb_aligned && codes_aligned) {
    switch (a.sd) {
      case 4: launch_scan_mu<4, METRIC>(ctx, a, grid, lds); return;
      case 8: launch_scan_mu<8, METRIC>(ctx, a, grid, lds); return;
      case 16: launch_scan_mu<16, METRIC>(ctx, a, grid, lds); return;
      default: break;
    }
  }
  ScanArgs b = a;
  hipLaunchKernelGGL((ivfpq_scan_kernel<0, METRIC, 0>), dim3(grid), dim3(256), lds, ctx->stream, b);
}

// The whole query pipeline, enqueued on ctx->stream.  flags_out: device [nq] (zeroed here).
// valid ids of a search result row are a prefix (empty slots hold ~0): their count, for the refine kernels
__global__ __launch_bounds__(256) void cand_count_kernel(const uint64_t *__restrict__ ids, uint32_t nq, uint32_t keff, uint32_t *__restrict__ cnt) {
  const uint32_t qi = blockIdx.x * 256 + threadIdx.x;
  if (qi >= nq) return;
  uint32_t c = 0;
  for (uint32_t i = 0; i < keff; ++i) c += ids[(uint64_t)qi * keff + i] != ~0ull ? 1u : 0u;
  cnt[qi] = c;
}

// refine of cand_rid [nq][keff] (cand_cnt [nq] valid entries each): exact distances in the index's metric against the raw column, then
// (dist, rowid) order, fetch k; cand_exact (optional): every candidate's exact distance in the list's order
static int launch_refine(lance_hip_ctx *ctx, const lance_hip_index *ix, const float *q, uint32_t nq, int d, const uint64_t *cand_rid, const uint32_t *cand_cnt,
                         uint32_t keff, uint32_t k, uint64_t *ids, float *dists, uint32_t *flags, float *cand_exact) {
  {
    const int P = next_pow2(std::max((int)keff, 64));
    ScopedTimer t(ctx, "refine");
    const float *rawf = static_cast<const float *>(ix->raw);
    const __half *rawh = static_cast<const __half *>(ix->raw);
    // flat_knn on the taken rows uses the index's metric with the ORIGINAL query (q_orig = widened q for f16)
    const int8_t *rawi = static_cast<const int8_t *>(ix->raw);
    if (ix->dtype == LANCE_HIP_I8 && ix->metric == LANCE_HIP_COSINE)
      hipLaunchKernelGGL((refine_kernel<METRIC_COSINE, int8_t>), dim3(nq), dim3(256), (size_t)P * 16, ctx->stream, q, d, rawi, ix->n_raw,
                         cand_rid, cand_cnt, (int)keff, (int)k, P, ids, dists, flags, cand_exact);
    else if (ix->dtype == LANCE_HIP_I8 && ix->metric == LANCE_HIP_DOT)
      hipLaunchKernelGGL((refine_kernel<METRIC_DOT, int8_t>), dim3(nq), dim3(256), (size_t)P * 16, ctx->stream, q, d, rawi, ix->n_raw,
                         cand_rid, cand_cnt, (int)keff, (int)k, P, ids, dists, flags, cand_exact);
    else if (ix->dtype == LANCE_HIP_I8)
      hipLaunchKernelGGL((refine_kernel<METRIC_L2, int8_t>), dim3(nq), dim3(256), (size_t)P * 16, ctx->stream, q, d, rawi, ix->n_raw,
                         cand_rid, cand_cnt, (int)keff, (int)k, P, ids, dists, flags, cand_exact);
    else if (ix->dtype == LANCE_HIP_F16 && ix->metric == LANCE_HIP_COSINE)
      hipLaunchKernelGGL((refine_kernel<METRIC_COSINE, __half, true>), dim3(nq), dim3(256), (size_t)P * 16, ctx->stream, q, d, rawh, ix->n_raw,
                         cand_rid, cand_cnt, (int)keff, (int)k, P, ids, dists, flags, cand_exact);
    else if (ix->dtype == LANCE_HIP_F16 && ix->metric == LANCE_HIP_DOT)
      hipLaunchKernelGGL((refine_kernel<METRIC_DOT, __half, true>), dim3(nq), dim3(256), (size_t)P * 16, ctx->stream, q, d, rawh, ix->n_raw,
                         cand_rid, cand_cnt, (int)keff, (int)k, P, ids, dists, flags, cand_exact);
    else if (ix->dtype == LANCE_HIP_F16)
      hipLaunchKernelGGL((refine_kernel<METRIC_L2, __half>), dim3(nq), dim3(256), (size_t)P * 16, ctx->stream, q, d, rawh, ix->n_raw,
                         cand_rid, cand_cnt, (int)keff, (int)k, P, ids, dists, flags, cand_exact);
    else if (ix->metric == LANCE_HIP_COSINE)
      hipLaunchKernelGGL((refine_kernel<METRIC_COSINE, float>), dim3(nq), dim3(256),
                         (size_t)P * 16 + (refine_wide_rows(d) ? (size_t)d * 4 : 0), ctx->stream, q, d, rawf, ix->n_raw,
                         cand_rid, cand_cnt, (int)keff, (int)k, P, ids, dists, flags, cand_exact);
    else {
      // f32 rows of whole 16-element chunks, 16-byte aligned: two lanes per candidate with the row in flight (LANCE_HIP_REFINE_V1=1: A/B)
      static const bool v1 = getenv("LANCE_HIP_REFINE_V1") != nullptr;
      const bool pair = !v1 && (d & 15) == 0 && ((reinterpret_cast<uintptr_t>(rawf) | reinterpret_cast<uintptr_t>(q)) & 15) == 0;
      const size_t lds_pair = (size_t)P * 16 + (size_t)d * 4;
      const uint8_t *raw8 = pair ? raw_compact_prepare(ctx, ix) : nullptr;      // integer-valued column: its lossless u8 copy (index.h)
      if (raw8) {
        ScopedTimer t8(ctx, "refine_u8");      // the same launch under its own name: tests assert which source the refine read
        if (ix->metric == LANCE_HIP_DOT)
          hipLaunchKernelGGL((refine_u8_kernel<METRIC_DOT>), dim3(nq), dim3(256), lds_pair, ctx->stream, q, d, raw8, ix->n_raw, cand_rid,
                             cand_cnt, (int)keff, (int)k, P, ids, dists, flags, cand_exact);
        else
          hipLaunchKernelGGL((refine_u8_kernel<METRIC_L2>), dim3(nq), dim3(256), lds_pair, ctx->stream, q, d, raw8, ix->n_raw, cand_rid,
                             cand_cnt, (int)keff, (int)k, P, ids, dists, flags, cand_exact);
      } else if (pair && ix->metric == LANCE_HIP_DOT)
        hipLaunchKernelGGL((refine_pair_kernel<METRIC_DOT>), dim3(nq), dim3(256), lds_pair, ctx->stream, q, d, rawf, ix->n_raw, cand_rid, cand_cnt,
                           (int)keff, (int)k, P, ids, dists, flags, cand_exact);
      else if (pair)
        hipLaunchKernelGGL((refine_pair_kernel<METRIC_L2>), dim3(nq), dim3(256), lds_pair, ctx->stream, q, d, rawf, ix->n_raw, cand_rid, cand_cnt,
                           (int)keff, (int)k, P, ids, dists, flags, cand_exact);
      else if (ix->metric == LANCE_HIP_DOT)
        hipLaunchKernelGGL((refine_kernel<METRIC_DOT, float>), dim3(nq), dim3(256), (size_t)P * 16, ctx->stream, q, d, rawf, ix->n_raw,
                           cand_rid, cand_cnt, (int)keff, (int)k, P, ids, dists, flags, cand_exact);
      else
        hipLaunchKernelGGL((refine_kernel<METRIC_L2, float>), dim3(nq), dim3(256), (size_t)P * 16, ctx->stream, q, d, rawf, ix->n_raw,
                           cand_rid, cand_cnt, (int)keff, (int)k, P, ids, dists, flags, cand_exact);
    }
  }
  LH_CHECK_HIP(hipGetLastError());
  return LANCE_HIP_OK;
}

static int ivfpq_search_enqueue_impl(lance_hip_ctx *ctx, const lance_hip_index *ix, const float *q, uint32_t nq, uint32_t k,
                                     uint32_t nprobes, uint32_t refine_factor, int has_range, float lower, float upper,
                                     uint64_t *ids, float *dists, uint32_t **flags_out, const uint32_t *allow);

// A batch is ~25 small launches (coarse quantiser, two groupings, bound pass, residuals, filter scan, rescan, merge, exact
// replay, refine) and a handful of memsets: at C3 a third of a 1000-query batch's wall time was the gaps between them
// (profiles/r03_c3_search_breakdown.json).  The second call with the same arguments (index, buffers, shape) is captured into
// a HIP graph, later ones replay it with one hipGraphLaunch (LANCE_HIP_GRAPH=0 keeps every call on the plain path).  The first call always runs uncaptured: it
// sizes the scratch arena (growth would need hipMalloc + a stream sync, neither of which a capture allows) and builds the
// index's lazy constants.  Timing runs (HIP events per kernel) and the diagnostic switches that synchronise mid-pipeline stay
// on the plain path.
int ivfpq_search_enqueue(lance_hip_ctx *ctx, const lance_hip_index *ix, const float *q, uint32_t nq, uint32_t k,
                         uint32_t nprobes, uint32_t refine_factor, int has_range, float lower, float upper,
                         uint64_t *ids, float *dists, uint32_t **flags_out, const uint32_t *allow) {
  static const bool on = [] {
    const char *e = getenv("LANCE_HIP_GRAPH");     // default on (r04d: +8 % on the C2 bench line); "0" switches it off
    return !(e && e[0] == '0') && !getenv("LANCE_HIP_Q_STATS") && !getenv("LANCE_HIP_PM_PROF") && !getenv("LANCE_HIP_QT_PROF");
  }();
  if (!on || ctx->timing || nq == 0 || ix->ephemeral)      // ephemeral: the one-call index of lance_hip_pq_scan_topk
    return ivfpq_search_enqueue_impl(ctx, ix, q, nq, k, nprobes, refine_factor, has_range, lower, upper, ids, dists, flags_out, allow);
  struct Key {
    const void *ix; uint64_t serial; const void *raw; uint64_t n_raw, raw_gen; const void *q, *ids, *dists, *allow;
    uint32_t nq, k, nprobes, rf; int has_range; float lo, hi;
  } key;
  memset(&key, 0, sizeof(key));
  key.ix = ix; key.serial = ix->serial; key.raw = ix->raw; key.n_raw = ix->n_raw; key.raw_gen = ix->raw_gen; key.q = q; key.ids = ids; key.dists = dists; key.allow = allow;
  key.nq = nq; key.k = k; key.nprobes = nprobes; key.rf = refine_factor; key.has_range = has_range; key.lo = lower; key.hi = upper;
  const std::string ks(reinterpret_cast<const char *>(&key), sizeof(key));
  auto plain = [&]() { return ivfpq_search_enqueue_impl(ctx, ix, q, nq, k, nprobes, refine_factor, has_range, lower, upper, ids, dists, flags_out, allow); };
  // Life of a key: first call -> plain path, remembered in the bounded `graph_seen_once` FIFO (it sized the scratch arena and built the
  // index's lazy constants); second call -> captured, instantiated, launched; later calls -> replayed.  A caller that never repeats a
  // call (fresh output buffers every time) therefore never creates a graph entry, and a full cache evicts ONE entry -- a failed one
  // first (no exec: nothing to wait for), else the least recently used -- instead of dropping every valid graph.
  constexpr size_t GRAPH_CACHE_MAX = 64, SEEN_ONCE_MAX = 128;
  auto it = ctx->graphs.find(ks);
  if (it != ctx->graphs.end()) {
    GraphEntry &e = it->second;
    if (!e.exec) return plain();      // a capture of this key failed before
    e.last_use = ++ctx->graph_tick;
    LH_CHECK_HIP(hipGraphLaunch(e.exec, ctx->stream));
    ctx->count_stage("graph_replay");
    for (const char *nm : e.paths) ++ctx->stage_counts[nm];
    ctx->last_replay_counter = e.replay;
    if (flags_out) *flags_out = e.flags;
    return LANCE_HIP_OK;
  }
  {
    auto so = std::find(ctx->graph_seen_once.begin(), ctx->graph_seen_once.end(), ks);
    if (so == ctx->graph_seen_once.end()) {
      // remembered AFTER the call: a scratch slot that grows during it drops the cache and this FIFO with it -- the call that sized the
      // arena is the one that counts as "seen once"
      const int prc = plain();
      if (prc == LANCE_HIP_OK) {
        if (ctx->graph_seen_once.size() < SEEN_ONCE_MAX) ctx->graph_seen_once.push_back(ks);
        else { ctx->graph_seen_once[ctx->graph_seen_next] = ks; ctx->graph_seen_next = (ctx->graph_seen_next + 1) % SEEN_ONCE_MAX; }
      }
      return prc;
    }
    so->clear();      // promoted: the slot is reused by the FIFO in its own time
  }
  if (ctx->graphs.size() >= GRAPH_CACHE_MAX) {
    auto victim = ctx->graphs.end();
    for (auto g = ctx->graphs.begin(); g != ctx->graphs.end(); ++g) {
      if (!g->second.exec) { victim = g; break; }
      if (victim == ctx->graphs.end() || g->second.last_use < victim->second.last_use) victim = g;
    }
    if (victim != ctx->graphs.end()) {
      if (victim->second.exec) {      // its last replay may still be in flight on this stream
        LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        (void)hipGraphExecDestroy(victim->second.exec);
      }
      ctx->graphs.erase(victim);
    }
  }
  uint32_t *fl = nullptr;
  hipGraph_t g = nullptr;
  if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    (void)hipGetLastError();
    ctx->graphs[ks].failed = true;
    return plain();
  }
  std::vector<const char *> paths;
  ctx->capturing = true;
  ctx->capture_paths = &paths;
  const int rc = ivfpq_search_enqueue_impl(ctx, ix, q, nq, k, nprobes, refine_factor, has_range, lower, upper, ids, dists, &fl, allow);
  ctx->capturing = false;
  ctx->capture_paths = nullptr;
  const hipError_t ee = hipStreamEndCapture(ctx->stream, &g);
  hipGraphExec_t ex = nullptr;
  if (rc == LANCE_HIP_OK && ee == hipSuccess && g && hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) == hipSuccess && ex) {
    (void)hipGraphDestroy(g);
    GraphEntry &e = ctx->graphs[ks];
    e.exec = ex; e.flags = fl; e.replay = ctx->last_replay_counter; e.last_use = ++ctx->graph_tick; e.paths = std::move(paths);
    LH_CHECK_HIP(hipGraphLaunch(ex, ctx->stream));
    ctx->count_stage("graph_capture");
    if (flags_out) *flags_out = fl;
    return LANCE_HIP_OK;
  }
  if (g) (void)hipGraphDestroy(g);
  (void)hipGetLastError();
  ctx->graphs[ks].failed = true;
  // nothing was executed (the launches went into the discarded graph) and the stage counters counted a call that never ran: undo them,
  // then run the batch on the plain path.  The failed capture may have left its reason in the thread's error string (a scratch slot that
  // would have had to grow): the plain run below succeeds or sets its own.
  for (const char *nm : paths) --ctx->stage_counts[nm];
  // A capture can also be invalidated from OUTSIDE this call -- another host thread freeing device memory (an index going out of scope: hipFree
  // synchronises the device, which a capturing stream must not be part of): make sure the stream has left capture mode and that no error of the
  // dead capture is left for the plain run's first launch to report (tests/test_zz_gpu_threads.py after the large-index tests, gpurun r06zi)
  {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(ctx->stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone) {
      hipGraph_t g2 = nullptr;
      (void)hipStreamEndCapture(ctx->stream, &g2);
      if (g2) (void)hipGraphDestroy(g2);
    }
    for (int i = 0; i < 4 && hipGetLastError() != hipSuccess; ++i) {}
  }
  set_error("");
  int prc = plain();
  if (prc == LANCE_HIP_ERUNTIME) {      // once more on a drained stream
    (void)hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < 4 && hipGetLastError() != hipSuccess; ++i) {}
    set_error("");
    prc = plain();
  }
  return prc;
}

static int ivfpq_search_enqueue_impl(lance_hip_ctx *ctx, const lance_hip_index *ix, const float *q, uint32_t nq, uint32_t k,
                                     uint32_t nprobes, uint32_t refine_factor, int has_range, float lower, float upper,
                                     uint64_t *ids, float *dists, uint32_t **flags_out, const uint32_t *allow) {
  LH_REQUIRE(k > 0, "search: k must be > 0");
  if (nprobes > ix->nlist) nprobes = ix->nlist;
  LH_REQUIRE(ix->nlist <= 8192 || nprobes <= 2048, "search: nprobes=%u > 2048 with more than 8192 partitions is not supported", nprobes);
  LH_REQUIRE(nprobes > 0, "search: nprobes must be > 0");
  const uint32_t rf = refine_factor == 0 ? 1 : refine_factor;
  const uint64_t keff64 = (uint64_t)k * rf;
  // the reference has no limit on k * refine_factor; here the exact kernel's heap and the refine sort live in LDS (8192 entries)
  LH_REQUIRE(keff64 <= 8192, "search: k * refine_factor = %llu > 8192 is not supported", (unsigned long long)keff64);
  const uint32_t keff = (uint32_t)keff64;
  const bool fast = keff <= (uint32_t)SCAN_MAX_KEFF;  // larger k: every query takes the exact (slow) kernel
  const bool do_refine = refine_factor >= 1;  // Some(rf): re-rank even when rf == 1 (scanner.rs:2884)
  LH_REQUIRE(!do_refine || ix->raw != nullptr, "search: refine_factor needs raw vectors (lance_hip_index_set_raw)");
  const int d = (int)ix->d, m = (int)ix->m, sd = d / m, nlist = (int)ix->nlist;
  const int scan_metric = ix->metric == LANCE_HIP_COSINE ? LANCE_HIP_L2 : ix->metric;

  uint32_t *flags = ctx->scratch_t<uint32_t>("search.flags", (size_t)nq + 1);
  if (!flags) return LANCE_HIP_ENOMEM;
  LH_CHECK_HIP(lh::memset_async(flags, 0, ((size_t)nq + 1) * 4, ctx->stream));
  uint32_t *n_fallback = flags + nq;
  ctx->last_replay_counter = n_fallback;
  if (flags_out) *flags_out = flags;
  if (nq == 0) return LANCE_HIP_OK;

  const float *qs = q;
  if (ix->metric == LANCE_HIP_COSINE) {  // knn.rs:495-498
    float *qn = ctx->scratch_t<float>("search.qnorm", (size_t)nq * d);
    if (!qn) return LANCE_HIP_ENOMEM;
    LH_TRY(launch_normalize(ctx, q, (int64_t)nq, d, qn, ix->dtype == LANCE_HIP_F16));   // an f16 key is normalised in f16 arithmetic
    qs = qn;
  }
  // coarse quantiser: all distances, then per-query partial sort
  float *matrix = ctx->scratch_t<float>("search.matrix", (size_t)nq * nlist);
  uint32_t *probes = ctx->scratch_t<uint32_t>("search.probes", (size_t)nq * nprobes);
  if (!matrix || !probes) return LANCE_HIP_ENOMEM;
  const bool coarse_l32 = ix->dtype == LANCE_HIP_F16 && scan_metric == LANCE_HIP_DOT && d > 16;
  if (coarse_mfma_supported(scan_metric, d, nq, (uint32_t)nlist, nprobes, coarse_l32, qs, ix->centroids)) {
    // kmeans.rs:1134-1158 on the matrix cores: bf16x3 surrogate matrix, exact re-check of the nprobes + few candidates
    const uint16_t *cpl = nullptr;
    const uint32_t *cmb = nullptr;
    if (coarse_groups_shape(d, (uint32_t)nlist)) {      // thousands of lists: the centroids' bf16 planes are constants of the index
      LH_TRY(coarse_planes_prepare(ctx, ix, scan_metric));
      if (ix->cq) { cpl = ix->cq->cpl; cmb = ix->cq->maxbits; }
    }
    LH_TRY(find_partitions_mfma(ctx, scan_metric, qs, nq, d, ix->centroids, (uint32_t)nlist, nprobes, matrix, probes, nullptr, cpl, cmb));
  } else {
    PairwiseArgs pa;
    pa.x = qs; pa.n = nq; pa.ldx = d; pa.cent = ix->centroids; pa.k = nlist; pa.matrix = matrix;
    pa.lanes32 = coarse_l32;
    LH_TRY(launch_dist_matrix(ctx, pa, d, scan_metric, 1));
    ScopedTimer t(ctx, "select_probes");
    launch_select_probes(ctx, matrix, nlist, (int)nprobes, (int)nq, probes, nullptr);
  }
  uint64_t *cand_rid = nullptr;
  uint32_t *cand_cnt = nullptr;
  if (do_refine) {
    cand_rid = ctx->scratch_t<uint64_t>("search.cand_rid", (size_t)nq * keff);
    cand_cnt = ctx->scratch_t<uint32_t>("search.cand_cnt", nq);
    if (!cand_rid || !cand_cnt) return LANCE_HIP_ENOMEM;
  }
  // scan: partition-major (two queries per LDS gather) when the batch is large enough to pair queries,
  // query-major otherwise
  static const bool no_pm = getenv("LANCE_HIP_NO_PM") != nullptr;
  // (tiled shapes, M >= 48: a work item's table build is as long as a query-major workgroup's, so sharing it pays earlier)
  const uint64_t pm_min_pairs = qscan_tiled_shape(m, sd) ? 2048 : 4096;
  const bool use_pm = fast && !no_pm && pm_supported(ix, keff, has_range, nq, nprobes) && (uint64_t)nq * nprobes >= pm_min_pairs;
  int nsplit = 1;
  if (nq < (uint32_t)(2 * ctx->num_cus)) {
    nsplit = (int)std::min<uint64_t>({8ull, (uint64_t)nprobes, cdiv(2ull * ctx->num_cus, nq)});
    if (nsplit < 1) nsplit = 1;
  }
  const size_t nblk = (size_t)nq * nsplit;
  uint32_t *ckeys = nullptr, *cpos = nullptr, *ccnt = nullptr;
  if (!use_pm) {
    ckeys = ctx->scratch_t<uint32_t>("search.ckeys", nblk * SCAN_LCAP);
    cpos = ctx->scratch_t<uint32_t>("search.cpos", nblk * SCAN_LCAP);
    ccnt = ctx->scratch_t<uint32_t>("search.ccnt", nblk);
    if (!ckeys || !cpos || !ccnt) return LANCE_HIP_ENOMEM;
  }
  ScanArgs a;
  {
    a.q = qs; a.probes = probes; a.centroids = ix->centroids; a.codebook = ix->codebook;
    a.part_offsets = ix->part_offsets; a.codes = ix->codes;
    a.d = d; a.m = m; a.sd = sd; a.nprobes = (int)nprobes; a.nsplit = nsplit; a.keff = (int)keff;
    a.nbits = (int)ix->nbits;
    a.residual = scan_metric == LANCE_HIP_L2 ? 1 : 0;
    a.round_f16 = ix->dtype == LANCE_HIP_F16 ? 1 : 0;
    a.lanes32 = (ix->dtype == LANCE_HIP_F16 && scan_metric == LANCE_HIP_DOT && sd > 16) ? 1 : 0;
    a.has_range = has_range;
    a.allow = allow;
    a.lo_key = 0; a.hi_key = 0xFFFFFFFFu;
    if (has_range) {
      uint32_t lb, ub;
      memcpy(&lb, &lower, 4); memcpy(&ub, &upper, 4);
      a.lo_key = (lb & 0x80000000u) ? ~lb : (lb | 0x80000000u);
      a.hi_key = (ub & 0x80000000u) ? ~ub : (ub | 0x80000000u);
    }
    a.out_keys = ckeys; a.out_pos = cpos; a.out_cnt = ccnt; a.flags = flags;
#ifdef LH_TIMING_EXPERIMENTS
    { static const int ablate = getenv("LANCE_HIP_ABLATE") ? atoi(getenv("LANCE_HIP_ABLATE")) : 0; a.ablate = ablate; }
#endif
    const int dpad = (d + 3) & ~3;
    const size_t lds = (size_t)dpad * 4 + (size_t)m * 256 * 4 + (size_t)SCAN_CAP * 8 + 256 * 4 + 8 * 4;
    LH_REQUIRE(lds <= 160 * 1024, "search: LUT of %d sub-vectors does not fit in LDS", m);
    if (use_pm) {
      LH_TRY(ivfpq_scan_merge_pm(ctx, ix, qs, nq, probes, nprobes, keff, k, do_refine, ids, dists, cand_rid, cand_cnt, flags, allow));
    } else if (fast && ix->nbits == 4) {
      const size_t lds4 = (size_t)dpad * 4 + (size_t)m * 16 * 4 + (size_t)SCAN_CAP * 8 + 256 * 4 + 8 * 4 + 16 + 256 * 4 + (size_t)m * 16 + 16;
      ScopedTimer t(ctx, "ivfpq_scan");
      if (scan_metric == LANCE_HIP_DOT) hipLaunchKernelGGL((ivfpq_scan4_kernel<METRIC_DOT>), dim3((unsigned)nblk), dim3(256), lds4, ctx->stream, a);
      else hipLaunchKernelGGL((ivfpq_scan4_kernel<METRIC_L2>), dim3((unsigned)nblk), dim3(256), lds4, ctx->stream, a);
    } else if (fast) {
      ScopedTimer t(ctx, "ivfpq_scan");
      if (scan_metric == LANCE_HIP_DOT) launch_scan<METRIC_DOT>(ctx, a, (int)nblk, lds);
      else launch_scan<METRIC_L2>(ctx, a, (int)nblk, lds);
    } else {
      hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)cdiv(nq, 256)), dim3(256), 0, ctx->stream, flags, FLAG_OVERFLOW, (int64_t)nq);
    }
  }
  // merge (+ refine)
  if (fast && !use_pm) {
    MergeArgs ma;
    ma.keys = ckeys; ma.pos = cpos; ma.cnt = ccnt; ma.row_ids = ix->row_ids; ma.part_offsets = ix->part_offsets;
    ma.nlist = nlist; ma.nsplit = nsplit; ma.keff = (int)keff; ma.k = (int)k;
    ma.P = next_pow2(std::max(nsplit * SCAN_LCAP, 64));
    ma.refine = do_refine ? 1 : 0;
    ma.out_ids = ids; ma.out_dists = dists; ma.cand_rid = cand_rid; ma.cand_cnt = cand_cnt; ma.flags = flags;
    ScopedTimer t(ctx, "ivfpq_merge");
    hipLaunchKernelGGL(ivfpq_merge_kernel, dim3(nq), dim3(256), (size_t)ma.P * 16, ctx->stream, ma);
  }
  {
    // exact replay of flagged queries (no-op workgroups for the rest)
    ExactArgs ea;
    ea.s = a; ea.row_ids = ix->row_ids; ea.k = (int)k; ea.refine = do_refine ? 1 : 0;
    ea.out_ids = ids; ea.out_dists = dists; ea.cand_rid = cand_rid; ea.cand_cnt = cand_cnt; ea.n_fallback = n_fallback;
    const int dpad = (d + 3) & ~3;
    const size_t lds = (size_t)dpad * 4 + (size_t)m * 256 * 4 + (size_t)keff * 12 + ((size_t)keff + 1) * 8 + 64 * 4 + 16 +
                       16 + (size_t)std::max<uint32_t>(200, keff) * 4 + (size_t)m * 16 + 16;
    LH_REQUIRE(lds <= 160 * 1024, "search: exact kernel does not fit in LDS (m=%d, k*refine=%u)", m, keff);
    ScopedTimer t(ctx, "ivfpq_exact");
    if (ix->nbits == 4) {
      if (scan_metric == LANCE_HIP_DOT) hipLaunchKernelGGL((ivfpq_exact_kernel<METRIC_DOT, 4>), dim3(nq), dim3(64), lds, ctx->stream, ea);
      else hipLaunchKernelGGL((ivfpq_exact_kernel<METRIC_L2, 4>), dim3(nq), dim3(64), lds, ctx->stream, ea);
    } else {
      if (scan_metric == LANCE_HIP_DOT) hipLaunchKernelGGL((ivfpq_exact_kernel<METRIC_DOT, 8>), dim3(nq), dim3(64), lds, ctx->stream, ea);
      else hipLaunchKernelGGL((ivfpq_exact_kernel<METRIC_L2, 8>), dim3(nq), dim3(64), lds, ctx->stream, ea);
    }
  }
  if (do_refine) LH_TRY(launch_refine(ctx, ix, q, nq, d, cand_rid, cand_cnt, keff, k, ids, dists, flags, nullptr));
  LH_CHECK_HIP(hipGetLastError());
  return LANCE_HIP_OK;
}

// find_partitions on f32 operands (kmeans.rs:1134-1158): all distances, per-query partial sort.  lanes32: the operands are
// widened f16 values and the metric is dot -- 32 lane accumulators.  Synchronises the stream.
int find_partitions_f32(lance_hip_ctx *ctx, int metric, const float *qf, uint32_t nq, uint32_t d, const float *cf, uint32_t nlist,
                        uint32_t nprobes, uint32_t *part_ids, float *dists, bool lanes32) {
  if (nprobes > nlist) nprobes = nlist;
  LH_REQUIRE(nlist <= 8192 || nprobes <= 2048, "find_partitions: nprobes=%u > 2048 with more than 8192 partitions is not supported", nprobes);
  if (nq == 0 || nprobes == 0) return LANCE_HIP_OK;
  float *matrix = ctx->scratch_t<float>("search.matrix", (size_t)nq * nlist);
  if (!matrix) return LANCE_HIP_ENOMEM;
  const int km = metric == LANCE_HIP_COSINE ? LANCE_HIP_L2 : metric;
  if (coarse_mfma_supported(km, (int)d, nq, nlist, nprobes, lanes32, qf, cf)) {
    LH_TRY(find_partitions_mfma(ctx, km, qf, nq, (int)d, cf, nlist, nprobes, matrix, part_ids, dists));
  } else {
    PairwiseArgs pa;
    pa.x = qf; pa.n = nq; pa.ldx = d;
    pa.cent = cf; pa.k = (int)nlist; pa.matrix = matrix;
    pa.lanes32 = lanes32;
    LH_TRY(launch_dist_matrix(ctx, pa, (int)d, km, 1));
    launch_select_probes(ctx, matrix, (int)nlist, (int)nprobes, (int)nq, part_ids, dists);
  }
  LH_CHECK_HIP(hipGetLastError());
  LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return LANCE_HIP_OK;
}

static int check_flags(lance_hip_ctx *ctx, const uint32_t *flags, uint32_t nq) {
  std::vector<uint32_t> fh(nq);
  LH_CHECK_HIP(hipMemcpyAsync(fh.data(), flags, (size_t)nq * 4, hipMemcpyDeviceToHost, ctx->stream));
  LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  uint32_t n_over = 0, n_amb = 0, n_bad = 0;
  for (uint32_t i = 0; i < nq; ++i) { n_over += (fh[i] & FLAG_OVERFLOW) ? 1 : 0; n_amb += (fh[i] & FLAG_AMBIGUOUS) ? 1 : 0; n_bad += (fh[i] & FLAG_BADROW) ? 1 : 0; }
  if (n_bad) {
    set_error("search: refine met row ids beyond the %s raw vectors given to lance_hip_index_set_raw (%u queries): the raw array must be "
              "indexed by the stored row ids", "attached", n_bad);
    return LANCE_HIP_EINVAL;
  }
  if (n_over || n_amb) {
    set_error("search: internal error, %u overflow / %u ambiguous queries were not resolved by the exact kernel", n_over, n_amb);
    return LANCE_HIP_ERUNTIME;
  }
  return LANCE_HIP_OK;
}

}  // namespace lh

using namespace lh;

extern "C" {

int lance_hip_index_prewarm(lance_hip_ctx *ctx, lance_hip_index *idx) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(ctx && idx, "index_prewarm: NULL argument");
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  if (idx->m > 0 && idx->nbits == 8) LH_TRY(mscan_prewarm(ctx, idx));
  if (coarse_groups_shape((int)idx->d, idx->nlist) && (reinterpret_cast<uintptr_t>(idx->centroids) & 15) == 0)
    LH_TRY(coarse_planes_prepare(ctx, idx, idx->metric == LANCE_HIP_COSINE ? LANCE_HIP_L2 : idx->metric));
  (void)raw_compact_prepare(ctx, idx);      // nullptr = the column stays f32 (not integer-valued, or not an f32 L2 / dot index): not an error
  LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return LANCE_HIP_OK;
}

int lance_hip_find_partitions(lance_hip_ctx *ctx, int dtype, int metric, const void *q, uint32_t nq, uint32_t d,
                              const void *centroids, uint32_t nlist, uint32_t nprobes, uint32_t *part_ids, float *dists) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(ctx && q && centroids && part_ids, "find_partitions: NULL argument");
  LH_TRY(check_dtype(dtype, "find_partitions"));
  LH_REQUIRE(nlist > 0 && nlist <= 65536, "find_partitions: nlist=%u not supported in this version (1..65536)", nlist);
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  const float *qf, *cf;
  LH_TRY(as_f32(ctx, dtype, q, (size_t)nq * d, "f16.q", &qf));
  LH_TRY(as_f32(ctx, model_dtype(dtype), centroids, (size_t)nlist * d, "f16.cent", &cf));
  // f16 columns under dot: dot_scalar::<f16, f32, 32> (dot.rs:91-102)
  return find_partitions_f32(ctx, metric, qf, nq, d, cf, nlist, nprobes, part_ids, dists, dtype == LANCE_HIP_F16 && metric == LANCE_HIP_DOT && d > 16);
}

int lance_hip_ivfpq_search_async(lance_hip_ctx *ctx, const lance_hip_index *idx, const void *q, uint32_t nq, uint32_t k,
                                 uint32_t nprobes, uint32_t refine_factor, uint64_t *ids, float *dists) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(ctx && idx && (nq == 0 || (q && ids && dists)), "search: NULL argument");
  LH_REQUIRE(ctx->device == idx->device, "search: context and index live on different devices");
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  const float *qf;
  LH_TRY(as_f32(ctx, idx->dtype, q, (size_t)nq * idx->d, "f16.q", &qf));
  return ivfpq_search_enqueue(ctx, idx, qf, nq, k, nprobes, refine_factor, 0, 0.f, 0.f, ids, dists, nullptr, nullptr);
}

int lance_hip_ivfpq_search(lance_hip_ctx *ctx, const lance_hip_index *idx, const void *q, uint32_t nq, uint32_t k,
                           uint32_t nprobes, uint32_t refine_factor, uint64_t *ids, float *dists) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(ctx && idx && (nq == 0 || (q && ids && dists)), "search: NULL argument");
  LH_REQUIRE(ctx->device == idx->device, "search: context and index live on different devices");
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  uint32_t *flags = nullptr;
  const float *qf;
  LH_TRY(as_f32(ctx, idx->dtype, q, (size_t)nq * idx->d, "f16.q", &qf));
  LH_TRY(ivfpq_search_enqueue(ctx, idx, qf, nq, k, nprobes, refine_factor, 0, 0.f, 0.f, ids, dists, &flags, nullptr));
  return check_flags(ctx, flags, nq);
}

int lance_hip_ivfpq_search_candidates(lance_hip_ctx *ctx, const lance_hip_index *idx, const void *q, uint32_t nq, uint32_t keff, uint32_t nprobes,
                                      uint64_t *ids, float *pq_dists, float *exact_dists) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(ctx && idx && (nq == 0 || (q && ids && pq_dists)), "search_candidates: NULL argument");
  LH_REQUIRE(ctx->device == idx->device, "search_candidates: context and index live on different devices");
  LH_REQUIRE(idx->m != 0, "search_candidates: not an IVF_PQ index");
  LH_REQUIRE(!exact_dists || idx->raw != nullptr, "search_candidates: exact distances need raw vectors (lance_hip_index_set_raw)");
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  if (nq == 0) return LANCE_HIP_OK;
  uint32_t *flags = nullptr;
  const float *qf;
  LH_TRY(as_f32(ctx, idx->dtype, q, (size_t)nq * idx->d, "f16.q", &qf));
  // ONE scan: the keff best by PQ distance, (dist, rowid) order ...
  LH_TRY(ivfpq_search_enqueue(ctx, idx, qf, nq, keff, nprobes, 0, 0, 0.f, 0.f, ids, pq_dists, &flags, nullptr));
  if (exact_dists) {
    // ... and the exact distance of each, in that order (the refine kernels' arithmetic, their selection output discarded)
    uint32_t *cnt = ctx->scratch_t<uint32_t>("search.cx_cnt", nq);
    uint64_t *oi = ctx->scratch_t<uint64_t>("search.cx_ids", (size_t)nq * keff);
    float *od = ctx->scratch_t<float>("search.cx_dists", (size_t)nq * keff);
    if (!cnt || !oi || !od) return LANCE_HIP_ENOMEM;
    hipLaunchKernelGGL(cand_count_kernel, dim3((unsigned)cdiv(nq, 256)), dim3(256), 0, ctx->stream, ids, nq, keff, cnt);
    LH_TRY(launch_refine(ctx, idx, qf, nq, (int)idx->d, ids, cnt, keff, keff, oi, od, flags, exact_dists));
  }
  return check_flags(ctx, flags, nq);
}

int lance_hip_ivfpq_search_range(lance_hip_ctx *ctx, const lance_hip_index *idx, const void *q, uint32_t nq, uint32_t k,
                                 uint32_t nprobes, uint32_t refine_factor, float lower, float upper, uint64_t *ids, float *dists) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(ctx && idx && (nq == 0 || (q && ids && dists)), "search_range: NULL argument");
  LH_REQUIRE(ctx->device == idx->device, "search_range: context and index live on different devices");
  LH_REQUIRE(idx->m != 0, "search_range: not an IVF_PQ index");
  LH_REQUIRE(!(lower > upper), "search_range: lower bound %g is above the upper bound %g", (double)lower, (double)upper);
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  uint32_t *flags = nullptr;
  const float *qf;
  LH_TRY(as_f32(ctx, idx->dtype, q, (size_t)nq * idx->d, "f16.q", &qf));
  LH_TRY(ivfpq_search_enqueue(ctx, idx, qf, nq, k, nprobes, refine_factor, 1, lower, upper, ids, dists, &flags, nullptr));
  return check_flags(ctx, flags, nq);
}

// bit[pos] = allow_by_rowid[row_ids[pos]] (rows whose id lies beyond the array are filtered out)
__global__ __launch_bounds__(256) void build_allow_bits_kernel(const uint64_t *__restrict__ row_ids, uint64_t n, const uint8_t *__restrict__ allow,
                                                               uint64_t n_allow, uint32_t *__restrict__ bits) {
  const uint64_t pos = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  bool ok = false;
  if (pos < n) {
    const uint64_t r = row_ids[pos];
    ok = r < n_allow && allow[r] != 0;
  }
  const uint64_t m = __ballot(ok);
  const int lane = threadIdx.x & 63;
  const uint64_t w0 = (pos - lane) >> 5;   // first 32-bit word of this wave's 64 positions
  if (lane == 0 && pos < n + 64) { bits[w0] = (uint32_t)m; bits[w0 + 1] = (uint32_t)(m >> 32); }
}

}  // extern "C"

// prefilter by row id -> one bit per storage position, in the context's scratch arena ("search.allow_bits")
int lh::build_allow_bits(lance_hip_ctx *ctx, const uint64_t *row_ids, uint64_t n, const uint8_t *allow_by_rowid, uint64_t n_allow,
                     const uint32_t **bits_out) {
  uint32_t *bits = ctx->scratch_t<uint32_t>("search.allow_bits", (size_t)(n / 32 + 4));
  if (!bits) return LANCE_HIP_ENOMEM;
  if (n > 0)
    hipLaunchKernelGGL(build_allow_bits_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, ctx->stream, row_ids, n, allow_by_rowid, n_allow, bits);
  *bits_out = bits;
  return LANCE_HIP_OK;
}

extern "C" {

static int search_filtered_impl(lance_hip_ctx *ctx, const lance_hip_index *idx, const void *q, uint32_t nq, uint32_t k, uint32_t nprobes,
                                uint32_t refine_factor, const uint8_t *allow_by_rowid, uint64_t n_allow, int has_range, float lower, float upper,
                                uint64_t *ids, float *dists) {
  LH_REQUIRE(ctx && idx && (nq == 0 || (q && ids && dists)), "search_filtered: NULL argument");
  LH_REQUIRE(ctx->device == idx->device, "search_filtered: context and index live on different devices");
  LH_REQUIRE(idx->m != 0, "search_filtered: not an IVF_PQ index");
  LH_REQUIRE(allow_by_rowid || n_allow == 0, "search_filtered: NULL filter");
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  const uint32_t *bits = nullptr;
  LH_TRY(build_allow_bits(ctx, idx->row_ids, idx->n, allow_by_rowid, n_allow, &bits));
  uint32_t *flags = nullptr;
  const float *qf;
  LH_TRY(as_f32(ctx, idx->dtype, q, (size_t)nq * idx->d, "f16.q", &qf));
  LH_TRY(ivfpq_search_enqueue(ctx, idx, qf, nq, k, nprobes, refine_factor, has_range, lower, upper, ids, dists, &flags, bits));
  return check_flags(ctx, flags, nq);
}

int lance_hip_ivfpq_search_filtered(lance_hip_ctx *ctx, const lance_hip_index *idx, const void *q, uint32_t nq, uint32_t k,
                                    uint32_t nprobes, uint32_t refine_factor, const uint8_t *allow_by_rowid, uint64_t n_allow,
                                    uint64_t *ids, float *dists) {
  lh::CtxLock _ctx_lock(ctx);
  return search_filtered_impl(ctx, idx, q, nq, k, nprobes, refine_factor, allow_by_rowid, n_allow, 0, 0.f, 0.f, ids, dists);
}

int lance_hip_ivfpq_search_filtered_range(lance_hip_ctx *ctx, const lance_hip_index *idx, const void *q, uint32_t nq, uint32_t k,
                                          uint32_t nprobes, uint32_t refine_factor, const uint8_t *allow_by_rowid, uint64_t n_allow,
                                          float lower, float upper, uint64_t *ids, float *dists) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(!(lower > upper), "search_filtered_range: lower bound %g is above the upper bound %g", (double)lower, (double)upper);
  return search_filtered_impl(ctx, idx, q, nq, k, nprobes, refine_factor, allow_by_rowid, n_allow, 1, lower, upper, ids, dists);
}

int lance_hip_search_stats(lance_hip_ctx *ctx, uint32_t *n_exact_replays_host) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(ctx && n_exact_replays_host, "search_stats: NULL argument");
  *n_exact_replays_host = 0;
  if (!ctx->last_replay_counter) return LANCE_HIP_OK;
  LH_CHECK_HIP(hipMemcpyAsync(n_exact_replays_host, ctx->last_replay_counter, 4, hipMemcpyDeviceToHost, ctx->stream));
  LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return LANCE_HIP_OK;
}

int lance_hip_pq_scan_topk(lance_hip_ctx *ctx, int dtype, int metric, const void *q_residual, uint32_t d,
                           const void *codebook, uint32_t m, uint32_t nbits, const uint8_t *codes_transposed,
                           const uint64_t *row_ids, uint64_t n_p, uint32_t k, int has_range, float lower, float upper,
                           uint64_t *out_ids, float *out_dists, uint32_t *out_n_host) {
  lh::CtxLock _ctx_lock(ctx);
  LH_REQUIRE(ctx && q_residual && codebook && out_ids && out_dists, "pq_scan_topk: NULL argument");
  LH_TRY(check_dtype(dtype, "pq_scan_topk"));
  LH_REQUIRE(n_p == 0 || (codes_transposed && row_ids), "pq_scan_topk: NULL codes");
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  // a single-partition index whose centroid is 0: q_residual - 0 == q_residual exactly
  std::vector<float> zero(d, 0.0f);
  float *zc = ctx->scratch_t<float>("scan1.zero", d);
  if (!zc) return LANCE_HIP_ENOMEM;
  LH_CHECK_HIP(hipMemcpyAsync(zc, zero.data(), (size_t)d * 4, hipMemcpyHostToDevice, ctx->stream));
  LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  uint32_t offs[2] = {0, (uint32_t)n_p};
  lance_hip_index *ix = nullptr;
  const int scan_metric = metric == LANCE_HIP_COSINE ? LANCE_HIP_L2 : metric;
  LH_TRY(lance_hip_index_from_storage(ctx, dtype, scan_metric, d, zc, 1, codebook, m, nbits, offs, codes_transposed, 1, row_ids, n_p, &ix));
  ix->ephemeral = true;
  uint32_t *flags = nullptr;
  const float *qf = nullptr;
  int r = as_f32(ctx, model_dtype(dtype), q_residual, d, "f16.q", &qf);
  if (r == LANCE_HIP_OK) r = ivfpq_search_enqueue(ctx, ix, qf, 1, k, 1, 0, has_range, lower, upper, out_ids, out_dists, &flags, nullptr);
  if (r == LANCE_HIP_OK) r = check_flags(ctx, flags, 1);
  if (r == LANCE_HIP_OK && out_n_host) {
    std::vector<uint64_t> ih(k);
    if (hipMemcpy(ih.data(), out_ids, (size_t)k * 8, hipMemcpyDeviceToHost) != hipSuccess) r = LANCE_HIP_ERUNTIME;
    uint32_t c = 0;
    for (uint32_t i = 0; i < k; ++i) c += ih[i] != ~0ull ? 1 : 0;
    *out_n_host = c;
  }
  lance_hip_index_destroy(ix);
  return r;
}

}  // extern "C"
