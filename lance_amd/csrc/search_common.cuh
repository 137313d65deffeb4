// search_common.cuh -- helpers shared by the query-major (search.hip) and partition-major
// (search_pm.hip) scan pipelines.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "exact.cuh"

namespace lh {

constexpr uint32_t FLAG_OVERFLOW = 1u, FLAG_AMBIGUOUS = 2u, FLAG_BADROW = 4u;
constexpr int SCAN_LCAP = 256;    // entries handed to the merge kernel per (query, split)
constexpr int SCAN_MAX_KEFF = 128;

// prefilter (flat/index.rs:129-165, RowIdMask): one bit per STORAGE position, NULL = no filter
__device__ __forceinline__ bool row_allowed(const uint32_t *__restrict__ allow, uint32_t pos) {
  return allow == nullptr || ((allow[pos >> 5] >> (pos & 31u)) & 1u) != 0u;
}

static inline int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

__device__ __forceinline__ uint32_t find_partition_dev(const uint32_t *__restrict__ offs, int nlist, uint32_t slot) {
  int lo = 0, hi = nlist;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (offs[mid] <= slot) lo = mid; else hi = mid;
  }
  return (uint32_t)lo;
}

// sort entries by (key, rowid) -- the SortExec order -- with pos as payload (BS threads)
template <int BS = 256>
__device__ __forceinline__ void bitonic_sort_kr(uint32_t *key, uint64_t *rid, uint32_t *pos, int P) {
  for (int k2 = 2; k2 <= P; k2 <<= 1) {
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < P / 2; i += BS) {
        const int ix = 2 * j * (i / j) + (i % j);
        const int px = ix + j;
        const bool up = (ix & k2) == 0;
        const uint32_t kx = key[ix], ky = key[px];
        const uint64_t rx = rid[ix], ry = rid[px];
        const bool gt = kx > ky || (kx == ky && rx > ry);
        if (gt == up) {
          key[ix] = ky; key[px] = kx; rid[ix] = ry; rid[px] = rx;
          const uint32_t t = pos[ix]; pos[ix] = pos[px]; pos[px] = t;
        }
      }
      __syncthreads();
    }
  }
}

// Outputs of the per-query final selection (shared by both merge kernels)
struct SelectOut {
  int keff, k, refine;
  uint64_t *out_ids;      // [nq][k]     (refine == 0)
  float *out_dists;
  uint64_t *cand_rid;     // [nq][keff]  (refine == 1)
  uint32_t *cand_cnt;     // [nq]
  uint32_t *flags;
  const uint32_t *part_offsets;
  int nlist;
};

// After key/rid/pos[0..P) are sorted by (key,rowid) with `total` real entries: tie check + outputs.
template <int BS = 256>
__device__ __forceinline__ void select_and_emit(const SelectOut &o, int q, uint32_t *key, uint64_t *rid, uint32_t *pos, int total,
                                                int *s_amb) {
  const int got = min(total, o.keff);
  // more rows at the boundary distance than fit?  If one partition alone holds more than keff of them the
  // survivors depend on the reference heap's internals -> flag for the exact replay kernel.
  if (total > o.keff && key[o.keff] == key[o.keff - 1]) {
    const uint32_t tf = key[o.keff - 1];
    int lo = o.keff, hi = total;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (key[mid] <= tf) lo = mid + 1; else hi = mid; }
    const int L = lo;
    for (int i = threadIdx.x; i < L; i += BS) {
      const uint32_t pi = find_partition_dev(o.part_offsets, o.nlist, pos[i]);
      const uint32_t a = o.part_offsets[pi], b = o.part_offsets[pi + 1];
      int same = 0;
      for (int j = 0; j < L; ++j) same += (pos[j] >= a && pos[j] < b) ? 1 : 0;
      if (same > o.keff) *s_amb = 1;
    }
    __syncthreads();
    if (threadIdx.x == 0 && *s_amb) atomicOr(&o.flags[q], FLAG_AMBIGUOUS);
  }
  if (o.refine) {
    for (int i = threadIdx.x; i < o.keff; i += BS) o.cand_rid[(int64_t)q * o.keff + i] = i < got ? rid[i] : ~0ull;
    if (threadIdx.x == 0) o.cand_cnt[q] = (uint32_t)got;
  } else {
    for (int i = threadIdx.x; i < o.k; i += BS) {
      o.out_ids[(int64_t)q * o.k + i] = i < got ? rid[i] : ~0ull;
      o.out_dists[(int64_t)q * o.k + i] = i < got ? key_to_float(key[i]) : INFINITY;
    }
  }
}

// ---- std BinaryHeap emulation (shared by the IVF_PQ and IVF_FLAT exact-replay kernels) ----------------
__device__ __forceinline__ void heap_sift_up(uint32_t *hk, uint32_t *hp, int start, int pos) {
  const uint32_t ek = hk[pos], ep = hp[pos];
  while (pos > start) {
    const int parent = (pos - 1) / 2;
    if (ek <= hk[parent]) break;
    hk[pos] = hk[parent]; hp[pos] = hp[parent];
    pos = parent;
  }
  hk[pos] = ek; hp[pos] = ep;
}
__device__ __forceinline__ void heap_push(uint32_t *hk, uint32_t *hp, int &len, uint32_t key, uint32_t pos) {
  hk[len] = key; hp[len] = pos;
  heap_sift_up(hk, hp, 0, len);
  ++len;
}
// std BinaryHeap::pop: swap last into the root, sift_down_to_bottom(0), then sift_up
__device__ __forceinline__ void heap_pop(uint32_t *hk, uint32_t *hp, int &len) {
  --len;
  if (len == 0) return;
  const uint32_t ek = hk[len], ep = hp[len];
  const int end = len;
  int pos = 0, child = 1;
  while (end >= 2 && child <= end - 2) {
    if (hk[child] <= hk[child + 1]) child += 1;
    hk[pos] = hk[child]; hp[pos] = hp[child];
    pos = child;
    child = 2 * pos + 1;
  }
  if (child == end - 1) { hk[pos] = hk[child]; hp[pos] = hp[child]; pos = child; }
  hk[pos] = ek; hp[pos] = ep;
  heap_sift_up(hk, hp, 0, pos);
}


}  // namespace lh
