// pm_common.cuh -- threshold machinery shared by the partition-major scan kernels (search_pm.hip: exact f32 pair scan;
// search_q.hip: quantised 4-query filter scan + exact re-evaluation).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lh {

// ---- threshold machinery, generic in the workgroup size ---------------------------------------------
struct CandBuf {
  uint32_t *key, *pos;  // [CAP]
  uint32_t *cnt;        // current entries
  uint32_t *T;          // current threshold (key)
};

// k-th smallest (0-based rank kk) of one value per lane; result in *out (LDS), broadcast after the barrier
template <int BS>
__device__ __forceinline__ void kth_smallest_bs(uint32_t v, int kk, uint32_t *sorted, uint32_t *out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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
  sorted[threadIdx.x] = v;
  __syncthreads();
  int rank = lane;
  for (int w = 0; w < BS / 64; ++w) {
    if (w == wave) continue;
    const uint32_t *run = sorted + w * 64;
    int lo = 0, hi = 64;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      const bool before = w < wave ? run[mid] <= v : run[mid] < v;
      if (before) lo = mid + 1; else hi = mid;
    }
    rank += lo;
  }
  if (rank == kk) *out = v;
  __syncthreads();
}

// T <- upper bound of the keff-th smallest key in the buffer (exact once <= BS entries remain); drop key > T
template <int BS, int CAP>
__device__ __forceinline__ void tighten_bs(const CandBuf &b, int keff, uint32_t *sorted, uint32_t *tnew_slot) {
  __syncthreads();
  const int c = min((int)*b.cnt, CAP);
  if (c < keff) return;  // uniform
  constexpr int PER = (CAP + BS - 1) / BS;
  uint32_t ek[PER], ep[PER];
  uint32_t mymin = 0xFFFFFFFFu;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int i = threadIdx.x + BS * j;
    ek[j] = 0xFFFFFFFFu; ep[j] = 0;
    if (i < c) { ek[j] = b.key[i]; ep[j] = b.pos[i]; mymin = min(mymin, ek[j]); }
  }
  kth_smallest_bs<BS>(mymin, keff - 1, sorted, tnew_slot);
  const uint32_t tnew = *tnew_slot;
  __syncthreads();
  if (threadIdx.x == 0) { *b.cnt = 0; *b.T = tnew; }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int i = threadIdx.x + BS * j;
    if (i < c && ek[j] <= tnew) {
      const uint32_t slot = atomicAdd(b.cnt, 1u);
      b.key[slot] = ek[j]; b.pos[slot] = ep[j];
    }
  }
  __syncthreads();
}

}  // namespace lh
