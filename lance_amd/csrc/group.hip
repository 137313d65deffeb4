// group.hip -- stable group-by of row indices by a small integer key (cluster id).
//
// Used twice on the hot path:
//   * k-means M-step (kmeans.rs:371-446): the reference sums each centroid's members
//     SEQUENTIALLY IN ROW ORDER in the element type, so bit-exact centroids need each
//     cluster's member list in ascending row order;
//   * shuffle / per-partition storage (v3/shuffler.rs:105-218, builder.rs:685-846):
//     rows grouped by IVF partition, canonical (stable) order inside a partition.
// Three kernels: per-block histogram -> exclusive scan over [key][block] -> stable scatter
// using wave ballots as a match-any (ranks inside a 64-row chunk are popcounts of the lanes
// below with the same key, so the output order is deterministic and stable).
#include "common.h"
#include "kernels.h"

namespace lh {

constexpr int GROUP_ROWS_PER_BLOCK = 1024;  // one wave, 16 chunks of 64 rows

__global__ __launch_bounds__(64) void group_hist_kernel(const uint32_t *__restrict__ ids, int64_t n, int64_t id_stride,
                                                        int k, int nblocks, uint32_t *__restrict__ blockhist,
                                                        const uint8_t *__restrict__ active) {
  extern __shared__ uint32_t hist[];
  const int b = blockIdx.y;
  if (active && !active[b]) return;
  const int blk = blockIdx.x;
  for (int c = threadIdx.x; c < k; c += 64) hist[c] = 0;
  __syncthreads();
  const uint32_t *idb = ids + (int64_t)b * id_stride;
  const int64_t r0 = (int64_t)blk * GROUP_ROWS_PER_BLOCK;
  for (int ch = 0; ch < GROUP_ROWS_PER_BLOCK / 64; ++ch) {
    const int64_t r = r0 + ch * 64 + threadIdx.x;
    if (r < n) {
      const uint32_t key = idb[r];
      if (key < (uint32_t)k) atomicAdd(&hist[key], 1u);
    }
  }
  __syncthreads();
  uint32_t *out = blockhist + (int64_t)b * k * nblocks;
  for (int c = threadIdx.x; c < k; c += 64) out[(int64_t)c * nblocks + blk] = hist[c];
}

// Two-level exclusive scan.  (1) one wave per key: exclusive scan of that key's per-block
// counts (in place) and the key total; (2) one workgroup per batch entry: exclusive scan of
// the k totals -> starts[k+1].  The scatter adds starts[key] to the per-block offset.
__global__ __launch_bounds__(64) void group_scan_blocks_kernel(uint32_t *__restrict__ blockhist, int k, int nblocks,
                                                               uint32_t *__restrict__ totals,
                                                               const uint8_t *__restrict__ active) {
  const int b = blockIdx.y;
  if (active && !active[b]) return;
  const int c = blockIdx.x;
  uint32_t *h = blockhist + ((int64_t)b * k + c) * nblocks;
  const int lane = threadIdx.x;
  uint32_t carry = 0;
  for (int base = 0; base < nblocks; base += 64) {
    const int i = base + lane;
    const uint32_t v = i < nblocks ? h[i] : 0;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(incl, off, 64);
      if (lane >= off) incl += t;
    }
    if (i < nblocks) h[i] = carry + incl - v;
    carry += __shfl(incl, 63, 64);
  }
  if (lane == 0) totals[(int64_t)b * k + c] = carry;
}

__global__ __launch_bounds__(256) void group_scan_totals_kernel(const uint32_t *__restrict__ totals, int k,
                                                                uint32_t *__restrict__ starts,
                                                                const uint8_t *__restrict__ active) {
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t carry_s;
  const int b = blockIdx.x;
  if (active && !active[b]) return;
  const uint32_t *tt = totals + (int64_t)b * k;
  uint32_t *st = starts + (int64_t)b * (k + 1);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < k; base += 256) {
    const int i = base + threadIdx.x;
    const uint32_t v = i < k ? tt[i] : 0;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(incl, off, 64);
      if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    const uint32_t carry = carry_s;
    if (i < k) st[i] = carry + woff + incl - v;
    __syncthreads();
    if (threadIdx.x == 255) carry_s = carry + woff + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) st[k] = carry_s;
}

// The same exclusive scan for tens of thousands of keys (the 2 x 65,536 virtual partitions of a C5 search batch: the 256-at-a-time loop
// above is 512 rounds of three barriers, most of the grouping's 0.6 ms per batch): every thread sums a contiguous run, one scan of the 256
// partial sums, then the run is walked again.  totals and starts must not alias.
__global__ __launch_bounds__(256) void group_scan_totals_big_kernel(const uint32_t *__restrict__ totals, int k, uint32_t *__restrict__ starts) {
  __shared__ uint32_t part[256];
  const int per = ((k + 255) / 256 + 3) & ~3;
  const int lo = min(k, (int)threadIdx.x * per), hi = min(k, lo + per);
  const bool al = (reinterpret_cast<uintptr_t>(totals) & 15) == 0;      // (lo is a multiple of 4)
  uint32_t s = 0;
  int i = lo;
  for (; al && i + 4 <= hi; i += 4) { const uint4 v = *reinterpret_cast<const uint4 *>(totals + i); s += v.x + v.y + v.z + v.w; }
  for (; i < hi; ++i) s += totals[i];
  part[threadIdx.x] = s;
  __syncthreads();
  uint32_t off = 0;
  for (int t = 0; t < (int)threadIdx.x; ++t) off += part[t];
  for (i = lo; i < hi; ++i) { const uint32_t v = totals[i]; starts[i] = off; off += v; }
  if (threadIdx.x == 255) starts[k] = off;      // (the last thread's run ends at k, or is empty and off is the grand total)
}

// Both scans in one launch for the sizes the k-means loop has (k <= 1024 keys, <= 256 blocks: 65,536 sampled rows): one workgroup per batch
// entry, thread c walks key c's per-block counts (contiguous) into exclusive offsets, then the block scans the key totals into starts[k + 1].
// Same outputs as group_scan_blocks_kernel + group_scan_totals_kernel, one kernel boundary less per Lloyd iteration.
__global__ __launch_bounds__(256) void group_scan_fused_kernel(uint32_t *__restrict__ blockhist, int k, int nblocks, uint32_t *__restrict__ starts,
                                                               const uint8_t *__restrict__ active) {
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t carry_s;
  const int b = blockIdx.x;
  if (active && !active[b]) return;
  uint32_t *st = starts + (int64_t)b * (k + 1);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < k; base += 256) {
    const int c = base + threadIdx.x;
    uint32_t tot = 0;
    if (c < k) {
      uint32_t *h = blockhist + ((int64_t)b * k + c) * nblocks;
      const bool al = (reinterpret_cast<uintptr_t>(h) & 15) == 0;
      int i = 0;
      // 64 counts at a time: all sixteen 16-byte loads in flight, THEN the prefix and the stores (the in-place update otherwise orders
      // every load behind the previous store: one L2 round trip per four blocks -- +23 us on the search path's 98-block grouping, gpurun r05v)
      for (; al && i + 64 <= nblocks; i += 64) {
        uint4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const uint4 *>(h + i + 4 * u);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const uint4 o = make_uint4(tot, tot + v[u].x, tot + v[u].x + v[u].y, tot + v[u].x + v[u].y + v[u].z);
          tot += v[u].x + v[u].y + v[u].z + v[u].w;
          *reinterpret_cast<uint4 *>(h + i + 4 * u) = o;
        }
      }
      for (; al && i + 16 <= nblocks; i += 16) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const uint4 *>(h + i + 4 * u);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint4 o = make_uint4(tot, tot + v[u].x, tot + v[u].x + v[u].y, tot + v[u].x + v[u].y + v[u].z);
          tot += v[u].x + v[u].y + v[u].z + v[u].w;
          *reinterpret_cast<uint4 *>(h + i + 4 * u) = o;
        }
      }
      for (; i < nblocks; ++i) { const uint32_t v = h[i]; h[i] = tot; tot += v; }
    }
    uint32_t incl = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(incl, off, 64);
      if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    const uint32_t carry = carry_s;
    if (c < k) st[c] = carry + woff + incl - tot;
    __syncthreads();
    if (threadIdx.x == 255) carry_s = carry + woff + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) st[k] = carry_s;
}

__global__ __launch_bounds__(64) void group_scatter_kernel(const uint32_t *__restrict__ ids, int64_t n, int64_t id_stride,
                                                           int k, int kbits, int nblocks,
                                                           const uint32_t *__restrict__ blockoffs,
                                                           const uint32_t *__restrict__ starts,
                                                           uint32_t *__restrict__ sorted_rows, int64_t out_stride,
                                                           const uint8_t *__restrict__ active) {
  extern __shared__ uint32_t cursor[];
  const int b = blockIdx.y;
  if (active && !active[b]) return;
  const int blk = blockIdx.x;
  const uint32_t *offs = blockoffs + (int64_t)b * k * nblocks;
  const uint32_t *st = starts + (int64_t)b * (k + 1);
  for (int c = threadIdx.x; c < k; c += 64) cursor[c] = st[c] + offs[(int64_t)c * nblocks + blk];
  __syncthreads();
  const uint32_t *idb = ids + (int64_t)b * id_stride;
  uint32_t *outb = sorted_rows + (int64_t)b * out_stride;
  const int lane = threadIdx.x;
  const uint64_t below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const int64_t r0 = (int64_t)blk * GROUP_ROWS_PER_BLOCK;
  for (int ch = 0; ch < GROUP_ROWS_PER_BLOCK / 64; ++ch) {
    const int64_t r = r0 + ch * 64 + lane;
    uint32_t key = LANCE_HIP_NONE;
    if (r < n) key = idb[r];
    const bool valid = key < (uint32_t)k;
    uint64_t mask = __ballot(valid);
    for (int bit = 0; bit < kbits; ++bit) {
      const bool one = (key >> bit) & 1u;
      const uint64_t bal = __ballot(one);
      mask &= one ? bal : ~bal;
    }
    uint32_t base = 0;
    if (valid) base = cursor[key];
    __syncthreads();
    if (valid) {
      const uint32_t rank = (uint32_t)__popcll(mask & below);
      outb[base + rank] = (uint32_t)r;
      if (rank == 0) cursor[key] = base + (uint32_t)__popcll(mask);
    }
    __syncthreads();
  }
}

// ---- keys with more than 16384 distinct values (nlist up to 65536 and the 2 x nlist virtual partitions of the
// partition-major scan): two stable passes, low 8 bits then the remaining high bits (LSD radix), each one the
// LDS-histogram sort below; offsets come from a global histogram.
__global__ __launch_bounds__(256) void group_digit_lo_kernel(const uint32_t *__restrict__ ids, int64_t n, int k, uint32_t *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = ids[i] < (uint32_t)k ? (ids[i] & 255u) : LANCE_HIP_NONE;
}
__global__ __launch_bounds__(256) void group_digit_hi_kernel(const uint32_t *__restrict__ ids, const uint32_t *__restrict__ p1,
                                                             const uint32_t *__restrict__ nvalid, int64_t n, uint32_t *__restrict__ out) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j < n) out[j] = j < (int64_t)*nvalid ? (ids[p1[j]] >> 8) : LANCE_HIP_NONE;
}
__global__ __launch_bounds__(256) void group_compose_kernel(const uint32_t *__restrict__ p1, const uint32_t *__restrict__ p2,
                                                            const uint32_t *__restrict__ nvalid, int64_t n, uint32_t *__restrict__ out) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j < n && j < (int64_t)*nvalid) out[j] = p1[p2[j]];
}
__global__ __launch_bounds__(256) void group_count_kernel(const uint32_t *__restrict__ ids, int64_t n, int k, uint32_t *__restrict__ counts) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n && ids[i] < (uint32_t)k) atomicAdd(&counts[ids[i]], 1u);
}

int stable_group(lance_hip_ctx *ctx, const uint32_t *ids, int64_t n, int64_t id_stride, int k, int batches,
                 uint32_t *starts, uint32_t *sorted_rows, int64_t out_stride, const uint8_t *active);

static int stable_group_wide(lance_hip_ctx *ctx, const uint32_t *ids, int64_t n, int k, uint32_t *starts, uint32_t *sorted_rows) {
  LH_REQUIRE(k <= (1 << 22), "stable_group: k=%d not supported (<= 4194304)", k);
  const int k2 = (k + 255) / 256;
  const size_t nn = (size_t)(n > 0 ? n : 1);
  uint32_t *klo = ctx->scratch_t<uint32_t>("group.klo", nn), *khi = ctx->scratch_t<uint32_t>("group.khi", nn);
  uint32_t *p1 = ctx->scratch_t<uint32_t>("group.p1", nn), *p2 = ctx->scratch_t<uint32_t>("group.p2", nn);
  uint32_t *st1 = ctx->scratch_t<uint32_t>("group.st1", 257), *st2 = ctx->scratch_t<uint32_t>("group.st2", (size_t)k2 + 1);
  uint32_t *counts = ctx->scratch_t<uint32_t>("group.counts", (size_t)k);
  if (!klo || !khi || !p1 || !p2 || !st1 || !st2 || !counts) return LANCE_HIP_ENOMEM;
  const unsigned grid = (unsigned)cdiv(nn, 256);
  hipLaunchKernelGGL(group_digit_lo_kernel, dim3(grid), dim3(256), 0, ctx->stream, ids, n, k, klo);
  LH_TRY(stable_group(ctx, klo, n, n, 256, 1, st1, p1, n, nullptr));
  hipLaunchKernelGGL(group_digit_hi_kernel, dim3(grid), dim3(256), 0, ctx->stream, ids, p1, st1 + 256, n, khi);
  LH_TRY(stable_group(ctx, khi, n, n, k2, 1, st2, p2, n, nullptr));
  hipLaunchKernelGGL(group_compose_kernel, dim3(grid), dim3(256), 0, ctx->stream, p1, p2, st1 + 256, n, sorted_rows);
  LH_CHECK_HIP(lh::memset_async(counts, 0, (size_t)k * 4, ctx->stream));
  hipLaunchKernelGGL(group_count_kernel, dim3(grid), dim3(256), 0, ctx->stream, ids, n, k, counts);
  hipLaunchKernelGGL(group_scan_totals_big_kernel, dim3(1), dim3(256), 0, ctx->stream, counts, k, starts);
  LH_CHECK_HIP(hipGetLastError());
  return LANCE_HIP_OK;
}

// ids: [batches][n] (stride id_stride) keys in [0,k) or NONE.  Outputs per batch:
// starts[k+1] (exclusive offsets, starts[k] = number of grouped rows) and sorted_rows
// (row indices grouped by key, ascending inside a group).
int stable_group(lance_hip_ctx *ctx, const uint32_t *ids, int64_t n, int64_t id_stride, int k, int batches,
                 uint32_t *starts, uint32_t *sorted_rows, int64_t out_stride, const uint8_t *active) {
  LH_REQUIRE(k > 0, "stable_group: k=%d not supported", k);
  LH_REQUIRE(n < (1ll << 32), "stable_group: n too large");
  if (batches == 0) return LANCE_HIP_OK;
  if (k > 16384) {
    LH_REQUIRE(batches == 1 && active == nullptr, "stable_group: more than 16384 keys only for a single problem");
    return stable_group_wide(ctx, ids, n, k, starts, sorted_rows);
  }
  const int nblocks = (int)cdiv(n > 0 ? n : 1, GROUP_ROWS_PER_BLOCK);
  uint32_t *blockhist = ctx->scratch_t<uint32_t>("group.blockhist", (size_t)batches * k * nblocks);
  uint32_t *totals = ctx->scratch_t<uint32_t>("group.totals", (size_t)batches * k);
  if (!blockhist || !totals) return LANCE_HIP_ENOMEM;
  int kbits = 0;
  while ((1 << kbits) < k) ++kbits;
  const size_t lds = (size_t)k * sizeof(uint32_t);
  hipLaunchKernelGGL(group_hist_kernel, dim3(nblocks, batches), dim3(64), lds, ctx->stream, ids, n, id_stride, k,
                     nblocks, blockhist, active);
  if (k <= 256 && nblocks <= 256) {      // (one pass of the 256 threads over the keys; wider key sets keep a wave per key)
    hipLaunchKernelGGL(group_scan_fused_kernel, dim3(batches), dim3(256), 0, ctx->stream, blockhist, k, nblocks, starts, active);
  } else {
    hipLaunchKernelGGL(group_scan_blocks_kernel, dim3(k, batches), dim3(64), 0, ctx->stream, blockhist, k, nblocks, totals, active);
    hipLaunchKernelGGL(group_scan_totals_kernel, dim3(batches), dim3(256), 0, ctx->stream, totals, k, starts, active);
  }
  hipLaunchKernelGGL(group_scatter_kernel, dim3(nblocks, batches), dim3(64), lds, ctx->stream, ids, n, id_stride, k,
                     kbits, nblocks, blockhist, starts, sorted_rows, out_stride, active);
  LH_CHECK_HIP(hipGetLastError());
  return LANCE_HIP_OK;
}

}  // namespace lh
