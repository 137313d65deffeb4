// flat_small.hip -- exhaustive KNN for ONE to FOUR queries: a single streaming pass at the HBM roofline.
//
//   KNNVectorDistanceExec / compute_distance   lance/src/io/exec/knn.rs:218-246, lance-index flat.rs:95-148
//   l2_scalar / dot_scalar (16 lane accumulators)  lance-linalg l2.rs:57-91, dot.rs:30-58
//   SortExec(dist asc, rowid asc).fetch(k)     lance/src/dataset/scanner.rs:3386-3406
//
// flat.hip's batch path (lanes own rows, queries stream through LDS, epochs of growing size with a select kernel after each)
// is built for hundreds of queries; for a single query its three epochs, three pool sorts and the overflow round trip cost as
// much as the 512 MB it reads (C1: 0.25 ms = 2.06 TB/s, profiles/r03_grid.json).  Here the scan is one launch:
//   * a workgroup owns a contiguous slice of the rows; a 16-lane group owns a row at a time, lane i = lane accumulator i of the
//     reference's l2_scalar / dot_scalar (sums[i] over the 16-chunks in order), so a group's loads are 64 contiguous bytes per
//     chunk and consecutive groups read consecutive rows -- fully coalesced, every byte used once;
//   * the 16 partial sums are folded in lane order through shuffles (((0 + s0) + s1) + ...), the d % 16 remainder is summed
//     first in element order: the reference's value, bit for bit;
//   * rows at or under the workgroup's running threshold go to an LDS list; when it fills, the threshold drops to the k-th
//     smallest of the lane minima (pm_common.cuh) -- it is always the k-th key of a SUBSET of the slice, never below the slice's
//     true k-th key, so no row of the answer is filtered;
//   * every workgroup leaves its k best (key, rowid) pairs; a second small kernel merges the G x k pairs per query in the
//     SortExec order (a total order: ties by row id, as the reference's final sort) -- two parallel passes since round 6
//     (lane minima -> their k-th smallest -> the pairs at or under it), the barrier-separated step loop only as the refill.
// More rows tied at a threshold than the list holds (thousands of duplicate vectors) raise a flag and the caller takes the
// batch path, which repairs such cases.  For one query the flag reaches the host through a pinned word the merge kernel
// writes (and it clears the device word behind itself): a call is two launches and a stream wait.
// C1 (1M x 128 f32, k = 10): 0.140 ms per call, of which the scan kernel 96 us (5.3 TB/s) and the merge 16 us
// (profiles/r06zzo_flat_one_*).
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "exact.cuh"
#include "kernels.h"
#include "pm_common.cuh"
#include "search_common.cuh"

#pragma clang fp contract(off)

namespace lh {

constexpr int FS_BS = 256;       // 16 row groups of 16 lanes
constexpr int FS_CAP = 1024;     // list entries per query and workgroup
constexpr int FS_RPI = 8;        // rows per group between two capacity checks
#ifndef FS_UNROLL_R
#define FS_UNROLL_R 2         // rows of a group in flight (x 16-lane chunk loads each)
#endif
constexpr int FS_MAXQ = 4;       // instantiated; flat_small_supported() decides how many queries take this path

struct FsArgs {
  const void *x;                 // [n][d] rows in the column's element type
  const uint64_t *row_ids;       // or NULL: the row number is the id
  int64_t n, rows_per_wg;
  int d, nq, k, G;
  const float *q;                // [nq][d]
  uint32_t *lkeys;               // [nq][G][k]
  uint64_t *lrids;
  uint32_t *overflow;
  uint32_t *host_flag = nullptr; // one query: pinned host word the merge kernel leaves the overflow flag in (it also clears `overflow` for the next call)
};

template <int METRIC, typename TX, int NQ>
__global__ __launch_bounds__(FS_BS) void flat_small_scan_kernel(FsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int dpad = (a.d + 15) & ~15;
  float *qs = reinterpret_cast<float *>(smem);                                   // [NQ][dpad]
  uint32_t *ckey = reinterpret_cast<uint32_t *>(qs + (size_t)NQ * dpad);        // [NQ][CAP]
  uint32_t *cpos = ckey + NQ * FS_CAP;                                           // [NQ][CAP] row - r_begin
  uint64_t *srid = reinterpret_cast<uint64_t *>(cpos + NQ * FS_CAP);            // [CAP]  (final sort)
  uint32_t *sorted = reinterpret_cast<uint32_t *>(srid + FS_CAP);               // [BS]
  uint32_t *misc = sorted + FS_BS;                                               // [NQ][4]: count, threshold, scratch, lost
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, gi = lane & 15;
  const int g16 = wave * 4 + (lane >> 4);
  const int64_t r_begin = (int64_t)blockIdx.x * a.rows_per_wg, r_end = min(a.n, r_begin + a.rows_per_wg);
  for (int i = threadIdx.x; i < NQ * dpad; i += FS_BS) {
    const int qi = i / dpad, e = i - qi * dpad;
    qs[i] = e < a.d ? a.q[(int64_t)qi * a.d + e] : 0.0f;
  }
  if (threadIdx.x < NQ * 4) misc[threadIdx.x] = (threadIdx.x & 3) == 1 ? 0xFFFFFFFFu : 0u;
  __syncthreads();
  const TX *x = static_cast<const TX *>(a.x);
  const int d = a.d, full = d / 16 * 16;
  for (int64_t base = r_begin; base < r_end; base += 16 * FS_RPI) {
    bool need = false;
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) need |= misc[qi * 4] > (uint32_t)(FS_CAP - 16 * FS_RPI);   // read, barrier, decide
    __syncthreads();
    if (need) {
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi) {
        CandBuf b{ckey + qi * FS_CAP, cpos + qi * FS_CAP, &misc[qi * 4], &misc[qi * 4 + 1]};
        tighten_bs<FS_BS, FS_CAP>(b, a.k, sorted, &misc[qi * 4 + 2]);
      }
    }
    uint32_t T[NQ];
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) T[qi] = misc[qi * 4 + 1];
#pragma unroll FS_UNROLL_R
    for (int r = 0; r < FS_RPI; ++r) {
      const int64_t row = base + r * 16 + g16;
      const bool valid = row < r_end;
      const TX *xr = x + (valid ? row : r_begin) * d;
      float acc[NQ], s[NQ];
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi) { acc[qi] = 0.0f; s[qi] = 0.0f; }
      if (full != d) {        // remainder first, in element order (every lane of the group computes the same value)
        for (int e = full; e < d; ++e) {
          const float xv = ld_elem(xr, e);
#pragma unroll
          for (int qi = 0; qi < NQ; ++qi) {
            if constexpr (METRIC == METRIC_DOT) s[qi] = s[qi] + xv * qs[qi * dpad + e];
            else { const float diff = xv - qs[qi * dpad + e]; s[qi] = s[qi] + diff * diff; }
          }
        }
      }
      if constexpr (NQ == 1) {
#pragma unroll 8
        for (int ch = 0; ch < full; ch += 16) {
          const float xv = ld_elem(xr, ch + gi);
          if constexpr (METRIC == METRIC_DOT) acc[0] = acc[0] + xv * qs[ch + gi];
          else { const float diff = xv - qs[ch + gi]; acc[0] = acc[0] + diff * diff; }
        }
      } else {
        // two to four queries: the row's chunk loads are issued eight at a time BEFORE anything consumes them -- left to itself the compiler
        // put an `s_waitcnt vmcnt(0)` behind every load of this loop (the queries' LDS reads in between), i.e. one chunk in flight per group
        for (int ch0 = 0; ch0 < full; ch0 += 128) {
          float xv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) xv[u] = ch0 + 16 * u < full ? ld_elem(xr, ch0 + 16 * u + gi) : 0.0f;
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int ch = ch0 + 16 * u;
            if (ch < full) {
#pragma unroll
              for (int qi = 0; qi < NQ; ++qi) {
                if constexpr (METRIC == METRIC_DOT) acc[qi] = acc[qi] + xv[u] * qs[qi * dpad + ch + gi];
                else { const float diff = xv[u] - qs[qi * dpad + ch + gi]; acc[qi] = acc[qi] + diff * diff; }
              }
            }
          }
        }
      }
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi) {
        // ((0 + a0) + a1) + ... + a15 as a running sum handed from lane to lane inside the 16-lane DPP row (row_shr:1; lane 0 of a
        // row receives +0.0): after step j lanes 0..j hold their prefix, recomputing a settled lane gives the same value, so after
        // 15 steps lane 15 holds the reference's total.  (The first version fetched the 16 partial sums with 16 ds_bpermute per row
        // and query: r04c, 135 us per 512 MB for one query and 2.4 x that per further query -- the fold, not HBM, set the pace.)
        float run = 0.0f + acc[qi];
#pragma unroll
        for (int t = 1; t < 16; ++t) {
          const float prev = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, run), 0x111, 0xF, 0xF, false));
          run = prev + acc[qi];
        }
        const uint32_t key = order_key(finish_metric<METRIC>(s[qi] + run));
        if (valid && gi == 15 && key <= T[qi]) {
          const uint32_t slot = atomicAdd(&misc[qi * 4], 1u);
          if (slot < (uint32_t)FS_CAP) { ckey[qi * FS_CAP + slot] = key; cpos[qi * FS_CAP + slot] = (uint32_t)(row - r_begin); }
          else misc[qi * 4 + 3] = 1u;
        }
      }
    }
    __syncthreads();     // every append of this round is counted before the next round's capacity check reads the counters
  }
  // the slice's k best per query: shrink the list to about k entries, sort by (key, rowid), write k pairs
#pragma unroll 1
  for (int qi = 0; qi < NQ; ++qi) {
    CandBuf b{ckey + qi * FS_CAP, cpos + qi * FS_CAP, &misc[qi * 4], &misc[qi * 4 + 1]};
    for (int iter = 0; iter < 4; ++iter) {
      __syncthreads();
      const bool more = (int)misc[qi * 4] > a.k + 28;      // read, barrier (inside tighten_bs), decide
      __syncthreads();
      if (!more) break;
      tighten_bs<FS_BS, FS_CAP>(b, a.k, sorted, &misc[qi * 4 + 2]);
    }
    __syncthreads();
    if (threadIdx.x == 0 && (misc[qi * 4 + 3] || misc[qi * 4] > (uint32_t)FS_CAP)) atomicOr(a.overflow, 1u);
    const int c = min((int)misc[qi * 4], FS_CAP);
    int P = 64;
    while (P < c) P <<= 1;
    uint32_t *key = ckey + qi * FS_CAP, *pos = cpos + qi * FS_CAP;
    for (int i = threadIdx.x; i < P; i += FS_BS) {
      if (i < c) {
        const int64_t row = r_begin + pos[i];
        srid[i] = a.row_ids ? a.row_ids[row] : (uint64_t)row;
      } else {
        key[i] = 0xFFFFFFFFu; pos[i] = 0; srid[i] = ~0ull;
      }
    }
    __syncthreads();
    bitonic_sort_kr<FS_BS>(key, srid, pos, P);
    const int64_t ob = ((int64_t)qi * a.G + blockIdx.x) * a.k;
    for (int i = threadIdx.x; i < a.k; i += FS_BS) {
      const bool ok = i < c;
      a.lkeys[ob + i] = ok ? key[i] : 0xFFFFFFFFu;
      a.lrids[ob + i] = ok ? srid[i] : ~0ull;
    }
    __syncthreads();
  }
}

// one workgroup per query: the k smallest (key, rowid) pairs of the G x k the scan left (empty slots: key ~0, rowid ~0)
__global__ __launch_bounds__(FS_BS) void flat_small_merge_kernel(FsArgs a, uint64_t *__restrict__ out_ids, float *__restrict__ out_dists) {
  __shared__ uint32_t ckey[FS_CAP], cpos[FS_CAP], sorted[FS_BS], misc[4];
  __shared__ uint64_t srid[FS_CAP];
  const int qi = blockIdx.x;
  const int64_t total = (int64_t)a.G * a.k;
  const uint32_t *lk = a.lkeys + (int64_t)qi * total;
  const uint64_t *lr = a.lrids + (int64_t)qi * total;
  if (threadIdx.x == 0) { misc[0] = 0; misc[1] = 0xFFFFFFFFu; misc[3] = 0; }
  __syncthreads();
  CandBuf b{ckey, cpos, &misc[0], &misc[1]};
  // Two parallel passes instead of total / 256 barrier-separated steps (round 6: the step loop below was 31 us of C1's 0.169 ms per
  // query at k = 10 -- G x k = 10,240 pairs -- and 192 us at k = 100): every lane takes the minimum key of its strided share of the
  // pairs, T = the k-th smallest lane minimum (k different pairs lie at or under it, so the k-th pair's key is <= T; an empty slot's
  // key ~0 is the minimum's identity), and the pairs at or under T -- little more than k of them unless keys tie in masses -- go to
  // the list.  A list that does not hold them is started again by the step loop, which tightens as it goes.
  {
    // (16-byte loads, sixteen in flight per lane: the pairs were written by workgroups of every XCD, so each first touch is a trip
    // to memory, and a single workgroup hides those only by having all of them on the way at once)
    const bool vec = (reinterpret_cast<uintptr_t>(lk) & 15) == 0;
    const int64_t nch = vec ? total >> 2 : 0;
    const uint4 *lk4 = reinterpret_cast<const uint4 *>(lk);
    uint32_t mymin = 0xFFFFFFFFu;
#pragma unroll 16
    for (int64_t c = threadIdx.x; c < nch; c += FS_BS) {
      const uint4 v = lk4[c];
      mymin = min(mymin, min(min(v.x, v.y), min(v.z, v.w)));
    }
    for (int64_t i = nch * 4 + threadIdx.x; i < total; i += FS_BS) mymin = min(mymin, lk[i]);
    kth_smallest_bs<FS_BS>(mymin, a.k - 1, sorted, &misc[2]);
    const uint32_t T = misc[2];
    auto take = [&](uint32_t key, int64_t i) {
      if (key <= T && (key != 0xFFFFFFFFu || lr[i] != ~0ull)) {      // empty slots never enter
        const uint32_t slot = atomicAdd(&misc[0], 1u);
        if (slot < (uint32_t)FS_CAP) { ckey[slot] = key; cpos[slot] = (uint32_t)i; }
      }
    };
#pragma unroll 4
    for (int64_t c = threadIdx.x; c < nch; c += FS_BS) {
      const uint4 v = lk4[c];
      if (min(min(v.x, v.y), min(v.z, v.w)) <= T) { take(v.x, 4 * c); take(v.y, 4 * c + 1); take(v.z, 4 * c + 2); take(v.w, 4 * c + 3); }
    }
    for (int64_t i = nch * 4 + threadIdx.x; i < total; i += FS_BS) take(lk[i], i);
    __syncthreads();
  }
  const bool refill = misc[0] > (uint32_t)FS_CAP;      // (uniform)
  __syncthreads();
  if (refill && threadIdx.x == 0) misc[0] = 0;
  __syncthreads();
  for (int64_t base = 0; refill && base < total; base += FS_BS) {
    const bool need = misc[0] > (uint32_t)(FS_CAP - FS_BS);
    __syncthreads();
    if (need) tighten_bs<FS_BS, FS_CAP>(b, a.k, sorted, &misc[2]);
    const uint32_t T = misc[1];
    const int64_t i = base + threadIdx.x;
    if (i < total) {
      const uint32_t key = lk[i];
      if (lr[i] != ~0ull && key <= T) {       // empty slots never enter
        const uint32_t slot = atomicAdd(&misc[0], 1u);
        if (slot < (uint32_t)FS_CAP) { ckey[slot] = key; cpos[slot] = (uint32_t)i; } else misc[3] = 1u;
      }
    }
    __syncthreads();
  }
  for (int iter = 0; iter < 4; ++iter) {
    __syncthreads();
    const bool more = (int)misc[0] > a.k + 28;
    __syncthreads();
    if (!more) break;
    tighten_bs<FS_BS, FS_CAP>(b, a.k, sorted, &misc[2]);
  }
  __syncthreads();
  if (threadIdx.x == 0 && (misc[3] || misc[0] > (uint32_t)FS_CAP)) atomicOr(a.overflow, 1u);
  const int c = min((int)misc[0], FS_CAP);
  int P = 64;
  while (P < c) P <<= 1;
  for (int i = threadIdx.x; i < P; i += FS_BS) {
    if (i < c) srid[i] = lr[cpos[i]];
    else { ckey[i] = 0xFFFFFFFFu; cpos[i] = 0; srid[i] = ~0ull; }
  }
  __syncthreads();
  bitonic_sort_kr<FS_BS>(ckey, srid, cpos, P);
  for (int i = threadIdx.x; i < a.k; i += FS_BS) {
    const bool ok = i < c;
    out_ids[(int64_t)qi * a.k + i] = ok ? srid[i] : ~0ull;
    out_dists[(int64_t)qi * a.k + i] = ok ? key_to_float(ckey[i]) : INFINITY;
  }
  if (a.host_flag && threadIdx.x == 0) {      // one query = one workgroup: nobody else touches the flag any more (its own atomicOr above was this thread's)
    const uint32_t o = __hip_atomic_load(a.overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(a.host_flag, o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(a.overflow, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

bool flat_small_supported(int metric, int dtype, uint32_t d, uint32_t nq, uint32_t k, uint64_t n) {
  static const bool off = getenv("LANCE_HIP_NO_FLAT_SMALL") != nullptr;
  // r04c, C1 (1M x 128 f32): one query 0.135 ms here against 0.25 ms on the batch path (wall 0.177 / 0.247).  Round 6 (r06zzzv), wall per
  // call at k = 10, this kernel / the batch path: two queries 0.193 / 0.256 ms (0.328 before the scan's chunk loads were issued eight at a
  // time), four queries 0.399 / 0.271 -- so ONE OR TWO queries come here; LANCE_HIP_FLAT_SMALL_MAXQ overrides
  static const uint32_t maxq = getenv("LANCE_HIP_FLAT_SMALL_MAXQ") ? (uint32_t)atoi(getenv("LANCE_HIP_FLAT_SMALL_MAXQ")) : 2u;
  if (off || nq == 0 || nq > std::min<uint32_t>(maxq, (uint32_t)FS_MAXQ) || k > 128 || d == 0 || d > 2048) return false;
  if (metric != LANCE_HIP_L2 && metric != LANCE_HIP_DOT) return false;      // cosine_fast has its own arithmetic (flat.hip)
  if (dtype == LANCE_HIP_F16 && metric == LANCE_HIP_DOT && d > 16) return false;   // 32 lane accumulators (dot.rs:91-102): batch path
  return n >= 4096;     // tiny tables: the batch path's single epoch is as good
}

template <int METRIC, typename TX>
static void fs_launch_nq(lance_hip_ctx *ctx, const FsArgs &a, size_t lds_fixed) {
  const int dpad = (a.d + 15) & ~15;
  auto lds = [&](int nq) { return (size_t)nq * dpad * 4 + (size_t)nq * FS_CAP * 8 + lds_fixed; };
  const dim3 grid((unsigned)a.G), block(FS_BS);
  if (a.nq == 1) hipLaunchKernelGGL((flat_small_scan_kernel<METRIC, TX, 1>), grid, block, lds(1), ctx->stream, a);
  else if (a.nq == 2) hipLaunchKernelGGL((flat_small_scan_kernel<METRIC, TX, 2>), grid, block, lds(2), ctx->stream, a);
  else if (a.nq == 3) hipLaunchKernelGGL((flat_small_scan_kernel<METRIC, TX, 3>), grid, block, lds(3), ctx->stream, a);
  else hipLaunchKernelGGL((flat_small_scan_kernel<METRIC, TX, 4>), grid, block, lds(4), ctx->stream, a);
}

// x: the rows in their own element type; q: f32 queries.  *done = false: the list overflowed (mass ties) -- take the batch path.
int flat_topk_small(lance_hip_ctx *ctx, int metric, int dtype, const void *x, const uint64_t *row_ids, uint64_t n, uint32_t d, const float *q,
                    uint32_t nq, uint32_t k, uint64_t *ids, float *dists, bool *done) {
  *done = false;
  FsArgs a;
  a.x = x; a.row_ids = row_ids; a.n = (int64_t)n; a.d = (int)d; a.nq = (int)nq; a.k = (int)k; a.q = q;
  const uint64_t step = 16 * FS_RPI;
  static const int wgs_per_cu = getenv("LANCE_HIP_FS_WGS_PER_CU") ? std::max(1, atoi(getenv("LANCE_HIP_FS_WGS_PER_CU"))) : 4;      // A/B
  uint64_t G = std::min<uint64_t>((uint64_t)ctx->num_cus * wgs_per_cu, std::max<uint64_t>(1, cdiv(n, 4 * step)));
  a.rows_per_wg = (int64_t)(cdiv(cdiv(n, G), step) * step);
  G = cdiv(n, (uint64_t)a.rows_per_wg);
  a.G = (int)G;
  a.lkeys = ctx->scratch_t<uint32_t>("fs.lkeys", (size_t)nq * G * k);
  a.lrids = ctx->scratch_t<uint64_t>("fs.lrids", (size_t)nq * G * k);
  // One query (the latency case): the merge kernel leaves the flag in a pinned host word and clears the device word behind itself, so
  // the call is two launches and a stream wait -- no fill kernel in front, no copy kernel behind (2 + 4 us of kernels and their
  // launch gaps, r06zzm).  The device word of that protocol has a slot of its own, zeroed when it is made.
  static const bool no_host_flag = getenv("LANCE_HIP_NO_FLAT_HOST_FLAG") != nullptr;
  const bool host_flag = nq == 1 && !no_host_flag;
  if (host_flag) {
    const bool fresh = ctx->slots.find("fs.ovf1") == ctx->slots.end();
    a.overflow = ctx->scratch_t<uint32_t>("fs.ovf1", 1);
    a.host_flag = ctx->host_flag_word();
    if (!a.host_flag) return LANCE_HIP_ENOMEM;
    if (a.overflow && fresh) LH_CHECK_HIP(lh::memset_async(a.overflow, 0, 4, ctx->stream));
  } else {
    a.overflow = ctx->scratch_t<uint32_t>("fs.ovf", 1);
  }
  if (!a.lkeys || !a.lrids || !a.overflow) return LANCE_HIP_ENOMEM;
  if (!host_flag) LH_CHECK_HIP(lh::memset_async(a.overflow, 0, 4, ctx->stream));
  const size_t lds_fixed = (size_t)FS_CAP * 8 + (size_t)FS_BS * 4 + (size_t)FS_MAXQ * 16;
  {
    ScopedTimer t(ctx, "flat_scan");
    auto go = [&](auto tag) {
      using TX = decltype(tag);
      if (metric == LANCE_HIP_DOT) fs_launch_nq<METRIC_DOT, TX>(ctx, a, lds_fixed);
      else fs_launch_nq<METRIC_L2, TX>(ctx, a, lds_fixed);
    };
    if (dtype == LANCE_HIP_F16) go(__half());
    else if (dtype == LANCE_HIP_I8) go(int8_t());
    else go(float());
    hipLaunchKernelGGL(flat_small_merge_kernel, dim3(nq), dim3(FS_BS), 0, ctx->stream, a, ids, dists);
  }
  LH_CHECK_HIP(hipGetLastError());
  uint32_t ovf = 0;
  if (host_flag) {
    LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    ovf = *static_cast<volatile uint32_t *>(a.host_flag);
  } else {
    LH_CHECK_HIP(hipMemcpyAsync(&ovf, a.overflow, 4, hipMemcpyDeviceToHost, ctx->stream));
    LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  }
  *done = ovf == 0;
  return LANCE_HIP_OK;
}

}  // namespace lh
