// index.h -- device-resident IVF_PQ index (see build.hip for the layout rationale).
#pragma once
#include <atomic>
#include <cstdint>
#include <mutex>
#include <vector>

struct int2_host { int x, y; };

inline uint64_t lance_hip_next_index_serial() {
  static std::atomic<uint64_t> next{1};
  return next.fetch_add(1, std::memory_order_relaxed);
}

struct lance_hip_index {
  uint64_t serial = lance_hip_next_index_serial();   // distinguishes indices that reuse an address (captured search graphs are keyed on it)
  bool ephemeral = false;         // built for one call (lance_hip_pq_scan_topk): its searches are never captured into graphs
  int device = 0;
  int metric = 0;
  int dtype = 0;                  // element type of queries / raw vectors (model is kept widened to f32)
  uint32_t d = 0, nlist = 0, m = 0, nbits = 8;
  uint64_t n = 0;                 // rows stored (rows with part id NONE are dropped)
  float *centroids = nullptr;     // [nlist][d]
  float *codebook = nullptr;      // [m][256][d/m]
  // cb_mean is followed by one flag word (cb_mean[d + 1], as bits): != 0 when a centroid or codeword is NaN / infinite
  // (an overflowed f16 model).  model_finite mirrors it on the host; the integer filter scans -- whose NaN entries quantise to 0
  // and would enter the merge kernel's sum cut as small sums -- are only taken for finite models
  bool model_finite = true;
  float *cb_mean = nullptr;       // 8-bit PQ: [d] mean codeword of every sub-quantiser, then [1] sum over m of the mean |c|^2 (bound pass scale)
  // LANCE_HIP_QPT=1 (search_qt.hip, per-query tables): constants of that filter, created by the first such search
  struct PtConst {
    float *g = nullptr;           // [d] mean centroid (the translation that keeps the table's resolution)
    float *cen_t = nullptr;       // [nlist][d] centroids - g
    float *row_beta = nullptr;    // [n] sum over m of 2 cen_t[p][m] . codeword[m][code] for every stored row
    float *beta_min = nullptr;    // [nlist] smallest row_beta of the partition
    float *beta_abs = nullptr;    // [nlist] largest |row_beta| of the partition
    float *beta_mean = nullptr;   // [nlist] 2 cen_t[p] . mu (mu = the mean codeword of every sub-quantiser): row_beta of an average code
  } *pt = nullptr;
  // search_ms.hip (the filter scan on the matrix cores): constants of that filter, created by the first such search
  struct MsConst {
    bool usable = false;          // false: nothing to scale by (all-zero codebook) -- the integer scan serves the index
    float sigma = 1.0f;           // power of two: the largest |2 sigma c| sits in [2^13, 2^14)
    void *cbh = nullptr;          // [m][256][d/m] binary16: -2 sigma c
    float *cbn2 = nullptr;        // [m][256] |c|^2
    float *row_cn2 = nullptr;     // [n] sigma^2 |reconstruction of the stored row|^2 (dot metric: zeros)
    float cmax = 0.0f;            // dot metric: upper bound of |centred reconstruction of any stored row| (codewords minus their sub-quantiser's mean: the plane cbh holds those); search_ms.hip
    float cmax_full = 0.0f;       // dot metric: upper bound of |reconstruction of any stored row|
    float *cmaxp = nullptr;       // dot metric: [nlist] upper bound of |centred reconstruction| over the rows of one list
    uint32_t sum_rs = 0, max_rs = 0;   // row slices (of 2048 rows) summed over the partitions / of the largest partition: bound the slice table
  } *ms = nullptr;
  // find_partitions over thousands of lists (xform_fused.hip MODE 2): the centroids' bf16 planes [nlist ^ 64][2 d + 16] and max |c|^2, built by the
  // first such search (they were rebuilt by every call: 65,536 x 128 centroids are 35 MB of planes, ~0.25 of the batch's 1.2 ms sweep stage)
  struct CqConst {
    uint16_t *cpl = nullptr;
    uint32_t *maxbits = nullptr;  // [4] words, [0] = max |c|^2 as float bits
  } *cq = nullptr;
  std::mutex lazy_mu;             // guards the creation of `pt` / `ms` / `cq` (several contexts / host threads may search one index)
  uint32_t *part_offsets = nullptr;  // [nlist+1] device
  std::vector<uint32_t> part_offsets_h;
  uint8_t *codes = nullptr;       // [n][code_bytes()] row-major, rows grouped by partition
  uint64_t *row_ids = nullptr;    // [n] in the same order
  int2_host *flat_items = nullptr;   // IVF_FLAT only: (partition, first row) of every 256-row block, device
  uint32_t n_flat_items = 0;
  float *vectors = nullptr;       // IVF_FLAT only (m == 0): [n][d] f32 vectors in the same order (flat/storage.rs FlatFloatStorage)
  const void *raw = nullptr;      // borrowed raw vectors (dtype elements) for refine, indexed by row id
  uint64_t n_raw = 0;
  // Refine source (search.hip, raw_compact_prepare): when EVERY element of an f32 raw column is an integer in [0, 255] (SIFT descriptors
  // are: u8 values stored as f32) the engine keeps a lossless u8 copy [n_raw][d] and the refine kernel reads that -- a quarter of the
  // bytes of the random row reads that bound it, the same f32 values after widening, hence the same bits out.  Created by the first
  // refining search of the index or by lance_hip_index_prewarm; dropped by lance_hip_index_set_raw.  Guarded by lazy_mu.
  uint8_t *raw_u8 = nullptr;
  uint64_t raw_gen = 0;           // bumped by every lance_hip_index_set_raw: captured search graphs are keyed on it (they hold raw / raw_u8 pointers)
  int raw_compact_state = 0;      // 0: not tried yet; 1: raw_u8 holds the column; -1: the column is not representable (or no memory for the copy)
  uint32_t max_part = 0;
  uint32_t code_bytes() const { return nbits == 4 ? m / 2 : m; }   // bytes of PQ code per row
  ~lance_hip_index();
};
