// kernels.h -- internal launcher prototypes shared by the translation units.
#pragma once
#include <stdint.h>

#include "common.h"

namespace lh {

// One operand per lane (rows of x), the other staged in LDS (centroids).
// Batched views: batch b reads columns [b*x_batch_off, +d) of x (row stride ldx) against
// centroid block cent + b*cent_batch_stride -- this is how the M PQ sub-quantisers run
// in one launch (pq/builder.rs:109-138, pq.rs:146-166).
struct PairwiseArgs {
  const void *x_native = nullptr;   // optional: the rows in the column's own element type (x_dtype); kernels that can read it
  int x_dtype = 0;                  // natively (mfma_assign.hip) then never touch `x` (which may be NULL)
  const float *x = nullptr;
  bool lanes32 = false;             // dot products in the 32-lane order of f16 columns (dot_scalar::<f16, f32, 32>, dot.rs:91-102);
                                    // set by the callers only when the operands are widened f16 values, metric = dot and d > 16
  int64_t n = 0;
  int64_t ldx = 0;
  int x_batch_off = 0;
  const float *cent = nullptr;
  int k = 0;
  int64_t cent_batch_stride = 0;
  const float *bias = nullptr;
  int64_t bias_batch_stride = 0;
  uint32_t *ids = nullptr;
  float *dists = nullptr;
  int64_t out_batch_stride = 0;
  uint8_t *codes = nullptr;  // codes[row*codes_ld + b] = id (0 if none), pq.rs:165
  int codes_ld = 0;
  float *matrix = nullptr;  // MODE 1: [b][n][k]
  const uint8_t *active = nullptr;
  bool x_aligned = false;
  bool cent_aligned = false;
  bool check_finite = false;  // MODE 0: rows with a non-finite element get id NONE
  // long UNIT-LENGTH rows (cosine indices: the normalise kernel writes them): the rows once more as a binary16 plane scaled by 2^14, row stride
  // x_plane_dp (d rounded up to 32, zero padded), and an upper bound of their squared norms (NaN for a row that has none) -- the K-tiled
  // coarse quantiser then runs one f16 product instead of three bf16 ones (mfma_assign.hip: ma_top3_wide_kernel<.., F16>)
  const uint16_t *x_plane16 = nullptr;
  const float *x_plane_n2 = nullptr;
  int x_plane_dp = 0;
  float *part_vb = nullptr;   // k-split partials (set by the launcher)
  float *part_v = nullptr;
  uint32_t *part_idx = nullptr;
};

// PQ sub-quantiser argmin (k = 256, sub-dimension 4 / 8 / 16, L2) on the matrix cores with exact re-check (pq_mfma.hip)
// its argument block
struct PqmArgs {
  PairwiseArgs p;
  // encode mode (fused residual + PQ encode, encode_fused.hip's contract): rows are read from `xn` in the column's own element
  // type (row stride p.ldx elements), row r's operand is xn[r] - rcent[rpart[r]] (rounded to f16 for Float16 columns,
  // residual.rs:96); rows without a partition encode the zero vector.  rcent == NULL: no residual (dot metric).
  const void *xn = nullptr;
  const float *rcent = nullptr;
  const uint32_t *rpart = nullptr;
  int round_f16 = 0;
  uint32_t *fb_cnt;      // [batches] undecided rows per sub-quantiser
  uint32_t *fb_items;    // [batches][n] their row numbers
  int batches;
};
bool pq_mfma_supported(const PairwiseArgs &p, int d, int metric, int batches);
int launch_pq_mfma(lance_hip_ctx *ctx, const PairwiseArgs &p, int d, int batches);
bool pq_mfma_encode_supported(int dtype, int d, int m, int nbits, const void *x, const float *cent, const float *codebook, int64_t n);
int launch_pq_mfma_encode(lance_hip_ctx *ctx, int dtype, const void *x, int64_t n, int d, const float *cent, const uint32_t *part_ids,
                          int residual, const float *codebook, int m, uint8_t *codes);

// flat scan v2: per-query candidate pools + threshold pairs (flat.hip)
struct FlatPool {
  const float *x;       // rows as f32 (NULL when the fixed-dimension kernels read the column in its own element type)
  const void *x_native = nullptr;   // the column itself: f32 / f16 / int8 rows, widened per element in registers
  int x_dtype = 0;                  // LANCE_HIP_F32 / F16 / I8
  const uint64_t *row_ids;
  int64_t r0, r1;       // rows of this epoch
  const float *q;       // queries of this chunk
  int nq, k, cap;
  uint32_t *tkey;       // [nq]
  uint64_t *trid;       // [nq]
  uint32_t *cnt;        // [nq]
  uint32_t *pkeys;      // [nq][cap]
  uint64_t *prids;      // [nq][cap]
  uint32_t *overflow;   // [1]
  const float *row_sy = nullptr;   // cosine: sqrt(y_norm) per row   (cosine.rs:143-175)
  const float *q_norm = nullptr;   // cosine: norm_l2(query)
};

int launch_wide_filter(lance_hip_ctx *ctx, const FlatPool &fp, int d, int metric);   // wide.hip, any d
// search.hip: prefilter by row id -> one bit per storage position (scratch "search.allow_bits")
int build_allow_bits(lance_hip_ctx *ctx, const uint64_t *row_ids, uint64_t n, const uint8_t *allow_by_rowid, uint64_t n_allow, const uint32_t **bits_out);
// bf16x3 MFMA surrogate + exact re-check for query batches (flat_mfma.hip)
bool flat_mfma_supported(int metric, int d, int nq, const void *x, const float *q);
int flat_mfma_prepare(lance_hip_ctx *ctx, const float *q, int nq, int d, const uint16_t **qhi, const uint16_t **qlo, const float **qn);
int launch_flat_filter_mfma(lance_hip_ctx *ctx, const FlatPool &e, int d, int metric, const uint16_t *qhi, const uint16_t *qlo, const float *qn);
// K-tiled bf16 MFMA surrogate + exact re-check for query batches against LONG f32 rows, L2 / dot / cosine (flat_mfma_wide.hip)
bool flat_mfma_wide_supported(int metric, int d, int nq, int64_t n, const float *x, const float *q);
int flat_mfma_wide_prepare_rows(lance_hip_ctx *ctx, const float *x, int64_t n, int d, const uint16_t **xb, const float **xn2);
int flat_mfma_wide_prepare_queries(lance_hip_ctx *ctx, const float *q, int nq, int d, const uint16_t **qb, const float **qn2);
int launch_flat_filter_mfma_wide(lance_hip_ctx *ctx, const FlatPool &e, int d, int metric, const uint16_t *xb, const float *xn2, const uint16_t *qb,
                                 const float *qn2);

int launch_assign(lance_hip_ctx *ctx, const PairwiseArgs &p, int d, int metric, int batches);
bool assign_reads_native(PairwiseArgs p, int d, int batches);
// residual + PQ encode in one kernel, rows read in the column's own element type (encode_fused.hip)
bool encode_fused_supported(int dtype, int d, int m, int nbits, const void *x, const float *cent, const float *codebook);
int launch_encode_fused(lance_hip_ctx *ctx, int dtype, const void *x, int64_t n, int d, const float *cent, const uint32_t *part_ids,
                        int residual, const float *codebook, int m, uint8_t *codes);
// the whole transform in one pass over the rows: coarse assign + exact re-check + residual + PQ encode (xform_fused.hip)
bool xform_fused_supported(int dtype, int metric, int d, int m, int nbits, int64_t n, int nlist, const void *x, const float *cent,
                           const float *codebook, uint8_t *codes, bool lanes32);
int launch_xform_fused(lance_hip_ctx *ctx, int dtype, int metric, const void *x, int64_t n, int d, const float *cent, int nlist,
                       const float *codebook, int m, uint32_t *part_ids, float *dists, uint8_t *codes, bool round_f16);
// long rows (d > 128): residual + PQ encode of 128-column blocks on the same machinery, after the K-tiled coarse quantiser (xform_fused.hip)
bool xform_tail_supported(int d, int m, int nbits, int64_t n, const float *x, const float *cent, const float *codebook);
int launch_xform_tail(lance_hip_ctx *ctx, const float *x, int64_t n, int d, const float *cent, const uint32_t *part_ids, int residual, bool round_f16,
                      const float *codebook, int m, uint8_t *codes);
// the codebook training's E-step (all sub-quantisers, one launch) on the transform's PQ phase (xform_fused.hip); the undecided items go to pq_mfma_fix_kernel
bool xform_pqtrain_supported(const PairwiseArgs &p, int sd, int batches);
int launch_xform_pqtrain(lance_hip_ctx *ctx, const PairwiseArgs &p, int sd, int batches, uint32_t *fb_cnt, uint32_t *fb_items);
// f32 rows of d <= 128: sweep + exact re-check in one kernel (phases 1-3 of the transform kernel, with the k-means bias); same outputs as launch_assign
bool xform_assign_supported(const PairwiseArgs &p, int d, int metric, int batches);
int launch_xform_assign(lance_hip_ctx *ctx, const PairwiseArgs &p, int d, int metric);
// find_partitions over thousands of lists: the transform kernel's sweep with per-group keys (xform_fused.hip; select: coarse_select_kernel, mfma_assign.hip)
// cpl_ready / maxbits_ready: the centroid planes and their maxima built earlier by xform_coarse_planes (an index's constants), or NULL
int launch_xform_sweep_groups(lance_hip_ctx *ctx, int metric, const float *q, uint32_t nq, int d, const float *cent, uint32_t nlist, uint32_t *maxbits,
                              float *gkey, int ng, float *e2, const uint16_t *cpl_ready = nullptr, const uint32_t *maxbits_ready = nullptr);
size_t xform_coarse_planes_elems(uint32_t nlist, int d);      // bf16 elements of the planes
int xform_coarse_planes(lance_hip_ctx *ctx, int metric, const float *cent, uint32_t nlist, int d, uint16_t *cpl, uint32_t *maxbits /* zeroed */);
bool coarse_groups_shape(int d, uint32_t nlist);             // mfma_assign.hip: find_partitions takes the per-group keys for this shape
// bf16x3 MFMA candidates + exact re-check (mfma_assign.hip); same outputs as launch_assign
bool mfma_assign_supported(const PairwiseArgs &p, int d, int batches);
int launch_assign_mfma(lance_hip_ctx *ctx, const PairwiseArgs &p, int d, int metric);
int launch_dist_matrix(lance_hip_ctx *ctx, const PairwiseArgs &p, int d, int metric, int batches);

int stable_group(lance_hip_ctx *ctx, const uint32_t *ids, int64_t n, int64_t id_stride, int k, int batches,
                 uint32_t *starts, uint32_t *sorted_rows, int64_t out_stride, const uint8_t *active);

// f16_arith: T = f16 -- x / cent hold f16-representable values and the M-step rounds like half::f16
int kmeans_train_batched(lance_hip_ctx *ctx, int metric, const float *x, int64_t n, int64_t ldx, int x_batch_off, int d,
                         int k, int B, uint32_t max_iters, double tol, float balance_factor_scaled, bool have_init,
                         const uint64_t *seeds, float *cent, double *loss_out, uint32_t *iters_out, bool f16_arith = false);

int kmeans_train_hierarchical(lance_hip_ctx *ctx, int metric, const float *x, int64_t n, int d, int target_k, uint32_t max_iters,
                              double tol, float bf_scaled, int hierarchical_k, uint64_t seed, float *cent_out, uint32_t *n_out,
                              bool f16_arith = false);
// element-type plumbing (dtype.hip)
int check_dtype(int dtype, const char *what);
int model_dtype(int dtype);   // f32 for int8 data, else the data type itself
int as_f32(lance_hip_ctx *ctx, int dtype, const void *p, size_t count, const char *slot, const float **out);
int widen_into(lance_hip_ctx *ctx, int dtype, const void *p, size_t count, float *dst);
int from_f32(lance_hip_ctx *ctx, int dtype, const float *src, void *dst, size_t count);
int launch_residual(lance_hip_ctx *ctx, const float *x, int64_t n, int d, const float *cent, const uint32_t *part_ids, float *out, bool f16);


bool pm_supported(const lance_hip_index *ix, uint32_t keff, int has_range, uint32_t nq, uint32_t nprobes);
int ivfpq_scan_merge_pm(lance_hip_ctx *ctx, const lance_hip_index *ix, const float *qs, uint32_t nq, const uint32_t *probes,
                        uint32_t nprobes, uint32_t keff, uint32_t k, bool do_refine, uint64_t *ids, float *dists,
                        uint64_t *cand_rid, uint32_t *cand_cnt, uint32_t *flags, const uint32_t *allow);
int find_partitions_f32(lance_hip_ctx *ctx, int metric, const float *qf, uint32_t nq, uint32_t d, const float *cf, uint32_t nlist,
                        uint32_t nprobes, uint32_t *part_ids, float *dists, bool lanes32);   // search.hip
// flat_small.hip: exhaustive KNN for one to four queries in a single streaming pass (*done = false: mass ties, take the batch path)
bool flat_small_supported(int metric, int dtype, uint32_t d, uint32_t nq, uint32_t k, uint64_t n);
int flat_topk_small(lance_hip_ctx *ctx, int metric, int dtype, const void *x, const uint64_t *row_ids, uint64_t n, uint32_t d, const float *q,
                    uint32_t nq, uint32_t k, uint64_t *ids, float *dists, bool *done);
// mfma_assign.hip: the coarse quantiser at query time on the matrix cores (surrogate matrix + exact re-check of the candidates)
bool coarse_mfma_supported(int metric, int d, uint32_t nq, uint32_t nlist, uint32_t nprobes, bool lanes32, const float *q, const float *cent);
int find_partitions_mfma(lance_hip_ctx *ctx, int metric, const float *q, uint32_t nq, int d, const float *cent, uint32_t nlist, uint32_t nprobes,
                         float *matrix, uint32_t *part_ids, float *dists, const uint16_t *cpl_ready = nullptr, const uint32_t *maxbits_ready = nullptr);
int launch_normalize(lance_hip_ctx *ctx, const float *x, int64_t n, int d, float *out, bool f16);   // f16: half-precision arithmetic on f32 containers
// the same, and the normalised rows once more as a binary16 plane x 2^14 (stride dp = d rounded up to 32) + the squared-norm bound per row
int launch_normalize_planes(lance_hip_ctx *ctx, const float *x, int64_t n, int d, float *out, bool f16, uint16_t *plane16, int dp, float *n2);

// quantised 4-query filter scan + exact re-evaluation (search_q.hip), driven by ivfpq_scan_merge_pm
int qscan_index_constants(lance_hip_ctx *ctx, lance_hip_index *ix);
constexpr int QSCAN_SEG_CAP = 256;   // survivors kept per (query, probe)
struct SelectOut;
bool qscan_supported(const lance_hip_index *ix, uint32_t nq, uint32_t nprobes);
// search_qt.hip: M = 48 / 64 / 96 (table tiled over the sub-quantisers); class-B queries of those shapes go to the rescan kernel
bool qscan_tiled_shape(int m, int sd);
int qscan_classb_to_rescan(lance_hip_ctx *ctx, const uint32_t *tbound, uint32_t nq, uint32_t nprobes, uint32_t *seg_cnt, uint32_t *qovf);
int qscan_nearest_keys(lance_hip_ctx *ctx, const uint32_t *probes, uint32_t nq, uint32_t nprobes, uint32_t *keys, uint32_t nb = 1);
int qscan_item_tables(lance_hip_ctx *ctx, const uint32_t *pair_starts, int nlist, int G, uint32_t *item_start, int4 *desc, uint32_t max_items);
// G = queries per work item of the main pass
int qscan_group(lance_hip_ctx *ctx, const uint32_t *probes, uint32_t nq, uint32_t nprobes, int nlist, const uint32_t *tglobal,
                uint32_t *keys, uint32_t *tbound, uint32_t *pair_starts, uint32_t *pair_idx, uint32_t *item_start4, int4 *desc4,
                uint32_t max_items4, int G = 4, int dot = 0);
int qbound_launch(lance_hip_ctx *ctx, const lance_hip_index *ix, const float *qs, uint32_t nq, uint32_t keff, const uint32_t *pair_starts0,
                  const uint32_t *pair_idx0, uint32_t *item_start, int4 *desc, uint32_t max_items, uint32_t *tglobal, const uint32_t *allow);
int qscan_launch(lance_hip_ctx *ctx, const lance_hip_index *ix, const float *qs, uint32_t nq, uint32_t nprobes, const uint32_t *pair_idx,
                 const uint32_t *item_start4, const int4 *desc4, uint32_t max_items4, const uint32_t *tbound, uint32_t *seg_cnt,
                 uint32_t *seg_pos, uint32_t *qovf, const uint32_t *allow, const uint32_t *probes = nullptr);
// search_qt.hip: per-query tables + per-row bias instead of a table per (query, partition) (M = 48 / 64 / 96; LANCE_HIP_QPT=1)
bool qscan_pt_enabled(const lance_hip_index *ix);
int qscan_pt_mode(const lance_hip_index *ix);      // 0 off, 1 tables after the bound pass, 2 tables before it (shared by both passes)
int qbound_pt_launch(lance_hip_ctx *ctx, const lance_hip_index *ix, const float *qs, uint32_t nq, uint32_t nprobes, uint32_t keff,
                     const uint32_t *probes, const uint32_t *pair_starts0, const uint32_t *pair_idx0, uint32_t *item_start, int4 *desc,
                     uint32_t max_items, uint32_t *tglobal, const uint32_t *allow);
int qmerge_launch(lance_hip_ctx *ctx, const lance_hip_index *ix, const float *qs, uint32_t nq, const uint32_t *probes, uint32_t nprobes,
                  const uint32_t *tbound, uint32_t *tglobal, const uint32_t *seg_cnt, const uint32_t *seg_pos, const uint32_t *qovf,
                  uint32_t *pool_key, uint32_t *pool_pos, uint32_t *pool_cnt, int pool_cap, const SelectOut &o, const uint32_t *allow,
                  const uint32_t *qslack = nullptr, const float *seg_val = nullptr, const float *seg_scale = nullptr);
// "this launcher does not serve the call, take the other route": POSITIVE, because every LANCE_HIP_E* error code is negative (ADVICE r05:
// -1 doubled as EINVAL and a failure inside the launcher fell back silently)
constexpr int LH_NOT_TAKEN = 1;
// search_ms.hip: the filter scan as a [rows x d] x [d x queries] product per partition on the matrix cores (8-bit PQ, d = 64 / 128, M = 16 / 32)
bool mscan_supported(const lance_hip_index *ix, uint32_t nq, uint32_t nprobes);
bool mscan_batch_shape(const lance_hip_index *ix, uint32_t nq, uint32_t nprobes);
bool mscan_dot_ready(const lance_hip_index *ix, uint32_t nq, uint32_t nprobes);   // dot metric: the batch can take the matrix-core bound pass + scan   // shape + batch-size part of mscan_supported
int mscan_launch(lance_hip_ctx *ctx, const lance_hip_index *ix, const float *qs, uint32_t nq, uint32_t nprobes, const uint32_t *probes,
                 const uint32_t *pair_starts, const uint32_t *pair_idx, const uint32_t *tbound, uint32_t *seg_cnt, uint32_t *seg_pos,
                 uint32_t *qovf, const uint32_t *allow, uint32_t **qslack_out, float **seg_val_out, float **seg_scale_out);
void mscan_cut_params(int *cut_shift, uint32_t *cut_slack);
int mscan_prewarm(lance_hip_ctx *ctx, const lance_hip_index *ix);
int msbound_launch(lance_hip_ctx *ctx, const lance_hip_index *ix, const float *qs, uint32_t nq, uint32_t keff, const uint32_t *pair_starts0,
                   const uint32_t *pair_idx0, uint32_t *item_start, int4 *desc, uint32_t max_items, uint32_t *tglobal, const uint32_t *allow,
                   uint32_t nb = 1);   // search_ms.hip: LH_NOT_TAKEN = the integer bound pass serves the batch
int qscan_items(lance_hip_ctx *ctx, const uint32_t *pair_starts, int nvp, int G, uint32_t *item_start, int4 *desc, uint32_t max_items);   // search_q.hip: work items of G grouped pairs      // builds the scan's index constants now (lance_hip_index_prewarm)
const uint8_t *raw_compact_prepare(lance_hip_ctx *ctx, const lance_hip_index *ix);   // search.hip: lossless u8 refine copy (index.h), or nullptr

}  // namespace lh
