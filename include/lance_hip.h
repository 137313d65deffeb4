/*
 * lance_hip.h -- C ABI of liblance_hip.so: the MI355X (gfx950) engine for Lance's
 * IVF-PQ hot path (k-means training, PQ codebook learning/encoding, flat and IVF-PQ
 * distance scans with top-k).
 *
 * This is the drop-in boundary: a Rust `extern "C"` block (or JNI / ctypes) binds
 * exactly these symbols; INTEGRATION.md shows the reference-side call sites.
 * Every entry point cites the reference interface it replaces (paths relative to the
 * lancedb/lance tree).
 *
 * Conventions
 *   - Plain C types only.  All data pointers are DEVICE pointers (HBM) unless the
 *     parameter name ends in `_host`; buffers are caller-owned, row-major, and no
 *     allocation crosses the ABI except opaque handles.  lance_hip_malloc/free/memcpy
 *     are provided so a host without its own HIP binding can stage data.
 *   - Return 0 on success, a negative LANCE_HIP_E* code on failure; the message is
 *     available from lance_hip_last_error() (thread-local).  Never throws or aborts.
 *   - Work is issued on the context's stream; calls return after the stream has been
 *     synchronised unless stated otherwise.
 *   - Threads (the reference calls these paths from rayon / tokio worker threads,
 *     rust/lance/src/index/vector/ivf/v2.rs:232-306): every entry point is re-entrant.  A
 *     context is a stream + scratch arena; its entry points hold a per-context lock, so
 *     threads sharing one context take turns, and threads with a context each overlap on
 *     the device.  An index is read-only during searches and may be searched through any
 *     number of contexts at once (tests/test_zz_gpu_threads.py); building / destroying it
 *     while it is being searched is the caller's error.  lance_hip_last_error() is
 *     thread-local.  A context's own stream is non-blocking (no implicit ordering against the
 *     legacy default stream).  While one thread's repeated search is being captured into a HIP
 *     graph (its second call with the same arguments), a DEVICE-WIDE synchronise from another
 *     thread (hipDeviceSynchronize, torch.cuda.synchronize()) is refused by the runtime and
 *     invalidates that capture: the library then runs the batch on the plain path, but the
 *     synchronising thread sees the runtime's error.  Hosts that search from several threads
 *     wait on streams or events, not on the device (lance_amd/engine.py does), or set
 *     LANCE_HIP_GRAPH=0.
 *   - "No partition" (all-NaN row; kmeans.rs:1447-1486) is id 0xFFFFFFFF.
 *   - Results are bit-identical to the reference CPU path for ids / codes / distances
 *     (see DESIGN.md for the exact statement and the two documented tie rules).
 */
#ifndef LANCE_HIP_H
#define LANCE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LANCE_HIP_OK 0
#define LANCE_HIP_EINVAL -1    /* bad argument */
#define LANCE_HIP_ERUNTIME -2  /* HIP runtime error */
#define LANCE_HIP_ENOTSUP -3   /* combination not implemented */
#define LANCE_HIP_ENOMEM -4
#define LANCE_HIP_EIO -5       /* file cannot be opened / read / written */

#define LANCE_HIP_NONE 0xFFFFFFFFu

/* lance_linalg::distance::DistanceType (distance.rs:36).  Cosine is handled the way
 * the reference index does: normalise, then L2 (lance-index ivf.rs:198-205).        */
enum { LANCE_HIP_L2 = 0, LANCE_HIP_COSINE = 1, LANCE_HIP_DOT = 2 };
/* element type of vectors / centroids / codebook.
 * LANCE_HIP_I8: the DATA operands (vectors x, queries q, raw vectors for refine) are int8 and are widened to
 * f32 exactly, as the reference does for Int8 columns (kmeans.rs:1216-1224 convert_to_floating_point,
 * l2.rs:253-260); the MODEL operands (centroids, codebook, init centroids, residual query, every output) are f32. */
enum { LANCE_HIP_F32 = 0, LANCE_HIP_F16 = 1, LANCE_HIP_I8 = 2 };

typedef struct lance_hip_ctx lance_hip_ctx;
typedef struct lance_hip_index lance_hip_index;

/* ---- context ------------------------------------------------------------------ */
/* stream == NULL: the context creates and owns a stream; otherwise it borrows the
 * caller's hipStream_t (e.g. torch's current stream).                               */
int lance_hip_ctx_create(int device_id, void *stream, lance_hip_ctx **out);
void lance_hip_ctx_destroy(lance_hip_ctx *ctx);
const char *lance_hip_last_error(void);
const char *lance_hip_version(void);
int lance_hip_synchronize(lance_hip_ctx *ctx);

int lance_hip_malloc(lance_hip_ctx *ctx, size_t bytes, void **out);
int lance_hip_free(lance_hip_ctx *ctx, void *ptr);
int lance_hip_memcpy_h2d(lance_hip_ctx *ctx, void *dst, const void *src_host, size_t bytes);
int lance_hip_memcpy_d2h(lance_hip_ctx *ctx, void *dst_host, const void *src, size_t bytes);

/* ---- a4: normalize_fsl (lance-linalg kernels.rs:141-146,172-211) ---------------- */
int lance_hip_normalize(lance_hip_ctx *ctx, int dtype, const void *x, uint64_t n, uint32_t d, void *out);

/* ---- a5/a6/a9: argmin over centroids ------------------------------------------- */
/* KMeansAlgoFloat::compute_membership_and_dist (lance-index kmeans.rs:317-369),
 * compute_partitions_arrow_array (:1187-1246), compute_partition (:1350-1369) with
 * argmin_value_float[_with_bias] (lance-linalg kernels.rs:79-111).
 * bias: k floats added before the comparison (un-biased distance returned) or NULL.
 * ids: n (LANCE_HIP_NONE = None); dists: n or NULL.                                  */
int lance_hip_assign(lance_hip_ctx *ctx, int dtype, int metric, const void *x, uint64_t n, uint32_t d,
                     const void *centroids, uint32_t k, const float *bias, uint32_t *ids, float *dists);

/* ---- a7/a8: KMeans::train_kmeans (kmeans.rs:610-719) ---------------------------- */
/* Lloyd iterations on exactly the n rows given (the caller applies the sample_rate*k
 * slice of kmeans.rs:1328-1340; the k*512 cap of :623-627 is applied here).
 * balance_factor is the caller's value (1.0 for IVF, ivf.rs:1859); it is divided by n
 * as train_kmeans does (:1344).  init_centroids: k*d or NULL (random rows from `seed`,
 * kmeans_random_init :149-170).  Given the same inputs, init and seed the centroids,
 * loss and iteration count are bit-identical to the reference algorithm (M-step sums
 * are accumulated per centroid in row order, :371-446).  As in the reference, k > 256
 * takes the hierarchical path (see lance_hip_kmeans_train_ex).                          */
int lance_hip_kmeans_train(lance_hip_ctx *ctx, int dtype, int metric, const void *x, uint64_t n, uint32_t d,
                           uint32_t k, uint32_t max_iters, double tol, float balance_factor,
                           const void *init_centroids, uint64_t seed, void *centroids_out,
                           double *loss_out_host, uint32_t *iters_out_host);

/* KMeans::new_with_params (kmeans.rs:1008-1073) with every KMeansParams field: k > 256 and
 * hierarchical_k > 1 selects train_hierarchical_kmeans (:746-1003; init centroids unused, loss 0,
 * *k_out_host = clusters produced, normally k).  lance_hip_kmeans_train == hierarchical_k 16.  */
int lance_hip_kmeans_train_ex(lance_hip_ctx *ctx, int dtype, int metric, const void *x, uint64_t n, uint32_t d,
                              uint32_t k, uint32_t max_iters, double tol, float balance_factor,
                              uint32_t hierarchical_k, const void *init_centroids, uint64_t seed,
                              void *centroids_out, double *loss_out_host, uint32_t *iters_out_host,
                              uint32_t *k_out_host);

/* Building blocks of one Lloyd iteration, for a host that owns the loop (multi-GPU:
 * one process per GPU, rows sharded, one all-reduce per iteration).
 * estep_partial assigns the LOCAL rows and accumulates, per centroid and in local row
 * order: buf = [k*d sums | k counts] as f32 (one all-reduce(sum) buffer), and optionally
 * losses[k] (f64, all-reduce sum) and radius[k] (f32, all-reduce max) -- the inputs of
 * compute_cluster_sizes / compute_balance_loss (kmeans.rs:210-237).                     */
int lance_hip_kmeans_estep_partial(lance_hip_ctx *ctx, int dtype, int metric, const void *x, uint64_t n,
                                   uint32_t d, const void *centroids, uint32_t k, const float *bias,
                                   float *buf /* k*d + k */, double *losses /* k or NULL */,
                                   float *radius /* k or NULL */, double *loss_out_host);
/* finalize: centroids = sums * (1/count) for count > 0 (kmeans.rs:410-418). */
int lance_hip_kmeans_finalize(lance_hip_ctx *ctx, int dtype, const float *buf, uint32_t k, uint32_t d,
                              void *centroids_out);

/* One split of the hierarchical trainer -- train_hierarchical_kmeans, kmeans.rs:746-1003: k-means with k centroids over the rows
 * `rows_host` (n_rows indices into x, ascending; NULL = all rows: the first level) of the training sample x [n][d] (device, f32 values;
 * dtype = the COLUMN's type: LANCE_HIP_F16 says the values are binary16-representable -- widened once by the caller -- and the M-step
 * rounds like half::f16, as the single-GPU trainer of an f16 column does), then the membership of those rows
 * (kmeans.rs:866-905).  balance_factor_scaled is the factor ALREADY divided by the sample size (train_kmeans :1344 divides once, by the
 * whole sample, and every split inherits it); seed = the run's seed (the trainer uses seed + number of k-means runs so far).  Outputs on
 * the host: centroids [k][d] f32, membership [n_rows] (LANCE_HIP_NONE: no centroid).
 * This is the unit of work of the multi-GPU hierarchical trainer (lance_amd/dist.py train_kmeans_hierarchical_sharded, SURVEY 8(e);
 * BASELINE config 5: nlist 65,536): every rank holds the sample, the splits of the largest clusters are computed on different ranks at
 * the same time and applied in the reference's order -- the result is the single-GPU trainer's bit for bit.                         */
int lance_hip_kmeans_split(lance_hip_ctx *ctx, int dtype, int metric, const float *x, uint64_t n, uint32_t d, const uint32_t *rows_host,
                           uint64_t n_rows, uint32_t k, uint32_t max_iters, double tol, float balance_factor_scaled, uint64_t seed,
                           float *centroids_out_host, uint32_t *membership_out_host);

/* Row-sharded Lloyd iteration WITHOUT host round trips (multi-GPU build).  Per iteration the caller enqueues, on
 * the stream the context was created on (e.g. torch's current stream, so that RCCL collectives are ordered with it):
 *   shard_estep  -- local E-step + local partials: buf = [k*d sums | k counts] (f32), losses [k] (f64), radius [k] (f32);
 *   all-reduce SUM of buf and of losses, all-reduce MAX of radius (the caller's collective library);
 *   shard_update -- on the reduced numbers: centroids = sums / counts, loss, balance-factor update, convergence test,
 *                   empty-cluster split (RNG seeded identically on every rank), bias of the next iteration.
 * `state` is an opaque device block of LANCE_HIP_KMEANS_SHARD_STATE_BYTES bytes; `bias` k device floats.  Once the run has
 * converged the state turns inactive: further estep / update calls leave every buffer untouched.  shard_end synchronises and
 * reports (loss, iterations, still active).  Reference loop: KMeans::train_kmeans, kmeans.rs:610-719.                     */
#define LANCE_HIP_KMEANS_SHARD_STATE_BYTES 128
/* The k row indices kmeans_random_init draws (kmeans.rs:149-170: k distinct rows; here the engine's seeded reservoir so that
 * seeded runs are reproducible -- the reference seeds from the OS).  Host-only helper for callers that gather the initial
 * centroids themselves (the sharded trainer: rank 0 draws from its rows and broadcasts).                                */
int lance_hip_kmeans_init_indices(uint64_t n, uint32_t k, uint64_t seed, uint64_t *out_host);
int lance_hip_kmeans_shard_begin(lance_hip_ctx *ctx, uint32_t k, float balance_factor_scaled, uint64_t seed, void *state, float *bias);
int lance_hip_kmeans_shard_estep(lance_hip_ctx *ctx, int metric, const float *x, uint64_t n, uint32_t d, const float *centroids, uint32_t k,
                                 const float *bias, const void *state, float *buf, double *losses, float *radius);
/* the same E-step on a shard in the column's own element type (dtype: f32 / f16 / int8 -- widened exactly inside; VERDICT r05: the f32-only
 * entry points made the host widen f16 / int8 shards) */
int lance_hip_kmeans_shard_estep_x(lance_hip_ctx *ctx, int dtype, int metric, const void *x, uint64_t n, uint32_t d, const float *centroids, uint32_t k,
                                   const float *bias, const void *state, float *buf, double *losses, float *radius);
int lance_hip_kmeans_shard_update(lance_hip_ctx *ctx, void *state, const float *buf, const double *losses, const float *radius,
                                  float *centroids, float *bias, uint32_t k, uint32_t d, uint64_t n_total, float balance_factor_scaled,
                                  double tol, uint32_t it);
int lance_hip_kmeans_shard_end(lance_hip_ctx *ctx, const void *state, double *loss_host, uint32_t *iters_host, int *active_host);

/* ---- the same loop with its collectives behind the ABI (RCCL over xGMI) ----------------------------------------
 * For hosts without torch.distributed.  One process per GPU; rank 0 draws an id (lance_hip_comm_unique_id, 128 bytes = an
 * ncclUniqueId), ships it to the other ranks by whatever channel the host has, every rank calls lance_hip_comm_create on its own
 * context -- or lance_hip_comm_adopt wraps an ncclComm_t the host already owns (it stays the host's to destroy).
 * lance_hip_kmeans_train_sharded runs KMeans::train_kmeans (kmeans.rs:610-719) over rows sharded across the ranks: per Lloyd
 * iteration a local E-step + partial sums, ONE ncclAllReduce(sum) of the fused f32 buffer [k*d sums | k counts] (129 KiB at
 * nlist 256, d 128), one of the k f64 losses and one max-reduce of the k radii, then the update kernel -- all enqueued on the
 * context's stream; the host reads the state every 8 iterations.  `centroids` holds the initial centroids (identical on every
 * rank: e.g. rank 0's kmeans_random_init rows, broadcast by the host) and receives the trained ones (identical on every rank).
 * comm == NULL: single process, no collective.  balance_factor is the unscaled parameter (divided by n_total as train_kmeans does).
 * Sums arrive in rank order rather than row order: against the single-GPU trainer the centroids agree to f32 round-off for more
 * than one rank and bit for bit for one.  An RCCL the process already maps answers; otherwise librccl.so.1 is loaded on first use.
 * Whatever can fail on one rank only (its row count, scratch, the f32 copy of an f16 / int8 shard, its first E-step) is exchanged as one
 * status word before the first all-reduce of the loop: every rank returns.                                                          */
typedef struct lance_hip_comm lance_hip_comm;
int lance_hip_comm_unique_id(char *id_out_host /* 128 bytes */);
int lance_hip_comm_create(lance_hip_ctx *ctx, const char *id_host /* 128 bytes */, int nranks, int rank, lance_hip_comm **out);
int lance_hip_comm_adopt(void *nccl_comm, int nranks, int rank, lance_hip_comm **out);
/* A host with its own transport (MPI, gloo, ...) hands over an in-place all-reduce instead of an RCCL communicator: `buf` is a DEVICE
 * pointer holding `count` elements (LANCE_HIP_COMM_F32 / _F64), `op` LANCE_HIP_COMM_SUM / _MAX, `stream` the hipStream_t the producing
 * kernels were enqueued on and the consuming ones will be (the callback must order itself against it -- e.g. synchronise it, reduce
 * through host memory, copy back -- and return 0, or non-zero to abort the training on this rank).  lance_hip_kmeans_train_sharded
 * issues the same exchanges through it as through RCCL (rust/lance-index/src/vector/kmeans.rs:610-719: the rayon reduction's place). */
enum { LANCE_HIP_COMM_F32 = 0, LANCE_HIP_COMM_F64 = 1 };
enum { LANCE_HIP_COMM_SUM = 0, LANCE_HIP_COMM_MAX = 1 };
typedef int (*lance_hip_allreduce_fn)(void *user, void *buf, uint64_t count, int dtype, int op, void *stream);
int lance_hip_comm_from_callback(lance_hip_allreduce_fn fn, void *user, int nranks, int rank, lance_hip_comm **out);
void lance_hip_comm_destroy(lance_hip_comm *comm);
int lance_hip_kmeans_train_sharded(lance_hip_ctx *ctx, lance_hip_comm *comm, int metric, const float *x_local, uint64_t n_local, uint32_t d,
                                   uint32_t k, uint64_t n_total, uint32_t max_iters, double tol, float balance_factor, uint64_t seed,
                                   float *centroids, double *loss_out_host, uint32_t *iters_out_host);

/* ... and on a shard in the column's own element type (f32 / f16 / int8; the shard is widened once, inside, for the whole loop) */
int lance_hip_kmeans_train_sharded_x(lance_hip_ctx *ctx, lance_hip_comm *comm, int dtype, int metric, const void *x_local, uint64_t n_local, uint32_t d,
                                     uint32_t k, uint64_t n_total, uint32_t max_iters, double tol, float balance_factor, uint64_t seed,
                                     float *centroids, double *loss_out_host, uint32_t *iters_out_host);

/* ---- a11: PQBuildParams::build_from_fsl (pq/builder.rs:89-157) ------------------- */
/* M independent k-means (k = 2^nbits, L2, no balance) over the sub-vector columns of
 * `residuals`; sub-quantiser m uses seed + m.  codebook_out: [m][2^nbits][d/m].      */
int lance_hip_pq_train(lance_hip_ctx *ctx, int dtype, const void *residuals, uint64_t n, uint32_t d,
                       uint32_t m, uint32_t nbits, uint32_t max_iters, uint32_t sample_rate, uint64_t seed,
                       void *codebook_out, uint32_t *iters_out_host /* m or NULL */);

/* ---- a10: do_compute_residual (residual.rs:58-102) ------------------------------- */
int lance_hip_residual(lance_hip_ctx *ctx, int dtype, const void *x, uint64_t n, uint32_t d,
                       const void *centroids, const uint32_t *part_ids, void *out);

/* ---- a12: ProductQuantizer::transform_impl (pq.rs:116-191) ----------------------- */
/* codes: [n][m] (nbits = 8) or [n][m/2] (nbits = 4).  `metric` is the QUANTIZER's distance type (pq.rs:143): a quantizer
 * produced by an index build is always an L2 one (lance/src/index/vector/builder.rs:456), whatever the index metric. */
int lance_hip_pq_encode(lance_hip_ctx *ctx, int dtype, int metric, const void *x, uint64_t n, uint32_t d,
                        const void *codebook, uint32_t m, uint32_t nbits, uint8_t *codes);

/* ---- a9+a10+a12: the IvfTransformer chain for one batch (lance-index ivf.rs:188-236;
 * precedent: one_pass_assign_ivf_pq_on_accelerator, python/lance/vector.py:607-755) --- */
/* [cosine: normalise] -> assign -> residual (L2/cosine) -> PQ encode.  Emits the
 * shuffle-buffer columns (__ivf_part_id u32, __pq_code u8[m]).  Non-finite rows get
 * part id LANCE_HIP_NONE (KeepFiniteVectors, utils.rs:263-286).  `metric` is the INDEX metric: it selects the
 * partition assignment (dot indices assign by dot product and take no residual); the PQ step always encodes with the
 * L2-nearest codeword, as the reference's build does (builder.rs:456 + pq.rs:143,165).   */
int lance_hip_ivfpq_encode(lance_hip_ctx *ctx, int dtype, int metric, const void *x, uint64_t n, uint32_t d,
                           const void *centroids, uint32_t nlist, const void *codebook, uint32_t m,
                           uint32_t nbits, uint32_t *part_ids, uint8_t *codes, double *loss_out_host);

/* ---- a13 + builder.rs:685-846: per-partition storage ------------------------------ */
/* Builds a device-resident index from the shuffle-buffer columns: rows are grouped by
 * partition in ascending input order (stable), rows with part id NONE are dropped.
 * row_ids: n u64 labels or NULL (row i gets id i).  The index keeps its own copies.   */
int lance_hip_index_create(lance_hip_ctx *ctx, int dtype, int metric, uint32_t d, const void *centroids,
                           uint32_t nlist, const void *codebook, uint32_t m, uint32_t nbits,
                           const uint32_t *part_ids, const uint8_t *codes, const uint64_t *row_ids, uint64_t n,
                           lance_hip_index **out);
/* Same, from the reference's on-disk/storage layout (pq/storage.rs:183-290): rows
 * already grouped by partition (part_offsets_host[nlist+1]) and each partition's codes
 * transposed to [m][n_p] (transposed != 0) or row-major.                              */
int lance_hip_index_from_storage(lance_hip_ctx *ctx, int dtype, int metric, uint32_t d, const void *centroids,
                                 uint32_t nlist, const void *codebook, uint32_t m, uint32_t nbits,
                                 const uint32_t *part_offsets_host, const uint8_t *codes, int transposed,
                                 const uint64_t *row_ids, uint64_t n, lance_hip_index **out);
void lance_hip_index_destroy(lance_hip_index *idx);
/* Optional raw vectors for refine (scanner.rs:2884-2904 `take` + flat_knn): x[n_raw][d],
 * indexed by row id (row id r -> x[r]); borrowed, must outlive the index and must not
 * change while attached: the engine may keep a lossless compact copy of it (an f32 column
 * whose every element is an integer in [0, 255] is read as bytes by the refine kernel) and
 * captured searches hold its address.  After changing the contents call set_raw again
 * (same pointer allowed): it drops the copy and invalidates the captured searches.       */
int lance_hip_index_set_raw(lance_hip_index *idx, const void *x, uint64_t n_raw);
/* Index::prewarm (rust/lance/src/index/vector/ivf/v2.rs:349-352, python dataset.py:2991 prewarm_index): builds NOW, on ctx's
 * stream, the per-index search constants the first search would otherwise build (the matrix-core scan's f16 codebook and row norms;
 * the lossless u8 refine copy of an integer-valued f32 raw column).  Optional: results never depend on it, only the first search's
 * latency does.  Synchronises the stream.                                                                                        */
int lance_hip_index_prewarm(lance_hip_ctx *ctx, lance_hip_index *idx);
int lance_hip_index_info(const lance_hip_index *idx, uint64_t *n_rows, uint32_t *nlist, uint32_t *m, uint32_t *d);
/* Copies out the storage layout (host pointers, any may be NULL): part_offsets[nlist+1],
 * codes transposed per partition (the reference layout), row ids in partition order. */
int lance_hip_index_export(lance_hip_ctx *ctx, const lance_hip_index *idx, uint32_t *part_offsets_host,
                           uint8_t *codes_transposed_host, uint64_t *row_ids_host);

/* ---- a14: IvfModel::find_partitions (ivf/storage.rs:107-119, kmeans.rs:1134-1158) -- */
/* Batched.  Ascending by distance; equal distances ordered by partition id (the
 * reference's partial sort is unstable, so any order of equals is a valid outcome).
 * part_ids / dists: [nq][nprobes].  Cosine: q must already be normalised.
 * nlist <= 65536; with more than 8192 partitions nprobes <= 256.                        */
int lance_hip_find_partitions(lance_hip_ctx *ctx, int dtype, int metric, const void *q, uint32_t nq, uint32_t d,
                              const void *centroids, uint32_t nlist, uint32_t nprobes, uint32_t *part_ids,
                              float *dists);

/* ---- a16+a17+a19: one partition (PQDistCalculator::new + distance_all, pq/storage.rs:
 * 854-960; FlatIndex::search, flat/index.rs:82-177) --------------------------------- */
/* q_residual: [d] already residualised (v2.rs:316-332).  codes_transposed: [m][n_p].
 * has_range: keep only lower <= dist < upper.  out_*: capacity k; *out_n_host = count;
 * output is sorted by (dist, row id).                                                 */
int lance_hip_pq_scan_topk(lance_hip_ctx *ctx, int dtype, int metric, const void *q_residual, uint32_t d,
                           const void *codebook, uint32_t m, uint32_t nbits, const uint8_t *codes_transposed,
                           const uint64_t *row_ids, uint64_t n_p, uint32_t k, int has_range, float lower,
                           float upper, uint64_t *out_ids, float *out_dists, uint32_t *out_n_host);

/* ---- a14..a21 batched: the ANN query (knn.rs:359-1075, v2.rs:455-505, scanner.rs:
 * 3440-3468, refine :2884-2904) -------------------------------------------------------- */
/* q: [nq][d] raw queries (cosine: normalised internally, knn.rs:495-498).  nprobes =
 * minimum_nprobes = maximum_nprobes.  refine_factor 0 = None (no refine); rf >= 1 = Some(rf):
 * k*rf candidates re-ranked with exact distances on the raw vectors.  ids/dists: [nq][k],
 * missing results are id UINT64_MAX / dist +inf.                                      */
int lance_hip_ivfpq_search(lance_hip_ctx *ctx, const lance_hip_index *idx, const void *q, uint32_t nq,
                           uint32_t k, uint32_t nprobes, uint32_t refine_factor, uint64_t *ids, float *dists);
/* Same, but only enqueues on the stream (no synchronisation, no host reads): for
 * callers that time or pipeline batches.  Scratch is owned by the context.           */
int lance_hip_ivfpq_search_async(lance_hip_ctx *ctx, const lance_hip_index *idx, const void *q, uint32_t nq,
                                 uint32_t k, uint32_t nprobes, uint32_t refine_factor, uint64_t *ids,
                                 float *dists);

/* Distance-range query (`nearest={..., "distance_range": (lower, upper)}`; Query::lower_bound / upper_bound): inside
 * every probed partition only rows with lower <= d < upper enter the k-heap (FlatIndex::search, flat/index.rs:98-113),
 * d being the ADC distance (after the dot offset).  An open end is -FLT_MAX / FLT_MAX as in the reference
 * (`unwrap_or(f32::MIN)` / `unwrap_or(f32::MAX)`).  refine_factor as in lance_hip_ivfpq_search: the k * refine_factor
 * candidates that passed the ADC range are re-ranked by exact distance -- the exact distances are NOT filtered here; a
 * caller reproducing the reference plan (scanner.rs:3334-3377 filters them before the final fetch) asks for all
 * candidates (k = k * rf, refine_factor = 1) and applies the range to the returned exact distances.                 */
int lance_hip_ivfpq_search_range(lance_hip_ctx *ctx, const lance_hip_index *idx, const void *q, uint32_t nq, uint32_t k,
                                 uint32_t nprobes, uint32_t refine_factor, float lower, float upper, uint64_t *ids,
                                 float *dists);

/* One scan, two distances per candidate -- the local half of a LIST-SHARDED multi-GPU search with refine (search is
 * embarrassingly parallel over IVF lists; scanner.rs:2884-2904 re-ranks the k * refine_factor best by PQ distance with exact
 * distances): ids / pq_dists [nq][keff] = the keff nearest rows of this index by PQ distance in (dist, rowid) order (~0 / +inf
 * beyond the found ones), exactly lance_hip_ivfpq_search(k = keff, refine_factor = 0); exact_dists [nq][keff] (may be NULL) = each
 * candidate's exact distance to the query in the index's metric, the refine kernels' arithmetic, in the SAME order.  The ranks
 * all-gather the three arrays and lance_hip_merge_topk picks the global keff by PQ distance and orders them by the exact one.
 * (Round 5 obtained the exact distances from a second scan with refine_factor = 1 and aligned the two lists with sorts.)          */
int lance_hip_ivfpq_search_candidates(lance_hip_ctx *ctx, const lance_hip_index *idx, const void *q, uint32_t nq, uint32_t keff,
                                      uint32_t nprobes, uint64_t *ids, float *pq_dists, float *exact_dists);

/* The same search under a row-id prefilter (`nearest=..., filter=..., prefilter=True`: scanner.rs -> DatasetPreFilter ->
 * FlatIndex::search's RowIdMask branch, flat/index.rs:129-165).  allow_by_rowid[r] != 0 <=> row id r may be returned; rows
 * whose id is >= n_allow are filtered out.  The mask is applied INSIDE the scan kernels (one bit per stored row, tested
 * before a row can become a candidate, bound pass included): no per-filter copy of the index.  The result is identical
 * to the reference's per-row distance(id) loop.  4-bit PQ: the reference scores filtered rows with the UNQUANTISED f32
 * table, byte-wise terms (pq/storage.rs:893-921) -- a different arithmetic from its own unfiltered fast-scan -- and so do
 * the 4-bit scan kernel and its exact replay here when a mask is given (search.hip pq4_masked_row).                     */
int lance_hip_ivfpq_search_filtered(lance_hip_ctx *ctx, const lance_hip_index *idx, const void *q, uint32_t nq, uint32_t k,
                                    uint32_t nprobes, uint32_t refine_factor, const uint8_t *allow_by_rowid, uint64_t n_allow,
                                    uint64_t *ids, float *dists);

/* Prefilter and distance range together: FlatIndex::search's RowIdMask branch with lower / upper bounds
 * (flat/index.rs:131-149: selected rows only, each scored with distance(id), kept when lower <= d < upper). */
int lance_hip_ivfpq_search_filtered_range(lance_hip_ctx *ctx, const lance_hip_index *idx, const void *q, uint32_t nq, uint32_t k,
                                          uint32_t nprobes, uint32_t refine_factor, const uint8_t *allow_by_rowid, uint64_t n_allow,
                                          float lower, float upper, uint64_t *ids, float *dists);

/* Number of queries of the most recent search on this context that had to be replayed by the exact
 * (heap-emulating) kernel -- ties at a partition's k-th distance, or candidate-buffer overflow.      */
int lance_hip_search_stats(lance_hip_ctx *ctx, uint32_t *n_exact_replays_host);

/* ---- a20+a21: flat KNN (flat.rs:95-148, l2.rs:245-266, scanner.rs:3386-3411) ------- */
/* Exhaustive scan of x[n][d] for nq queries; result sorted by (dist, row id).
 * row_ids NULL -> row index.                                                          */
int lance_hip_flat_topk(lance_hip_ctx *ctx, int dtype, int metric, const void *x, const uint64_t *row_ids,
                        uint64_t n, uint32_t d, const void *q, uint32_t nq, uint32_t k, uint64_t *ids,
                        float *dists);

/* ---- N4: IVF_FLAT (FlatIndex sub-index over raw vectors: flat/index.rs:82-177, flat/storage.rs:345-402) ------ */
/* Builds the per-partition FlatFloatStorage on the device: x[n][d] (dtype elements, widened exactly to f32) is
 * gathered into partition order (stable, rows with part id LANCE_HIP_NONE dropped).  L2, Dot and Cosine (f32 and f16 columns;
 * an int8 column under cosine is refused).
 * Cosine (IvfTransformer::new_flat, ivf.rs:147-175): the caller passes the rows ALREADY NORMALISED (lance_hip_normalize) with
 * part ids assigned in L2 -- the reference stores the normalised rows too; lance_hip_ivfflat_search normalises the query
 * key (knn.rs:498), finds the partitions in L2 (ivf/v2.rs:455-465) and scores rows with cosine_distance (ivf/v2.rs:405-411).
 * The handle is destroyed with lance_hip_index_destroy.                                               */
int lance_hip_ivfflat_create(lance_hip_ctx *ctx, int dtype, int metric, uint32_t d, const void *centroids,
                             uint32_t nlist, const void *x, const uint32_t *part_ids, const uint64_t *row_ids,
                             uint64_t n, lance_hip_index **out);
/* find_partitions + exact scan of the nprobes partitions + SortExec(dist, rowid).fetch(k); k <= 128.
 * When one partition holds more than k rows at or below a tied k-th distance, the survivors are those FlatIndex's
 * BinaryHeap keeps (such queries are replayed through a heap with std's push/pop, as in the IVF_PQ path). */
int lance_hip_ivfflat_search(lance_hip_ctx *ctx, const lance_hip_index *idx, const void *q, uint32_t nq, uint32_t k,
                             uint32_t nprobes, uint64_t *ids, float *dists);
/* IVF_FLAT under a row-id prefilter (FlatIndex::search's RowIdMask branch, flat/index.rs:129-165): allow_by_rowid[r] != 0
 * <=> row id r may be returned.  The mask is tested inside the scan kernels (bound pass, main pass, exact replay) -- the
 * selected rows are scored by the same distance function, so no filtered copy of the index is needed.      */
int lance_hip_ivfflat_search_filtered(lance_hip_ctx *ctx, const lance_hip_index *idx, const void *q, uint32_t nq, uint32_t k,
                                      uint32_t nprobes, const uint8_t *allow_by_rowid, uint64_t n_allow, uint64_t *ids, float *dists);

/* ---- a22 / 8(f) N3: index files (lance/src/index/vector/builder.rs:938-1079 merge_partitions) ---------------------- */
/* The `index.idx` + `auxiliary.idx` pair of an IVF_PQ / IVF_FLAT index directory, Lance file format 2.0 (the
 * FileWriter default, lance-file/src/writer.rs:553-561).  Host-side: nothing here needs a GPU except load/save, and every
 * pointer of the view below is a HOST pointer (the exception to this header's device-pointer convention).
 * Readers replaced: IvfQuantizationStorage::try_new (lance-index/src/vector/storage.rs:182-243),
 * ProductQuantizationMetadata (pq/storage.rs:52-144), IvfModel <-> pb (ivf/storage.rs:181-244).                    */
enum { LANCE_HIP_IVF_PQ = 0, LANCE_HIP_IVF_FLAT = 1 };
typedef struct lance_hip_index_file lance_hip_index_file;
typedef struct lance_hip_index_file_view {
  int index_type;               /* LANCE_HIP_IVF_PQ / LANCE_HIP_IVF_FLAT */
  int metric;
  int dtype;                    /* element type of the stored tensors / flat vectors: LANCE_HIP_F32 or LANCE_HIP_F16 */
  uint32_t d, nlist, m, nbits;  /* m = nbits = 0 for IVF_FLAT */
  uint64_t n_rows;
  int transposed;               /* codes are [code bytes][n_p] inside each partition (pq/storage.rs:430-449) */
  int has_loss;
  double loss;
  const float *centroids;       /* [nlist][d], widened to f32 */
  const float *codebook;        /* [m][2^nbits][d/m] f32 (pq/builder.rs:139-154 layout); NULL for IVF_FLAT */
  const uint32_t *part_offsets; /* [nlist+1] row offsets */
  const uint64_t *row_ids;      /* [n_rows], partition order */
  const uint8_t *codes;         /* IVF_PQ: n_rows * (nbits == 4 ? m/2 : m) bytes */
  const void *vectors;          /* IVF_FLAT: [n_rows][d] of dtype */
} lance_hip_index_file_view;
/* Maps and validates both files; the view's pointers stay valid until close.  Files written by Lance <= 0.27 (PQ
 * codebook inline in the schema metadata) are read too, and so is the legacy single-file layout of Lance <= 0.21
 * (`index.idx` only: pb Index behind a 16-byte footer; lance/src/index/vector/ivf.rs IVFIndex::try_new, pq.rs
 * PQIndex::load) -- its codes come back row-major (transposed = 0).  Other index types are refused.             */
int lance_hip_index_file_open(const char *index_dir, lance_hip_index_file **out);
int lance_hip_index_file_get(const lance_hip_index_file *f, lance_hip_index_file_view *view);
void lance_hip_index_file_close(lance_hip_index_file *f);
/* Writes the pair (creating index_dir if needed) in the layout merge_partitions produces: global buffers
 * (pb IVF, pb Tensor codebook) ahead of the pages, 64-byte aligned buffers, transposed codes.                      */
int lance_hip_index_file_write(const char *index_dir, const lance_hip_index_file_view *view);
/* Files -> device-resident index.  dtype is the indexed column's element type (F32 / F16 / I8).                   */
int lance_hip_index_load(lance_hip_ctx *ctx, const char *index_dir, int dtype, lance_hip_index **out);
/* The same for one list shard of a multi-GPU search: only the IVF lists p with p % list_mod == list_rem are
 * copied to this GPU, the others are left empty; centroids and codebook are complete, row ids are the stored ones.   */
int lance_hip_index_load_lists(lance_hip_ctx *ctx, const char *index_dir, int dtype, uint32_t list_mod, uint32_t list_rem,
                               lance_hip_index **out);
/* Device-resident index -> files.  loss is the k-means loss recorded in index.idx (has_loss = 0 to omit).        */
int lance_hip_index_save(lance_hip_ctx *ctx, const lance_hip_index *idx, const char *index_dir, int has_loss, double loss);
/* One top-level fixed-width column (scalar or fixed-size list, uncompressed, no nulls) of a format-2.0 Lance file,
 * e.g. the vector column of a data file.  dst NULL: only reports rows / bytes per row.                           */
int lance_hip_file_read_column(const char *path, const char *column, void *dst, uint64_t dst_bytes, uint64_t *rows,
                               uint32_t *row_bytes);

/* Shuffle buffers -- the artefact the reference's `precomputed_shuffle_buffers` hand-off carries: rows of (row_id u64,
 * __ivf_part_id u32, __pq_code FSL<u8>[code_bytes]), un-transposed (python/lance/vector.py:659-665; read back by
 * IvfIndexBuilder::shuffle_dataset, lance/src/index/vector/builder.rs:509-546).  One Lance v2.0 FILE, written with the same
 * page encodings the reference's FileWriter uses; rows with part id LANCE_HIP_NONE are dropped; row_ids NULL = row numbers.
 * HOST pointers.  (Wrapping the file in a dataset directory -- manifest, version files -- stays with the caller's pylance:
 * the storage engine is outside this library's scope.)                                                           */
int lance_hip_shuffle_buffer_write(const char *path, const uint64_t *row_ids, const uint32_t *part_ids, const uint8_t *codes,
                                   uint64_t n, uint32_t code_bytes, uint64_t *rows_written);

/* ---- multi-GPU: merge of per-shard candidate lists (list-sharded search) -------------------------------- */
/* SortExec([_distance asc, _rowid asc]).with_fetch(k) (lance scanner.rs:3440-3468) over `c` candidates per query gathered
 * from the list shards: ids [nq][c] (int64, -1 = none), dists [nq][c].  With exact_dists != NULL (refine, scanner.rs:2884-
 * 2904) the keff best by (dists, id) are re-ranked by (exact_dists, id) before the fetch.  Outputs [nq][k] (-1 / +inf pad). */
int lance_hip_merge_topk(lance_hip_ctx *ctx, const int64_t *ids, const float *dists, const float *exact_dists, uint32_t nq,
                         uint32_t c, uint32_t keff, uint32_t k, int64_t *out_ids, float *out_dists);

/* ---- measurement hooks (bench.py): per-kernel HIP-event timing on the ctx stream --- */
/* When enabled, each internal launch of the named hot kernels is bracketed by events;
 * query returns accumulated milliseconds and launch count, then resets.              */
int lance_hip_timing_enable(lance_hip_ctx *ctx, int on);
int lance_hip_timing_query(lance_hip_ctx *ctx, const char *kernel, double *ms_total, uint64_t *launches);
/* Ceilings measured in-process for bench.py's roofline denominators (no reference counterpart; SURVEY 8(d) asks for
 * peaks "re-measured on the box").  what: 0 / 1 / 2 = random LDS gathers of 4 / 8 / 16-byte PQ-LUT entries, result in
 * lane-gathers per second; 3 = device copy, bytes (read + written) per second; 4 / 5 = v_add_f32 / v_pk_add_f32
 * wave-instructions per second; 6-9 = LDS table-layout variants (lane-gathers per second); 10 / 11 = dense
 * v_mfma_f32_32x32x16_f16 / _bf16 issue rate in flop per second (round 6: the MFMA roofline's measured peak).          */
int lance_hip_ubench(lance_hip_ctx *ctx, int what, double *result);

#ifdef __cplusplus
}
#endif
#endif /* LANCE_HIP_H */
