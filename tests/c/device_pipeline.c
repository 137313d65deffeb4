/* C99 consumer of include/lance_hip.h driving the DEVICE entry points in the order the Rust adapter does
 * (integration/rust/lance-linalg/src/hip.rs, integration/rust/lance-index/src/vector/hip_*.rs): every buffer crosses the ABI
 * through lance_hip_malloc / lance_hip_memcpy_h2d / _d2h, nothing here knows about torch or HIP.
 *   IVF training (KMeans::new_with_params)       -> lance_hip_kmeans_train_ex
 *   assign + residual of the PQ training sample  -> lance_hip_assign, lance_hip_residual
 *   PQ codebook (PQBuildParams::build_from_fsl)  -> lance_hip_pq_train
 *   transform of every row (IvfTransformer)      -> lance_hip_ivfpq_encode
 *   shuffle + per-partition transpose ON THE HOST, as the reference's shuffler / ProductQuantizationStorage::new do
 *   index_from_storage(transposed = 1) + set_raw -> the handle a VectorIndex would hold
 *   find_partitions, search (plain, refine, prefilter)
 * usage: device_pipeline <in.bin> <out.bin>
 * in.bin : u32 n, d, nlist, m, nq, k, nprobes, refine, ivf_iters, pq_iters, seed, metric; f32 x[n][d]; f32 q[nq][d]; u8 allow[n]
 * out.bin: f32 centroids[nlist][d]; f32 codebook[m][256][d/m]; u32 part[n]; u8 codes[n][m]; u32 probes[nq][nprobes];
 *          u64 ids[nq][k], f32 dists[nq][k]  x3 (plain, refine, prefilter)                                              */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lance_hip.h"

#define CHECK(call)                                                                   \
  do {                                                                                \
    int rc_ = (call);                                                                 \
    if (rc_ != LANCE_HIP_OK) {                                                        \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, lance_hip_last_error());          \
      return 1;                                                                       \
    }                                                                                 \
  } while (0)

static void *dalloc(lance_hip_ctx *ctx, size_t bytes) {
  void *p = NULL;
  if (lance_hip_malloc(ctx, bytes ? bytes : 1, &p) != LANCE_HIP_OK) { fprintf(stderr, "malloc: %s\n", lance_hip_last_error()); exit(1); }
  return p;
}

int main(int argc, char **argv) {
  uint32_t h[12];
  FILE *fi, *fo;
  if (argc < 3) return 2;
  fi = fopen(argv[1], "rb");
  if (!fi || fread(h, 4, 12, fi) != 12) return 2;
  const uint32_t n = h[0], d = h[1], nlist = h[2], m = h[3], nq = h[4], k = h[5], nprobes = h[6], refine = h[7], ivf_iters = h[8],
                 pq_iters = h[9], seed = h[10];
  const int metric = (int)h[11];
  const uint32_t sd = d / m;
  float *x = malloc((size_t)n * d * 4), *q = malloc((size_t)nq * d * 4);
  uint8_t *allow = malloc(n);
  if (fread(x, 4, (size_t)n * d, fi) != (size_t)n * d || fread(q, 4, (size_t)nq * d, fi) != (size_t)nq * d || fread(allow, 1, n, fi) != n) return 2;
  fclose(fi);

  lance_hip_ctx *ctx = NULL;
  CHECK(lance_hip_ctx_create(0, NULL, &ctx));
  float *dx = dalloc(ctx, (size_t)n * d * 4), *dq = dalloc(ctx, (size_t)nq * d * 4);
  CHECK(lance_hip_memcpy_h2d(ctx, dx, x, (size_t)n * d * 4));
  CHECK(lance_hip_memcpy_h2d(ctx, dq, q, (size_t)nq * d * 4));

  /* IVF: the first min(n, nlist * 256) rows are the training sample here (the caller applies the sampling, kmeans.rs:1328-1340) */
  const uint32_t n_ivf = n < nlist * 256u ? n : nlist * 256u;
  float *dcent = dalloc(ctx, (size_t)nlist * d * 4);
  double loss = 0;
  uint32_t iters = 0, kout = 0;
  CHECK(lance_hip_kmeans_train_ex(ctx, LANCE_HIP_F32, metric == LANCE_HIP_COSINE ? LANCE_HIP_L2 : metric, dx, n_ivf, d, nlist, ivf_iters, 1e-4, 1.0f,
                                  16, NULL, seed, dcent, &loss, &iters, &kout));
  if (kout != nlist) { fprintf(stderr, "trained %u of %u centroids\n", kout, nlist); return 1; }

  /* PQ training sample: residuals of the first min(n, 65536) rows */
  const uint32_t n_pq = n < 65536u ? n : 65536u;
  uint32_t *dpart_s = dalloc(ctx, (size_t)n_pq * 4);
  float *dres = dalloc(ctx, (size_t)n_pq * d * 4), *dcb = dalloc(ctx, (size_t)256 * d * 4);
  CHECK(lance_hip_assign(ctx, LANCE_HIP_F32, LANCE_HIP_L2, dx, n_pq, d, dcent, nlist, NULL, dpart_s, NULL));
  CHECK(lance_hip_residual(ctx, LANCE_HIP_F32, dx, n_pq, d, dcent, dpart_s, dres));
  uint32_t *pq_it = malloc((size_t)m * 4);
  CHECK(lance_hip_pq_train(ctx, LANCE_HIP_F32, dres, n_pq, d, m, 8, pq_iters, 256, (uint64_t)seed + 1, dcb, pq_it));

  /* transform of every row */
  uint32_t *dpart = dalloc(ctx, (size_t)n * 4);
  uint8_t *dcodes = dalloc(ctx, (size_t)n * m);
  double tloss = 0;
  CHECK(lance_hip_ivfpq_encode(ctx, LANCE_HIP_F32, metric, dx, n, d, dcent, nlist, dcb, m, 8, dpart, dcodes, &tloss));
  uint32_t *part = malloc((size_t)n * 4);
  uint8_t *codes = malloc((size_t)n * m);
  CHECK(lance_hip_memcpy_d2h(ctx, part, dpart, (size_t)n * 4));
  CHECK(lance_hip_memcpy_d2h(ctx, codes, dcodes, (size_t)n * m));

  /* the shuffler: stable sort by partition; storage: codes transposed inside each partition (pq/storage.rs:430-449) */
  uint32_t *offs = calloc((size_t)nlist + 1, 4), *fill = calloc(nlist, 4);
  uint64_t kept = 0;
  for (uint32_t i = 0; i < n; ++i) if (part[i] != LANCE_HIP_NONE) { offs[part[i] + 1]++; kept++; }
  for (uint32_t p = 0; p < nlist; ++p) offs[p + 1] += offs[p];
  uint64_t *rid = malloc(kept * 8 + 8);
  uint8_t *codes_t = malloc(kept * m + 1);
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t p = part[i];
    if (p == LANCE_HIP_NONE) continue;
    const uint32_t np = offs[p + 1] - offs[p], j = fill[p]++;
    rid[offs[p] + j] = i;
    for (uint32_t mm = 0; mm < m; ++mm) codes_t[(size_t)offs[p] * m + (size_t)mm * np + j] = codes[(size_t)i * m + mm];
  }
  uint8_t *dcodes_t = dalloc(ctx, kept * m);
  uint64_t *drid = dalloc(ctx, kept * 8);
  CHECK(lance_hip_memcpy_h2d(ctx, dcodes_t, codes_t, kept * m));
  CHECK(lance_hip_memcpy_h2d(ctx, drid, rid, kept * 8));
  lance_hip_index *ix = NULL;
  CHECK(lance_hip_index_from_storage(ctx, LANCE_HIP_F32, metric, d, dcent, nlist, dcb, m, 8, offs, dcodes_t, 1, drid, kept, &ix));
  CHECK(lance_hip_index_set_raw(ix, dx, n));

  /* queries */
  uint32_t *dprobes = dalloc(ctx, (size_t)nq * nprobes * 4);
  float *dpd = dalloc(ctx, (size_t)nq * nprobes * 4);
  float *qn = dq;
  if (metric == LANCE_HIP_COSINE) {        /* find_partitions wants normalised queries under cosine (knn.rs:495-498) */
    qn = dalloc(ctx, (size_t)nq * d * 4);
    CHECK(lance_hip_normalize(ctx, LANCE_HIP_F32, dq, nq, d, qn));
  }
  CHECK(lance_hip_find_partitions(ctx, LANCE_HIP_F32, metric, qn, nq, d, dcent, nlist, nprobes, dprobes, dpd));
  uint64_t *dids = dalloc(ctx, (size_t)nq * k * 8);
  float *ddists = dalloc(ctx, (size_t)nq * k * 4);
  uint8_t *dallow = dalloc(ctx, n);
  CHECK(lance_hip_memcpy_h2d(ctx, dallow, allow, n));

  float *cent = malloc((size_t)nlist * d * 4), *cb = malloc((size_t)256 * d * 4);
  uint32_t *probes = malloc((size_t)nq * nprobes * 4);
  uint64_t *ids = malloc((size_t)nq * k * 8);
  float *dists = malloc((size_t)nq * k * 4);
  CHECK(lance_hip_memcpy_d2h(ctx, cent, dcent, (size_t)nlist * d * 4));
  CHECK(lance_hip_memcpy_d2h(ctx, cb, dcb, (size_t)256 * d * 4));
  CHECK(lance_hip_memcpy_d2h(ctx, probes, dprobes, (size_t)nq * nprobes * 4));
  fo = fopen(argv[2], "wb");
  if (!fo) return 2;
  fwrite(cent, 4, (size_t)nlist * d, fo);
  fwrite(cb, 4, (size_t)256 * d, fo);
  fwrite(part, 4, n, fo);
  fwrite(codes, 1, (size_t)n * m, fo);
  fwrite(probes, 4, (size_t)nq * nprobes, fo);
  for (int pass = 0; pass < 3; ++pass) {
    if (pass == 0) CHECK(lance_hip_ivfpq_search(ctx, ix, dq, nq, k, nprobes, 0, dids, ddists));
    if (pass == 1) CHECK(lance_hip_ivfpq_search(ctx, ix, dq, nq, k, nprobes, refine, dids, ddists));
    if (pass == 2) CHECK(lance_hip_ivfpq_search_filtered(ctx, ix, dq, nq, k, nprobes, 0, dallow, n, dids, ddists));
    CHECK(lance_hip_memcpy_d2h(ctx, ids, dids, (size_t)nq * k * 8));
    CHECK(lance_hip_memcpy_d2h(ctx, dists, ddists, (size_t)nq * k * 4));
    fwrite(ids, 8, (size_t)nq * k, fo);
    fwrite(dists, 4, (size_t)nq * k, fo);
  }
  fclose(fo);
  printf("ivf iters %u loss %.17g, pq iters[0] %u, kept %llu rows, sub-dimension %u\nok\n", iters, loss, pq_it[0], (unsigned long long)kept, sd);
  lance_hip_index_destroy(ix);
  void *bufs[] = {dx, dq, dcent, dpart_s, dres, dcb, dpart, dcodes, dcodes_t, drid, dprobes, dpd, dids, ddists, dallow};
  for (size_t i = 0; i < sizeof(bufs) / sizeof(bufs[0]); ++i) CHECK(lance_hip_free(ctx, bufs[i]));
  if (qn != dq) CHECK(lance_hip_free(ctx, qn));
  lance_hip_ctx_destroy(ctx);
  return 0;
}
