/* Plain-C consumer of include/lance_hip.h (what a cgo / Rust `extern "C"` binding sees): opens an index directory,
 * prints what it holds, writes it back out to another directory and re-opens that.  No GPU is touched.
 * usage: index_file_roundtrip <index_dir> <out_dir>                                                               */
#include <stdio.h>
#include <string.h>

#include "lance_hip.h"

static int fail(const char *what) {
  fprintf(stderr, "%s: %s\n", what, lance_hip_last_error());
  return 1;
}

int main(int argc, char **argv) {
  lance_hip_index_file *f = NULL, *g = NULL;
  lance_hip_index_file_view v, w;
  uint64_t rows = 0;
  uint32_t row_bytes = 0;
  if (argc < 3) return 2;
  if (lance_hip_index_file_open(argv[1], &f) != LANCE_HIP_OK) return fail("open");
  if (lance_hip_index_file_get(f, &v) != LANCE_HIP_OK) return fail("get");
  printf("type=%d metric=%d dtype=%d d=%u nlist=%u m=%u nbits=%u rows=%llu transposed=%d loss=%.17g\n", v.index_type, v.metric,
         v.dtype, v.d, v.nlist, v.m, v.nbits, (unsigned long long)v.n_rows, v.transposed, v.has_loss ? v.loss : -1.0);
  if (lance_hip_index_file_write(argv[2], &v) != LANCE_HIP_OK) return fail("write");
  if (lance_hip_index_file_open(argv[2], &g) != LANCE_HIP_OK) return fail("reopen");
  if (lance_hip_index_file_get(g, &w) != LANCE_HIP_OK) return fail("get");
  if (w.n_rows != v.n_rows || w.d != v.d || w.m != v.m || memcmp(w.codes, v.codes, (size_t)v.n_rows * v.m) != 0 ||
      memcmp(w.row_ids, v.row_ids, (size_t)v.n_rows * 8) != 0 || memcmp(w.centroids, v.centroids, (size_t)v.nlist * v.d * 4) != 0 ||
      memcmp(w.codebook, v.codebook, (size_t)256 * v.d * 4) != 0 || w.loss != v.loss) {
    fprintf(stderr, "round trip differs\n");
    return 1;
  }
  lance_hip_index_file_close(g);
  lance_hip_index_file_close(f);
  if (argc > 3) {
    if (lance_hip_file_read_column(argv[3], "vec", NULL, 0, &rows, &row_bytes) != LANCE_HIP_OK) return fail("read_column");
    printf("column vec: rows=%llu row_bytes=%u\n", (unsigned long long)rows, row_bytes);
  }
  if (lance_hip_index_file_open("/nonexistent/dir", &f) != LANCE_HIP_EIO) return 1;
  printf("error channel: %s\nok\n", lance_hip_last_error());
  return 0;
}
