// TEST INFRASTRUCTURE (CPU): compiles the product's lance_amd/csrc/{index_file,lance_file}.cpp against tests/c/hip_shim and
// supplies host stand-ins for the three engine entry points that glue code calls (lance_hip_index_from_storage,
// lance_hip_index_export, lance_hip_ivfflat_create) plus the context's staging buffer.  It then drives
// lance_hip_index_load / _load_lists / _save exactly as a caller would and checks what reached the "device" and what came
// back in the files -- under AddressSanitizer / UBSan when the test builds it that way.  What it proves: the new native
// glue (pinned-staging uploads in chunks, list-shard packing, f16 narrowing, save) moves the right bytes and touches no
// memory it should not.  What it cannot prove: anything about kernels.
//
// usage: index_io_host_harness <pq_index_dir> <legacy_index_dir> <scratch_dir>
#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "../../lance_amd/csrc/common.h"
#include "../../lance_amd/csrc/f16.h"
#include "../../lance_amd/csrc/index.h"

// ---- engine pieces the glue needs ----------------------------------------------------------------------------------
namespace lh {
static thread_local char g_err[1024] = "";
void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); }
}  // namespace lh
extern "C" const char *lance_hip_last_error(void) { return lh::g_err; }

void *lance_hip_ctx::scratch(const char *, size_t) { return nullptr; }
void lance_hip_ctx::drop_graphs() {}
void lance_hip_ctx::time_begin(const char *) {}
void lance_hip_ctx::time_end(const char *) {}
static size_t g_max_stage = 0, g_stage_calls = 0;
void *lance_hip_ctx::host_staging(size_t bytes) {
  ++g_stage_calls;
  if (bytes > g_max_stage) g_max_stage = bytes;
  if (bytes <= pinned_bytes) return pinned;
  free(pinned);
  pinned_bytes = bytes;          // EXACT size: a chunk copy one byte too long trips the sanitizer
  pinned = malloc(bytes);
  return pinned;
}

lance_hip_index::~lance_hip_index() {
  free(centroids); free(codebook); free(part_offsets); free(codes); free(row_ids); free(vectors); free(flat_items);
}

static float *widen(int model_f16, const void *src, size_t count) {
  float *out = static_cast<float *>(malloc(count * 4 + 4));
  if (model_f16) for (size_t i = 0; i < count; ++i) { uint16_t h; memcpy(&h, static_cast<const uint8_t *>(src) + 2 * i, 2); out[i] = lh::h2f_host(h); }
  else memcpy(out, src, count * 4);
  return out;
}

// stores row-major codes per partition (what the real index keeps), from either layout
extern "C" int lance_hip_index_from_storage(lance_hip_ctx *, int dtype, int metric, uint32_t d, const void *centroids, uint32_t nlist,
                                            const void *codebook, uint32_t m, uint32_t nbits, const uint32_t *part_offsets_host,
                                            const uint8_t *codes, int transposed, const uint64_t *row_ids, uint64_t n, lance_hip_index **out) {
  auto *ix = new lance_hip_index();
  ix->metric = metric; ix->dtype = dtype; ix->d = d; ix->nlist = nlist; ix->m = m; ix->nbits = nbits; ix->n = n;
  const int f16 = dtype == LANCE_HIP_F16;
  ix->centroids = widen(f16, centroids, (size_t)nlist * d);
  ix->codebook = widen(f16, codebook, ((size_t)1 << nbits) * d);
  ix->part_offsets_h.assign(part_offsets_host, part_offsets_host + nlist + 1);
  const uint32_t cb = ix->code_bytes();
  ix->codes = static_cast<uint8_t *>(malloc((size_t)n * cb + 1));
  ix->row_ids = static_cast<uint64_t *>(malloc((size_t)n * 8 + 8));
  memcpy(ix->row_ids, row_ids, (size_t)n * 8);
  for (uint32_t p = 0; p < nlist; ++p) {
    const size_t a = part_offsets_host[p], np_ = part_offsets_host[p + 1] - a;
    for (size_t r = 0; r < np_; ++r)
      for (uint32_t c = 0; c < cb; ++c)
        ix->codes[(a + r) * cb + c] = transposed ? codes[a * cb + (size_t)c * np_ + r] : codes[(a + r) * cb + c];
  }
  *out = ix;
  return LANCE_HIP_OK;
}

extern "C" int lance_hip_index_export(lance_hip_ctx *, const lance_hip_index *ix, uint32_t *offs, uint8_t *codes_t, uint64_t *rid) {
  const uint32_t cb = ix->code_bytes();
  if (offs) memcpy(offs, ix->part_offsets_h.data(), (ix->nlist + 1) * 4);
  if (rid) memcpy(rid, ix->row_ids, (size_t)ix->n * 8);
  if (codes_t)
    for (uint32_t p = 0; p < ix->nlist; ++p) {
      const size_t a = ix->part_offsets_h[p], np_ = ix->part_offsets_h[p + 1] - a;
      for (size_t r = 0; r < np_; ++r)
        for (uint32_t c = 0; c < cb; ++c) codes_t[a * cb + (size_t)c * np_ + r] = ix->codes[(a + r) * cb + c];
    }
  return LANCE_HIP_OK;
}

extern "C" int lance_hip_ivfflat_create(lance_hip_ctx *, int dtype, int metric, uint32_t d, const void *centroids, uint32_t nlist,
                                        const void *x, const uint32_t *part_ids, const uint64_t *row_ids, uint64_t n, lance_hip_index **out) {
  auto *ix = new lance_hip_index();
  ix->metric = metric; ix->dtype = dtype; ix->d = d; ix->nlist = nlist; ix->m = 0;
  const int f16 = dtype == LANCE_HIP_F16;
  ix->centroids = widen(f16, centroids, (size_t)nlist * d);
  ix->part_offsets_h.assign(nlist + 1, 0);
  for (uint64_t r = 0; r < n; ++r) if (part_ids[r] < nlist) ix->part_offsets_h[part_ids[r] + 1]++;
  for (uint32_t p = 0; p < nlist; ++p) ix->part_offsets_h[p + 1] += ix->part_offsets_h[p];
  ix->n = ix->part_offsets_h[nlist];
  ix->vectors = static_cast<float *>(malloc((size_t)ix->n * d * 4 + 4));
  ix->row_ids = static_cast<uint64_t *>(malloc((size_t)ix->n * 8 + 8));
  std::vector<uint32_t> cur(ix->part_offsets_h.begin(), ix->part_offsets_h.end() - 1);
  float *xf = widen(f16, x, (size_t)n * d);
  for (uint64_t r = 0; r < n; ++r) {
    if (part_ids[r] >= nlist) continue;
    const uint32_t s = cur[part_ids[r]]++;
    memcpy(ix->vectors + (size_t)s * d, xf + (size_t)r * d, (size_t)d * 4);
    ix->row_ids[s] = row_ids ? row_ids[r] : r;
  }
  free(xf);
  *out = ix;
  return LANCE_HIP_OK;
}

// ---- the checks ----------------------------------------------------------------------------------------------------
#define CHECK(cond, ...)                                                              \
  do {                                                                                \
    if (!(cond)) { fprintf(stderr, "FAILED %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, " [%s]\n", lance_hip_last_error()); return 1; } \
  } while (0)

static int same_as_file(lance_hip_ctx *ctx, const lance_hip_index *ix, const lance_hip_index_file_view &v, uint32_t mod, uint32_t rem) {
  const uint32_t cb = v.nbits == 4 ? v.m / 2 : v.m;
  CHECK(ix->nlist == v.nlist && ix->d == v.d && ix->m == v.m && ix->nbits == v.nbits && ix->metric == v.metric, "index shape");
  CHECK(memcmp(ix->centroids, v.centroids, (size_t)v.nlist * v.d * 4) == 0, "centroids");
  CHECK(memcmp(ix->codebook, v.codebook, ((size_t)1 << v.nbits) * v.d * 4) == 0, "codebook");
  uint64_t expect = 0;
  for (uint32_t p = 0; p < v.nlist; ++p) {
    const size_t a = v.part_offsets[p], np_ = v.part_offsets[p + 1] - a;
    const size_t la = ix->part_offsets_h[p], ln = ix->part_offsets_h[p + 1] - la;
    if (p % mod != rem) { CHECK(ln == 0, "foreign list %u not empty", p); continue; }
    CHECK(ln == np_, "list %u length %zu != %zu", p, ln, np_);
    expect += np_;
    CHECK(memcmp(ix->row_ids + la, v.row_ids + a, np_ * 8) == 0, "row ids of list %u", p);
    for (size_t r = 0; r < np_; ++r)
      for (uint32_t c = 0; c < cb; ++c) {
        const uint8_t want = v.transposed ? v.codes[a * cb + (size_t)c * np_ + r] : v.codes[(a + r) * cb + c];
        CHECK(ix->codes[(la + r) * cb + c] == want, "code byte list %u row %zu col %u", p, r, c);
      }
  }
  CHECK(ix->n == expect, "row count %llu != %llu", (unsigned long long)ix->n, (unsigned long long)expect);
  (void)ctx;
  return 0;
}

static int roundtrip_dir(lance_hip_ctx *ctx, const char *dir, const std::string &scratch, const char *tag) {
  lance_hip_index_file *f = nullptr;
  lance_hip_index_file_view v;
  CHECK(lance_hip_index_file_open(dir, &f) == LANCE_HIP_OK && lance_hip_index_file_get(f, &v) == LANCE_HIP_OK, "open %s", dir);
  const int dtype = v.dtype;
  // whole index and every list shard of 2- and 3-way placements
  for (uint32_t mod = 1; mod <= 3; ++mod)
    for (uint32_t rem = 0; rem < mod; ++rem) {
      lance_hip_index *ix = nullptr;
      CHECK(lance_hip_index_load_lists(ctx, dir, dtype, mod, rem, &ix) == LANCE_HIP_OK, "load_lists %u/%u", rem, mod);
      if (same_as_file(ctx, ix, v, mod, rem)) return 1;
      if (mod == 1) {   // save, reopen, compare with the source files
        const std::string out = scratch + "/" + tag;
        CHECK(lance_hip_index_save(ctx, ix, out.c_str(), v.has_loss, v.loss) == LANCE_HIP_OK, "save");
        lance_hip_index_file *g = nullptr;
        lance_hip_index_file_view w;
        CHECK(lance_hip_index_file_open(out.c_str(), &g) == LANCE_HIP_OK && lance_hip_index_file_get(g, &w) == LANCE_HIP_OK, "reopen");
        CHECK(w.n_rows == v.n_rows && w.transposed == 1 && w.dtype == v.dtype && w.has_loss == v.has_loss && w.loss == v.loss, "saved header");
        lance_hip_index *ix2 = nullptr;
        CHECK(lance_hip_index_load(ctx, out.c_str(), dtype, &ix2) == LANCE_HIP_OK, "reload");
        if (same_as_file(ctx, ix2, v, 1, 0)) return 1;
        lance_hip_index_file_close(g);
        delete ix2;
      }
      delete ix;
    }
  lance_hip_index *bad = nullptr;
  CHECK(lance_hip_index_load_lists(ctx, dir, dtype, 2, 2, &bad) == LANCE_HIP_EINVAL && bad == nullptr, "shard 2 of 2 accepted");
  CHECK(lance_hip_index_load_lists(ctx, dir, dtype, 0, 0, &bad) == LANCE_HIP_EINVAL, "0-way placement accepted");
  lance_hip_index_file_close(f);
  return 0;
}

int main(int argc, char **argv) {
  if (argc < 4) return 2;
  lance_hip_ctx ctx;
  const std::string scratch = argv[3];
  if (roundtrip_dir(&ctx, argv[1], scratch, "pq")) return 1;
  if (roundtrip_dir(&ctx, argv[2], scratch, "legacy")) return 1;

  // a synthetic f16 IVF_PQ index with 4-bit codes, big enough for several staging chunks, and an IVF_FLAT one
  {
    const uint32_t d = 32, nlist = 5, m = 8, nbits = 4, cb = m / 2;
    const uint64_t n = 3000;
    std::vector<float> cent((size_t)nlist * d), book((size_t)16 * d);
    for (size_t i = 0; i < cent.size(); ++i) cent[i] = lh::round_f16_host((float)(i % 97) * 0.25f - 3.0f);
    for (size_t i = 0; i < book.size(); ++i) book[i] = lh::round_f16_host((float)(i % 31) * 0.125f);
    std::vector<uint32_t> offs = {0, 700, 700, 1500, 2999, 3000};
    std::vector<uint64_t> rid(n);
    std::vector<uint8_t> codes((size_t)n * cb);
    for (uint64_t i = 0; i < n; ++i) rid[i] = (i * 2654435761ull) % 100000;
    for (size_t i = 0; i < codes.size(); ++i) codes[i] = (uint8_t)(i * 37 + 11);
    lance_hip_index_file_view v{};
    v.index_type = LANCE_HIP_IVF_PQ; v.metric = LANCE_HIP_L2; v.dtype = LANCE_HIP_F16; v.d = d; v.nlist = nlist; v.m = m; v.nbits = nbits;
    v.n_rows = n; v.transposed = 1; v.has_loss = 1; v.loss = 42.5; v.centroids = cent.data(); v.codebook = book.data();
    v.part_offsets = offs.data(); v.row_ids = rid.data(); v.codes = codes.data();
    const std::string dir = scratch + "/f16";
    CHECK(lance_hip_index_file_write(dir.c_str(), &v) == LANCE_HIP_OK, "write f16");
    if (roundtrip_dir(&ctx, dir.c_str(), scratch, "f16_again")) return 1;
    lance_hip_index *ix = nullptr;
    CHECK(lance_hip_index_load(&ctx, dir.c_str(), LANCE_HIP_F32, &ix) != LANCE_HIP_OK, "f16 files accepted for an f32 column");
  }
  {
    const uint32_t d = 24, nlist = 4;
    const uint64_t n = 500;
    std::vector<float> cent((size_t)nlist * d, 1.5f), vec((size_t)n * d);
    for (size_t i = 0; i < vec.size(); ++i) vec[i] = (float)(i % 113) * 0.5f;
    std::vector<uint32_t> offs = {0, 100, 100, 420, 500};
    std::vector<uint64_t> rid(n);
    for (uint64_t i = 0; i < n; ++i) rid[i] = n - 1 - i;
    lance_hip_index_file_view v{};
    v.index_type = LANCE_HIP_IVF_FLAT; v.metric = LANCE_HIP_DOT; v.dtype = LANCE_HIP_F32; v.d = d; v.nlist = nlist; v.n_rows = n;
    v.centroids = cent.data(); v.part_offsets = offs.data(); v.row_ids = rid.data(); v.vectors = vec.data();
    const std::string dir = scratch + "/flat";
    CHECK(lance_hip_index_file_write(dir.c_str(), &v) == LANCE_HIP_OK, "write flat");
    lance_hip_index *ix = nullptr;
    CHECK(lance_hip_index_load(&ctx, dir.c_str(), LANCE_HIP_F32, &ix) == LANCE_HIP_OK, "load flat");
    CHECK(ix->n == n && ix->m == 0 && memcmp(ix->vectors, vec.data(), vec.size() * 4) == 0 && memcmp(ix->row_ids, rid.data(), n * 8) == 0, "flat content");
    CHECK(ix->part_offsets_h == offs, "flat offsets");
    const std::string out = scratch + "/flat_saved";
    CHECK(lance_hip_index_save(&ctx, ix, out.c_str(), 0, 0.0) == LANCE_HIP_OK, "save flat");
    lance_hip_index_file *g = nullptr;
    lance_hip_index_file_view w;
    CHECK(lance_hip_index_file_open(out.c_str(), &g) == LANCE_HIP_OK && lance_hip_index_file_get(g, &w) == LANCE_HIP_OK, "reopen flat");
    CHECK(w.index_type == LANCE_HIP_IVF_FLAT && w.n_rows == n && !w.has_loss && memcmp(w.vectors, vec.data(), vec.size() * 4) == 0, "saved flat");
    lance_hip_index_file_close(g);
    lance_hip_index *shard = nullptr;
    CHECK(lance_hip_index_load_lists(&ctx, dir.c_str(), LANCE_HIP_F32, 2, 0, &shard) != LANCE_HIP_OK, "flat list shard accepted");
    delete ix;
  }
  free(ctx.pinned);
  printf("staging calls %zu, largest chunk %zu bytes\nok\n", g_stage_calls, g_max_stage);
  return 0;
}
