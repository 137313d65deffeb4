// index_file.cpp -- the on-disk form of an IVF_PQ / IVF_FLAT index: reads the
// `index.idx` + `auxiliary.idx` pair the reference's IvfIndexBuilder::merge_partitions writes
// (rust/lance/src/index/vector/builder.rs:938-1079) straight into the device-resident index, and writes the same pair
// from one.  What is in the files (reference readers: lance-index/src/vector/storage.rs:182-243,
// pq/storage.rs:52-144, ivf/storage.rs:181-244):
//   auxiliary.idx  columns `_rowid` u64 and `__pq_code` FSL<u8>[code bytes] (or `flat` FSL<f32|f16>[d]), rows sorted by
//                  partition, each partition's codes transposed to [code bytes][n_p]; schema metadata `distance_type`,
//                  `lance:ivf` -> global buffer holding pb IVF{offsets, lengths}, `storage_metadata` -> JSON list with one
//                  JSON document {nbits, num_sub_vectors, dimension, transposed, codebook_position} and the codebook as a
//                  pb Tensor global buffer (files up to v0.27 carry it inline as `codebook_tensor`, still readable)
//   index.idx      the FLAT sub-index (no rows); metadata `lance:index` {"type","distance_type"}, `lance:ivf` -> pb IVF
//                  with centroids_tensor + loss, `lance:flat` per-partition metadata (empty strings)
#include <sys/stat.h>

#include <algorithm>
#include <cerrno>
#include <memory>

#include "common.h"
#include "f16.h"
#include "index.h"
#include "lance_file.h"

using namespace lh;
using namespace lancefile;

namespace {

struct Tensor {
  int data_type = 0;   // index.proto:36-45: 1 = FLOAT16, 2 = FLOAT32
  std::vector<uint64_t> shape;
  const uint8_t *data = nullptr;
  size_t size = 0;
};

bool parse_tensor(const uint8_t *p, size_t n, Tensor *t) {
  PbReader r(p, n);
  bool ok = true;
  for (PbField f; r.next(&f, &ok);) {
    if (f.number == 1 && f.wire == 0) t->data_type = (int)f.value;
    else if (f.number == 2) { if (!PbReader::append_varints(f, &t->shape)) return false; }
    else if (f.number == 3 && f.wire == 2) { t->data = f.data; t->size = f.size; }
  }
  return ok;
}

struct IvfPb {
  std::vector<uint64_t> offsets, lengths;
  bool has_tensor = false;
  Tensor centroids;
  bool has_loss = false;
  double loss = 0.0;
  size_t n_legacy_centroids = 0;
};

bool parse_ivf(const uint8_t *p, size_t n, IvfPb *out) {
  PbReader r(p, n);
  bool ok = true;
  for (PbField f; r.next(&f, &ok);) {
    if (f.number == 2) { if (!PbReader::append_varints(f, &out->offsets)) return false; }
    else if (f.number == 3) { if (!PbReader::append_varints(f, &out->lengths)) return false; }
    else if (f.number == 4 && f.wire == 2) { out->has_tensor = true; if (!parse_tensor(f.data, f.size, &out->centroids)) return false; }
    else if (f.number == 5 && f.wire == 1) { out->has_loss = true; memcpy(&out->loss, &f.value, 8); }
    else if (f.number == 1) out->n_legacy_centroids += f.wire == 2 ? f.size / 4 : 1;
  }
  return ok;
}

// widens a FLOAT32 / FLOAT16 tensor to f32
bool tensor_to_f32(const Tensor &t, size_t count, std::vector<float> *out, int *dtype, std::string *err) {
  if (t.data_type == 2) {
    if (t.size != count * 4) { *err = "tensor holds " + std::to_string(t.size) + " bytes, expected " + std::to_string(count * 4); return false; }
    out->resize(count);
    if (count) memcpy(out->data(), t.data, count * 4);
    *dtype = LANCE_HIP_F32;
  } else if (t.data_type == 1) {
    if (t.size != count * 2) { *err = "tensor holds " + std::to_string(t.size) + " bytes, expected " + std::to_string(count * 2); return false; }
    out->resize(count);
    for (size_t i = 0; i < count; ++i) { uint16_t h; memcpy(&h, t.data + 2 * i, 2); (*out)[i] = h2f_host(h); }
    *dtype = LANCE_HIP_F16;
  } else {
    *err = "tensor data type " + std::to_string(t.data_type) + " is not supported (FLOAT32 and FLOAT16 only)";
    return false;
  }
  return true;
}

int metric_from_string(std::string s) {
  std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return (char)tolower(c); });
  if (s == "l2" || s == "euclidean") return LANCE_HIP_L2;
  if (s == "cosine") return LANCE_HIP_COSINE;
  if (s == "dot") return LANCE_HIP_DOT;
  return -1;
}
const char *metric_name(int m) { return m == LANCE_HIP_L2 ? "l2" : m == LANCE_HIP_COSINE ? "cosine" : "dot"; }

bool json_u64(const Json *j, uint64_t *out) {
  if (!j || j->kind != Json::Number || j->num < 0 || j->num != (double)(uint64_t)j->num) return false;
  *out = (uint64_t)j->num;
  return true;
}

}  // namespace

struct lance_hip_index_file {
  std::unique_ptr<FileReader> aux, idx;
  lance_hip_index_file_view v{};
  std::vector<float> centroids, codebook;
  std::vector<uint32_t> part_offsets;
  std::vector<uint8_t> rowid_buf, payload_buf;   // only filled when a column spans several pages
  std::vector<uint8_t> inline_codebook;          // v0.27-style codebook_tensor bytes
  std::vector<uint8_t> legacy;                   // whole legacy (v1) index.idx, when that is what the directory holds
  std::vector<uint64_t> legacy_row_ids;
};

#define IO_FAIL(code, ...)       \
  do {                           \
    set_error(__VA_ARGS__);      \
    return code;                 \
  } while (0)

// Legacy (v1) vector index, written by Lance up to 0.21 and still opened by the reference (lance/src/index/vector/ivf.rs
// IVFIndex::try_new + pq.rs PQIndex::load): ONE file `index.idx` = for every partition [PQ codes n_p x m, row-major]
// [row ids n_p x u64] at the byte offset IVF.offsets[p], then a length-prefixed pb `Index` message whose position is in
// the 16-byte footer [u64 position][u16 major = 0][u16 minor <= 2]["LANC"] (protos/index.proto:13-34,131-160).
static int open_legacy(const std::string &path, lance_hip_index_file *f) {
  FILE *fp = fopen(path.c_str(), "rb");
  if (!fp) IO_FAIL(LANCE_HIP_EIO, "index_file_open: cannot open %s: %s", path.c_str(), strerror(errno));
  fseek(fp, 0, SEEK_END);
  const long sz = ftell(fp);
  fseek(fp, 0, SEEK_SET);
  f->legacy.resize(sz > 0 ? (size_t)sz : 0);
  const bool rd = sz > 0 && fread(f->legacy.data(), 1, (size_t)sz, fp) == (size_t)sz;
  fclose(fp);
  if (!rd || sz < 20) IO_FAIL(LANCE_HIP_EIO, "index_file_open: cannot read %s", path.c_str());
  const uint8_t *b = f->legacy.data();
  const uint64_t size = (uint64_t)sz;
  uint64_t pos;
  memcpy(&pos, b + size - 16, 8);
  uint32_t len = 0;
  if (pos > size - 20) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: %s: corrupt legacy footer", path.c_str());
  memcpy(&len, b + pos, 4);
  if (len > size - 16 - 4 - pos) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: %s: corrupt legacy index message", path.c_str());
  lance_hip_index_file_view &v = f->v;
  const uint8_t *vi = nullptr; size_t vi_n = 0;
  {
    PbReader r(b + pos + 4, len);
    bool ok = true;
    for (PbField g; r.next(&g, &ok);)
      if (g.number == 5 && g.wire == 2) { vi = g.data; vi_n = g.size; }
    if (!ok || !vi) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: %s holds no vector index", path.c_str());
  }
  IvfPb ivf;
  bool have_ivf = false, have_pq = false;
  uint64_t dim = 0, metric = 0, nbits = 0, m = 0, pq_dim = 0;
  Tensor cb_tensor;
  bool cb_has_tensor = false;
  std::vector<float> cb_floats, cent_floats;
  {
    PbReader r(vi, vi_n);
    bool ok = true;
    for (PbField g; r.next(&g, &ok);) {
      if (g.number == 2 && g.wire == 0) dim = g.value;
      else if (g.number == 4 && g.wire == 0) metric = g.value;
      else if (g.number == 3 && g.wire == 2) {                       // VectorIndexStage
        PbReader rs(g.data, g.size);
        bool oks = true;
        for (PbField h; rs.next(&h, &oks);) {
          if (h.number == 2 && h.wire == 2) {                        // IVF
            have_ivf = true;
            if (!parse_ivf(h.data, h.size, &ivf)) oks = false;
            if (!ivf.has_tensor && ivf.n_legacy_centroids) {         // repeated float centroids = 1 (packed)
              PbReader ri(h.data, h.size);
              bool oki = true;
              for (PbField k; ri.next(&k, &oki);)
                if (k.number == 1 && k.wire == 2) { const size_t o = cent_floats.size(); cent_floats.resize(o + k.size / 4); memcpy(cent_floats.data() + o, k.data, k.size / 4 * 4); }
            }
          } else if (h.number == 3 && h.wire == 2) {                 // PQ
            have_pq = true;
            PbReader rp(h.data, h.size);
            bool okp = true;
            for (PbField k; rp.next(&k, &okp);) {
              if (k.number == 1 && k.wire == 0) nbits = k.value;
              else if (k.number == 2 && k.wire == 0) m = k.value;
              else if (k.number == 3 && k.wire == 0) pq_dim = k.value;
              else if (k.number == 4 && k.wire == 2) { const size_t o = cb_floats.size(); cb_floats.resize(o + k.size / 4); memcpy(cb_floats.data() + o, k.data, k.size / 4 * 4); }
              else if (k.number == 5 && k.wire == 2) { cb_has_tensor = true; if (!parse_tensor(k.data, k.size, &cb_tensor)) okp = false; }
            }
            if (!okp) oks = false;
          } else if (h.wire == 2) {
            IO_FAIL(LANCE_HIP_ENOTSUP, "index_file_open: %s: legacy index stage %u (OPQ transform / flat / DiskANN) is not supported", path.c_str(), h.number);
          }
        }
        if (!oks) ok = false;
      }
    }
    if (!ok) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: %s: malformed legacy vector index", path.c_str());
  }
  if (!have_ivf || !have_pq) IO_FAIL(LANCE_HIP_ENOTSUP, "index_file_open: %s: only legacy IVF_PQ indices are supported", path.c_str());
  if (metric != 0 && metric != 2) IO_FAIL(LANCE_HIP_ENOTSUP, "index_file_open: %s: legacy index with metric %llu (cosine / hamming) is not supported", path.c_str(), (unsigned long long)metric);
  if (nbits != 8 || m == 0 || dim == 0 || dim > (1u << 20) || m > dim || pq_dim != dim || dim % m) IO_FAIL(LANCE_HIP_ENOTSUP, "index_file_open: %s: legacy PQ nbits=%llu m=%llu dim=%llu is not supported", path.c_str(), (unsigned long long)nbits, (unsigned long long)m, (unsigned long long)dim);
  v.index_type = LANCE_HIP_IVF_PQ;
  v.metric = metric == 0 ? LANCE_HIP_L2 : LANCE_HIP_DOT;
  v.d = (uint32_t)dim; v.m = (uint32_t)m; v.nbits = 8; v.transposed = 0;
  v.nlist = (uint32_t)ivf.lengths.size();
  if (v.nlist == 0 || ivf.offsets.size() != v.nlist) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: %s: legacy IVF without partitions / offsets", path.c_str());
  std::string err;
  if (ivf.has_tensor) {
    if (ivf.centroids.shape.size() != 2 || ivf.centroids.shape[0] != v.nlist || ivf.centroids.shape[1] != v.d) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: %s: centroid tensor shape mismatch", path.c_str());
    if (!tensor_to_f32(ivf.centroids, (size_t)v.nlist * v.d, &f->centroids, &v.dtype, &err)) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: centroids: %s", err.c_str());
  } else {
    if (cent_floats.size() != (size_t)v.nlist * v.d) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: %s: %zu centroid values for [%u][%u]", path.c_str(), cent_floats.size(), v.nlist, v.d);
    f->centroids = std::move(cent_floats);
    v.dtype = LANCE_HIP_F32;
  }
  int cb_dtype = LANCE_HIP_F32;
  if (cb_has_tensor) {
    if (cb_tensor.shape.size() != 2 || cb_tensor.shape[0] != 256 || cb_tensor.shape[1] != v.d) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: %s: codebook tensor shape mismatch", path.c_str());
    if (!tensor_to_f32(cb_tensor, (size_t)256 * v.d, &f->codebook, &cb_dtype, &err)) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: codebook: %s", err.c_str());
  } else {
    if (cb_floats.size() != (size_t)256 * v.d) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: %s: %zu codebook values for [256][%u]", path.c_str(), cb_floats.size(), v.d);
    f->codebook = std::move(cb_floats);
  }
  if (cb_dtype != v.dtype) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: codebook and centroids have different element types");
  v.centroids = f->centroids.data();
  v.codebook = f->codebook.data();
  v.has_loss = ivf.has_loss; v.loss = ivf.loss;
  // gather the per-partition [codes][row ids] blocks into the two arrays of the view
  f->part_offsets.assign(v.nlist + 1, 0);
  uint64_t total = 0;
  for (uint32_t p = 0; p < v.nlist; ++p) {
    if (ivf.lengths[p] >= (1ull << 32)) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: partition %u length is not a u32", p);
    total += ivf.lengths[p];
    if (total >= (1ull << 32)) IO_FAIL(LANCE_HIP_ENOTSUP, "index_file_open: more than 2^32 rows");
    f->part_offsets[p + 1] = (uint32_t)total;
  }
  f->payload_buf.resize((size_t)total * v.m);
  f->legacy_row_ids.resize((size_t)total);
  for (uint32_t p = 0; p < v.nlist; ++p) {
    const uint64_t n = ivf.lengths[p], off = ivf.offsets[p], need = n * (v.m + 8);
    if (off > pos || need > pos - off) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: %s: partition %u lies outside the data section", path.c_str(), p);
    memcpy(f->payload_buf.data() + (size_t)f->part_offsets[p] * v.m, b + off, (size_t)(n * v.m));
    memcpy(f->legacy_row_ids.data() + f->part_offsets[p], b + off + n * v.m, (size_t)(n * 8));
  }
  v.part_offsets = f->part_offsets.data();
  v.n_rows = total;
  v.codes = f->payload_buf.data();
  v.row_ids = f->legacy_row_ids.data();
  return LANCE_HIP_OK;
}

static int open_impl(const std::string &dir, lance_hip_index_file *f) {
  std::string err;
  {   // a directory whose index.idx carries the legacy footer (format 0.1 / 0.2) has no auxiliary.idx
    const std::string ip = dir + "/index.idx";
    FILE *fp = fopen(ip.c_str(), "rb");
    if (fp) {
      uint8_t foot[8] = {0};
      const bool got = fseek(fp, -8, SEEK_END) == 0 && fread(foot, 1, 8, fp) == 8;
      fclose(fp);
      uint16_t major, minor;
      memcpy(&major, foot, 2); memcpy(&minor, foot + 2, 2);
      if (got && memcmp(foot + 4, "LANC", 4) == 0 && major == 0 && minor < 3) return open_legacy(ip, f);
    }
  }
  f->idx = FileReader::open(dir + "/index.idx", &err);
  if (!f->idx) IO_FAIL(LANCE_HIP_EIO, "index_file_open: %s", err.c_str());
  f->aux = FileReader::open(dir + "/auxiliary.idx", &err);
  if (!f->aux) IO_FAIL(LANCE_HIP_EIO, "index_file_open: %s", err.c_str());
  lance_hip_index_file_view &v = f->v;

  // ---- index.idx: type, metric, centroids, loss (v2.rs IVFIndex::try_new; lance-index/src/lib.rs INDEX_METADATA_SCHEMA_KEY)
  const std::string *im = f->idx->meta("lance:index");
  if (!im) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: index.idx has no 'lance:index' metadata (legacy v1 index files are not supported)");
  Json jm;
  if (!Json::parse(*im, &jm, &err)) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: lance:index: %s", err.c_str());
  const Json *jt = jm.get("type"), *jd = jm.get("distance_type");
  if (!jt || jt->kind != Json::String || !jd || jd->kind != Json::String) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: lance:index lacks type / distance_type");
  if (jt->str == "IVF_PQ") v.index_type = LANCE_HIP_IVF_PQ;
  else if (jt->str == "IVF_FLAT") v.index_type = LANCE_HIP_IVF_FLAT;
  else IO_FAIL(LANCE_HIP_ENOTSUP, "index_file_open: index type %s is not supported (IVF_PQ and IVF_FLAT are)", jt->str.c_str());
  v.metric = metric_from_string(jd->str);
  if (v.metric < 0) IO_FAIL(LANCE_HIP_ENOTSUP, "index_file_open: distance type '%s' is not supported", jd->str.c_str());

  auto ivf_of = [&](const FileReader &r, const char *what, IvfPb *out) -> int {
    const std::string *pos = r.meta("lance:ivf");
    if (!pos) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: %s has no 'lance:ivf' metadata", what);
    char *stop = nullptr;
    const unsigned long gi = strtoul(pos->c_str(), &stop, 10);
    const uint8_t *p; size_t n;
    if (*stop || pos->empty() || !r.global_buffer(gi, &p, &n) || gi == 0) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: %s: bad 'lance:ivf' buffer index '%s'", what, pos->c_str());
    if (!parse_ivf(p, n, out)) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: %s: malformed IVF message", what);
    return LANCE_HIP_OK;
  };
  IvfPb ivf_idx, ivf_aux;
  LH_TRY(ivf_of(*f->idx, "index.idx", &ivf_idx));
  LH_TRY(ivf_of(*f->aux, "auxiliary.idx", &ivf_aux));
  if (!ivf_idx.has_tensor) IO_FAIL(LANCE_HIP_ENOTSUP, "index_file_open: index.idx has no centroids_tensor%s", ivf_idx.n_legacy_centroids ? " (v1 repeated-float centroids are not supported)" : "");
  if (ivf_idx.centroids.shape.size() != 2) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: centroids tensor is not 2-D");
  v.nlist = (uint32_t)ivf_idx.centroids.shape[0];
  v.d = (uint32_t)ivf_idx.centroids.shape[1];
  if (v.nlist == 0 || v.d == 0 || ivf_idx.lengths.size() != v.nlist) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: %zu partitions in index.idx but centroids are [%u][%u]", ivf_idx.lengths.size(), v.nlist, v.d);
  if (!tensor_to_f32(ivf_idx.centroids, (size_t)v.nlist * v.d, &f->centroids, &v.dtype, &err)) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: centroids: %s", err.c_str());
  v.centroids = f->centroids.data();
  v.has_loss = ivf_idx.has_loss; v.loss = ivf_idx.loss;

  // ---- auxiliary.idx: partition lengths, storage metadata, columns (storage.rs:182-243)
  const std::string *dt = f->aux->meta("distance_type");
  if (!dt || metric_from_string(*dt) != v.metric) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: auxiliary.idx distance_type '%s' disagrees with index.idx", dt ? dt->c_str() : "(missing)");
  if (ivf_aux.lengths.size() != v.nlist) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: auxiliary.idx lists %zu partitions, index.idx %u", ivf_aux.lengths.size(), v.nlist);
  f->part_offsets.assign(v.nlist + 1, 0);
  uint64_t total = 0;
  for (uint32_t p = 0; p < v.nlist; ++p) {
    // ivf/storage.rs:216-229: offsets absent -> prefix sums of lengths; present -> row offsets, which the writer makes the same
    if (!ivf_aux.offsets.empty() && (ivf_aux.offsets.size() != v.nlist || ivf_aux.offsets[p] != total)) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: partition %u offset is not the running row count", p);
    if (ivf_aux.lengths[p] >= (1ull << 32)) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: partition %u length is not a u32", p);
    total += ivf_aux.lengths[p];
    if (total >= (1ull << 32)) IO_FAIL(LANCE_HIP_ENOTSUP, "index_file_open: more than 2^32 rows");
    f->part_offsets[p + 1] = (uint32_t)total;
  }
  if (total != f->aux->num_rows()) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: partition lengths sum to %llu but auxiliary.idx has %llu rows", (unsigned long long)total, (unsigned long long)f->aux->num_rows());
  v.part_offsets = f->part_offsets.data();
  v.n_rows = total;

  const std::string *sm = f->aux->meta("storage_metadata");
  if (!sm) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: auxiliary.idx has no 'storage_metadata'");
  Json jl, js;
  if (!Json::parse(*sm, &jl, &err) || jl.kind != Json::Array || jl.arr.empty() || jl.arr[0].kind != Json::String) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: storage_metadata is not a list of JSON strings");
  if (!Json::parse(jl.arr[0].str, &js, &err) || js.kind != Json::Object) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: storage_metadata[0]: %s", err.c_str());

  std::string ioerr;
  auto column_view = [&](const char *name, uint32_t want_row_bytes, std::vector<uint8_t> *buf, const uint8_t **out) -> int {
    const int c = f->aux->column_of(name);
    if (c < 0) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: auxiliary.idx has no column '%s'", name);
    const Column &col = f->aux->column((size_t)c);
    if (col.rows != total) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: column '%s' has %llu rows, expected %llu", name, (unsigned long long)col.rows, (unsigned long long)total);
    if (total && col.row_bytes != want_row_bytes) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: column '%s' has %u bytes per row, expected %u", name, col.row_bytes, want_row_bytes);
    *out = f->aux->contiguous((size_t)c);
    if (!*out && total) {
      buf->resize((size_t)total * want_row_bytes);
      if (!f->aux->read_rows((size_t)c, 0, total, buf->data(), &ioerr)) IO_FAIL(LANCE_HIP_EIO, "index_file_open: %s", ioerr.c_str());
      *out = buf->data();
    }
    return LANCE_HIP_OK;
  };
  const uint8_t *rid = nullptr, *payload = nullptr;
  LH_TRY(column_view("_rowid", 8, &f->rowid_buf, &rid));
  v.row_ids = reinterpret_cast<const uint64_t *>(rid);

  if (v.index_type == LANCE_HIP_IVF_PQ) {
    uint64_t nbits = 0, m = 0, dim = 0, cpos = 0;
    if (!json_u64(js.get("nbits"), &nbits) || !json_u64(js.get("num_sub_vectors"), &m) || !json_u64(js.get("dimension"), &dim)) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: PQ metadata lacks nbits / num_sub_vectors / dimension");
    if (dim != v.d) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: PQ dimension %llu differs from the centroid dimension %u", (unsigned long long)dim, v.d);
    if ((nbits != 8 && nbits != 4) || m == 0 || v.d % m || (nbits == 4 && m % 2)) IO_FAIL(LANCE_HIP_ENOTSUP, "index_file_open: PQ nbits=%llu num_sub_vectors=%llu is not supported", (unsigned long long)nbits, (unsigned long long)m);
    v.m = (uint32_t)m; v.nbits = (uint32_t)nbits;
    const Json *jtr = js.get("transposed");
    v.transposed = jtr && jtr->kind == Json::Bool && jtr->b;
    json_u64(js.get("codebook_position"), &cpos);
    Tensor cb;
    if (cpos > 0) {   // pq/storage.rs:89-108: a global buffer index (they start at 1)
      const uint8_t *p; size_t n;
      if (!f->aux->global_buffer(cpos, &p, &n) || !parse_tensor(p, n, &cb)) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: codebook global buffer %llu is missing or malformed", (unsigned long long)cpos);
    } else {           // v0.27 and older: the pb Tensor inline as a JSON byte list
      const Json *jc = js.get("codebook_tensor");
      if (!jc || jc->kind != Json::Array || jc->arr.empty()) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: PQ metadata has neither codebook_position nor codebook_tensor");
      f->inline_codebook.resize(jc->arr.size());
      for (size_t i = 0; i < jc->arr.size(); ++i) f->inline_codebook[i] = (uint8_t)jc->arr[i].num;
      if (!parse_tensor(f->inline_codebook.data(), f->inline_codebook.size(), &cb)) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: inline codebook tensor is malformed");
    }
    const size_t ksub = (size_t)1 << nbits;
    if (cb.shape.size() != 2 || cb.shape[0] != ksub || cb.shape[1] != v.d) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: codebook tensor shape is not [%zu][%u]", ksub, v.d);
    int cb_dtype = 0;
    if (!tensor_to_f32(cb, ksub * v.d, &f->codebook, &cb_dtype, &err)) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: codebook: %s", err.c_str());
    if (cb_dtype != v.dtype) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: codebook and centroids have different element types");
    v.codebook = f->codebook.data();
    LH_TRY(column_view("__pq_code", nbits == 4 ? (uint32_t)m / 2 : (uint32_t)m, &f->payload_buf, &payload));
    v.codes = payload;
  } else {
    uint64_t dim = 0;
    if (!json_u64(js.get("dim"), &dim) || dim != v.d) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: flat storage dim disagrees with the centroid dimension %u", v.d);
    const int c = f->aux->column_of("flat");
    if (c < 0) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: auxiliary.idx has no column 'flat'");
    std::string item; uint32_t fdim = 0, ib = 0;
    if (!parse_logical_type(f->aux->fields()[(size_t)c].logical_type, &item, &fdim, &ib) || fdim != v.d || (item != "float" && item != "halffloat"))
      IO_FAIL(LANCE_HIP_ENOTSUP, "index_file_open: flat column type '%s' is not supported", f->aux->fields()[(size_t)c].logical_type.c_str());
    const int fdt = item == "float" ? LANCE_HIP_F32 : LANCE_HIP_F16;
    if (fdt != v.dtype) IO_FAIL(LANCE_HIP_EINVAL, "index_file_open: flat vectors and centroids have different element types");
    LH_TRY(column_view("flat", v.d * ib, &f->payload_buf, &payload));
    v.vectors = payload;
    v.transposed = 0;
  }
  return LANCE_HIP_OK;
}

extern "C" int lance_hip_index_file_open(const char *index_dir, lance_hip_index_file **out) {
  LH_REQUIRE(index_dir && out, "index_file_open: NULL argument");
  std::unique_ptr<lance_hip_index_file> f(new lance_hip_index_file());
  LH_TRY(open_impl(index_dir, f.get()));
  *out = f.release();
  return LANCE_HIP_OK;
}

extern "C" int lance_hip_index_file_get(const lance_hip_index_file *f, lance_hip_index_file_view *view) {
  LH_REQUIRE(f && view, "index_file_get: NULL argument");
  *view = f->v;
  return LANCE_HIP_OK;
}

extern "C" void lance_hip_index_file_close(lance_hip_index_file *f) { delete f; }

// ---------------------------------------------------------------------------------------------------------------------
// writer
// ---------------------------------------------------------------------------------------------------------------------
namespace {
std::string tensor_pb(int dtype, uint64_t rows, uint64_t cols, const float *vals) {
  PbWriter t;
  const uint64_t shape[2] = {rows, cols};
  const size_t count = (size_t)(rows * cols);
  if (dtype == LANCE_HIP_F16) {
    std::vector<uint16_t> h(count);
    for (size_t i = 0; i < count; ++i) h[i] = f2h_host(vals[i]);
    t.varint_field(1, 1);
    t.packed_varints(2, shape, 2);
    t.bytes_field(3, h.data(), count * 2);
  } else {
    t.varint_field(1, 2);
    t.packed_varints(2, shape, 2);
    t.bytes_field(3, vals, count * 4);
  }
  return t.str();
}
}  // namespace

extern "C" int lance_hip_index_file_write(const char *index_dir, const lance_hip_index_file_view *v) {
  LH_REQUIRE(index_dir && v, "index_file_write: NULL argument");
  LH_REQUIRE(v->index_type == LANCE_HIP_IVF_PQ || v->index_type == LANCE_HIP_IVF_FLAT, "index_file_write: bad index type %d", v->index_type);
  LH_REQUIRE(v->metric == LANCE_HIP_L2 || v->metric == LANCE_HIP_COSINE || v->metric == LANCE_HIP_DOT, "index_file_write: bad metric %d", v->metric);
  LH_REQUIRE(v->dtype == LANCE_HIP_F32 || v->dtype == LANCE_HIP_F16, "index_file_write: model element type must be F32 or F16");
  LH_REQUIRE(v->nlist > 0 && v->d > 0 && v->centroids && v->part_offsets, "index_file_write: centroids / part_offsets missing");
  LH_REQUIRE(v->part_offsets[0] == 0 && v->part_offsets[v->nlist] == v->n_rows, "index_file_write: offsets do not cover n_rows");
  LH_REQUIRE(v->n_rows == 0 || v->row_ids, "index_file_write: row_ids missing");
  const bool pq = v->index_type == LANCE_HIP_IVF_PQ;
  uint32_t code_bytes = 0;
  if (pq) {
    LH_REQUIRE((v->nbits == 8 || v->nbits == 4) && v->m > 0 && v->d % v->m == 0 && !(v->nbits == 4 && v->m % 2), "index_file_write: bad PQ shape m=%u nbits=%u", v->m, v->nbits);
    LH_REQUIRE(v->codebook && (v->n_rows == 0 || v->codes), "index_file_write: codebook / codes missing");
    code_bytes = v->nbits == 4 ? v->m / 2 : v->m;
  } else {
    LH_REQUIRE(v->n_rows == 0 || v->vectors, "index_file_write: vectors missing");
  }
  if (mkdir(index_dir, 0777) != 0 && errno != EEXIST) IO_FAIL(LANCE_HIP_EIO, "index_file_write: cannot create %s: %s", index_dir, strerror(errno));
  const std::string dir(index_dir);
  std::string err;

  // ---- auxiliary.idx (builder.rs:958-966, 1030-1050)
  {
    std::vector<Field> fields(2);
    fields[0].name = "_rowid"; fields[0].logical_type = "uint64"; fields[0].nullable = true; fields[0].id = 0;
    fields[1].id = 1;
    if (pq) { fields[1].name = "__pq_code"; fields[1].logical_type = "fixed_size_list:uint8:" + std::to_string(code_bytes); }
    else { fields[1].name = "flat"; fields[1].logical_type = std::string("fixed_size_list:") + (v->dtype == LANCE_HIP_F16 ? "halffloat:" : "float:") + std::to_string(v->d); fields[1].nullable = true; }
    std::vector<uint64_t> offs(v->nlist), lens(v->nlist);
    for (uint32_t p = 0; p < v->nlist; ++p) {
      LH_REQUIRE(v->part_offsets[p + 1] >= v->part_offsets[p], "index_file_write: part_offsets not monotone");
      offs[p] = v->part_offsets[p]; lens[p] = v->part_offsets[p + 1] - v->part_offsets[p];
    }
    auto w = FileWriter::create(dir + "/auxiliary.idx", fields, &err);
    if (!w) IO_FAIL(LANCE_HIP_EIO, "index_file_write: %s", err.c_str());
    PbWriter ivf;
    ivf.packed_varints(2, offs.data(), offs.size());
    ivf.packed_varints(3, lens.data(), lens.size());
    w->add_schema_metadata("distance_type", metric_name(v->metric));
    const uint32_t ivf_pos = w->add_global_buffer(ivf.str().data(), ivf.str().size());
    w->add_schema_metadata("lance:ivf", std::to_string(ivf_pos));
    std::string meta;
    if (pq) {
      const std::string cb = tensor_pb(v->dtype, (uint64_t)1 << v->nbits, v->d, v->codebook);
      const uint32_t cb_pos = w->add_global_buffer(cb.data(), cb.size());
      // field order of ProductQuantizationMetadata (pq/storage.rs:52-67)
      meta = "{\"codebook_position\":" + std::to_string(cb_pos) + ",\"nbits\":" + std::to_string(v->nbits) + ",\"num_sub_vectors\":" + std::to_string(v->m) +
             ",\"dimension\":" + std::to_string(v->d) + ",\"codebook_tensor\":[],\"transposed\":true}";
    } else {
      meta = "{\"dim\":" + std::to_string(v->d) + "}";
    }
    w->add_schema_metadata("storage_metadata", "[" + json_quote(meta) + "]");
    w->set_column(0, v->row_ids, v->n_rows, 64, 1);
    std::vector<uint8_t> tcodes;
    const uint8_t *codes = v->codes;
    if (pq && !v->transposed && v->n_rows) {   // builder.rs:1039-1043 always stores transposed:true (pq/storage.rs:430-449)
      tcodes.resize((size_t)v->n_rows * code_bytes);
      for (uint32_t p = 0; p < v->nlist; ++p) {
        const size_t a = v->part_offsets[p], np_ = v->part_offsets[p + 1] - a;
        const uint8_t *src = v->codes + a * code_bytes;
        uint8_t *dst = tcodes.data() + a * code_bytes;
        for (size_t r = 0; r < np_; ++r)
          for (uint32_t c = 0; c < code_bytes; ++c) dst[(size_t)c * np_ + r] = src[r * code_bytes + c];
      }
      codes = tcodes.data();
    }
    if (pq) w->set_column(1, codes, v->n_rows, 8, code_bytes);
    else w->set_column(1, v->vectors, v->n_rows, v->dtype == LANCE_HIP_F16 ? 16 : 32, v->d);
    if (!w->finish(&err)) IO_FAIL(LANCE_HIP_EIO, "index_file_write: auxiliary.idx: %s", err.c_str());
  }
  // ---- index.idx (builder.rs:967-971, 1052-1071): the FLAT sub-index stores nothing per partition
  {
    std::vector<Field> fields(1);
    fields[0].name = "__flat_marker"; fields[0].logical_type = "uint64";
    auto w = FileWriter::create(dir + "/index.idx", fields, &err);
    if (!w) IO_FAIL(LANCE_HIP_EIO, "index_file_write: %s", err.c_str());
    std::vector<uint64_t> zeros(v->nlist, 0);
    PbWriter ivf;
    ivf.packed_varints(2, zeros.data(), zeros.size());
    ivf.packed_varints(3, zeros.data(), zeros.size());
    ivf.bytes_field(4, tensor_pb(v->dtype, v->nlist, v->d, v->centroids));
    if (v->has_loss) { uint64_t bits; memcpy(&bits, &v->loss, 8); ivf.fixed64_field(5, bits); }
    w->add_schema_metadata("lance:index", std::string("{\"type\":\"") + (pq ? "IVF_PQ" : "IVF_FLAT") + "\",\"distance_type\":\"" + metric_name(v->metric) + "\"}");
    const uint32_t ivf_pos = w->add_global_buffer(ivf.str().data(), ivf.str().size());
    w->add_schema_metadata("lance:ivf", std::to_string(ivf_pos));
    std::string parts = "[";
    for (uint32_t p = 0; p < v->nlist; ++p) parts += p ? ",\"\"" : "\"\"";
    parts += "]";
    w->add_schema_metadata("lance:flat", parts);
    w->set_column(0, nullptr, 0, 64, 1);
    if (!w->finish(&err)) IO_FAIL(LANCE_HIP_EIO, "index_file_write: index.idx: %s", err.c_str());
  }
  return LANCE_HIP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// generic flat column access (vector columns of v2.0 data files)
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int lance_hip_file_read_column(const char *path, const char *column, void *dst, uint64_t dst_bytes, uint64_t *rows,
                                          uint32_t *row_bytes) {
  LH_REQUIRE(path && column, "file_read_column: NULL argument");
  std::string err;
  auto r = FileReader::open(path, &err);
  if (!r) IO_FAIL(LANCE_HIP_EIO, "file_read_column: %s", err.c_str());
  const int c = r->column_of(column);
  if (c < 0) IO_FAIL(LANCE_HIP_EINVAL, "file_read_column: %s has no top-level column '%s'", path, column);
  const Column &col = r->column((size_t)c);
  if (rows) *rows = col.rows;
  if (row_bytes) *row_bytes = col.row_bytes;
  if (!dst) return LANCE_HIP_OK;
  uint64_t need = 0;
  LH_REQUIRE(!__builtin_mul_overflow(col.rows, (uint64_t)col.row_bytes, &need), "file_read_column: column size overflows");
  LH_REQUIRE(dst_bytes >= need, "file_read_column: destination holds %llu bytes, column needs %llu", (unsigned long long)dst_bytes, (unsigned long long)need);
  if (!r->read_rows((size_t)c, 0, col.rows, dst, &err)) IO_FAIL(LANCE_HIP_EIO, "file_read_column: %s", err.c_str());
  return LANCE_HIP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// shuffle buffers: the (row_id, __ivf_part_id, __pq_code) rows an accelerator hands to the reference's index builder
// (python/lance/vector.py:659-665 output_schema; consumed by IvfIndexBuilder::shuffle_dataset, builder.rs:509-546, which
// renames row_id -> _rowid).  Written as one Lance v2.0 file with the generic FileWriter (the encodings the reference's own
// FileWriter emits for u64 / u32 / FSL<u8>); rows whose partition id is LANCE_HIP_NONE are dropped, as the reference's
// transform drops non-finite vectors before the shuffle.  Host pointers.
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int lance_hip_shuffle_buffer_write(const char *path, const uint64_t *row_ids, const uint32_t *part_ids, const uint8_t *codes,
                                              uint64_t n, uint32_t code_bytes, uint64_t *rows_written) {
  LH_REQUIRE(path && part_ids && codes && code_bytes > 0, "shuffle_buffer_write: NULL / empty argument");
  std::vector<uint64_t> rid;
  std::vector<uint32_t> part;
  std::vector<uint8_t> cd;
  rid.reserve(n); part.reserve(n); cd.reserve((size_t)n * code_bytes);
  for (uint64_t i = 0; i < n; ++i) {
    if (part_ids[i] == LANCE_HIP_NONE) continue;
    rid.push_back(row_ids ? row_ids[i] : i);
    part.push_back(part_ids[i]);
    cd.insert(cd.end(), codes + (size_t)i * code_bytes, codes + (size_t)(i + 1) * code_bytes);
  }
  std::vector<Field> fields(3);
  fields[0].name = "row_id"; fields[0].id = 0; fields[0].logical_type = "uint64";
  fields[1].name = "__ivf_part_id"; fields[1].id = 1; fields[1].logical_type = "uint32";
  fields[2].name = "__pq_code"; fields[2].id = 2; fields[2].logical_type = "fixed_size_list:uint8:" + std::to_string(code_bytes);
  std::string err;
  auto w = FileWriter::create(path, fields, &err);
  if (!w) IO_FAIL(LANCE_HIP_EIO, "shuffle_buffer_write: %s", err.c_str());
  w->set_column(0, rid.data(), rid.size(), 64, 1);
  w->set_column(1, part.data(), part.size(), 32, 1);
  w->set_column(2, cd.data(), rid.size(), 8, code_bytes);
  if (!w->finish(&err)) IO_FAIL(LANCE_HIP_EIO, "shuffle_buffer_write: %s", err.c_str());
  if (rows_written) *rows_written = rid.size();
  return LANCE_HIP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// files <-> HBM
// ---------------------------------------------------------------------------------------------------------------------
namespace {
struct DevTmp {   // device staging freed on scope exit
  std::vector<void *> ptrs;
  ~DevTmp() { for (void *p : ptrs) (void)hipFree(p); }
  // src may point into the read-only file mapping: it is never handed to HIP directly (the runtime pins large pageable
  // sources, which a PROT_READ file mapping need not allow) but copied through the context's pinned staging buffer
  int upload(lance_hip_ctx *ctx, const void *src, size_t bytes, void **out) {
    void *p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) IO_FAIL(LANCE_HIP_ENOMEM, "index_load: hipMalloc(%zu) failed", bytes);
    ptrs.push_back(p);
    size_t kChunk = (size_t)64 << 20;
    if (const char *e = getenv("LANCE_HIP_STAGE_CHUNK")) {   // tests: many small chunks
      const unsigned long long ov = strtoull(e, nullptr, 10);
      if (ov) kChunk = (size_t)ov;
    }
    for (size_t off = 0; off < bytes; off += kChunk) {
      const size_t nb = std::min(kChunk, bytes - off);
      void *stage = ctx->host_staging(nb);
      if (!stage) return LANCE_HIP_ENOMEM;
      memcpy(stage, static_cast<const uint8_t *>(src) + off, nb);
      LH_CHECK_HIP(hipMemcpyAsync(static_cast<uint8_t *>(p) + off, stage, nb, hipMemcpyHostToDevice, ctx->stream));
      LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));   // the staging buffer is reused by the next chunk
    }
    *out = p;
    return LANCE_HIP_OK;
  }
};
}  // namespace

extern "C" int lance_hip_index_load(lance_hip_ctx *ctx, const char *index_dir, int dtype, lance_hip_index **out) {
  return lance_hip_index_load_lists(ctx, index_dir, dtype, 1, 0, out);
}

extern "C" int lance_hip_index_load_lists(lance_hip_ctx *ctx, const char *index_dir, int dtype, uint32_t list_mod, uint32_t list_rem,
                                          lance_hip_index **out) {
  LH_REQUIRE(ctx && index_dir && out, "index_load: NULL argument");
  LH_REQUIRE(list_mod >= 1 && list_rem < list_mod, "index_load: bad list shard %u of %u", list_rem, list_mod);
  lance_hip_index_file *f = nullptr;
  LH_TRY(lance_hip_index_file_open(index_dir, &f));
  std::unique_ptr<lance_hip_index_file> guard(f);
  const lance_hip_index_file_view &v = f->v;
  // dtype is the element type of the indexed column (queries, raw vectors); the stored model must be its model type
  LH_REQUIRE(dtype == LANCE_HIP_F32 || dtype == LANCE_HIP_F16 || dtype == LANCE_HIP_I8, "index_load: bad dtype %d", dtype);
  LH_REQUIRE((dtype == LANCE_HIP_F16) == (v.dtype == LANCE_HIP_F16), "index_load: the files hold %s tensors, which does not fit dtype %d", v.dtype == LANCE_HIP_F16 ? "f16" : "f32", dtype);
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  DevTmp tmp;
  int r;
  if (v.index_type == LANCE_HIP_IVF_PQ) {
    // model tensors go in as f32 host arrays for f32/int8 columns; an f16 model is narrowed back (exact) and staged on the device
    const void *cent = v.centroids, *cb = v.codebook;
    std::vector<uint16_t> hc, hb;
    if (dtype == LANCE_HIP_F16) {
      const size_t nc = (size_t)v.nlist * v.d, nb = ((size_t)1 << v.nbits) * v.d;
      hc.resize(nc); hb.resize(nb);
      for (size_t i = 0; i < nc; ++i) hc[i] = f2h_host(v.centroids[i]);
      for (size_t i = 0; i < nb; ++i) hb[i] = f2h_host(v.codebook[i]);
      void *dc, *db;
      LH_TRY(tmp.upload(ctx, hc.data(), nc * 2, &dc));
      LH_TRY(tmp.upload(ctx, hb.data(), nb * 2, &db));
      cent = dc; cb = db;
    }
    const uint32_t code_bytes = v.nbits == 4 ? v.m / 2 : v.m;
    void *dcodes = nullptr, *drid = nullptr;
    if (list_mod == 1) {
      LH_TRY(tmp.upload(ctx, v.codes, (size_t)v.n_rows * code_bytes, &dcodes));
      LH_TRY(tmp.upload(ctx, v.row_ids, (size_t)v.n_rows * 8, &drid));
      r = lance_hip_index_from_storage(ctx, dtype, v.metric, v.d, cent, v.nlist, cb, v.m, v.nbits, v.part_offsets,
                                       static_cast<const uint8_t *>(dcodes), v.transposed, static_cast<const uint64_t *>(drid),
                                       v.n_rows, out);
    } else {
      // list shard (lists p with p % list_mod == list_rem): the foreign lists are emptied, the owned code blocks and row
      // ids -- each contiguous in the file, in either layout -- are packed into host buffers and uploaded once
      std::vector<uint32_t> offs(v.nlist + 1, 0);
      for (uint32_t p = 0; p < v.nlist; ++p)
        offs[p + 1] = offs[p] + (p % list_mod == list_rem ? v.part_offsets[p + 1] - v.part_offsets[p] : 0);
      const uint64_t nloc = offs[v.nlist];
      std::vector<uint8_t> hcodes((size_t)nloc * code_bytes);
      std::vector<uint64_t> hrid((size_t)nloc);
      for (uint32_t p = 0; p < v.nlist; ++p) {
        const size_t np_ = offs[p + 1] - offs[p];
        if (np_ == 0) continue;
        memcpy(hcodes.data() + (size_t)offs[p] * code_bytes, v.codes + (size_t)v.part_offsets[p] * code_bytes, np_ * code_bytes);
        memcpy(hrid.data() + offs[p], v.row_ids + v.part_offsets[p], np_ * 8);
      }
      LH_TRY(tmp.upload(ctx, hcodes.data(), hcodes.size(), &dcodes));
      LH_TRY(tmp.upload(ctx, hrid.data(), hrid.size() * 8, &drid));
      r = lance_hip_index_from_storage(ctx, dtype, v.metric, v.d, cent, v.nlist, cb, v.m, v.nbits, offs.data(),
                                       static_cast<const uint8_t *>(dcodes), v.transposed, static_cast<const uint64_t *>(drid),
                                       nloc, out);
    }
  } else {
    LH_REQUIRE(dtype != LANCE_HIP_I8, "index_load: IVF_FLAT files hold float vectors; int8 columns are not stored this way");
    LH_REQUIRE(list_mod == 1, "index_load: list shards are implemented for IVF_PQ indices only");
    std::vector<uint32_t> part_ids((size_t)v.n_rows);
    for (uint32_t p = 0; p < v.nlist; ++p) std::fill(part_ids.begin() + v.part_offsets[p], part_ids.begin() + v.part_offsets[p + 1], p);
    const size_t es = dtype == LANCE_HIP_F16 ? 2 : 4;
    void *dx, *dp, *dr, *dc;
    LH_TRY(tmp.upload(ctx, v.vectors, (size_t)v.n_rows * v.d * es, &dx));
    LH_TRY(tmp.upload(ctx, part_ids.data(), (size_t)v.n_rows * 4, &dp));
    LH_TRY(tmp.upload(ctx, v.row_ids, (size_t)v.n_rows * 8, &dr));
    std::vector<uint16_t> hc;
    const void *cent_src = v.centroids;
    if (dtype == LANCE_HIP_F16) {
      hc.resize((size_t)v.nlist * v.d);
      for (size_t i = 0; i < hc.size(); ++i) hc[i] = f2h_host(v.centroids[i]);
      cent_src = hc.data();
    }
    LH_TRY(tmp.upload(ctx, cent_src, (size_t)v.nlist * v.d * es, &dc));
    r = lance_hip_ivfflat_create(ctx, dtype, v.metric, v.d, dc, v.nlist, dx, static_cast<const uint32_t *>(dp),
                                 static_cast<const uint64_t *>(dr), v.n_rows, out);
  }
  if (r == LANCE_HIP_OK) LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return r;
}

extern "C" int lance_hip_index_save(lance_hip_ctx *ctx, const lance_hip_index *idx, const char *index_dir, int has_loss, double loss) {
  LH_REQUIRE(ctx && idx && index_dir, "index_save: NULL argument");
  LH_CHECK_HIP(hipSetDevice(ctx->device));
  lance_hip_index_file_view v{};
  const bool pq = idx->m != 0;
  v.index_type = pq ? LANCE_HIP_IVF_PQ : LANCE_HIP_IVF_FLAT;
  v.metric = idx->metric;
  v.dtype = idx->dtype == LANCE_HIP_F16 ? LANCE_HIP_F16 : LANCE_HIP_F32;
  v.d = idx->d; v.nlist = idx->nlist; v.m = idx->m; v.nbits = idx->nbits; v.n_rows = idx->n;
  v.transposed = pq; v.has_loss = has_loss; v.loss = loss;
  std::vector<float> cent((size_t)idx->nlist * idx->d), cb;
  std::vector<uint64_t> rid((size_t)idx->n);
  std::vector<uint8_t> codes;
  std::vector<float> vec;
  std::vector<uint16_t> vech;
  LH_CHECK_HIP(hipMemcpyAsync(cent.data(), idx->centroids, cent.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (pq) {
    cb.resize(((size_t)1 << idx->nbits) * idx->d);
    LH_CHECK_HIP(hipMemcpyAsync(cb.data(), idx->codebook, cb.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    codes.resize((size_t)idx->n * idx->code_bytes());
    LH_TRY(lance_hip_index_export(ctx, idx, nullptr, codes.data(), rid.data()));
    v.codebook = cb.data(); v.codes = codes.data();
  } else {
    vec.resize((size_t)idx->n * idx->d);
    if (idx->n) {
      LH_CHECK_HIP(hipMemcpyAsync(vec.data(), idx->vectors, vec.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
      LH_CHECK_HIP(hipMemcpyAsync(rid.data(), idx->row_ids, rid.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    if (v.dtype == LANCE_HIP_F16) {
      vech.resize(vec.size());
      for (size_t i = 0; i < vec.size(); ++i) vech[i] = f2h_host(vec[i]);
      v.vectors = vech.data();
    } else {
      v.vectors = vec.data();
    }
  }
  LH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  v.centroids = cent.data();
  v.part_offsets = idx->part_offsets_h.data();
  v.row_ids = rid.data();
  return lance_hip_index_file_write(index_dir, &v);
}
