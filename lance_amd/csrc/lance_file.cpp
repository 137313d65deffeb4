// lance_file.cpp -- see lance_file.h.  Host only; no HIP.
#include "lance_file.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace lancefile {

// =====================================================================================================================
// protobuf wire
// =====================================================================================================================
bool PbReader::varint(const uint8_t *&p, const uint8_t *end, uint64_t *out) {
  uint64_t r = 0;
  for (int shift = 0; shift < 64; shift += 7) {
    if (p >= end) return false;
    const uint8_t c = *p++;
    r |= (uint64_t)(c & 0x7f) << shift;
    if (!(c & 0x80)) { *out = r; return true; }
  }
  return false;
}

bool PbReader::next(PbField *f, bool *ok) {
  *ok = true;
  if (p_ >= end_) return false;
  uint64_t key;
  if (!varint(p_, end_, &key)) { *ok = false; return false; }
  f->number = (uint32_t)(key >> 3);
  f->wire = (uint32_t)(key & 7);
  f->data = nullptr; f->size = 0; f->value = 0;
  switch (f->wire) {
    case 0:
      if (!varint(p_, end_, &f->value)) { *ok = false; return false; }
      return true;
    case 1:
      if (end_ - p_ < 8) { *ok = false; return false; }
      memcpy(&f->value, p_, 8); p_ += 8;
      return true;
    case 5: {
      if (end_ - p_ < 4) { *ok = false; return false; }
      uint32_t v; memcpy(&v, p_, 4); p_ += 4; f->value = v;
      return true;
    }
    case 2: {
      uint64_t n;
      if (!varint(p_, end_, &n) || n > (uint64_t)(end_ - p_)) { *ok = false; return false; }
      f->data = p_; f->size = (size_t)n; p_ += n;
      return true;
    }
    default:
      *ok = false;
      return false;
  }
}

bool PbReader::append_varints(const PbField &f, std::vector<uint64_t> *out) {
  if (f.wire == 0) { out->push_back(f.value); return true; }
  if (f.wire != 2) return false;
  const uint8_t *p = f.data, *end = f.data + f.size;
  while (p < end) {
    uint64_t v;
    if (!varint(p, end, &v)) return false;
    out->push_back(v);
  }
  return true;
}

void PbWriter::raw_varint(uint64_t v) {
  while (v >= 0x80) { buf_.push_back((char)(uint8_t)(v | 0x80)); v >>= 7; }
  buf_.push_back((char)(uint8_t)v);
}
void PbWriter::varint_field(uint32_t number, uint64_t v) { raw_varint((uint64_t)number << 3); raw_varint(v); }
void PbWriter::bytes_field(uint32_t number, const void *p, size_t n) {
  raw_varint(((uint64_t)number << 3) | 2);
  raw_varint(n);
  buf_.append(reinterpret_cast<const char *>(p), n);
}
void PbWriter::packed_varints(uint32_t number, const uint64_t *v, size_t n) {
  if (n == 0) return;
  PbWriter inner;
  for (size_t i = 0; i < n; ++i) inner.raw_varint(v[i]);
  bytes_field(number, inner.str());
}
void PbWriter::fixed64_field(uint32_t number, uint64_t bits) {
  raw_varint(((uint64_t)number << 3) | 1);
  buf_.append(reinterpret_cast<const char *>(&bits), 8);
}

// =====================================================================================================================
// JSON
// =====================================================================================================================
namespace {
struct JsonParser {
  const char *p, *end;
  std::string err;
  void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p; }
  bool fail(const char *m) { if (err.empty()) err = m; return false; }
  static void utf8(uint32_t cp, std::string *s) {
    if (cp < 0x80) s->push_back((char)cp);
    else if (cp < 0x800) { s->push_back((char)(0xC0 | (cp >> 6))); s->push_back((char)(0x80 | (cp & 0x3F))); }
    else if (cp < 0x10000) { s->push_back((char)(0xE0 | (cp >> 12))); s->push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s->push_back((char)(0x80 | (cp & 0x3F))); }
    else { s->push_back((char)(0xF0 | (cp >> 18))); s->push_back((char)(0x80 | ((cp >> 12) & 0x3F))); s->push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s->push_back((char)(0x80 | (cp & 0x3F))); }
  }
  bool hex4(uint32_t *out) {
    if (end - p < 4) return fail("truncated \\u escape");
    uint32_t v = 0;
    for (int i = 0; i < 4; ++i) {
      const char c = *p++;
      v <<= 4;
      if (c >= '0' && c <= '9') v |= (uint32_t)(c - '0');
      else if (c >= 'a' && c <= 'f') v |= (uint32_t)(c - 'a' + 10);
      else if (c >= 'A' && c <= 'F') v |= (uint32_t)(c - 'A' + 10);
      else return fail("bad \\u escape");
    }
    *out = v;
    return true;
  }
  bool string(std::string *out) {
    if (p >= end || *p != '"') return fail("expected string");
    ++p;
    out->clear();
    while (p < end && *p != '"') {
      if (*p == '\\') {
        if (++p >= end) return fail("truncated escape");
        const char c = *p++;
        switch (c) {
          case '"': out->push_back('"'); break;
          case '\\': out->push_back('\\'); break;
          case '/': out->push_back('/'); break;
          case 'b': out->push_back('\b'); break;
          case 'f': out->push_back('\f'); break;
          case 'n': out->push_back('\n'); break;
          case 'r': out->push_back('\r'); break;
          case 't': out->push_back('\t'); break;
          case 'u': {
            uint32_t cp;
            if (!hex4(&cp)) return false;
            if (cp >= 0xD800 && cp < 0xDC00 && end - p >= 6 && p[0] == '\\' && p[1] == 'u') {
              p += 2;
              uint32_t lo;
              if (!hex4(&lo)) return false;
              cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
            }
            utf8(cp, out);
            break;
          }
          default: return fail("bad escape");
        }
      } else {
        out->push_back(*p++);
      }
    }
    if (p >= end) return fail("unterminated string");
    ++p;
    return true;
  }
  bool value(Json *out, int depth) {
    if (depth > 64) return fail("nesting too deep");
    ws();
    if (p >= end) return fail("unexpected end");
    const char c = *p;
    if (c == '{') {
      ++p; out->kind = Json::Object;
      ws();
      if (p < end && *p == '}') { ++p; return true; }
      for (;;) {
        ws();
        std::string k;
        if (!string(&k)) return false;
        ws();
        if (p >= end || *p != ':') return fail("expected ':'");
        ++p;
        Json v;
        if (!value(&v, depth + 1)) return false;
        out->obj.emplace_back(std::move(k), std::move(v));
        ws();
        if (p < end && *p == ',') { ++p; continue; }
        if (p < end && *p == '}') { ++p; return true; }
        return fail("expected ',' or '}'");
      }
    }
    if (c == '[') {
      ++p; out->kind = Json::Array;
      ws();
      if (p < end && *p == ']') { ++p; return true; }
      for (;;) {
        Json v;
        if (!value(&v, depth + 1)) return false;
        out->arr.push_back(std::move(v));
        ws();
        if (p < end && *p == ',') { ++p; continue; }
        if (p < end && *p == ']') { ++p; return true; }
        return fail("expected ',' or ']'");
      }
    }
    if (c == '"') { out->kind = Json::String; return string(&out->str); }
    if (end - p >= 4 && !strncmp(p, "true", 4)) { p += 4; out->kind = Json::Bool; out->b = true; return true; }
    if (end - p >= 5 && !strncmp(p, "false", 5)) { p += 5; out->kind = Json::Bool; out->b = false; return true; }
    if (end - p >= 4 && !strncmp(p, "null", 4)) { p += 4; out->kind = Json::Null; return true; }
    if (c == '-' || (c >= '0' && c <= '9')) {
      const char *q = p;
      while (q < end && (*q == '-' || *q == '+' || *q == '.' || *q == 'e' || *q == 'E' || (*q >= '0' && *q <= '9'))) ++q;
      std::string tok(p, q);
      char *stop = nullptr;
      out->num = strtod(tok.c_str(), &stop);
      if (stop == tok.c_str() || *stop) return fail("bad number");
      out->kind = Json::Number;
      p = q;
      return true;
    }
    return fail("unexpected character");
  }
};
}  // namespace

const Json *Json::get(const char *key) const {
  if (kind != Object) return nullptr;
  for (const auto &kv : obj)
    if (kv.first == key) return &kv.second;
  return nullptr;
}

bool Json::parse(const std::string &text, Json *out, std::string *err) {
  JsonParser ps{text.data(), text.data() + text.size(), {}};
  *out = Json();
  if (!ps.value(out, 0)) { if (err) *err = "JSON: " + ps.err; return false; }
  ps.ws();
  if (ps.p != ps.end) { if (err) *err = "JSON: trailing characters"; return false; }
  return true;
}

std::string json_quote(const std::string &s) {
  std::string o = "\"";
  for (const char ch : s) {
    const unsigned char c = (unsigned char)ch;
    switch (c) {
      case '"': o += "\\\""; break;
      case '\\': o += "\\\\"; break;
      case '\n': o += "\\n"; break;
      case '\r': o += "\\r"; break;
      case '\t': o += "\\t"; break;
      case '\b': o += "\\b"; break;
      case '\f': o += "\\f"; break;
      default:
        if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); o += b; }
        else o.push_back(ch);
    }
  }
  o.push_back('"');
  return o;
}

// =====================================================================================================================
// schema helpers
// =====================================================================================================================
static bool item_width(const std::string &item, uint32_t *bytes) {
  static const struct { const char *name; uint32_t bytes; } kTypes[] = {
      {"uint8", 1}, {"int8", 1}, {"uint16", 2}, {"int16", 2}, {"halffloat", 2}, {"uint32", 4}, {"int32", 4},
      {"float", 4}, {"uint64", 8}, {"int64", 8}, {"double", 8}};
  for (const auto &t : kTypes)
    if (item == t.name) { *bytes = t.bytes; return true; }
  return false;
}

bool parse_logical_type(const std::string &t, std::string *item, uint32_t *dim, uint32_t *item_bytes) {
  static const char kFsl[] = "fixed_size_list:";
  if (t.compare(0, sizeof kFsl - 1, kFsl) == 0) {
    const size_t a = sizeof kFsl - 1, b = t.rfind(':');
    if (b == std::string::npos || b <= a) return false;
    *item = t.substr(a, b - a);
    char *stop = nullptr;
    const unsigned long d = strtoul(t.c_str() + b + 1, &stop, 10);
    if (*stop || d == 0 || d > 0x7fffffffUL) return false;
    *dim = (uint32_t)d;
    return item_width(*item, item_bytes);
  }
  *item = t;
  *dim = 1;
  return item_width(t, item_bytes);
}

// =====================================================================================================================
// reader
// =====================================================================================================================
namespace {
bool ends_with(const uint8_t *p, size_t n, const char *suffix) {
  const size_t m = strlen(suffix);
  return n >= m && memcmp(p + n - m, suffix, m) == 0;
}

#define PB_LOOP(reader, f)                         \
  bool _ok = true;                                 \
  for (PbField f; (reader).next(&f, &_ok);)

// ArrayEncoding -> (bits per value, total list dimension); refuses everything but flat / FSL-of-flat without nulls
bool decode_array_encoding(const uint8_t *p, size_t n, uint32_t *bits, uint32_t *dim, std::string *err, int depth = 0) {
  if (depth > 8) { *err = "page encoding nested too deep"; return false; }
  PbReader r(p, n);
  bool seen = false;
  PB_LOOP(r, f) {
    if (f.wire != 2) { *err = "malformed ArrayEncoding"; return false; }
    seen = true;
    if (f.number == 2) {   // Nullable
      PbReader rn(f.data, f.size);
      bool got = false, okn = true;
      for (PbField g; rn.next(&g, &okn);) {
        if (g.number == 1 && g.wire == 2) {   // NoNull{values}
          PbReader rv(g.data, g.size);
          bool okv = true;
          for (PbField h; rv.next(&h, &okv);)
            if (h.number == 1 && h.wire == 2) { if (!decode_array_encoding(h.data, h.size, bits, dim, err, depth + 1)) return false; got = true; }
          if (!okv) { *err = "malformed NoNull encoding"; return false; }
        } else {
          *err = "page holds null values (SomeNull/AllNull encoding): not supported for index columns";
          return false;
        }
      }
      if (!okn || !got) { *err = "malformed Nullable encoding"; return false; }
    } else if (f.number == 1) {   // Flat
      PbReader rf(f.data, f.size);
      bool okf = true;
      for (PbField g; rf.next(&g, &okf);) {
        if (g.number == 1 && g.wire == 0) *bits = (uint32_t)g.value;
        else if (g.number == 2 && g.wire == 2) {
          PbReader rb(g.data, g.size);
          bool okb = true;
          for (PbField h; rb.next(&h, &okb);)
            if (h.wire == 0 && h.value != 0) { *err = "page encoding references a non-page / non-zero buffer: not supported"; return false; }
          if (!okb) { *err = "malformed Buffer"; return false; }
        } else if (g.number == 3) {
          *err = "compressed flat pages are not supported";
          return false;
        }
      }
      if (!okf) { *err = "malformed Flat encoding"; return false; }
    } else if (f.number == 3) {   // FixedSizeList
      PbReader rl(f.data, f.size);
      bool okl = true;
      uint32_t d = 0;
      const uint8_t *items = nullptr; size_t items_n = 0;
      for (PbField g; rl.next(&g, &okl);) {
        if (g.number == 1 && g.wire == 0) d = (uint32_t)g.value;
        else if (g.number == 2 && g.wire == 2) { items = g.data; items_n = g.size; }
        else if (g.number == 3 && g.wire == 0 && g.value) { *err = "fixed-size lists with validity are not supported"; return false; }
      }
      if (!okl || d == 0 || !items) { *err = "malformed FixedSizeList encoding"; return false; }
      if (__builtin_mul_overflow(*dim, d, dim)) { *err = "fixed-size-list dimension overflows"; return false; }
      if (!decode_array_encoding(items, items_n, bits, dim, err, depth + 1)) return false;
    } else {
      *err = "unsupported page encoding (ArrayEncoding field " + std::to_string(f.number) + "); only flat / fixed-size-list pages are read";
      return false;
    }
  }
  if (!_ok || !seen) { *err = "malformed ArrayEncoding"; return false; }
  return true;
}
}  // namespace

FileReader::~FileReader() {
  if (map_) munmap(const_cast<uint8_t *>(map_), size_);
  if (fd_ >= 0) close(fd_);
}

std::unique_ptr<FileReader> FileReader::open(const std::string &path, std::string *err) {
  std::unique_ptr<FileReader> r(new FileReader());
  r->fd_ = ::open(path.c_str(), O_RDONLY);
  if (r->fd_ < 0) { *err = "cannot open " + path + ": " + strerror(errno); return nullptr; }
  struct stat st;
  if (fstat(r->fd_, &st) != 0) { *err = "cannot stat " + path + ": " + strerror(errno); return nullptr; }
  r->size_ = (size_t)st.st_size;
  if (r->size_ < 40) { *err = path + ": too small to be a Lance file"; return nullptr; }
  void *m = mmap(nullptr, r->size_, PROT_READ, MAP_PRIVATE, r->fd_, 0);
  if (m == MAP_FAILED) { *err = "cannot mmap " + path + ": " + strerror(errno); return nullptr; }
  r->map_ = reinterpret_cast<const uint8_t *>(m);
  std::string perr;
  if (!r->parse(&perr)) { *err = path + ": " + perr; return nullptr; }
  return r;
}

bool FileReader::parse(std::string *err) {
  const uint8_t *foot = map_ + size_ - 40;
  if (memcmp(foot + 36, "LANC", 4) != 0) { *err = "bad magic (not a Lance file)"; return false; }
  uint64_t cmo, gbo;
  uint32_t ngb, ncol;
  memcpy(&cmo, foot + 8, 8); memcpy(&gbo, foot + 16, 8);
  memcpy(&ngb, foot + 24, 4); memcpy(&ncol, foot + 28, 4);
  memcpy(&major_, foot + 32, 2); memcpy(&minor_, foot + 34, 2);
  // version.rs try_from_major_minor: (0,3) and (2,0) are format 2.0; (0,<3) is the legacy format, (2,>=1) structural
  if (!((major_ == 0 && minor_ == 3) || (major_ == 2 && minor_ == 0))) {
    *err = "file format version " + std::to_string(major_) + "." + std::to_string(minor_) + " is not supported (only the 2.0 container)";
    return false;
  }
  const uint64_t body = size_ - 40;
  if (ngb == 0 || gbo > body || (uint64_t)ngb * 16 > body - gbo || cmo > body || (uint64_t)ncol * 16 > body - cmo) { *err = "corrupt footer"; return false; }
  global_.resize(ngb);
  for (uint32_t g = 0; g < ngb; ++g) {
    memcpy(&global_[g].first, map_ + gbo + 16 * (uint64_t)g, 8);
    memcpy(&global_[g].second, map_ + gbo + 16 * (uint64_t)g + 8, 8);
    if (global_[g].first > body || global_[g].second > body - global_[g].first) { *err = "global buffer out of bounds"; return false; }
  }
  // ---- file descriptor (global buffer 0): schema fields + metadata + row count
  {
    PbReader fd(map_ + global_[0].first, (size_t)global_[0].second);
    PB_LOOP(fd, f) {
      if (f.number == 2 && f.wire == 0) num_rows_ = f.value;
      if (f.number != 1 || f.wire != 2) continue;
      PbReader sc(f.data, f.size);
      bool oks = true;
      for (PbField g; sc.next(&g, &oks);) {
        if (g.number == 1 && g.wire == 2) {
          Field fl;
          PbReader fr(g.data, g.size);
          bool okf = true;
          for (PbField h; fr.next(&h, &okf);) {
            if (h.number == 2 && h.wire == 2) fl.name.assign(reinterpret_cast<const char *>(h.data), h.size);
            else if (h.number == 3 && h.wire == 0) fl.id = (int32_t)h.value;
            else if (h.number == 4 && h.wire == 0) fl.parent_id = (int32_t)h.value;
            else if (h.number == 5 && h.wire == 2) fl.logical_type.assign(reinterpret_cast<const char *>(h.data), h.size);
            else if (h.number == 6 && h.wire == 0) fl.nullable = h.value != 0;
          }
          if (!okf) { *err = "malformed schema field"; return false; }
          fields_.push_back(std::move(fl));
        } else if (g.number == 5 && g.wire == 2) {
          std::string k, v;
          PbReader kv(g.data, g.size);
          bool okk = true;
          for (PbField h; kv.next(&h, &okk);) {
            if (h.number == 1 && h.wire == 2) k.assign(reinterpret_cast<const char *>(h.data), h.size);
            else if (h.number == 2 && h.wire == 2) v.assign(reinterpret_cast<const char *>(h.data), h.size);
          }
          if (!okk) { *err = "malformed schema metadata"; return false; }
          metadata_.emplace_back(std::move(k), std::move(v));
        }
      }
      if (!oks) { *err = "malformed schema"; return false; }
    }
    if (!_ok) { *err = "malformed file descriptor"; return false; }
  }
  // ---- column metadata
  columns_.resize(ncol);
  for (uint32_t c = 0; c < ncol; ++c) {
    uint64_t pos, len;
    memcpy(&pos, map_ + cmo + 16 * (uint64_t)c, 8);
    memcpy(&len, map_ + cmo + 16 * (uint64_t)c + 8, 8);
    if (pos > body || len > body - pos) { *err = "column metadata out of bounds"; return false; }
    Column &col = columns_[c];
    PbReader cm(map_ + pos, (size_t)len);
    PB_LOOP(cm, f) {
      if (f.number != 2 || f.wire != 2) continue;
      Page pg;
      const uint8_t *enc = nullptr; size_t enc_n = 0;
      PbReader pr(f.data, f.size);
      bool okp = true;
      for (PbField g; pr.next(&g, &okp);) {
        if (g.number == 1) { if (!PbReader::append_varints(g, &pg.buffer_offsets)) okp = false; }
        else if (g.number == 2) { if (!PbReader::append_varints(g, &pg.buffer_sizes)) okp = false; }
        else if (g.number == 3 && g.wire == 0) pg.length = g.value;
        else if (g.number == 5 && g.wire == 0) pg.priority = g.value;
        else if (g.number == 4 && g.wire == 2) {   // Encoding{indirect|direct|none}
          PbReader er(g.data, g.size);
          bool oke = true;
          for (PbField h; er.next(&h, &oke);) {
            if (h.number == 2 && h.wire == 2) {        // DirectEncoding{encoding}
              PbReader dr(h.data, h.size);
              bool okd = true;
              for (PbField k; dr.next(&k, &okd);)
                if (k.number == 1 && k.wire == 2) { enc = k.data; enc_n = k.size; }
              if (!okd) oke = false;
            } else if (h.number == 1 && h.wire == 2) {  // DeferredEncoding{buffer_location, buffer_length}
              uint64_t loc = 0, ln = 0;
              PbReader dr(h.data, h.size);
              bool okd = true;
              for (PbField k; dr.next(&k, &okd);) {
                if (k.number == 1 && k.wire == 0) loc = k.value;
                else if (k.number == 2 && k.wire == 0) ln = k.value;
              }
              if (!okd || loc > body || ln > body - loc) oke = false;
              else { enc = map_ + loc; enc_n = (size_t)ln; }
            }
          }
          if (!oke) okp = false;
        }
        if (!okp) break;
      }
      if (!okp) { *err = "malformed page metadata (column " + std::to_string(c) + ")"; return false; }
      if (!enc) { *err = "page without an encoding (column " + std::to_string(c) + ")"; return false; }
      // google.protobuf.Any{type_url, value}
      const uint8_t *url = nullptr, *val = nullptr; size_t url_n = 0, val_n = 0;
      PbReader ar(enc, enc_n);
      bool oka = true;
      for (PbField g; ar.next(&g, &oka);) {
        if (g.number == 1 && g.wire == 2) { url = g.data; url_n = g.size; }
        else if (g.number == 2 && g.wire == 2) { val = g.data; val_n = g.size; }
      }
      if (!oka || !url) { *err = "malformed page encoding"; return false; }
      if (!ends_with(url, url_n, "lance.encodings.ArrayEncoding")) {
        *err = "page encoding " + std::string(reinterpret_cast<const char *>(url), url_n) + " is not a v2.0 ArrayEncoding";
        return false;
      }
      std::string eerr;
      pg.bits_per_value = 0; pg.dimension = 1;
      if (!decode_array_encoding(val, val_n, &pg.bits_per_value, &pg.dimension, &eerr)) { *err = "column " + std::to_string(c) + ": " + eerr; return false; }
      if (pg.bits_per_value == 0 || pg.bits_per_value % 8) { *err = "column " + std::to_string(c) + ": " + std::to_string(pg.bits_per_value) + "-bit values are not supported"; return false; }
      const uint64_t rb = (uint64_t)(pg.bits_per_value / 8) * pg.dimension;
      if (pg.buffer_offsets.size() != 1 || pg.buffer_sizes.size() != 1) { *err = "column " + std::to_string(c) + ": flat page with " + std::to_string(pg.buffer_offsets.size()) + " buffers"; return false; }
      // rb > 0 here; the length check divides instead of multiplying: two file-controlled u64 values must not wrap
      if (pg.buffer_offsets[0] > body || pg.buffer_sizes[0] > body - pg.buffer_offsets[0] || pg.length > pg.buffer_sizes[0] / rb) { *err = "column " + std::to_string(c) + ": page buffer out of bounds"; return false; }
      if (col.pages.empty()) col.row_bytes = (uint32_t)rb;
      else if (col.row_bytes != rb) { *err = "column " + std::to_string(c) + ": pages disagree on the value width"; return false; }
      if (__builtin_add_overflow(col.rows, pg.length, &col.rows)) { *err = "column " + std::to_string(c) + ": row count overflows"; return false; }
      col.pages.push_back(std::move(pg));
    }
    if (!_ok) { *err = "malformed column metadata"; return false; }
  }
  return true;
}

const std::string *FileReader::meta(const char *key) const {
  for (const auto &kv : metadata_)
    if (kv.first == key) return &kv.second;
  return nullptr;
}

int FileReader::column_of(const char *field_name) const {
  // v2.0: one column per leaf field, in schema order; the files read here have flat schemas (every field top-level)
  if (fields_.size() != columns_.size()) return -1;
  for (size_t i = 0; i < fields_.size(); ++i) {
    if (fields_[i].parent_id != -1) return -1;
    if (fields_[i].name == field_name) return (int)i;
  }
  return -1;
}

bool FileReader::global_buffer(size_t i, const uint8_t **p, size_t *n) const {
  if (i >= global_.size()) return false;
  *p = map_ + global_[i].first;
  *n = (size_t)global_[i].second;
  return true;
}

bool FileReader::read_rows(size_t c, uint64_t row0, uint64_t rows, void *dst, std::string *err) const {
  if (c >= columns_.size()) { *err = "no such column"; return false; }
  const Column &col = columns_[c];
  if (row0 > col.rows || rows > col.rows - row0) { *err = "row range out of bounds"; return false; }
  uint8_t *out = reinterpret_cast<uint8_t *>(dst);
  uint64_t start = 0;
  for (const Page &pg : col.pages) {
    const uint64_t lo = row0 > start ? row0 : start;
    const uint64_t hi = (row0 + rows) < (start + pg.length) ? (row0 + rows) : (start + pg.length);
    if (lo < hi)
      memcpy(out + (lo - row0) * col.row_bytes, map_ + pg.buffer_offsets[0] + (lo - start) * col.row_bytes, (size_t)((hi - lo) * col.row_bytes));
    start += pg.length;
  }
  return true;
}

const uint8_t *FileReader::contiguous(size_t c) const {
  if (c >= columns_.size() || columns_[c].pages.size() != 1) return nullptr;
  return map_ + columns_[c].pages[0].buffer_offsets[0];
}

// =====================================================================================================================
// writer
// =====================================================================================================================
FileWriter::~FileWriter() {
  if (f_) fclose(f_);
}

std::unique_ptr<FileWriter> FileWriter::create(const std::string &path, std::vector<Field> fields, std::string *err) {
  std::unique_ptr<FileWriter> w(new FileWriter());
  w->f_ = fopen(path.c_str(), "wb");
  if (!w->f_) { *err = "cannot create " + path + ": " + strerror(errno); return nullptr; }
  w->cols_.resize(fields.size());
  w->fields_ = std::move(fields);
  return w;
}

bool FileWriter::write(const void *p, size_t n) {
  if (failed_) return false;
  if (n && fwrite(p, 1, n, f_) != n) { failed_ = true; return false; }
  pos_ += n;
  return true;
}

bool FileWriter::write_padded(const void *p, size_t n) {
  static const uint8_t kPad[64] = {72, 72, 72, 72, 72, 72, 72, 72, 72, 72, 72, 72, 72, 72, 72, 72, 72, 72, 72, 72, 72, 72,
                                   72, 72, 72, 72, 72, 72, 72, 72, 72, 72, 72, 72, 72, 72, 72, 72, 72, 72, 72, 72, 72, 72,
                                   72, 72, 72, 72, 72, 72, 72, 72, 72, 72, 72, 72, 72, 72, 72, 72, 72, 72, 72, 72};
  if (!write(p, n)) return false;
  const size_t rem = n % 64;
  return rem == 0 || write(kPad, 64 - rem);
}

uint32_t FileWriter::add_global_buffer(const void *p, size_t n) {
  const uint64_t at = pos_;
  if (!write_padded(p, n)) return 0;
  global_.emplace_back(at, (uint64_t)n);
  return (uint32_t)global_.size();
}

void FileWriter::add_schema_metadata(const std::string &key, const std::string &value) {
  for (auto &kv : metadata_)
    if (kv.first == key) { kv.second = value; return; }
  metadata_.emplace_back(key, value);
}

void FileWriter::set_column(size_t c, const void *data, uint64_t rows, uint32_t bits, uint32_t dim) {
  if (c >= cols_.size()) { failed_ = true; return; }
  cols_[c] = Pending{reinterpret_cast<const uint8_t *>(data), rows, bits, dim};
}

namespace {
// Nullable{NoNull{values: inner}}
std::string wrap_no_nulls(const std::string &inner) {
  PbWriter nn; nn.bytes_field(1, inner);         // NoNull.values
  PbWriter nl; nl.bytes_field(1, nn.str());      // Nullable.no_nulls
  PbWriter ae; ae.bytes_field(2, nl.str());      // ArrayEncoding.nullable
  return ae.str();
}
std::string flat_encoding(uint32_t bits) {
  PbWriter fl;
  fl.varint_field(1, bits);
  fl.bytes_field(2, "", 0);                      // Buffer{index 0, type page}: all defaults
  PbWriter ae; ae.bytes_field(1, fl.str());      // ArrayEncoding.flat
  return wrap_no_nulls(ae.str());
}
std::string page_encoding(uint32_t bits, uint32_t dim) {
  std::string enc = flat_encoding(bits);
  if (dim > 1) {
    PbWriter fsl;
    fsl.varint_field(1, dim);
    fsl.bytes_field(2, enc);
    PbWriter ae; ae.bytes_field(3, fsl.str());   // ArrayEncoding.fixed_size_list
    enc = wrap_no_nulls(ae.str());
  }
  return enc;
}
std::string any_direct(const char *type_url, const std::string &msg) {
  PbWriter any; any.bytes_field(1, type_url, strlen(type_url)); any.bytes_field(2, msg);
  PbWriter direct; direct.bytes_field(1, any.str());   // DirectEncoding.encoding
  PbWriter enc; enc.bytes_field(2, direct.str());      // Encoding.direct
  return enc.str();
}
}  // namespace

bool FileWriter::finish(std::string *err) {
  uint64_t rows = cols_.empty() ? 0 : cols_[0].rows;
  for (const Pending &c : cols_)
    if (c.rows != rows || (c.rows && (!c.data || c.bits == 0 || c.bits % 8))) { *err = "FileWriter: columns not set consistently"; failed_ = true; }
  // ---- pages (writer.rs:202-231); at most kMaxPageBytes each, cut at row boundaries
  std::vector<std::string> col_meta(cols_.size());
  for (size_t c = 0; c < cols_.size() && !failed_; ++c) {
    const Pending &pc = cols_[c];
    PbWriter cm;
    {
      PbWriter empty; empty.bytes_field(1, "", 0);   // ColumnEncoding.values = Empty
      cm.bytes_field(1, any_direct("/lance.encodings.ColumnEncoding", empty.str()));
    }
    const uint64_t rb = (uint64_t)(pc.bits / 8) * pc.dim;
    uint64_t page_bytes = kMaxPageBytes;
    if (const char *e = getenv("LANCE_HIP_MAX_PAGE_BYTES")) {   // tests: force multi-page columns on small inputs
      const unsigned long long ov = strtoull(e, nullptr, 10);
      if (ov) page_bytes = ov;
    }
    const uint64_t per_page = rb ? (page_bytes / rb ? page_bytes / rb : 1) : 0;
    const std::string enc = pc.rows ? any_direct("/lance.encodings.ArrayEncoding", page_encoding(pc.bits, pc.dim)) : std::string();
    for (uint64_t r0 = 0; r0 < pc.rows; r0 += per_page) {
      const uint64_t nr = pc.rows - r0 < per_page ? pc.rows - r0 : per_page;
      const uint64_t off = pos_, sz = nr * rb;
      if (!write_padded(pc.data + r0 * rb, (size_t)sz)) break;
      PbWriter pg;
      pg.packed_varints(1, &off, 1);
      pg.packed_varints(2, &sz, 1);
      pg.varint_field(3, nr);
      pg.bytes_field(4, enc);
      if (r0) pg.varint_field(5, r0);
      cm.bytes_field(2, pg.str());
    }
    col_meta[c] = cm.str();
  }
  // ---- global buffer 0: the file descriptor (writer.rs:451-477; datatypes.rs:53-83 for the field message)
  PbWriter schema;
  for (const Field &fl : fields_) {
    PbWriter pf;
    pf.bytes_field(2, fl.name);
    if (fl.id) pf.varint_field(3, (uint64_t)(int64_t)fl.id);
    if (fl.parent_id) pf.varint_field(4, (uint64_t)(int64_t)fl.parent_id);
    pf.bytes_field(5, fl.logical_type);
    if (fl.nullable) pf.varint_field(6, 1);
    pf.varint_field(7, 1);   // Encoding::Plain
    schema.bytes_field(1, pf.str());
  }
  for (const auto &kv : metadata_) {
    PbWriter e;
    if (!kv.first.empty()) e.bytes_field(1, kv.first);
    if (!kv.second.empty()) e.bytes_field(2, kv.second);
    schema.bytes_field(5, e.str());
  }
  PbWriter fd;
  fd.bytes_field(1, schema.str());
  if (rows) fd.varint_field(2, rows);
  std::vector<std::pair<uint64_t, uint64_t>> gbo;
  gbo.emplace_back(pos_, (uint64_t)fd.str().size());
  write(fd.str().data(), fd.str().size());
  gbo.insert(gbo.end(), global_.begin(), global_.end());
  // ---- column metadatas, offset tables, footer (writer.rs:570-625)
  const uint64_t cm_start = pos_;
  std::vector<std::pair<uint64_t, uint64_t>> cmo;
  for (const std::string &m : col_meta) { cmo.emplace_back(pos_, (uint64_t)m.size()); write(m.data(), m.size()); }
  const uint64_t cmo_start = pos_;
  for (const auto &e : cmo) { write(&e.first, 8); write(&e.second, 8); }
  const uint64_t gbo_start = pos_;
  for (const auto &e : gbo) { write(&e.first, 8); write(&e.second, 8); }
  const uint32_t ngb = (uint32_t)gbo.size(), ncol = (uint32_t)cols_.size();
  const uint16_t major = 0, minor = 3;   // format 2.0 (writer.rs:553-561)
  write(&cm_start, 8); write(&cmo_start, 8); write(&gbo_start, 8);
  write(&ngb, 4); write(&ncol, 4); write(&major, 2); write(&minor, 2);
  write("LANC", 4);
  if (f_ && fclose(f_) != 0) failed_ = true;
  f_ = nullptr;
  if (failed_ && err->empty()) *err = std::string("FileWriter: write failed: ") + strerror(errno);
  return !failed_;
}

}  // namespace lancefile
