// lance_file.h -- host-side reader/writer of the Lance v2.0 file container, just wide enough for the
// vector-index files (index.idx / auxiliary.idx) and flat vector columns of data files.
//
// Format (reference protos/file2.proto:31-100, rust/lance-file/src/writer.rs:190-625):
//   [data buffers, each padded to 64 B] [file descriptor pb] [column metadata pbs] [CMO table] [GBO table] [footer 40 B]
// Page encodings understood: the v2.0 `ArrayEncoding` tree Nullable{NoNull{Flat}} and
// Nullable{NoNull{FixedSizeList{Nullable{NoNull{Flat}}}}} (protos/encodings_v2_0.proto) -- what the reference's
// FileWriter emits for u64 row ids, FSL<u8> PQ codes and FSL<f32|f16> vectors.  Anything else is refused with a message
// (no silent mis-decoding): compressed, bit-packed, nullable-with-nulls, v2.1 structural pages.
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace lancefile {

// ---- protobuf wire format (only what the messages above need) -------------------------------------------------------
struct PbField {
  uint32_t number = 0;
  uint32_t wire = 0;          // 0 varint, 1 fixed64, 2 length-delimited, 5 fixed32
  uint64_t value = 0;         // varint / fixed payload
  const uint8_t *data = nullptr;  // length-delimited payload
  size_t size = 0;
};

class PbReader {
 public:
  PbReader(const uint8_t *p, size_t n) : p_(p), end_(p + n) {}
  // false at end of message; sets *ok = false on malformed input
  bool next(PbField *f, bool *ok);
  static bool varint(const uint8_t *&p, const uint8_t *end, uint64_t *out);
  // repeated scalar, packed or not: appends the values of field `f`
  static bool append_varints(const PbField &f, std::vector<uint64_t> *out);

 private:
  const uint8_t *p_, *end_;
};

class PbWriter {
 public:
  void varint_field(uint32_t number, uint64_t v);      // always written (caller applies the proto3 "skip default" rule)
  void bytes_field(uint32_t number, const void *p, size_t n);
  void bytes_field(uint32_t number, const std::string &s) { bytes_field(number, s.data(), s.size()); }
  void packed_varints(uint32_t number, const uint64_t *v, size_t n);   // nothing when n == 0
  void fixed64_field(uint32_t number, uint64_t bits);
  const std::string &str() const { return buf_; }

 private:
  void raw_varint(uint64_t v);
  std::string buf_;
};

// ---- minimal JSON (schema metadata values are serde_json documents) --------------------------------------------------
struct Json {
  enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
  bool b = false;
  double num = 0.0;
  std::string str;
  std::vector<Json> arr;
  std::vector<std::pair<std::string, Json>> obj;
  const Json *get(const char *key) const;
  static bool parse(const std::string &text, Json *out, std::string *err);
};
std::string json_quote(const std::string &s);

// ---- schema ------------------------------------------------------------------------------------------------------
struct Field {
  std::string name;
  int32_t id = 0;
  int32_t parent_id = -1;
  std::string logical_type;   // "uint64", "fixed_size_list:uint8:16", "fixed_size_list:float:128", ...
  bool nullable = false;
};

// "fixed_size_list:<item>:<dim>" or a bare item type -> element byte width and list dimension (1 for scalars);
// false for types this reader does not handle
bool parse_logical_type(const std::string &t, std::string *item, uint32_t *dim, uint32_t *item_bytes);

struct Page {
  std::vector<uint64_t> buffer_offsets, buffer_sizes;
  uint64_t length = 0;     // rows
  uint64_t priority = 0;   // first row number
  uint32_t bits_per_value = 0;   // decoded from the encoding
  uint32_t dimension = 1;        // FSL dimension (1 for scalars)
};

struct Column {
  std::vector<Page> pages;
  uint64_t rows = 0;
  uint32_t row_bytes = 0;   // bits_per_value / 8 * dimension
};

// Read-only memory map of one Lance v2.0 file with its metadata decoded.
class FileReader {
 public:
  ~FileReader();
  static std::unique_ptr<FileReader> open(const std::string &path, std::string *err);
  uint64_t num_rows() const { return num_rows_; }
  uint16_t major() const { return major_; }
  uint16_t minor() const { return minor_; }
  const std::vector<Field> &fields() const { return fields_; }
  const std::vector<std::pair<std::string, std::string>> &metadata() const { return metadata_; }
  const std::string *meta(const char *key) const;
  size_t num_columns() const { return columns_.size(); }
  const Column &column(size_t i) const { return columns_[i]; }
  int column_of(const char *field_name) const;   // -1 if absent
  size_t num_global_buffers() const { return global_.size(); }
  // 0 = the file descriptor, 1.. = user buffers in add order (writer.rs:493-499)
  bool global_buffer(size_t i, const uint8_t **p, size_t *n) const;
  // copies rows [row0, row0+rows) of a column, rows*row_bytes bytes
  bool read_rows(size_t col, uint64_t row0, uint64_t rows, void *dst, std::string *err) const;
  // zero-copy view when the whole column is one page (nullptr otherwise)
  const uint8_t *contiguous(size_t col) const;

 private:
  bool parse(std::string *err);
  int fd_ = -1;
  const uint8_t *map_ = nullptr;
  size_t size_ = 0;
  uint64_t num_rows_ = 0;
  uint16_t major_ = 0, minor_ = 0;
  std::vector<Field> fields_;
  std::vector<std::pair<std::string, std::string>> metadata_;
  std::vector<Column> columns_;
  std::vector<std::pair<uint64_t, uint64_t>> global_;
};

// Sequential writer producing the layout of the reference's FileWriter (writer.rs:384-625): global buffers added before
// finish() land ahead of the pages, pages are written at finish(), 64 B alignment with the same pad byte.
class FileWriter {
 public:
  ~FileWriter();
  static std::unique_ptr<FileWriter> create(const std::string &path, std::vector<Field> fields, std::string *err);
  // returns the buffer's index (1-based), 0 on failure
  uint32_t add_global_buffer(const void *p, size_t n);
  void add_schema_metadata(const std::string &key, const std::string &value);
  // Column data is borrowed until finish(); every column must receive the same number of rows.  bits_per_value is the
  // element width, dimension the FSL width (1 = scalar).
  void set_column(size_t col, const void *data, uint64_t rows, uint32_t bits_per_value, uint32_t dimension);
  bool finish(std::string *err);
  static constexpr uint64_t kMaxPageBytes = 32ull << 20;   // encoder default max_page_bytes

 private:
  struct Pending { const uint8_t *data = nullptr; uint64_t rows = 0; uint32_t bits = 0, dim = 1; };
  bool write(const void *p, size_t n);
  bool write_padded(const void *p, size_t n);
  FILE *f_ = nullptr;
  uint64_t pos_ = 0;
  bool failed_ = false;
  std::vector<Field> fields_;
  std::vector<Pending> cols_;
  std::vector<std::pair<std::string, std::string>> metadata_;
  std::vector<std::pair<uint64_t, uint64_t>> global_;
};

}  // namespace lancefile
