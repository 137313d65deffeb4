"""Arrow forms of the artefacts pylance's accelerator seam consumes.

`Dataset.create_index(..., ivf_centroids=, pq_codebook=, precomputed_shuffle_buffers=)` lets Python hand a trained model
and pre-assigned rows to the Rust builder, which then skips its own training and transform:

  * ivf_centroids  -- RecordBatch whose first column is named `_ivf_centroids`: FixedSizeList<float>[d], nlist rows
                      (python/src/dataset.rs:3026-3051; cast to the column's element type there);
  * pq_codebook    -- RecordBatch whose first column is named `_pq_codebook`: FixedSizeList<float>[d / M] holding the
                      (M, 256, d/M) array flattened, i.e. M * 256 rows (python/python/lance/dataset.py:2928-2954;
                      Rust takes `.values()`, python/src/dataset.rs:3109-3118);
  * shuffle buffers -- batches with schema (row_id: uint64, __ivf_part_id: uint32, __pq_code: FixedSizeList<uint8>[M]),
                      codes row-major / un-transposed (python/python/lance/vector.py:659-665; consumed by
                      lance/src/index/vector/builder.rs:509-546).  Rows without a partition are dropped.

Everything here is host-side; no GPU is involved.  The shuffle buffers can be written as Arrow IPC (pyarrow) or, round 3, as
a Lance v2.0 FILE by the library's own writer (write_shuffle_buffers_lance -> lance_hip_shuffle_buffer_write) with the three
columns above; wrapping that file into the dataset directory the reference opens (manifest, version files) is one
`lance.write_dataset` call on a machine that has pylance -- this image does not.
"""
import os

import numpy as np
import pyarrow as pa

NONE = 0xFFFFFFFF


def ivf_centroids_batch(centroids):
    c = np.ascontiguousarray(centroids)
    if c.ndim != 2:
        raise ValueError(f"IVF centroids must be a 2-D (nlist, d) array, got {c.shape}")
    if c.dtype not in (np.float16, np.float32, np.float64):
        raise TypeError(f"IVF centroids must be floating point, got {c.dtype}")
    fsl = pa.FixedSizeListArray.from_arrays(pa.array(c.reshape(-1)), c.shape[1])
    return pa.RecordBatch.from_arrays([fsl], ["_ivf_centroids"])


def pq_codebook_batch(codebook):
    cb = np.ascontiguousarray(codebook)
    if cb.ndim != 3 or cb.shape[1] not in (16, 256):
        raise ValueError(f"PQ codebook must be 3D array: (sub_vectors, 256, dim), got {cb.shape}")
    if cb.dtype not in (np.float16, np.float32, np.float64):
        raise TypeError(f"PQ codebook must be floating number, got {cb.dtype}")
    fsl = pa.FixedSizeListArray.from_arrays(pa.array(cb.reshape(-1)), cb.shape[2])
    return pa.RecordBatch.from_arrays([fsl], ["_pq_codebook"])


def shuffle_buffer_schema(num_sub_vectors):
    return pa.schema([pa.field("row_id", pa.uint64()), pa.field("__ivf_part_id", pa.uint32()),
                      pa.field("__pq_code", pa.list_(pa.uint8(), list_size=num_sub_vectors))])


def shuffle_buffer_batches(row_ids, part_ids, codes, batch_size=10240):
    """yield RecordBatches of the shuffle-buffer schema; rows whose partition is NONE (non-finite vectors) are dropped"""
    rid = np.ascontiguousarray(row_ids, np.uint64)
    part = np.ascontiguousarray(part_ids).view(np.uint32) if np.asarray(part_ids).dtype == np.int32 else np.ascontiguousarray(part_ids, np.uint32)
    codes = np.ascontiguousarray(codes, np.uint8)
    if not (rid.shape[0] == part.shape[0] == codes.shape[0]):
        raise ValueError("row ids, partition ids and codes must have the same number of rows")
    keep = part != NONE
    rid, part, codes = rid[keep], part[keep], codes[keep]
    m = codes.shape[1]
    schema = shuffle_buffer_schema(m)
    for lo in range(0, rid.shape[0], batch_size):
        hi = min(lo + batch_size, rid.shape[0])
        fsl = pa.FixedSizeListArray.from_arrays(pa.array(codes[lo:hi].reshape(-1)), m)
        yield pa.RecordBatch.from_arrays([pa.array(rid[lo:hi]), pa.array(part[lo:hi]), fsl], schema=schema)


def write_shuffle_buffers_ipc(path, row_ids, part_ids, codes, batch_size=10240):
    """Arrow IPC file with the shuffle-buffer schema (what a pylance-equipped host re-writes as a Lance file)."""
    m = np.asarray(codes).shape[1]
    with pa.OSFile(path, "wb") as sink, pa.ipc.new_file(sink, shuffle_buffer_schema(m)) as w:
        n = 0
        for b in shuffle_buffer_batches(row_ids, part_ids, codes, batch_size):
            w.write_batch(b)
            n += b.num_rows
    return n


def write_shuffle_buffers_lance(path, row_ids, part_ids, codes):
    """The same rows as ONE Lance v2.0 file written by the native writer (lance_hip_shuffle_buffer_write): columns row_id u64,
    __ivf_part_id u32, __pq_code fixed_size_list<u8>[M] -- the schema IvfIndexBuilder::shuffle_dataset reads back
    (lance/src/index/vector/builder.rs:509-546; python/lance/vector.py:659-665).  Rows without a partition are dropped.
    -> rows written.  A pylance host wraps the file into the dataset directory `precomputed_shuffle_buffers` points at
    (lance.write_dataset(lance.file.LanceFileReader(path).read_all().to_table(), uri, data_storage_version="legacy")); the
    manifest / version files of a dataset belong to the storage engine, which this package does not rebuild."""
    import ctypes as C
    from . import _lib
    part = np.ascontiguousarray(part_ids).view(np.uint32) if np.asarray(part_ids).dtype == np.int32 else np.ascontiguousarray(part_ids, np.uint32)
    codes = np.ascontiguousarray(codes, np.uint8)
    rid = None if row_ids is None else np.ascontiguousarray(row_ids, np.uint64)
    if codes.ndim != 2 or codes.shape[0] != part.shape[0] or (rid is not None and rid.shape[0] != part.shape[0]):
        raise ValueError("row ids, partition ids and codes must have the same number of rows")
    written = C.c_uint64(0)
    _lib.check(_lib.load().lance_hip_shuffle_buffer_write(os.fsencode(path), None if rid is None else rid.ctypes.data_as(C.c_void_p),
                                                         part.ctypes.data_as(C.c_void_p), codes.ctypes.data_as(C.c_void_p),
                                                         part.shape[0], codes.shape[1], C.byref(written)))
    return int(written.value)


def read_shuffle_buffers_lance(path):
    """-> (row_ids u64 [n], part_ids u32 [n], codes u8 [n, M]) from a file written by write_shuffle_buffers_lance (or by the
    reference's FileWriter with the same three fixed-width columns), through the native reader."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    out = []
    for col, dt in (("row_id", np.uint64), ("__ivf_part_id", np.uint32), ("__pq_code", np.uint8)):
        rows, rb = C.c_uint64(0), C.c_uint32(0)
        _lib.check(lib.lance_hip_file_read_column(os.fsencode(path), col.encode(), None, 0, C.byref(rows), C.byref(rb)))
        buf = np.empty(rows.value * rb.value, np.uint8)
        _lib.check(lib.lance_hip_file_read_column(os.fsencode(path), col.encode(), buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(rows), C.byref(rb)))
        a = buf.view(dt)
        out.append(a.reshape(rows.value, -1) if col == "__pq_code" else a)
    return tuple(out)


def centroids_from_batch(batch):
    if batch.schema.field(0).name != "_ivf_centroids":
        raise ValueError("Expected '_ivf_centroids' as the first column name.")
    col = batch.column(0)
    d = col.type.list_size
    return np.asarray(col.values.to_numpy(zero_copy_only=False)).reshape(-1, d)


def codebook_from_batch(batch, num_sub_vectors):
    if batch.schema.field(0).name != "_pq_codebook":
        raise ValueError("Expected '_pq_codebook' as the first column name.")
    col = batch.column(0)
    sd = col.type.list_size
    flat = np.asarray(col.values.to_numpy(zero_copy_only=False))
    return flat.reshape(num_sub_vectors, -1, sd)
