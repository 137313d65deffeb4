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

Everything here is host-side pyarrow; no GPU is involved.  Writing the buffers as legacy-format Lance files is one
`lance.file` call on a machine that has pylance; this image does not, so the tests stop at Arrow.
"""
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
