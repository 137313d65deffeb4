"""Index files: the `index.idx` + `auxiliary.idx` pair of an IVF_PQ / IVF_FLAT index directory.

Host-side mirror of the reference's on-disk contract (rust/lance/src/index/vector/builder.rs:938-1079 writes it,
lance-index/src/vector/storage.rs:182-243 + pq/storage.rs:52-144 + ivf/storage.rs:181-244 read it) over the native
reader/writer in liblance_hip.so (lance_amd/csrc/{lance_file,index_file}.cpp).  Parsing and writing need no GPU;
`DeviceIndex.load/save` (engine.py) move an index between files and HBM.
"""
import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib

_METRIC_NAMES = {_lib.L2: "l2", _lib.COSINE: "cosine", _lib.DOT: "dot"}


@dataclass
class IndexFileContents:
    """What the two files hold.  `codes` keeps the stored layout: inside partition p the bytes are
    [code_bytes][n_p] when `transposed` (pq/storage.rs:430-449), rows in `row_ids` order."""
    index_type: str                 # "IVF_PQ" | "IVF_FLAT"
    metric: str
    dtype: str                      # element type of the stored model tensors: "float32" | "float16"
    centroids: np.ndarray           # [nlist, d] float32 (f16 widened exactly)
    part_offsets: np.ndarray        # [nlist + 1] uint32
    row_ids: np.ndarray             # [n] uint64
    codebook: Optional[np.ndarray] = None   # [m, 2^nbits, d/m] float32
    codes: Optional[np.ndarray] = None      # [n * code_bytes] uint8
    vectors: Optional[np.ndarray] = None    # IVF_FLAT: [n, d]
    num_sub_vectors: int = 0
    nbits: int = 0
    transposed: bool = True
    loss: Optional[float] = None

    @property
    def code_bytes(self) -> int:
        return self.num_sub_vectors // 2 if self.nbits == 4 else self.num_sub_vectors

    def codes_row_major(self) -> np.ndarray:
        """[n, code_bytes] with the per-partition transpose undone (the un-transposed shuffle-buffer layout)."""
        cb, n = self.code_bytes, len(self.row_ids)
        out = np.empty((n, cb), np.uint8)
        for p in range(len(self.part_offsets) - 1):
            a, b = int(self.part_offsets[p]), int(self.part_offsets[p + 1])
            blk = self.codes[a * cb:b * cb]
            out[a:b] = blk.reshape(cb, b - a).T if self.transposed else blk.reshape(b - a, cb)
        return out

    def part_ids(self) -> np.ndarray:
        return np.repeat(np.arange(len(self.part_offsets) - 1, dtype=np.uint32), np.diff(self.part_offsets.astype(np.int64)))


def _arr(ptr, count, dtype):
    if not ptr or count == 0:
        return np.empty(0, dtype)
    buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=count).copy()


def read_index_files(index_dir, with_rows=True) -> IndexFileContents:
    """Parses and validates `<index_dir>/index.idx` + `auxiliary.idx` (no GPU needed).  with_rows=False returns the model
    and the partition offsets only (row ids / codes / vectors stay in the file mapping: what DeviceIndex.load needs)."""
    lib = _lib.load()
    h = C.c_void_p()
    _lib.check(lib.lance_hip_index_file_open(os.fspath(index_dir).encode(), C.byref(h)))
    try:
        v = _lib.IndexFileView()
        _lib.check(lib.lance_hip_index_file_get(h, C.byref(v)))
        n, d, nlist = int(v.n_rows), int(v.d), int(v.nlist)
        out = IndexFileContents(
            index_type="IVF_PQ" if v.index_type == _lib.IVF_PQ else "IVF_FLAT",
            metric=_METRIC_NAMES[v.metric],
            dtype="float16" if v.dtype == _lib.F16 else "float32",
            centroids=_arr(v.centroids, nlist * d, np.float32).reshape(nlist, d),
            part_offsets=_arr(v.part_offsets, nlist + 1, np.uint32),
            row_ids=_arr(v.row_ids, n if with_rows else 0, np.uint64),
            transposed=bool(v.transposed),
            loss=float(v.loss) if v.has_loss else None,
        )
        if v.index_type == _lib.IVF_PQ:
            m, nbits = int(v.m), int(v.nbits)
            out.num_sub_vectors, out.nbits = m, nbits
            out.codebook = _arr(v.codebook, (1 << nbits) * d, np.float32).reshape(m, 1 << nbits, d // m)
            out.codes = _arr(v.codes, n * out.code_bytes if with_rows else 0, np.uint8)
        else:
            nv = n if with_rows else 0
            out.vectors = _arr(v.vectors, nv * d, np.float16 if v.dtype == _lib.F16 else np.float32).reshape(nv, d)
        return out
    finally:
        lib.lance_hip_index_file_close(h)


def write_index_files(index_dir, c: IndexFileContents) -> None:
    """Writes the pair in the layout merge_partitions produces (format 2.0, transposed codes)."""
    lib = _lib.load()
    keep = []

    def ptr(a, dtype):
        if a is None:
            return None
        a = np.ascontiguousarray(a, dtype=dtype)
        keep.append(a)
        return a.ctypes.data_as(C.c_void_p)

    v = _lib.IndexFileView()
    v.index_type = _lib.IVF_PQ if c.index_type == "IVF_PQ" else _lib.IVF_FLAT
    v.metric = _lib.METRICS[c.metric]
    v.dtype = _lib.F16 if c.dtype == "float16" else _lib.F32
    v.nlist, v.d = c.centroids.shape
    v.m, v.nbits = c.num_sub_vectors, c.nbits
    v.n_rows = len(c.row_ids)
    v.transposed = int(c.transposed)
    v.has_loss, v.loss = (0, 0.0) if c.loss is None else (1, float(c.loss))
    v.centroids = ptr(c.centroids, np.float32)
    v.codebook = ptr(c.codebook, np.float32)
    v.part_offsets = ptr(c.part_offsets, np.uint32)
    v.row_ids = ptr(c.row_ids, np.uint64)
    v.codes = ptr(c.codes, np.uint8)
    v.vectors = ptr(c.vectors, np.float16 if c.dtype == "float16" else np.float32) if c.vectors is not None else None
    _lib.check(lib.lance_hip_index_file_write(os.fspath(index_dir).encode(), C.byref(v)))


def read_column(path, column, dtype, width=1) -> np.ndarray:
    """One uncompressed fixed-width top-level column of a format-2.0 Lance file (e.g. a data file's vector column) as
    [rows] or [rows, width] of dtype."""
    lib = _lib.load()
    rows, rb = C.c_uint64(), C.c_uint32()
    _lib.check(lib.lance_hip_file_read_column(os.fspath(path).encode(), column.encode(), None, 0, C.byref(rows), C.byref(rb)))
    dt = np.dtype(dtype)
    if rows.value and rb.value != dt.itemsize * width:
        raise ValueError(f"column {column!r} has {rb.value} bytes per row, not {dt.itemsize * width}")
    out = np.empty((rows.value, width) if width > 1 else (rows.value,), dt)
    _lib.check(lib.lance_hip_file_read_column(os.fspath(path).encode(), column.encode(), out.ctypes.data_as(C.c_void_p),
                                              out.nbytes, None, None))
    return out


def describe(index_dir) -> dict:
    """Summary of an index directory (host-only): what `python -m lance_amd.index_file <dir>` prints."""
    c = read_index_files(index_dir)
    lens = np.diff(c.part_offsets.astype(np.int64))
    out = {
        "index_type": c.index_type, "metric": c.metric, "model_dtype": c.dtype, "dimension": int(c.centroids.shape[1]),
        "num_partitions": int(c.centroids.shape[0]), "rows": int(len(c.row_ids)), "loss": c.loss,
        "partition_rows": {"min": int(lens.min()), "max": int(lens.max()), "mean": float(lens.mean()), "empty": int((lens == 0).sum())},
    }
    if c.index_type == "IVF_PQ":
        out.update({"num_sub_vectors": c.num_sub_vectors, "num_bits": c.nbits, "code_bytes_per_row": c.code_bytes,
                    "codes_transposed": c.transposed, "code_bytes_total": int(c.codes.size)})
    return out


if __name__ == "__main__":
    import json
    import sys
    if len(sys.argv) != 2:
        sys.exit("usage: python -m lance_amd.index_file <index directory holding index.idx [+ auxiliary.idx]>")
    print(json.dumps(describe(sys.argv[1]), indent=1))
