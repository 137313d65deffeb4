"""ctypes binding of liblance_hip.so (the C ABI in include/lance_hip.h).

The product path has no CPU fallback: if the HIP library is missing or no GPU is
visible, loading / context creation raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LANCE_HIP_LIB") or os.path.join(_HERE, "liblance_hip.so")   # override: kernel-variant A/B runs

OK, EINVAL, ERUNTIME, ENOTSUP, ENOMEM, EIO = 0, -1, -2, -3, -4, -5
IVF_PQ, IVF_FLAT = 0, 1
L2, COSINE, DOT = 0, 1, 2
F32, F16, I8 = 0, 1, 2
NONE = 0xFFFFFFFF
U64_MAX = 0xFFFFFFFFFFFFFFFF

METRICS = {"l2": L2, "L2": L2, "euclidean": L2, "cosine": COSINE, "dot": DOT, 0: L2, 1: COSINE, 2: DOT}

# every symbol include/lance_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "lance_hip_ctx_create", "lance_hip_ctx_destroy", "lance_hip_last_error", "lance_hip_version",
    "lance_hip_synchronize", "lance_hip_malloc", "lance_hip_free", "lance_hip_memcpy_h2d", "lance_hip_memcpy_d2h",
    "lance_hip_normalize", "lance_hip_assign", "lance_hip_kmeans_train", "lance_hip_kmeans_train_ex",
    "lance_hip_kmeans_estep_partial", "lance_hip_kmeans_shard_begin", "lance_hip_kmeans_shard_estep", "lance_hip_kmeans_shard_update",
    "lance_hip_kmeans_shard_end", "lance_hip_kmeans_init_indices", "lance_hip_kmeans_split",
    "lance_hip_kmeans_finalize", "lance_hip_pq_train", "lance_hip_residual", "lance_hip_pq_encode",
    "lance_hip_ivfpq_encode", "lance_hip_index_create", "lance_hip_index_from_storage", "lance_hip_index_destroy",
    "lance_hip_index_set_raw", "lance_hip_index_prewarm", "lance_hip_index_info", "lance_hip_index_export", "lance_hip_find_partitions",
    "lance_hip_pq_scan_topk", "lance_hip_ivfpq_search", "lance_hip_ivfpq_search_async", "lance_hip_ivfpq_search_range",
    "lance_hip_ivfpq_search_candidates",
    "lance_hip_search_stats", "lance_hip_ivfpq_search_filtered", "lance_hip_ivfpq_search_filtered_range",
    "lance_hip_flat_topk", "lance_hip_ivfflat_create", "lance_hip_ivfflat_search", "lance_hip_ivfflat_search_filtered",
    "lance_hip_index_file_open", "lance_hip_index_file_get", "lance_hip_index_file_close", "lance_hip_index_file_write",
    "lance_hip_index_load", "lance_hip_index_load_lists", "lance_hip_index_save", "lance_hip_file_read_column",
    "lance_hip_timing_enable", "lance_hip_timing_query", "lance_hip_ubench", "lance_hip_merge_topk",
    "lance_hip_shuffle_buffer_write",
    "lance_hip_comm_unique_id", "lance_hip_comm_create", "lance_hip_comm_adopt", "lance_hip_comm_from_callback", "lance_hip_comm_destroy", "lance_hip_kmeans_train_sharded", "lance_hip_kmeans_train_sharded_x",
    "lance_hip_kmeans_shard_estep_x",
]


# lance_hip_allreduce_fn (include/lance_hip.h): int fn(void *user, void *buf, uint64_t count, int dtype, int op, void *stream)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_void_p)


class IndexFileView(C.Structure):
    """lance_hip_index_file_view (include/lance_hip.h)."""
    _fields_ = [
        ("index_type", C.c_int), ("metric", C.c_int), ("dtype", C.c_int),
        ("d", C.c_uint32), ("nlist", C.c_uint32), ("m", C.c_uint32), ("nbits", C.c_uint32),
        ("n_rows", C.c_uint64), ("transposed", C.c_int), ("has_loss", C.c_int), ("loss", C.c_double),
        ("centroids", C.c_void_p), ("codebook", C.c_void_p), ("part_offsets", C.c_void_p),
        ("row_ids", C.c_void_p), ("codes", C.c_void_p), ("vectors", C.c_void_p),
    ]


class LanceHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"lance_hip error {code}: {msg}")
        self.code = code


_lib = None


def load():
    """Load liblance_hip.so.  torch is imported first so that both share one HIP runtime
    (torch bundles libamdhip64.so.7 under the same soname)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C lance_amd/csrc`). lance_amd has no CPU fallback."
        )
    try:
        import torch  # noqa: F401  (runtime sharing)
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH)
    vp, u64, u32, i32, f32, f64 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_float, C.c_double
    sig = {
        "lance_hip_ctx_create": (i32, [i32, vp, C.POINTER(vp)]),
        "lance_hip_ctx_destroy": (None, [vp]),
        "lance_hip_last_error": (C.c_char_p, []),
        "lance_hip_version": (C.c_char_p, []),
        "lance_hip_synchronize": (i32, [vp]),
        "lance_hip_malloc": (i32, [vp, C.c_size_t, C.POINTER(vp)]),
        "lance_hip_free": (i32, [vp, vp]),
        "lance_hip_memcpy_h2d": (i32, [vp, vp, vp, C.c_size_t]),
        "lance_hip_memcpy_d2h": (i32, [vp, vp, vp, C.c_size_t]),
        "lance_hip_normalize": (i32, [vp, i32, vp, u64, u32, vp]),
        "lance_hip_assign": (i32, [vp, i32, i32, vp, u64, u32, vp, u32, vp, vp, vp]),
        "lance_hip_kmeans_train": (i32, [vp, i32, i32, vp, u64, u32, u32, u32, f64, f32, vp, u64, vp,
                                         C.POINTER(f64), C.POINTER(u32)]),
        "lance_hip_kmeans_train_ex": (i32, [vp, i32, i32, vp, u64, u32, u32, u32, f64, f32, u32, vp, u64, vp,
                                            C.POINTER(f64), C.POINTER(u32), C.POINTER(u32)]),
        "lance_hip_kmeans_estep_partial": (i32, [vp, i32, i32, vp, u64, u32, vp, u32, vp, vp, vp, vp, C.POINTER(f64)]),
        "lance_hip_kmeans_finalize": (i32, [vp, i32, vp, u32, u32, vp]),
        "lance_hip_kmeans_shard_begin": (i32, [vp, u32, f32, u64, vp, vp]),
        "lance_hip_kmeans_shard_estep": (i32, [vp, i32, vp, u64, u32, vp, u32, vp, vp, vp, vp, vp]),
        "lance_hip_kmeans_shard_estep_x": (i32, [vp, i32, i32, vp, u64, u32, vp, u32, vp, vp, vp, vp, vp]),
        "lance_hip_kmeans_shard_update": (i32, [vp, vp, vp, vp, vp, vp, vp, u32, u32, u64, f32, f64, u32]),
        "lance_hip_kmeans_shard_end": (i32, [vp, vp, C.POINTER(f64), C.POINTER(u32), C.POINTER(i32)]),
        "lance_hip_kmeans_init_indices": (i32, [u64, u32, u64, vp]),
        "lance_hip_kmeans_split": (i32, [vp, i32, i32, vp, u64, u32, vp, u64, u32, u32, f64, f32, u64, vp, vp]),
        "lance_hip_pq_train": (i32, [vp, i32, vp, u64, u32, u32, u32, u32, u32, u64, vp, vp]),
        "lance_hip_residual": (i32, [vp, i32, vp, u64, u32, vp, vp, vp]),
        "lance_hip_pq_encode": (i32, [vp, i32, i32, vp, u64, u32, vp, u32, u32, vp]),
        "lance_hip_ivfpq_encode": (i32, [vp, i32, i32, vp, u64, u32, vp, u32, vp, u32, u32, vp, vp, C.POINTER(f64)]),
        "lance_hip_index_create": (i32, [vp, i32, i32, u32, vp, u32, vp, u32, u32, vp, vp, vp, u64, C.POINTER(vp)]),
        "lance_hip_index_from_storage": (i32, [vp, i32, i32, u32, vp, u32, vp, u32, u32, vp, vp, i32, vp, u64,
                                               C.POINTER(vp)]),
        "lance_hip_index_destroy": (None, [vp]),
        "lance_hip_index_set_raw": (i32, [vp, vp, u64]),
        "lance_hip_index_prewarm": (i32, [vp, vp]),
        "lance_hip_index_info": (i32, [vp, C.POINTER(u64), C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]),
        "lance_hip_index_export": (i32, [vp, vp, vp, vp, vp]),
        "lance_hip_find_partitions": (i32, [vp, i32, i32, vp, u32, u32, vp, u32, u32, vp, vp]),
        "lance_hip_pq_scan_topk": (i32, [vp, i32, i32, vp, u32, vp, u32, u32, vp, vp, u64, u32, i32, f32, f32, vp, vp,
                                         C.POINTER(u32)]),
        "lance_hip_ivfpq_search": (i32, [vp, vp, vp, u32, u32, u32, u32, vp, vp]),
        "lance_hip_ivfpq_search_async": (i32, [vp, vp, vp, u32, u32, u32, u32, vp, vp]),
        "lance_hip_ivfpq_search_range": (i32, [vp, vp, vp, u32, u32, u32, u32, f32, f32, vp, vp]),
        "lance_hip_ivfpq_search_candidates": (i32, [vp, vp, vp, u32, u32, u32, vp, vp, vp]),
        "lance_hip_search_stats": (i32, [vp, C.POINTER(u32)]),
        "lance_hip_ivfpq_search_filtered": (i32, [vp, vp, vp, u32, u32, u32, u32, vp, u64, vp, vp]),
        "lance_hip_ivfpq_search_filtered_range": (i32, [vp, vp, vp, u32, u32, u32, u32, vp, u64, f32, f32, vp, vp]),
        "lance_hip_flat_topk": (i32, [vp, i32, i32, vp, vp, u64, u32, vp, u32, u32, vp, vp]),
        "lance_hip_ivfflat_create": (i32, [vp, i32, i32, u32, vp, u32, vp, vp, vp, u64, C.POINTER(vp)]),
        "lance_hip_ivfflat_search": (i32, [vp, vp, vp, u32, u32, u32, vp, vp]),
        "lance_hip_ivfflat_search_filtered": (i32, [vp, vp, vp, u32, u32, u32, vp, u64, vp, vp]),
        "lance_hip_index_file_open": (i32, [C.c_char_p, C.POINTER(vp)]),
        "lance_hip_index_file_get": (i32, [vp, C.POINTER(IndexFileView)]),
        "lance_hip_index_file_close": (None, [vp]),
        "lance_hip_index_file_write": (i32, [C.c_char_p, C.POINTER(IndexFileView)]),
        "lance_hip_index_load": (i32, [vp, C.c_char_p, i32, C.POINTER(vp)]),
        "lance_hip_index_load_lists": (i32, [vp, C.c_char_p, i32, u32, u32, C.POINTER(vp)]),
        "lance_hip_index_save": (i32, [vp, vp, C.c_char_p, i32, f64]),
        "lance_hip_file_read_column": (i32, [C.c_char_p, C.c_char_p, vp, u64, C.POINTER(u64), C.POINTER(u32)]),
        "lance_hip_timing_enable": (i32, [vp, i32]),
        "lance_hip_timing_query": (i32, [vp, C.c_char_p, C.POINTER(f64), C.POINTER(u64)]),
        "lance_hip_ubench": (i32, [vp, i32, C.POINTER(f64)]),
        "lance_hip_merge_topk": (i32, [vp, vp, vp, vp, u32, u32, u32, u32, vp, vp]),
        "lance_hip_shuffle_buffer_write": (i32, [C.c_char_p, vp, vp, vp, u64, u32, C.POINTER(u64)]),
        "lance_hip_comm_unique_id": (i32, [C.c_char_p]),
        "lance_hip_comm_create": (i32, [vp, C.c_char_p, i32, i32, C.POINTER(vp)]),
        "lance_hip_comm_adopt": (i32, [vp, i32, i32, C.POINTER(vp)]),
        "lance_hip_comm_from_callback": (i32, [ALLREDUCE_FN, vp, i32, i32, C.POINTER(vp)]),
        "lance_hip_comm_destroy": (None, [vp]),
        "lance_hip_kmeans_train_sharded": (i32, [vp, vp, i32, vp, u64, u32, u32, u64, u32, f64, f32, u64, vp, C.POINTER(f64), C.POINTER(u32)]),
        "lance_hip_kmeans_train_sharded_x": (i32, [vp, vp, i32, i32, vp, u64, u32, u32, u64, u32, f64, f32, u64, vp, C.POINTER(f64), C.POINTER(u32)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code):
    if code != OK:
        raise LanceHipError(code, load().lance_hip_last_error().decode("utf-8", "replace"))
