"""ctypes/numpy front-end of the CPU oracle (oracle/lance_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of lance_oracle.c.  Importable from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from
lance_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblance_oracle.so")

L2, COSINE, DOT = 0, 1, 2
NONE = 0xFFFFFFFF
_METRICS = {"l2": L2, "L2": L2, "cosine": COSINE, "dot": DOT, 0: 0, 1: 1, 2: 2}


def build(force=False):
    src = os.path.join(_HERE, "lance_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liblance_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def usable_cpus():
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota.  A container that sees 128+
    cores behind an 8-core quota would otherwise run every OpenMP loop with 128 threads on 8 cores."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                      # cgroup v2: "<quota> <period>" or "max <period>"
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:      # cgroup v1
                quota = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = int(f.read())
            if quota > 0:
                n = min(n, max(1, int(quota / period + 0.5)))
        except Exception:
            pass
    return max(1, n)


os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")    # idle OpenMP workers sleep instead of spinning between short calls


def use_native_build():
    """bench.py cpu_baseline: rebuild the oracle for the host it is timed on (-march=native) and switch to it.
    Same arithmetic (-ffp-contract=off), only the vector width of the compiler's code changes.  -> True if active."""
    global _SO, _lib
    out = os.path.join(_HERE, "liblance_oracle_native.so")
    try:
        subprocess.check_call(["make", "-C", _HERE, "native", "MARCH=native", "OUT=liblance_oracle_native.so"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except Exception:
        return False
    _SO, _lib = out, None
    return True


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_l2_f32.restype = C.c_float
        _lib.orc_l2_f16.restype = C.c_float
        _lib.orc_l2_u8.restype = C.c_float
        _lib.orc_dot_f32.restype = C.c_float
        _lib.orc_dot_f16.restype = C.c_float
        _lib.orc_norm_l2_f32.restype = C.c_float
        _lib.orc_cosine_f32.restype = C.c_float
        _lib.orc_cosine_f32.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_size_t]
        _lib.orc_dot32_f32.restype = C.c_float
        _lib.orc_norm_l2_32_f32.restype = C.c_float
        _lib.orc_cosine_scalar32_f32.restype = C.c_float
        _lib.orc_cosine_scalar32_f32.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_size_t]
        _lib.orc_f16_to_f32.restype = C.c_float
        _lib.orc_f32_to_f16.restype = C.c_uint16
        _lib.orc_f32_to_f16.argtypes = [C.c_float]
        _lib.orc_heap_topk.restype = C.c_size_t
        _lib.orc_sort_fetch.restype = C.c_size_t
        _lib.orc_partition_layout.restype = C.c_size_t
        _lib.orc_kmeans_train_f32.restype = C.c_int
        _lib.orc_kmeans_train_x.restype = C.c_int
        _lib.orc_round_f16.restype = C.c_float
        _lib.orc_round_f16.argtypes = [C.c_float]
        _lib.orc_kmeans_train_hierarchical_f32.restype = C.c_size_t
        _lib.orc_kmeans_train_hierarchical_x.restype = C.c_size_t
        _lib.orc_set_threads(min(int(_lib.orc_num_threads()), usable_cpus()))
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _m(metric):
    return _METRICS[metric]


DOT_H, COSINE_H = 3, 4      # lance_oracle.c ORC_DOT_H / ORC_COSINE_H: the metric on a Float16 column (half::f16's own arms)


def _is_f16(*arrays):
    return any(a is not None and np.asarray(a).dtype == np.float16 for a in arrays)


def _mh(metric, f16):
    """metric code; on a float16 column dot / cosine become the f16 variants (32-lane dot_scalar / norm_l2_impl, scalar
    cosine, half-precision normalize)"""
    m = metric if isinstance(metric, (int, np.integer)) else _m(metric)
    if f16 and m == _METRICS["dot"]:
        return DOT_H
    if f16 and m == _METRICS["cosine"]:
        return COSINE_H
    return int(m)


def set_threads(n):
    lib().orc_set_threads(int(n))


def num_threads():
    return int(lib().orc_num_threads())


# ---- distances ---------------------------------------------------------------
def l2(x, y):
    x = np.ascontiguousarray(x); y = np.ascontiguousarray(y)
    if x.dtype == np.float16:
        return float(lib().orc_l2_f16(_p(x.view(np.uint16)), _p(y.view(np.uint16)), C.c_size_t(x.size)))
    if x.dtype == np.uint8:
        return float(lib().orc_l2_u8(_p(x), _p(y), C.c_size_t(x.size)))
    x = _f32(x); y = _f32(y)
    return float(lib().orc_l2_f32(_p(x), _p(y), C.c_size_t(x.size)))


def dot(x, y):
    x = np.ascontiguousarray(x); y = np.ascontiguousarray(y)
    if x.dtype == np.float16:
        return float(lib().orc_dot_f16(_p(x.view(np.uint16)), _p(y.view(np.uint16)), C.c_size_t(x.size)))
    x = _f32(x); y = _f32(y)
    return float(lib().orc_dot_f32(_p(x), _p(y), C.c_size_t(x.size)))


def norm_l2(x):
    h = _is_f16(x)
    x = _f32(x)
    if h:       # norm_l2_impl::<f16, f32, 32>
        return float(lib().orc_norm_l2_32_f32(_p(x), C.c_size_t(x.size)))
    return float(lib().orc_norm_l2_f32(_p(x), C.c_size_t(x.size)))


def cosine(x, y):
    h = _is_f16(x, y)
    xn = norm_l2(x)
    x = _f32(x); y = _f32(y)
    if h:       # f16: the trait default cosine_scalar over the 32-lane dot
        return float(lib().orc_cosine_scalar32_f32(_p(x), C.c_float(xn), _p(y), C.c_size_t(x.size)))
    return float(lib().orc_cosine_f32(_p(x), C.c_float(xn), _p(y), C.c_size_t(x.size)))


def distance_batch(metric, q, x):
    h = _is_f16(x)
    q = _f32(q); x = _f32(x)
    n, d = x.shape
    out = np.empty(n, np.float32)
    lib().orc_distance_batch_f32(_mh(metric, h), _p(q), _p(x), C.c_size_t(n), C.c_size_t(d), _p(out))
    return out


def normalize(x):
    """float16 in -> float16 out, normalised in half-precision arithmetic (do_normalize_fsl::<Float16Type>)"""
    h = _is_f16(x)
    x = _f32(x)
    x2 = x.reshape(-1, x.shape[-1])
    out = np.empty_like(x2)
    if h:
        lib().orc_normalize_h(_p(x2), C.c_size_t(x2.shape[0]), C.c_size_t(x2.shape[1]), _p(out))
        return out.reshape(x.shape).astype(np.float16)
    lib().orc_normalize_f32(_p(x2), C.c_size_t(x2.shape[0]), C.c_size_t(x2.shape[1]), _p(out))
    return out.reshape(x.shape)


def is_finite(x):
    x = _f32(x)
    out = np.empty(x.shape[0], np.uint8)
    lib().orc_is_finite_f32(_p(x), C.c_size_t(x.shape[0]), C.c_size_t(x.shape[1]), _p(out))
    return out.astype(bool)


# ---- assignment / k-means ------------------------------------------------------
def assign(x, centroids, metric="l2", bias=None):
    """-> (ids u32 [NONE = no partition], dists f32)"""
    x = np.ascontiguousarray(x); centroids = np.ascontiguousarray(centroids)
    n, d = x.shape
    k = centroids.shape[0]
    ids = np.empty(n, np.uint32); dists = np.empty(n, np.float32)
    if x.dtype == np.float16:
        assert bias is None
        lib().orc_assign_f16(_m(metric), _p(x.view(np.uint16)), C.c_size_t(n), C.c_size_t(d),
                             _p(centroids.astype(np.float16).view(np.uint16)), C.c_size_t(k), _p(ids), _p(dists))
        return ids, dists
    x = _f32(x); centroids = _f32(centroids)
    b = None if bias is None else _f32(bias)
    lib().orc_assign_f32(_m(metric), _p(x), C.c_size_t(n), C.c_size_t(d), _p(centroids), C.c_size_t(k),
                         _p(b), _p(ids), _p(dists))
    return ids, dists


def kmeans_init_indices(n, k, seed):
    out = np.empty(k, np.uint64)
    lib().orc_kmeans_init_indices(C.c_uint64(n), C.c_uint32(k), C.c_uint64(seed), _p(out))
    return out


def kmeans_train(x, k, max_iters=50, tol=1e-4, balance_factor=0.0, init=None, seed=0, metric="l2"):
    """KMeans::train_kmeans on exactly the rows given (caller applies caps).
    float16 input -> the Float16Type instantiation (f16 M-step arithmetic), centroids returned as float16.
    -> (centroids [k,d], loss, iters, cluster_sizes)"""
    f16 = np.asarray(x).dtype == np.float16
    x = _f32(x)
    n, d = x.shape
    cent = np.empty((k, d), np.float32)
    loss = C.c_double(0)
    sizes = np.empty(k, np.uint64)
    init_a = None if init is None else _f32(init)
    it = lib().orc_kmeans_train_x(_m(metric), _p(x), C.c_size_t(n), C.c_size_t(d), C.c_size_t(k),
                                  C.c_uint32(max_iters), C.c_double(tol), C.c_float(balance_factor),
                                  _p(init_a), C.c_uint64(seed), _p(cent), C.byref(loss), _p(sizes), C.c_int(int(f16)))
    return (cent.astype(np.float16) if f16 else cent), loss.value, int(it), sizes


def kmeans_train_hierarchical(x, k, max_iters=50, tol=1e-4, balance_factor_scaled=0.0, hierarchical_k=16, seed=0, metric="l2"):
    """train_hierarchical_kmeans (kmeans.rs:746-1003) -> centroids [n_clusters, d].  float16 input -> the Float16Type
    instantiation (f16 M-step in every inner k-means), centroids returned as float16."""
    f16 = np.asarray(x).dtype == np.float16
    x = _f32(x)
    n, d = x.shape
    cent = np.zeros((k, d), np.float32)
    got = lib().orc_kmeans_train_hierarchical_x(_m(metric), _p(x), C.c_size_t(n), C.c_size_t(d), C.c_size_t(k), C.c_uint32(max_iters),
                                                C.c_double(tol), C.c_float(balance_factor_scaled), C.c_size_t(hierarchical_k),
                                                C.c_uint64(seed), _p(cent), C.c_int(int(f16)))
    return cent[:got].astype(np.float16) if f16 else cent[:got]


def kmeans_split(x, rows, k, max_iters=50, tol=1e-4, balance_factor_scaled=0.0, seed=0, metric="l2"):
    """One split of the hierarchical trainer (orc_kmeans_split_x): k-means with k centroids over x[rows] (rows None: all), then the
    membership of those rows.  float16 x -> the Float16Type instantiation.  -> (centroids [k, d] f32, membership u32 [len(rows)])"""
    f16 = np.asarray(x).dtype == np.float16
    x = _f32(x)
    n, d = x.shape
    r = None if rows is None else np.ascontiguousarray(rows, np.uint32)
    nr = n if r is None else len(r)
    cent = np.empty((k, d), np.float32)
    mem = np.empty(nr, np.uint32)
    lib().orc_kmeans_split_x(_m(metric), _p(x), C.c_size_t(n), C.c_size_t(d), _p(r), C.c_size_t(nr), C.c_size_t(k), C.c_uint32(max_iters),
                             C.c_double(tol), C.c_float(balance_factor_scaled), C.c_uint64(seed), _p(cent), _p(mem), C.c_int(int(f16)))
    return cent, mem


def residual(x, centroids, part_ids):
    f16 = np.asarray(x).dtype == np.float16
    x = _f32(x); centroids = _f32(centroids)
    part_ids = np.ascontiguousarray(part_ids, np.uint32)
    out = np.empty_like(x)
    lib().orc_residual_x(_p(x), C.c_size_t(x.shape[0]), C.c_size_t(x.shape[1]), _p(centroids), _p(part_ids), _p(out), C.c_int(int(f16)))
    return out.astype(np.float16) if f16 else out


def divide_to_subvectors(x, m):
    x = _f32(x)
    n, d = x.shape
    out = np.empty((m, n, d // m), np.float32)
    lib().orc_divide_to_subvectors_f32(_p(x), C.c_size_t(n), C.c_size_t(d), C.c_size_t(m), _p(out))
    return out


def pq_train(resid, m, nbits=8, max_iters=50, sample_rate=256, seed=0):
    f16 = np.asarray(resid).dtype == np.float16
    resid = _f32(resid)
    n, d = resid.shape
    kc = 1 << nbits
    cb = np.empty((m, kc, d // m), np.float32)
    iters = np.zeros(m, np.int32)
    lib().orc_pq_train_x(_p(resid), C.c_size_t(n), C.c_size_t(d), C.c_size_t(m), C.c_uint32(nbits),
                         C.c_uint32(max_iters), C.c_size_t(sample_rate), C.c_uint64(seed), _p(cb), _p(iters), C.c_int(int(f16)))
    return (cb.astype(np.float16) if f16 else cb), iters


def pq_encode(x, codebook, metric="l2", nbits=8):
    metric = _mh(metric, _is_f16(x))       # Float16 sub-vectors under dot: dot_scalar::<f16, f32, 32> (pq.rs:143,165 -> dot.rs:91-102)
    x = _f32(x); codebook = _f32(codebook)
    n, d = x.shape
    m = codebook.shape[0]
    if nbits == 4:
        codes = np.empty((n, m // 2), np.uint8)
        lib().orc_pq_encode4_f32(C.c_int(metric), _p(x), C.c_size_t(n), C.c_size_t(d), _p(codebook), C.c_size_t(m), _p(codes))
        return codes
    codes = np.empty((n, m), np.uint8)
    lib().orc_pq_encode_f32(C.c_int(metric), _p(x), C.c_size_t(n), C.c_size_t(d), _p(codebook), C.c_size_t(m),
                            C.c_uint32(nbits), _p(codes))
    return codes


def transpose(codes):
    codes = np.ascontiguousarray(codes, np.uint8)
    n, m = codes.shape
    out = np.empty((m, n), np.uint8)
    lib().orc_transpose_u8(_p(codes), C.c_size_t(n), C.c_size_t(m), _p(out))
    return out


def build_lut(q, codebook, metric="l2", nbits=8):
    q = _f32(q); codebook = _f32(codebook)
    m = codebook.shape[0]
    lut = np.empty((m, 1 << nbits), np.float32)
    lib().orc_build_lut_f32(_m(metric), _p(q), C.c_size_t(q.size), _p(codebook), C.c_size_t(m), C.c_uint32(nbits), _p(lut))
    return lut


def pq_scan(lut, codes_t, metric="l2"):
    lut = _f32(lut); codes_t = np.ascontiguousarray(codes_t, np.uint8)
    m, n_p = codes_t.shape
    out = np.empty(n_p, np.float32)
    lib().orc_pq_scan_f32(_m(metric), _p(lut), C.c_size_t(m), _p(codes_t), C.c_size_t(n_p), _p(out))
    return out


def pq_scan_rowmajor(lut, codes):
    lut = _f32(lut); codes = np.ascontiguousarray(codes, np.uint8)
    n_p, m = codes.shape
    out = np.empty(n_p, np.float32)
    lib().orc_pq_scan_rowmajor_f32(_p(lut), C.c_size_t(m), _p(codes), C.c_size_t(n_p), _p(out))
    return out


def pq_scan4(lut, codes_t, k_hint, metric="l2"):
    """compute_pq_distance_4bit over transposed packed codes [M/2][n]"""
    lut = _f32(lut); codes_t = np.ascontiguousarray(codes_t, np.uint8)
    mb, n_p = codes_t.shape
    out = np.empty(n_p, np.float32)
    lib().orc_pq_scan4_f32(_m(metric), _p(lut), C.c_size_t(mb * 2), _p(codes_t), C.c_size_t(n_p), C.c_size_t(k_hint), _p(out))
    return out


def sum_4bit_dist_table(codes, code_len, dist_table, n):
    codes = np.ascontiguousarray(codes, np.uint8); dist_table = np.ascontiguousarray(dist_table, np.uint8)
    dists = np.zeros(n, np.uint16)
    lib().orc_sum_4bit_dist_table(C.c_size_t(n), C.c_size_t(code_len), _p(codes), _p(dist_table), _p(dists))
    return dists


def heap_topk(dists, row_ids, k, lower=None, upper=None):
    dists = _f32(dists); row_ids = np.ascontiguousarray(row_ids, np.uint64)
    out_i = np.empty(max(k, 1), np.uint64); out_d = np.empty(max(k, 1), np.float32)
    has = lower is not None or upper is not None
    lo = np.float32(np.finfo(np.float32).min if lower is None else lower)
    hi = np.float32(np.finfo(np.float32).max if upper is None else upper)
    c = lib().orc_heap_topk(_p(dists), _p(row_ids), C.c_size_t(dists.size), C.c_size_t(k), C.c_int(int(has)),
                            C.c_float(lo), C.c_float(hi), _p(out_i), _p(out_d))
    return out_i[:c].copy(), out_d[:c].copy()


def sort_fetch(ids, dists, k):
    ids = np.array(ids, np.uint64); dists = np.array(dists, np.float32)
    c = lib().orc_sort_fetch(_p(ids), _p(dists), C.c_size_t(ids.size), C.c_size_t(k))
    return ids[:c].copy(), dists[:c].copy()


def find_partitions(q, centroids, nprobes, metric="l2"):
    h = _is_f16(q, centroids)
    metric = _mh(metric, h)
    q = _f32(q).reshape(-1, centroids.shape[1]); centroids = _f32(centroids)
    nq, d = q.shape
    nlist = centroids.shape[0]
    nprobes = min(nprobes, nlist)
    ids = np.empty((nq, nprobes), np.uint32); dists = np.empty((nq, nprobes), np.float32)
    lib().orc_find_partitions_f32(C.c_int(metric), _p(q), C.c_size_t(nq), C.c_size_t(d), _p(centroids), C.c_size_t(nlist),
                                  C.c_size_t(nprobes), _p(ids), _p(dists))
    return ids, dists


def flat_knn(x, q, k, metric="l2", row_ids=None):
    metric = _mh(metric, _is_f16(x))
    x = _f32(x); q = _f32(q).reshape(-1, x.shape[1])
    n, d = x.shape
    nq = q.shape[0]
    rid = None if row_ids is None else np.ascontiguousarray(row_ids, np.uint64)
    ids = np.empty((nq, k), np.uint64); dists = np.empty((nq, k), np.float32)
    lib().orc_flat_knn_f32(C.c_int(metric), _p(x), _p(rid), C.c_size_t(n), C.c_size_t(d), _p(q), C.c_size_t(nq),
                           C.c_size_t(k), _p(ids), _p(dists))
    return ids, dists


def partition_layout(part_ids, nlist):
    part_ids = np.ascontiguousarray(part_ids, np.uint32)
    n = part_ids.size
    offs = np.empty(nlist + 1, np.uint32); perm = np.empty(n, np.uint32)
    tot = lib().orc_partition_layout(_p(part_ids), C.c_size_t(n), C.c_size_t(nlist), _p(offs), _p(perm))
    return offs, perm[:tot].copy()


class IvfPqIndex:
    """Canonical CPU index (reference layout: per-partition transposed codes)."""

    def __init__(self, metric, centroids, codebook, part_offsets, codes_t, row_ids, f16=False, nbits=8):
        self.metric = _m(metric)
        self.nbits = nbits
        self.f16 = bool(f16) or np.asarray(centroids).dtype == np.float16
        self.centroids = _f32(centroids)
        self.codebook = _f32(codebook)
        self.part_offsets = np.ascontiguousarray(part_offsets, np.uint32)
        self.codes_t = np.ascontiguousarray(codes_t, np.uint8)
        self.row_ids = np.ascontiguousarray(row_ids, np.uint64)

    def search(self, queries, k, nprobes, refine=0, raw=None, prefilter=None, lower=None, upper=None):
        """prefilter: boolean array over row ids (True = selected), the RowIdMask of a prefiltered query
        (flat/index.rs:129-165); every selected row goes through DistCalculator::distance(id).
        lower / upper: distance range [lower, upper) applied inside every partition's heap (flat/index.rs:98-113)."""
        q = _f32(queries).reshape(-1, self.centroids.shape[1])
        nq, d = q.shape
        ids = np.empty((nq, k), np.uint64); dists = np.empty((nq, k), np.float32)
        r = None if raw is None else _f32(raw)
        if lower is not None or upper is not None:
            allow = None if prefilter is None else np.ascontiguousarray(prefilter, dtype=np.uint8)
            lo = np.float32(np.finfo(np.float32).min if lower is None else lower)
            hi = np.float32(np.finfo(np.float32).max if upper is None else upper)
            lib().orc_ivfpq_search_range(self.metric, _p(self.centroids), C.c_size_t(self.centroids.shape[0]), C.c_size_t(d),
                                         _p(self.codebook), C.c_size_t(self.codebook.shape[0]), C.c_uint32(self.nbits),
                                         _p(self.part_offsets), _p(self.codes_t), _p(self.row_ids), _p(q), C.c_size_t(nq),
                                         C.c_size_t(k), C.c_size_t(nprobes), _p(ids), _p(dists), C.c_int(int(self.f16)),
                                         _p(allow), C.c_size_t(0 if allow is None else allow.size), C.c_float(lo), C.c_float(hi),
                                         C.c_size_t(refine), _p(r))
            return ids, dists
        if prefilter is not None:
            allow = np.ascontiguousarray(prefilter, dtype=np.uint8)
            lib().orc_ivfpq_search_filtered(self.metric, _p(self.centroids), C.c_size_t(self.centroids.shape[0]), C.c_size_t(d),
                                            _p(self.codebook), C.c_size_t(self.codebook.shape[0]), C.c_uint32(self.nbits),
                                            _p(self.part_offsets), _p(self.codes_t), _p(self.row_ids), _p(q), C.c_size_t(nq),
                                            C.c_size_t(k), C.c_size_t(nprobes), C.c_size_t(refine), _p(r), _p(ids), _p(dists),
                                            C.c_int(int(self.f16)), _p(allow), C.c_size_t(allow.size))
            return ids, dists
        lib().orc_ivfpq_search_x2(self.metric, _p(self.centroids), C.c_size_t(self.centroids.shape[0]), C.c_size_t(d),
                                 _p(self.codebook), C.c_size_t(self.codebook.shape[0]), C.c_uint32(self.nbits), _p(self.part_offsets),
                                 _p(self.codes_t), _p(self.row_ids), _p(q), C.c_size_t(nq), C.c_size_t(k),
                                 C.c_size_t(nprobes), C.c_size_t(refine), _p(r), _p(ids), _p(dists), C.c_int(int(self.f16)))
        return ids, dists


def build_index(x, centroids, codebook, metric="l2", row_ids=None, nbits=8):
    """Transform chain of lance-index ivf.rs:188-236 + per-partition storage
    (builder.rs:685-846) in canonical stable row order:
    [normalise if cosine] -> keep finite -> assign -> residual (L2/cosine) -> PQ encode
    -> group by partition -> transpose each partition's codes."""
    f16 = np.asarray(x).dtype == np.float16
    xin = x
    x = _f32(x)
    m = _m(metric)
    if row_ids is None:
        row_ids = np.arange(x.shape[0], dtype=np.uint64)
    row_ids = np.asarray(row_ids, np.uint64)
    xs = _f32(normalize(x.astype(np.float16) if f16 else x)) if m == COSINE else x     # f16 rows: half-precision normalize
    keep = is_finite(xs)
    xs = xs[keep]; rid = row_ids[keep]
    sm = L2 if m == COSINE else m
    part, _ = assign(xs.astype(np.float16) if f16 else xs, centroids, sm)    # f16 rows: l2_scalar 16 lanes / dot_scalar 32 lanes
    res = residual(xs.astype(np.float16) if f16 else xs, centroids, np.where(part == NONE, 0, part)) if sm == L2 else xs
    # the quantizer is always BUILT with DistanceType::L2 (lance/src/index/vector/builder.rs:456 `Q::build(&training_data,
    # DistanceType::L2, ..)`), and ProductQuantizer::transform encodes with the quantizer's own distance type
    # (pq.rs:143,165): nearest codeword in L2 even for a dot index; only the query-side LUT uses the index metric
    codes = pq_encode(res, codebook, L2, nbits=nbits)
    nlist = centroids.shape[0]
    offs, perm = partition_layout(part, nlist)
    codes_sorted = codes[perm]
    mm = codes.shape[1]
    codes_t = np.empty(codes_sorted.size, np.uint8)
    for p in range(nlist):
        a, b = int(offs[p]), int(offs[p + 1])
        if b > a:
            codes_t[a * mm:b * mm] = transpose(codes_sorted[a:b]).ravel()
    idx = IvfPqIndex(m, centroids, codebook, offs, codes_t, rid[perm], f16=f16, nbits=nbits)
    idx.part_ids = part
    idx.codes_rowmajor = codes
    idx.perm = perm
    return idx


def ivfflat_search(x, centroids, queries, k, nprobes, metric="l2", row_ids=None):
    """IVF_FLAT: per probed partition FlatIndex::search over the raw vectors (flat/index.rs:82-177 -- a max-heap of
    k, the earlier-scanned row wins a tie; distances = distance_type.func()(query, vector), flat/storage.rs:345-402),
    then SortExec([_distance, _rowid]).fetch(k) (scanner.rs:3440-3468).  Rows are stored per partition in ascending
    input order; rows without a partition (non-finite) are dropped."""
    h = _is_f16(x)
    if h:       # Float16 column: rows, key and centroids are f16; distances are half::f16's (see ORC_DOT_H / ORC_COSINE_H)
        x = np.ascontiguousarray(x, np.float16); centroids = np.ascontiguousarray(centroids, np.float16)
        q = np.ascontiguousarray(queries, np.float16).reshape(-1, x.shape[1])
    else:
        x = _f32(x); centroids = _f32(centroids)
        q = _f32(queries).reshape(-1, x.shape[1])
    nlist = centroids.shape[0]
    rid = np.arange(x.shape[0], dtype=np.uint64) if row_ids is None else np.asarray(row_ids, np.uint64)
    coarse = metric
    if metric == "cosine":
        # IvfTransformer::new_flat (ivf.rs:147-175): NormalizeTransformer, then the partition transform in L2 -- the rows
        # are stored normalised; the query key is normalised by the ANN node (knn.rs:498) and find_partitions runs in L2
        # (ivf/v2.rs:455-465); the FlatIndex still scores with cosine_distance (ivf/v2.rs:405-411, flat/storage.rs:345-402)
        x = normalize(x); q = normalize(q); coarse = "l2"
    part, _ = assign(x, centroids, coarse)
    offs, perm = partition_layout(part, nlist)
    probes, _ = find_partitions(q, centroids, nprobes, coarse)
    out_i = np.full((q.shape[0], k), np.iinfo(np.uint64).max, np.uint64)
    out_d = np.full((q.shape[0], k), np.inf, np.float32)
    for qi in range(q.shape[0]):
        ci, cd = [], []
        for p in probes[qi]:
            rows = perm[int(offs[p]):int(offs[p + 1])]
            if len(rows) == 0:
                continue
            d = distance_batch(metric, q[qi], x[rows])
            hi, hd = heap_topk(d, rid[rows], k)
            ci.append(hi); cd.append(hd)
        if ci:
            si, sd = sort_fetch(np.concatenate(ci), np.concatenate(cd), k)
            out_i[qi, :len(si)] = si; out_d[qi, :len(sd)] = sd
    return out_i, out_d
