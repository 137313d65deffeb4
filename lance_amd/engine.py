"""Thin Python host layer over the C ABI: device memory and streams come from torch
(plumbing), every computation is a call into liblance_hip.so.
"""
import ctypes as C
import functools
import os

import numpy as np
import torch

from . import _lib
from ._lib import METRICS, check


def _dev():
    if not torch.cuda.is_available():
        raise RuntimeError("lance_amd needs an MI355X: no HIP device is visible and there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def to_device(a, dtype=None):
    """numpy / torch (cpu or cuda) -> contiguous cuda tensor"""
    if isinstance(a, torch.Tensor):
        t = a
    else:
        arr = np.ascontiguousarray(a)
        if arr.dtype == np.uint64:  # torch has limited uint64 support: move the bits
            t = torch.from_numpy(arr.view(np.int64))
        elif arr.dtype == np.uint32:
            t = torch.from_numpy(arr.view(np.int32))
        else:
            t = torch.from_numpy(arr)
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t.to(_dev()).contiguous()


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _vec(a):
    """vector data -> (cuda tensor in its own element type, dtype code); float16 stays float16"""
    t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
    if t.dtype == torch.float16:
        return t.to(_dev()).contiguous(), _lib.F16
    if t.dtype == torch.int8:      # Int8 columns: data stays int8, the model (centroids, codebook) is float32
        return t.to(_dev()).contiguous(), _lib.I8
    return t.to(torch.float32).to(_dev()).contiguous(), _lib.F32


def _nbits(codebook):
    """num_bits from the codebook shape (M, 2^nbits, d/M)"""
    kc = int(codebook.shape[1])
    if kc not in (16, 256):
        raise ValueError(f"codebook must have 16 or 256 centroids per sub-vector, got {kc}")
    return 4 if kc == 16 else 8


def _like(a, ref):
    """data arrays (queries, raw vectors) are cast to the element type of the vectors"""
    t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
    return t.to(ref.dtype).to(_dev()).contiguous()


def _model(a, ref):
    """model arrays (centroids, codebook) take the element type of the vectors, float32 for int8 vectors"""
    t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
    return t.to(torch.float32 if ref.dtype == torch.int8 else ref.dtype).to(_dev()).contiguous()


_DT = {"float32": (torch.float32, 0), "float16": (torch.float16, 1), "int8": (torch.int8, 2)}


def _on_engine_device(fn):
    """Every tensor a method allocates or moves must live on the GPU the context was created on (two Engines in one process
    may sit on different GPUs): run the method with that device current."""
    @functools.wraps(fn)
    def wrapper(self, *a, **kw):
        eng = self if isinstance(self, Engine) else getattr(self, "engine", None)
        if eng is None or not torch.cuda.is_available() or torch.cuda.current_device() == eng.device:
            return fn(self, *a, **kw)
        with torch.cuda.device(eng.device):
            return fn(self, *a, **kw)
    return wrapper


def _wrap_methods(cls, skip=()):
    for name, fn in list(vars(cls).items()):
        if callable(fn) and not name.startswith("__") and name not in skip and not isinstance(fn, (classmethod, staticmethod)):
            setattr(cls, name, _on_engine_device(fn))
    return cls


class Engine:
    """One context (stream + scratch arena) on the current device."""

    def __init__(self, device=None, use_torch_stream=False):
        self.lib = _lib.load()
        if device is None:
            device = _dev().index
        self.device = device
        self.use_torch_stream = bool(use_torch_stream)
        stream = None
        if use_torch_stream:
            stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        h = C.c_void_p()
        check(self.lib.lance_hip_ctx_create(device, stream, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.lance_hip_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        check(self.lib.lance_hip_synchronize(self.h))

    # ---- building blocks ---------------------------------------------------------
    def normalize(self, x):
        """normalize_fsl (kernels.rs:141-186): float32 rows in f32 arithmetic; float16 rows stay float16 and are normalised in
        half-precision arithmetic, as do_normalize_fsl::<Float16Type> does"""
        t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))
        f16 = t.dtype == torch.float16
        x = t.to(_dev()).contiguous() if f16 else to_device(x, torch.float32)
        out = torch.empty_like(x)
        n, d = x.shape
        torch.cuda.synchronize()
        check(self.lib.lance_hip_normalize(self.h, _lib.F16 if f16 else _lib.F32, _ptr(x), n, d, _ptr(out)))
        return out

    def assign(self, x, centroids, metric="l2", bias=None):
        x, dt = _vec(x); centroids = _model(centroids, x)
        n, d = x.shape
        k = centroids.shape[0]
        ids = torch.empty(n, dtype=torch.int32, device=x.device)
        dists = torch.empty(n, dtype=torch.float32, device=x.device)
        b = None if bias is None else to_device(bias, torch.float32)
        torch.cuda.synchronize()
        check(self.lib.lance_hip_assign(self.h, dt, METRICS[metric], _ptr(x), n, d, _ptr(centroids), k, _ptr(b),
                                        _ptr(ids), _ptr(dists)))
        return ids, dists

    def kmeans_train(self, x, k, max_iters=50, tol=1e-4, balance_factor=0.0, init=None, seed=0, metric="l2", hierarchical_k=16):
        """KMeans::new_with_params: flat Lloyd for k <= 256 (or hierarchical_k <= 1), hierarchical otherwise."""
        x, dt = _vec(x)
        n, d = x.shape
        mdt = torch.float32 if x.dtype == torch.int8 else x.dtype
        cent = torch.zeros((k, d), dtype=mdt, device=x.device)
        init_t = None if init is None else _model(init, x)
        loss = C.c_double(0); iters = C.c_uint32(0); kout = C.c_uint32(0)
        torch.cuda.synchronize()
        check(self.lib.lance_hip_kmeans_train_ex(self.h, dt, METRICS[metric], _ptr(x), n, d, k, max_iters, tol,
                                                 balance_factor, hierarchical_k, _ptr(init_t), seed, _ptr(cent), C.byref(loss),
                                                 C.byref(iters), C.byref(kout)))
        return cent[: kout.value], loss.value, iters.value

    def kmeans_split(self, x, rows, k, max_iters=50, tol=1e-4, balance_factor_scaled=0.0, seed=0, metric="l2", f16_arith=False):
        """One split of the hierarchical trainer (lance_hip_kmeans_split): k-means with k centroids over the rows `rows` (ascending u32
        indices, None: all) of the device sample x (f32; an f16 / int8 sample is widened -- f16_arith: the values are binary16 and the
        M-step rounds like half::f16), then their membership.  No device-wide synchronisation: host threads run splits side by side.
        -> (centroids np.float32 [k, d], membership np.uint32 [len(rows)])"""
        f16 = bool(f16_arith)
        if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()):
            x, dt0 = _vec(x)
            f16 = f16 or x.dtype == torch.float16
            x = x.float().contiguous()          # (callers that split many times widen once themselves and pass f16_arith)
        dt = _lib.F16 if f16 else _lib.F32
        n, d = x.shape
        r = None if rows is None else np.ascontiguousarray(rows, np.uint32)
        nr = n if r is None else len(r)
        cent = np.empty((k, d), np.float32)
        mem = np.empty(nr, np.uint32)
        check(self.lib.lance_hip_kmeans_split(self.h, dt, METRICS[metric], _ptr(x), n, d, None if r is None else r.ctypes.data_as(C.c_void_p), nr, k,
                                              max_iters, tol, float(balance_factor_scaled), seed, cent.ctypes.data_as(C.c_void_p),
                                              mem.ctypes.data_as(C.c_void_p)))
        return cent, mem

    def kmeans_estep_partial(self, x, centroids, metric="l2", bias=None):
        """-> (buf [k*d sums | k counts] f32, losses [k] f64, radius [k] f32) device tensors"""
        x = to_device(x, torch.float32); centroids = to_device(centroids, torch.float32)
        n, d = x.shape
        k = centroids.shape[0]
        buf = torch.empty(k * d + k, dtype=torch.float32, device=x.device)
        losses = torch.empty(k, dtype=torch.float64, device=x.device)
        radius = torch.empty(k, dtype=torch.float32, device=x.device)
        b = None if bias is None else to_device(bias, torch.float32)
        torch.cuda.synchronize()
        check(self.lib.lance_hip_kmeans_estep_partial(self.h, _lib.F32, METRICS[metric], _ptr(x), n, d, _ptr(centroids), k,
                                                      _ptr(b), _ptr(buf), _ptr(losses), _ptr(radius), None))
        return buf, losses, radius

    def kmeans_finalize(self, buf, k, d):
        cent = torch.empty((k, d), dtype=torch.float32, device=buf.device)
        torch.cuda.synchronize()
        check(self.lib.lance_hip_kmeans_finalize(self.h, _lib.F32, _ptr(buf), k, d, _ptr(cent)))
        return cent

    # ---- row-sharded Lloyd loop, enqueue-only (lance_hip_kmeans_shard_*): see lance_amd/dist.py -------------------------
    def kmeans_shard_begin(self, k, d, balance_factor_scaled, seed):
        """-> dict of the device buffers one sharded training run needs (state, bias, reduce buffers)"""
        dev = _dev()
        st = {"state": torch.zeros(128, dtype=torch.uint8, device=dev), "bias": torch.zeros(k, dtype=torch.float32, device=dev),
              "buf": torch.zeros(k * d + k, dtype=torch.float32, device=dev), "losses": torch.zeros(k, dtype=torch.float64, device=dev),
              "radius": torch.zeros(k, dtype=torch.float32, device=dev), "k": k, "d": d, "bfs": float(balance_factor_scaled)}
        check(self.lib.lance_hip_kmeans_shard_begin(self.h, k, st["bfs"], seed, _ptr(st["state"]), _ptr(st["bias"])))
        return st

    def kmeans_shard_estep(self, st, x_local, cent, metric="l2"):
        n = x_local.shape[0]
        check(self.lib.lance_hip_kmeans_shard_estep(self.h, METRICS[metric], _ptr(x_local) if n else None, n, st["d"], _ptr(cent), st["k"],
                                                    _ptr(st["bias"]) if st["bfs"] != 0 else None, _ptr(st["state"]), _ptr(st["buf"]),
                                                    _ptr(st["losses"]), _ptr(st["radius"])))

    def kmeans_shard_update(self, st, cent, n_total, tol, it):
        check(self.lib.lance_hip_kmeans_shard_update(self.h, _ptr(st["state"]), _ptr(st["buf"]), _ptr(st["losses"]), _ptr(st["radius"]),
                                                     _ptr(cent), _ptr(st["bias"]), st["k"], st["d"], n_total, st["bfs"], tol, it))

    def kmeans_shard_end(self, st):
        loss = C.c_double(0); iters = C.c_uint32(0); active = C.c_int(0)
        check(self.lib.lance_hip_kmeans_shard_end(self.h, _ptr(st["state"]), C.byref(loss), C.byref(iters), C.byref(active)))
        return loss.value, iters.value, bool(active.value)

    # ---- the sharded Lloyd loop with its collectives behind the C ABI (comm.cpp): for hosts without torch.distributed ----
    def comm_unique_id(self):
        """ncclGetUniqueId through the library: 128 bytes rank 0 ships to the other ranks"""
        buf = C.create_string_buffer(128)
        check(self.lib.lance_hip_comm_unique_id(buf))
        return buf.raw

    def comm_create(self, unique_id, nranks, rank):
        """ncclCommInitRank on this engine's device -> opaque communicator handle (comm_destroy releases it)"""
        h = C.c_void_p()
        check(self.lib.lance_hip_comm_create(self.h, C.create_string_buffer(bytes(unique_id), 128), nranks, rank, C.byref(h)))
        return h

    def comm_from_callback(self, fn, nranks, rank):
        """lance_hip_comm_from_callback: the sharded trainer's exchanges through a host transport.  fn(buf_ptr, count, dtype, op,
        stream_ptr) -> 0 reduces `count` elements (dtype 0 = f32, 1 = f64; op 0 = sum, 1 = max) at DEVICE address buf_ptr in place
        across the ranks.  The ctypes thunk is kept alive on the returned handle's engine."""
        def guarded(user, buf, count, dtype, op, stream):
            # an exception must reach the trainer as a failed exchange: ctypes would print the traceback and hand 0 ("reduced") to C,
            # and the ranks would go on with unreduced buffers
            try:
                return int(fn(buf, count, dtype, op, stream))
            except BaseException:
                import traceback
                traceback.print_exc()
                return 1
        thunk = _lib.ALLREDUCE_FN(guarded)
        h = C.c_void_p()
        check(self.lib.lance_hip_comm_from_callback(thunk, None, nranks, rank, C.byref(h)))
        if not hasattr(self, "_comm_thunks"):
            self._comm_thunks = {}
        self._comm_thunks[h.value] = thunk       # alive exactly as long as the communicator
        return h

    def comm_destroy(self, comm):
        self.lib.lance_hip_comm_destroy(comm)
        getattr(self, "_comm_thunks", {}).pop(getattr(comm, "value", comm), None)

    def kmeans_train_sharded(self, comm, x_local, init_centroids, n_total, max_iters=50, tol=1e-4, balance_factor=0.0, seed=0, metric="l2"):
        """lance_hip_kmeans_train_sharded: rows sharded over the ranks of `comm` (None: single process), one fused all-reduce per
        Lloyd iteration inside the library.  init_centroids: identical on every rank.  -> (centroids, loss, iterations)"""
        x, dt = _vec(x_local)                # the shard in the column's own element type (f16 / int8 are widened inside the library, once)
        cent = to_device(init_centroids, torch.float32).clone().contiguous()
        n, d = x.shape
        k = cent.shape[0]
        loss = C.c_double(0); iters = C.c_uint32(0)
        torch.cuda.synchronize()
        check(self.lib.lance_hip_kmeans_train_sharded_x(self.h, comm, dt, METRICS[metric], _ptr(x), n, d, k, int(n_total), max_iters, tol, balance_factor,
                                                        seed, _ptr(cent), C.byref(loss), C.byref(iters)))
        return cent, loss.value, iters.value

    def pq_train(self, residuals, m, nbits=8, max_iters=50, sample_rate=256, seed=0):
        r, dt = _vec(residuals)
        n, d = r.shape
        cb = torch.empty((m, 1 << nbits, d // m), dtype=torch.float32 if r.dtype == torch.int8 else r.dtype, device=r.device)
        iters = np.zeros(m, np.uint32)
        torch.cuda.synchronize()
        check(self.lib.lance_hip_pq_train(self.h, dt, _ptr(r), n, d, m, nbits, max_iters, sample_rate, seed, _ptr(cb),
                                          iters.ctypes.data_as(C.c_void_p)))
        return cb, iters

    def residual(self, x, centroids, part_ids):
        x, dt = _vec(x); centroids = _model(centroids, x)
        p = to_device(part_ids, torch.int32)
        out = torch.empty(x.shape, dtype=centroids.dtype, device=x.device)
        torch.cuda.synchronize()
        check(self.lib.lance_hip_residual(self.h, dt, _ptr(x), x.shape[0], x.shape[1], _ptr(centroids), _ptr(p), _ptr(out)))
        return out

    def pq_encode(self, x, codebook, metric="l2"):
        x, dt = _vec(x); codebook = _model(codebook, x)
        n, d = x.shape
        m = codebook.shape[0]
        nb = _nbits(codebook)
        codes = torch.empty((n, m if nb == 8 else m // 2), dtype=torch.uint8, device=x.device)
        torch.cuda.synchronize()
        check(self.lib.lance_hip_pq_encode(self.h, dt, METRICS[metric], _ptr(x), n, d, _ptr(codebook), m, nb, _ptr(codes)))
        return codes

    def ivfpq_encode(self, x, centroids, codebook, metric="l2", want_loss=True):
        """-> (partition ids, PQ codes, loss).  want_loss=False passes NULL for `loss_out_host`: the C side then skips copying the n
        assignment distances to the host and summing them there (0.5 ms per million rows; the index build does not use the figure)."""
        x, dt = _vec(x); centroids = _model(centroids, x)
        codebook = _model(codebook, x)
        n, d = x.shape
        m = codebook.shape[0]
        nb = _nbits(codebook)
        part = torch.empty(n, dtype=torch.int32, device=x.device)
        codes = torch.empty((n, m if nb == 8 else m // 2), dtype=torch.uint8, device=x.device)
        loss = C.c_double(0)
        torch.cuda.synchronize()
        check(self.lib.lance_hip_ivfpq_encode(self.h, dt, METRICS[metric], _ptr(x), n, d, _ptr(centroids),
                                              centroids.shape[0], _ptr(codebook), m, nb, _ptr(part), _ptr(codes),
                                              C.byref(loss) if want_loss else None))
        return part, codes, (loss.value if want_loss else None)

    def find_partitions(self, q, centroids, nprobes, metric="l2"):
        centroids, dt = _vec(centroids)
        if isinstance(q, (torch.Tensor, np.ndarray)) and str(q.dtype).endswith("int8"):
            q, dt = _vec(q)                       # int8 queries against the f32 centroids of an Int8 column
            q = q.reshape(-1, centroids.shape[1])
        else:
            q = _like(q, centroids).reshape(-1, centroids.shape[1])
        nq, d = q.shape
        nlist = centroids.shape[0]
        nprobes = min(nprobes, nlist)
        ids = torch.empty((nq, nprobes), dtype=torch.int32, device=q.device)
        dists = torch.empty((nq, nprobes), dtype=torch.float32, device=q.device)
        torch.cuda.synchronize()
        check(self.lib.lance_hip_find_partitions(self.h, dt, METRICS[metric], _ptr(q), nq, d, _ptr(centroids), nlist,
                                                 nprobes, _ptr(ids), _ptr(dists)))
        return ids, dists

    def pq_scan_topk(self, q_residual, codebook, codes_transposed, row_ids, k, metric="l2", lower=None, upper=None):
        cb, dt = _vec(codebook)
        q = _like(q_residual, cb).reshape(-1)
        ct = to_device(codes_transposed, torch.uint8)
        rid = to_device(row_ids, torch.int64)
        nb = _nbits(cb)
        m = cb.shape[0]
        n_p = ct.shape[1]
        out_i = torch.empty(k, dtype=torch.int64, device=q.device)
        out_d = torch.empty(k, dtype=torch.float32, device=q.device)
        cnt = C.c_uint32(0)
        has = lower is not None or upper is not None
        lo = float(np.finfo(np.float32).min) if lower is None else float(lower)
        hi = float(np.finfo(np.float32).max) if upper is None else float(upper)
        torch.cuda.synchronize()
        check(self.lib.lance_hip_pq_scan_topk(self.h, dt, METRICS[metric], _ptr(q), q.numel(), _ptr(cb), m, nb, _ptr(ct),
                                              _ptr(rid), n_p, k, int(has), lo, hi, _ptr(out_i), _ptr(out_d), C.byref(cnt)))
        return out_i[:cnt.value], out_d[:cnt.value]

    def flat_topk(self, x, q, k, metric="l2", row_ids=None):
        x, dt = _vec(x)
        q = _like(q, x).reshape(-1, x.shape[1])
        n, d = x.shape
        nq = q.shape[0]
        rid = None if row_ids is None else to_device(row_ids, torch.int64)
        ids = torch.empty((nq, k), dtype=torch.int64, device=x.device)
        dists = torch.empty((nq, k), dtype=torch.float32, device=x.device)
        torch.cuda.synchronize()
        check(self.lib.lance_hip_flat_topk(self.h, dt, METRICS[metric], _ptr(x), _ptr(rid), n, d, _ptr(q), nq, k,
                                           _ptr(ids), _ptr(dists)))
        return ids, dists

    def search_stats(self):
        """queries of the last search replayed by the exact (heap-emulating) kernel"""
        n = C.c_uint32(0)
        check(self.lib.lance_hip_search_stats(self.h, C.byref(n)))
        return n.value

    # ---- timing hooks --------------------------------------------------------------
    def timing(self, on=True):
        check(self.lib.lance_hip_timing_enable(self.h, int(on)))

    def timing_query(self, kernel):
        ms = C.c_double(0); n = C.c_uint64(0)
        check(self.lib.lance_hip_timing_query(self.h, kernel.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value


    def merge_topk(self, ids, dists, k, exact=None, keff=None):
        """(dist, rowid) merge of gathered candidate lists on the device (lance_hip_merge_topk): ids [nq, C] int64 (-1 = none),
        dists [nq, C]; with `exact` the keff best by PQ distance are re-ranked by the exact distances.  -> ([nq,k], [nq,k])"""
        ids = to_device(ids, torch.int64); dists = to_device(dists, torch.float32)
        ex = None if exact is None else to_device(exact, torch.float32)
        nq, c = ids.shape
        out_i = torch.empty((nq, k), dtype=torch.int64, device=ids.device)
        out_d = torch.empty((nq, k), dtype=torch.float32, device=ids.device)
        torch.cuda.synchronize()
        check(self.lib.lance_hip_merge_topk(self.h, _ptr(ids), _ptr(dists), _ptr(ex), nq, c, keff if keff is not None else k, k,
                                            _ptr(out_i), _ptr(out_d)))
        return out_i, out_d

    def ubench(self, what):
        """in-process ceilings for bench.py (lance_hip_ubench): "lds4" / "lds8" / "lds16" lane-gathers per second,
        "copy" bytes per second, "valu" / "valu_pk" f32 wave-instructions per second"""
        r = C.c_double(0)
        codes = {"lds4": 0, "lds8": 1, "lds16": 2, "copy": 3, "valu": 4, "valu_pk": 5,
                 # round 3: code-major m-staggered u16x4 table (M = 16 / 32), today's layout with the integer accumulate, and the
                 # conflict-free ds_read_b64 stream (the LDS pipe's own ceiling)
                 "lds8_stagger16": 6, "lds8_stagger32": 7, "lds8_u16x4": 8, "lds8_linear": 9,
                 # round 6: dense 32x32x16 MFMA rate (flop per second), f16 / bf16 operands
                 "mfma_f16": 10, "mfma_bf16": 11}
        check(self.lib.lance_hip_ubench(self.h, codes[what], C.byref(r)))
        return r.value


_wrap_methods(Engine, skip=("close",))



def _torch_inputs_ready():
    """What a search call needs from torch before the engine's stream reads the batch: the work torch's CURRENT stream has queued (the
    conversion / move of the queries).  Not torch.cuda.synchronize(): a device-wide synchronise is not permitted while ANY stream of the
    process is capturing -- another host thread's repeated search being captured into a HIP graph -- and both threads then fail (the
    synchronising one in torch, the capturing one in the runtime; tests/test_zz_gpu_threads.py, gpurun r06zm).  The engines' streams are
    non-blocking: torch's stream has no implicit dependency on them."""
    torch.cuda.current_stream().synchronize()

class DeviceFlatIndex:
    """Handle of a device-resident IVF_FLAT index (FlatIndex sub-index over the raw vectors of each partition)."""

    def __init__(self, engine, handle, metric, centroids, data_dtype):
        self.engine = engine
        self.h = handle
        self.metric = metric
        self.centroids = centroids
        self.data_dtype = data_dtype

    @classmethod
    def create(cls, engine, metric, centroids, x, part_ids, row_ids=None):
        x, dt = _vec(x)
        cent = _model(centroids, x)
        part = to_device(part_ids, torch.int32)
        rid = None if row_ids is None else to_device(row_ids, torch.int64)
        n, d = x.shape
        h = C.c_void_p()
        torch.cuda.synchronize()
        check(engine.lib.lance_hip_ivfflat_create(engine.h, dt, METRICS[metric], d, _ptr(cent), cent.shape[0], _ptr(x), _ptr(part),
                                                  _ptr(rid), n, C.byref(h)))
        return cls(engine, h, metric, cent, x.dtype)

    @classmethod
    def load(cls, engine, index_dir, dtype=None):
        """IVF_FLAT index files -> HBM (lance_hip_index_load)."""
        from . import index_file
        c = index_file.read_index_files(index_dir, with_rows=False)
        if c.index_type != "IVF_FLAT":
            raise ValueError(f"{index_dir} holds an {c.index_type} index; use DeviceIndex.load")
        ddt, dt = _DT[dtype if dtype is not None else c.dtype]
        h = C.c_void_p()
        torch.cuda.synchronize()
        check(engine.lib.lance_hip_index_load(engine.h, os.fspath(index_dir).encode(), dt, C.byref(h)))
        return cls(engine, h, c.metric, to_device(c.centroids, torch.float16 if c.dtype == "float16" else torch.float32), ddt)

    def save(self, index_dir, loss=None):
        torch.cuda.synchronize()
        check(self.engine.lib.lance_hip_index_save(self.engine.h, self.h, os.fspath(index_dir).encode(),
                                                   0 if loss is None else 1, 0.0 if loss is None else float(loss)))

    def search(self, q, k, nprobes, allow=None):
        """allow: boolean array indexed by row id (a prefilter) -- tested inside the scan kernels
        (lance_hip_ivfflat_search_filtered), no filtered copy of the index"""
        d = self.centroids.shape[1]
        t = q if isinstance(q, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(q))
        q = t.to(self.data_dtype).to(_dev()).contiguous().reshape(-1, d)
        nq = q.shape[0]
        ids = torch.empty((nq, k), dtype=torch.int64, device=q.device)
        dists = torch.empty((nq, k), dtype=torch.float32, device=q.device)
        if allow is not None:
            a = allow if isinstance(allow, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(allow, dtype=bool))
            a = a.to(torch.uint8).to(_dev()).contiguous()
            torch.cuda.synchronize()
            check(self.engine.lib.lance_hip_ivfflat_search_filtered(self.engine.h, self.h, _ptr(q), nq, k, nprobes, _ptr(a), a.numel(),
                                                                    _ptr(ids), _ptr(dists)))
            return ids, dists
        torch.cuda.synchronize()
        check(self.engine.lib.lance_hip_ivfflat_search(self.engine.h, self.h, _ptr(q), nq, k, nprobes, _ptr(ids), _ptr(dists)))
        return ids, dists

    def close(self):
        if getattr(self, "h", None):
            self.engine.lib.lance_hip_index_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceIndex:
    """Handle of a device-resident IVF_PQ index (lance_hip_index)."""

    def __init__(self, engine, handle, metric, centroids, codebook, raw=None, data_dtype=None):
        self.data_dtype = data_dtype if data_dtype is not None else centroids.dtype   # element type of queries / raw vectors
        self.engine = engine
        self.h = handle
        self.metric = metric
        self.centroids = centroids
        self.codebook = codebook
        self._raw = None
        if raw is not None:
            self.set_raw(raw)

    @classmethod
    def create(cls, engine, metric, centroids, codebook, part_ids, codes, row_ids=None, raw=None, dtype=None):
        cent, dt = _vec(centroids); cb = _like(codebook, cent)
        ddt = cent.dtype
        if dtype is not None:
            ddt, dt = _DT[dtype]
        part = to_device(part_ids, torch.int32); codes = to_device(codes, torch.uint8)
        rid = None if row_ids is None else to_device(row_ids, torch.int64)
        n = part.numel()
        nlist, d = cent.shape
        m = cb.shape[0]
        h = C.c_void_p()
        torch.cuda.synchronize()
        check(engine.lib.lance_hip_index_create(engine.h, dt, METRICS[metric], d, _ptr(cent), nlist, _ptr(cb), m, _nbits(cb),
                                                _ptr(part), _ptr(codes), _ptr(rid), n, C.byref(h)))
        return cls(engine, h, metric, cent, cb, raw, ddt)

    @classmethod
    def from_storage(cls, engine, metric, centroids, codebook, part_offsets, codes, row_ids, transposed=True, raw=None, dtype=None):
        cent, dt = _vec(centroids); cb = _like(codebook, cent)
        ddt = cent.dtype
        if dtype is not None:
            ddt, dt = _DT[dtype]
        offs = np.ascontiguousarray(part_offsets, np.uint32)
        codes = to_device(np.ascontiguousarray(codes, np.uint8).reshape(-1), torch.uint8)
        rid = to_device(row_ids, torch.int64)
        nlist, d = cent.shape
        m = cb.shape[0]
        n = rid.numel()
        h = C.c_void_p()
        torch.cuda.synchronize()
        check(engine.lib.lance_hip_index_from_storage(engine.h, dt, METRICS[metric], d, _ptr(cent), nlist, _ptr(cb), m, _nbits(cb),
                                                      offs.ctypes.data_as(C.c_void_p), _ptr(codes), int(transposed), _ptr(rid), n,
                                                      C.byref(h)))
        return cls(engine, h, metric, cent, cb, raw, ddt)

    @classmethod
    def load(cls, engine, index_dir, dtype=None, raw=None, lists=None):
        """`<index_dir>/index.idx` + `auxiliary.idx` (an IVF_PQ index the reference wrote, or `save`) -> HBM, through
        lance_hip_index_load.  dtype: element type of the indexed column ("float32" | "float16" | "int8"); default = the
        element type of the stored tensors.  lists = (world, rank): only the IVF lists p with p % world == rank."""
        from . import index_file
        c = index_file.read_index_files(index_dir, with_rows=False)
        if c.index_type != "IVF_PQ":
            raise ValueError(f"{index_dir} holds an {c.index_type} index; use DeviceFlatIndex.load")
        ddt, dt = _DT[dtype if dtype is not None else c.dtype]
        mdt = torch.float16 if c.dtype == "float16" else torch.float32
        h = C.c_void_p()
        torch.cuda.synchronize()
        mod, rem = (1, 0) if lists is None else (int(lists[0]), int(lists[1]))
        check(engine.lib.lance_hip_index_load_lists(engine.h, os.fspath(index_dir).encode(), dt, mod, rem, C.byref(h)))
        return cls(engine, h, c.metric, to_device(c.centroids, mdt), to_device(c.codebook, mdt), raw, ddt)

    def save(self, index_dir, loss=None):
        """HBM -> the file pair, in the layout merge_partitions writes (builder.rs:938-1079), through lance_hip_index_save."""
        torch.cuda.synchronize()
        check(self.engine.lib.lance_hip_index_save(self.engine.h, self.h, os.fspath(index_dir).encode(),
                                                   0 if loss is None else 1, 0.0 if loss is None else float(loss)))

    def set_raw(self, raw):
        t = raw if isinstance(raw, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(raw))
        self._raw = t.to(self.data_dtype).to(_dev()).contiguous()
        check(self.engine.lib.lance_hip_index_set_raw(self.h, _ptr(self._raw), self._raw.shape[0]))

    def prewarm(self):
        """Index::prewarm (ivf/v2.rs:349-352, dataset.py prewarm_index): build the per-index search constants now."""
        check(self.engine.lib.lance_hip_index_prewarm(self.engine.h, self.h))

    def info(self):
        n = C.c_uint64(); nlist = C.c_uint32(); m = C.c_uint32(); d = C.c_uint32()
        check(self.engine.lib.lance_hip_index_info(self.h, C.byref(n), C.byref(nlist), C.byref(m), C.byref(d)))
        return {"n": n.value, "nlist": nlist.value, "m": m.value, "d": d.value}

    def export(self):
        """-> (part_offsets u32[nlist+1], codes_transposed u8 (per-partition [m][n_p] blocks), row_ids u64[n])"""
        inf = self.info()
        offs = np.empty(inf["nlist"] + 1, np.uint32)
        mb = inf["m"] if _nbits(self.codebook) == 8 else inf["m"] // 2
        codes = np.empty(inf["n"] * mb, np.uint8)
        rid = np.empty(inf["n"], np.uint64)
        check(self.engine.lib.lance_hip_index_export(self.engine.h, self.h, offs.ctypes.data_as(C.c_void_p),
                                                     codes.ctypes.data_as(C.c_void_p), rid.ctypes.data_as(C.c_void_p)))
        return offs, codes, rid

    def part_offsets(self):
        """-> part_offsets u32[nlist+1] alone (no copy of the codes or row ids to the host: at a billion rows those are 40 GB)"""
        inf = self.info()
        offs = np.empty(inf["nlist"] + 1, np.uint32)
        check(self.engine.lib.lance_hip_index_export(self.engine.h, self.h, offs.ctypes.data_as(C.c_void_p), None, None))
        return offs

    MAX_PAIRS_PER_CALL = 1_500_000      # (query, probe) pairs per library call of a synchronous search (256 survivor slots x 8 B each)

    def search(self, q, k, nprobes, refine_factor=0, out=None, sync=True, engine=None):
        """engine: another Engine (own stream + scratch arena) on the same GPU to run this batch on -- the index is read-only
        during searches, so batches enqueued through different engines overlap on the device (sync=False)."""
        d = self.centroids.shape[1]
        t = q if isinstance(q, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(q))
        q = t.to(self.data_dtype).to(_dev()).contiguous().reshape(-1, d)
        nq = q.shape[0]
        if out is None:
            ids = torch.empty((nq, k), dtype=torch.int64, device=q.device)
            dists = torch.empty((nq, k), dtype=torch.float32, device=q.device)
        else:
            ids, dists = out
        eng = engine if engine is not None else self.engine
        fn = eng.lib.lance_hip_ivfpq_search if sync else eng.lib.lance_hip_ivfpq_search_async
        if sync and nq > 1 and nq * max(1, nprobes) > self.MAX_PAIRS_PER_CALL:
            # a batch whose survivor segments would not fit the library's 2 GiB scratch limit leaves the batched kernels for the
            # query-major ones (correct, many times slower): queries are independent, so the batch goes through in slices
            step = max(1, self.MAX_PAIRS_PER_CALL // max(1, nprobes))
            _torch_inputs_ready()
            for a in range(0, nq, step):
                b = min(nq, a + step)
                check(fn(eng.h, self.h, _ptr(q[a:b]), b - a, k, nprobes, refine_factor, _ptr(ids[a:b]), _ptr(dists[a:b])))
            return ids, dists
        if sync:
            _torch_inputs_ready()
        elif not eng.use_torch_stream and (q.data_ptr() != t.data_ptr() or out is None):
            # the batch was converted / moved / allocated by torch kernels on torch's stream: they must finish before the
            # engine's own stream reads them.  reshape() always makes a new tensor OBJECT, so the test is on the storage: a
            # caller that hands over ready device tensors of the index's element type (and `out`) pays nothing
            torch.cuda.current_stream().synchronize()
        check(fn(eng.h, self.h, _ptr(q), nq, k, nprobes, refine_factor, _ptr(ids), _ptr(dists)))
        return ids, dists

    def search_filtered(self, q, k, nprobes, allow, refine_factor=0, out=None):
        """Search under a row-id prefilter (lance_hip_ivfpq_search_filtered): `allow` = boolean array indexed by row id.  The
        mask is tested inside the scan kernels; no filtered copy of the index is built.  out = (ids, dists) device tensors to
        write into (a caller that keeps its query / output buffers gets the captured-graph path from its second call)."""
        d = self.centroids.shape[1]
        t = q if isinstance(q, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(q))
        q = t.to(self.data_dtype).to(_dev()).contiguous().reshape(-1, d)
        a = allow if isinstance(allow, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(allow, dtype=bool))
        a = a.to(torch.uint8).to(_dev()).contiguous()
        nq = q.shape[0]
        if out is None:
            ids = torch.empty((nq, k), dtype=torch.int64, device=q.device)
            dists = torch.empty((nq, k), dtype=torch.float32, device=q.device)
        else:
            ids, dists = out
        _torch_inputs_ready()
        check(self.engine.lib.lance_hip_ivfpq_search_filtered(self.engine.h, self.h, _ptr(q), nq, k, nprobes, refine_factor, _ptr(a),
                                                              a.numel(), _ptr(ids), _ptr(dists)))
        return ids, dists

    def search_candidates(self, q, keff, nprobes, exact=True, engine=None):
        """ONE scan -> (ids [nq, keff] int64 (-1 = none), PQ distances, exact distances or None): the keff nearest rows by PQ distance in
        (dist, rowid) order and, aligned with them, each one's exact distance to the query (lance_hip_ivfpq_search_candidates) -- the
        local half of a list-sharded search with refine (lance_amd/dist.py: search_list_sharded)."""
        eng = engine or self.engine
        d = self.centroids.shape[1]
        t = q if isinstance(q, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(q))
        q = t.to(self.data_dtype).to(_dev()).contiguous().reshape(-1, d)
        nq = q.shape[0]
        ids = torch.empty((nq, keff), dtype=torch.int64, device=q.device)
        pq = torch.empty((nq, keff), dtype=torch.float32, device=q.device)
        ex = torch.empty((nq, keff), dtype=torch.float32, device=q.device) if exact else None
        _torch_inputs_ready()
        check(eng.lib.lance_hip_ivfpq_search_candidates(eng.h, self.h, _ptr(q), nq, keff, min(nprobes, self.centroids.shape[0]), _ptr(ids), _ptr(pq),
                                                        _ptr(ex) if exact else None))
        return ids, pq, ex

    def search_range(self, q, k, nprobes, lower=None, upper=None, refine_factor=0, allow=None, out=None):
        """Distance-range query: only rows with lower <= d < upper (ADC distance) enter the per-partition heaps.  With a
        refine factor the reference also filters the exact distances before the final fetch (scanner.rs:3334-3377): all
        k * refine_factor candidates come back re-ranked from the device and the range is applied to them here.
        allow: boolean array indexed by row id -- the range combined with a row-id prefilter, tested inside the scan
        kernels (lance_hip_ivfpq_search_filtered_range; flat/index.rs:131-149)."""
        if refine_factor and refine_factor > 0:
            keff = k * refine_factor
            ci, cd = self.search_range(q, keff, nprobes, lower, upper, refine_factor=-1, allow=allow)       # -1: re-rank, keep all
            # the bounds are f32 values in the reference (Query::lower_bound: Option<f32>): round before comparing
            lo = float(np.finfo(np.float32).min) if lower is None else float(np.float32(lower))
            hi = float(np.finfo(np.float32).max) if upper is None else float(np.float32(upper))
            ok = (ci >= 0) & (cd >= lo) & (cd < hi)
            # first k in-range entries of every row, order kept (the device returned them sorted by (distance, row id))
            rank = torch.cumsum(ok.to(torch.int64), dim=1) - 1
            ids = torch.full((ci.shape[0], k), -1, dtype=torch.int64, device=ci.device)
            dists = torch.full((ci.shape[0], k), float("inf"), dtype=torch.float32, device=ci.device)
            take = ok & (rank < k)
            rows = torch.arange(ci.shape[0], device=ci.device).unsqueeze(1).expand_as(ci)
            ids[rows[take], rank[take]] = ci[take]
            dists[rows[take], rank[take]] = cd[take]
            return ids, dists
        d = self.centroids.shape[1]
        t = q if isinstance(q, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(q))
        q = t.to(self.data_dtype).to(_dev()).contiguous().reshape(-1, d)
        nq = q.shape[0]
        if out is None:
            ids = torch.empty((nq, k), dtype=torch.int64, device=q.device)
            dists = torch.empty((nq, k), dtype=torch.float32, device=q.device)
        else:
            ids, dists = out
        lo = float(np.finfo(np.float32).min) if lower is None else float(np.float32(lower))
        hi = float(np.finfo(np.float32).max) if upper is None else float(np.float32(upper))
        if allow is not None:
            a = allow if isinstance(allow, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(allow, dtype=bool))
            a = a.to(torch.uint8).to(_dev()).contiguous()
            _torch_inputs_ready()
            check(self.engine.lib.lance_hip_ivfpq_search_filtered_range(self.engine.h, self.h, _ptr(q), nq, k, nprobes,
                                                                        1 if refine_factor == -1 else 0, _ptr(a), a.numel(), lo, hi,
                                                                        _ptr(ids), _ptr(dists)))
            return ids, dists
        _torch_inputs_ready()
        check(self.engine.lib.lance_hip_ivfpq_search_range(self.engine.h, self.h, _ptr(q), nq, k, nprobes, 1 if refine_factor == -1 else 0,
                                                           lo, hi, _ptr(ids), _ptr(dists)))
        return ids, dists

    def close(self):
        if getattr(self, "h", None):
            self.engine.lib.lance_hip_index_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_wrap_methods(DeviceFlatIndex, skip=("close",))
_wrap_methods(DeviceIndex, skip=("close",))
