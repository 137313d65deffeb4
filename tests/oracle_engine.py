"""Oracle-backed stand-ins for lance_amd.engine.{Engine, DeviceIndex, DeviceFlatIndex} on CPU tensors -- TEST
INFRASTRUCTURE.  They let the host layer (lance_amd/vector.py, dist.py, accelerator.py) and the bodies of the `-m gpu`
tests run in this container, where no GPU exists: every device step is computed by the CPU oracle, the index-file steps by
the real native reader/writer.  Nothing here says anything about the kernels; it checks plumbing, argument order, shapes
and the tests' own logic before they meet hardware.  Used by tests/test_device_tests_dryrun.py (and the gloo tests)."""
import numpy as np
import torch

import oracle

f32 = np.float32


def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def cpu_to_device(a, dtype=None):
    """lance_amd.engine.to_device without a device"""
    if isinstance(a, torch.Tensor):
        t = a
    else:
        arr = np.ascontiguousarray(a)
        if arr.dtype == np.uint64:
            t = torch.from_numpy(arr.view(np.int64))
        elif arr.dtype == np.uint32:
            t = torch.from_numpy(arr.view(np.int32))
        else:
            t = torch.from_numpy(arr)
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


class OracleEngine:
    """Engine: CPU tensors in, CPU tensors out, same return conventions (int32 ids with -1 = none, float32 distances)."""

    def normalize(self, x):
        xn = _np(x)
        return torch.from_numpy(oracle.normalize(xn if xn.dtype == np.float16 else xn.astype(f32)))     # float16 stays float16

    def assign(self, x, cent, metric="l2", bias=None):
        ids, d = oracle.assign(np.ascontiguousarray(_np(x)), _np(cent), metric, bias=None if bias is None else _np(bias))
        return torch.from_numpy(ids.view(np.int32).copy()), torch.from_numpy(d)

    def kmeans_train(self, x, k, max_iters=50, tol=1e-4, balance_factor=0.0, init=None, seed=0, metric="l2", hierarchical_k=16):
        xn = np.ascontiguousarray(_np(x))
        n = xn.shape[0]
        if k > 256 and hierarchical_k > 1:
            c = oracle.kmeans_train_hierarchical(xn, k, max_iters=max_iters, tol=tol, balance_factor_scaled=f32(balance_factor) / f32(n),
                                                 hierarchical_k=hierarchical_k, seed=seed, metric=metric)
            return torch.from_numpy(c), 0.0, 0
        c, loss, iters, _ = oracle.kmeans_train(xn[: min(n, k * 512)], k, max_iters=max_iters, tol=tol,
                                                balance_factor=f32(balance_factor) / f32(n), init=None if init is None else _np(init),
                                                seed=seed, metric=metric)
        return torch.from_numpy(c), loss, iters

    def kmeans_split(self, x, rows, k, max_iters=50, tol=1e-4, balance_factor_scaled=0.0, seed=0, metric="l2"):
        return oracle.kmeans_split(np.ascontiguousarray(_np(x)), rows, k, max_iters=max_iters, tol=tol, balance_factor_scaled=balance_factor_scaled,
                                   seed=seed, metric=metric)

    def residual(self, x, cent, part):
        return torch.from_numpy(oracle.residual(_np(x), _np(cent), np.ascontiguousarray(_np(part)).view(np.uint32)))

    def pq_train(self, r, m, nbits=8, max_iters=50, sample_rate=256, seed=0):
        cb, it = oracle.pq_train(np.ascontiguousarray(_np(r)), m, nbits=nbits, max_iters=max_iters, sample_rate=sample_rate, seed=seed)
        return torch.from_numpy(cb), it.astype(np.uint32)

    def pq_encode(self, x, cb, metric="l2"):
        cbn = _np(cb)
        return torch.from_numpy(oracle.pq_encode(np.ascontiguousarray(_np(x)), cbn, metric, nbits=4 if cbn.shape[1] == 16 else 8))

    def ivfpq_encode(self, x, cent, cb, metric="l2", want_loss=True):
        xn = np.ascontiguousarray(_np(x))
        cbn = _np(cb)
        oi = oracle.build_index(xn, _np(cent), cbn, metric, nbits=4 if cbn.shape[1] == 16 else 8)
        h = xn.dtype == np.float16        # a Float16 column keeps its own normalize / dot (oracle: ORC_COSINE_H / ORC_DOT_H)
        xs = oracle.normalize(xn if h else xn.astype(f32)) if metric == "cosine" else xn
        xs = xs if h else xs.astype(f32)
        keep = oracle.is_finite(xs)
        part = np.full(xn.shape[0], oracle.NONE, np.uint32)
        part[keep] = oi.part_ids
        codes = np.zeros((xn.shape[0], oi.codes_rowmajor.shape[1]), np.uint8)
        codes[keep] = oi.codes_rowmajor
        _, dist = oracle.assign(xs[keep], _np(cent) if h else _np(cent).astype(f32), "l2" if metric == "cosine" else metric)
        ok = oi.part_ids != oracle.NONE
        return torch.from_numpy(part.view(np.int32).copy()), torch.from_numpy(codes), float(dist[ok].astype(np.float64).sum())

    def find_partitions(self, q, cent, nprobes, metric="l2"):
        qn, cn = _np(q), _np(cent)
        if not (qn.dtype == np.float16 and cn.dtype == np.float16):
            qn, cn = qn.astype(f32), cn.astype(f32)
        p, d = oracle.find_partitions(qn, cn, nprobes, metric)
        return torch.from_numpy(p.view(np.int32).copy()), torch.from_numpy(d)

    def flat_topk(self, x, q, k, metric="l2", row_ids=None):
        rid = None if row_ids is None else np.ascontiguousarray(_np(row_ids)).view(np.uint64)
        xn = np.ascontiguousarray(_np(x))
        i, d = oracle.flat_knn(xn if xn.dtype == np.float16 else xn.astype(f32), _np(q).astype(f32), k, metric, row_ids=rid)
        return torch.from_numpy(i.view(np.int64)), torch.from_numpy(d)


def _group(part, codes, rid, nlist):
    """what lance_hip_index_create does: stable grouping by partition, rows without one dropped, per-partition transpose"""
    offs, perm = oracle.partition_layout(part, nlist)
    cs = codes[perm]
    m = codes.shape[1]
    ct = np.empty(cs.size, np.uint8)
    for p in range(nlist):
        a, b = int(offs[p]), int(offs[p + 1])
        ct[a * m:b * m] = cs[a:b].T.reshape(-1)
    return offs, ct, rid[perm]


class OracleDeviceIndex:
    """DeviceIndex: create / from_storage / load / save / export / search / info / set_raw."""

    def __init__(self, engine, o, metric, centroids, codebook, raw=None, data_dtype=None):
        self.engine, self.o, self.metric = engine, o, metric
        self.centroids, self.codebook = centroids, codebook
        self.data_dtype = data_dtype if data_dtype is not None else centroids.dtype
        self._raw = None
        if raw is not None:
            self.set_raw(raw)

    @classmethod
    def create(cls, engine, metric, centroids, codebook, part_ids, codes, row_ids=None, raw=None, dtype=None):
        cent, cb = _np(centroids), _np(codebook)
        part = np.ascontiguousarray(_np(part_ids)).view(np.uint32)
        rid = np.arange(part.size, dtype=np.uint64) if row_ids is None else np.ascontiguousarray(_np(row_ids)).view(np.uint64)
        offs, ct, rids = _group(part, _np(codes), rid, cent.shape[0])
        o = oracle.IvfPqIndex(metric, cent, cb, offs, ct, rids, nbits=4 if cb.shape[1] == 16 else 8)
        return cls(engine, o, metric, cpu_to_device(centroids), cpu_to_device(codebook), raw,
                   {"float16": torch.float16, "int8": torch.int8}.get(dtype, None))

    @classmethod
    def from_storage(cls, engine, metric, centroids, codebook, part_offsets, codes, row_ids, transposed=True, raw=None, dtype=None):
        cent, cb = _np(centroids), _np(codebook)
        offs = np.asarray(part_offsets, np.uint32)
        codes = np.ascontiguousarray(_np(codes), np.uint8).reshape(-1)
        nb = 4 if cb.shape[1] == 16 else 8
        m = cb.shape[0] // 2 if nb == 4 else cb.shape[0]
        if not transposed and codes.size:
            codes = np.concatenate([codes[int(offs[p]) * m:int(offs[p + 1]) * m].reshape(-1, m).T.reshape(-1) for p in range(len(offs) - 1)])
        o = oracle.IvfPqIndex(metric, cent, cb, offs, codes, np.ascontiguousarray(_np(row_ids)).view(np.uint64), nbits=nb)
        return cls(engine, o, metric, cpu_to_device(centroids), cpu_to_device(codebook), raw)

    @classmethod
    def load(cls, engine, index_dir, dtype=None, raw=None, lists=None):
        from lance_amd import index_file
        from lance_amd.dist import shard_index_contents
        c = index_file.read_index_files(index_dir)
        if c.index_type != "IVF_PQ":
            raise ValueError("not IVF_PQ")
        model = np.float16 if c.dtype == "float16" else np.float32
        offs, codes, rid = (c.part_offsets, c.codes, c.row_ids) if lists is None else shard_index_contents(c, lists[0], lists[1])
        return cls.from_storage(engine, c.metric, c.centroids.astype(model), c.codebook.astype(model), offs, codes, rid,
                                transposed=c.transposed, raw=raw)

    def save(self, index_dir, loss=None):
        from lance_amd import index_file
        f16 = self.centroids.dtype == torch.float16
        index_file.write_index_files(index_dir, index_file.IndexFileContents(
            index_type="IVF_PQ", metric=self.metric, dtype="float16" if f16 else "float32", centroids=self.o.centroids,
            part_offsets=self.o.part_offsets, row_ids=self.o.row_ids, codebook=self.o.codebook, codes=self.o.codes_t,
            num_sub_vectors=self.o.codebook.shape[0], nbits=self.o.nbits, transposed=True, loss=loss))

    def set_raw(self, raw):
        self._raw = cpu_to_device(raw).to(self.data_dtype)

    def close(self):
        pass

    def prewarm(self):
        pass

    def info(self):
        return {"n": int(self.o.row_ids.size), "nlist": int(self.o.centroids.shape[0]), "m": int(self.o.codebook.shape[0]),
                "d": int(self.o.centroids.shape[1])}

    def export(self):
        return self.o.part_offsets.copy(), self.o.codes_t.copy(), self.o.row_ids.copy()

    def search(self, q, k, nprobes, refine_factor=0, out=None, sync=True, engine=None):
        raw = None if self._raw is None else self._raw.numpy().astype(f32)
        i, d = self.o.search(_np(q).astype(f32), k, nprobes, refine=refine_factor, raw=raw if refine_factor else None)
        return torch.from_numpy(i.view(np.int64)), torch.from_numpy(d)


    def search_filtered(self, q, k, nprobes, allow, refine_factor=0):
        raw = None if self._raw is None else self._raw.numpy().astype(f32)
        i, d = self.o.search(_np(q).astype(f32), k, nprobes, refine=refine_factor, raw=raw if refine_factor else None,
                             prefilter=np.ascontiguousarray(_np(allow), dtype=bool))
        return torch.from_numpy(i.view(np.int64)), torch.from_numpy(d)

    def search_range(self, q, k, nprobes, lower=None, upper=None, refine_factor=0, allow=None):
        lo = np.finfo(f32).min if lower is None else lower
        hi = np.finfo(f32).max if upper is None else upper
        raw = None if self._raw is None else self._raw.numpy().astype(f32)
        i, d = self.o.search(_np(q).astype(f32), k, nprobes, lower=lo, upper=hi, refine=refine_factor, raw=raw if refine_factor else None,
                             prefilter=None if allow is None else np.ascontiguousarray(_np(allow), dtype=bool))
        return torch.from_numpy(i.view(np.int64)), torch.from_numpy(d)


class OracleDeviceFlatIndex:
    """DeviceFlatIndex: create / load / save / search."""

    def __init__(self, engine, metric, centroids, x, part, rid, data_dtype):
        self.engine, self.metric, self.centroids, self.data_dtype = engine, metric, centroids, data_dtype
        self.x, self.part, self.rid = x, part, rid

    @classmethod
    def create(cls, engine, metric, centroids, x, part_ids, row_ids=None):
        xn = np.ascontiguousarray(_np(x))
        part = np.ascontiguousarray(_np(part_ids)).view(np.uint32)
        rid = np.arange(part.size, dtype=np.uint64) if row_ids is None else np.ascontiguousarray(_np(row_ids)).view(np.uint64)
        keep = part != oracle.NONE
        return cls(engine, metric, cpu_to_device(centroids), xn[keep], part[keep], rid[keep], cpu_to_device(x).dtype)

    @classmethod
    def load(cls, engine, index_dir, dtype=None):
        from lance_amd import index_file
        c = index_file.read_index_files(index_dir)
        if c.index_type != "IVF_FLAT":
            raise ValueError("not IVF_FLAT")
        return cls(engine, c.metric, cpu_to_device(c.centroids), c.vectors, c.part_ids(), c.row_ids, torch.float32)

    def save(self, index_dir, loss=None):
        from lance_amd import index_file
        nlist = self.centroids.shape[0]
        offs, perm = oracle.partition_layout(self.part, nlist)
        index_file.write_index_files(index_dir, index_file.IndexFileContents(
            index_type="IVF_FLAT", metric=self.metric, dtype="float32", centroids=_np(self.centroids).astype(f32), part_offsets=offs,
            row_ids=self.rid[perm], vectors=self.x[perm].astype(f32), loss=loss))

    def close(self):
        pass

    def search(self, q, k, nprobes, allow=None):
        if allow is not None:      # the fused mask == a compacted copy (same distance function per selected row)
            al = np.ascontiguousarray(_np(allow), dtype=bool)
            keep = (self.rid < al.size) & al[np.minimum(self.rid, max(al.size, 1) - 1).astype(np.int64)] if al.size else np.zeros(self.rid.size, bool)
            sub = OracleDeviceFlatIndex(self.engine, self.metric, self.centroids, self.x[keep], self.part[keep], self.rid[keep], self.data_dtype)
            return sub.search(q, k, nprobes)
        # the stored partition of every row is authoritative (it may come from a file or carry a prefilter's holes)
        et = np.float16 if self.data_dtype == torch.float16 else f32      # a Float16 column: key, centroids and rows are f16
        cent = _np(self.centroids).astype(et)
        nlist = cent.shape[0]
        offs, perm = oracle.partition_layout(self.part, nlist)
        qq = _np(q).astype(et).reshape(-1, cent.shape[1])
        if self.metric == "cosine":     # stored rows are normalised already; the query key is normalised here (knn.rs:498)
            qq = oracle.normalize(qq)
        probes, _ = oracle.find_partitions(qq, cent, nprobes, "l2" if self.metric == "cosine" else self.metric)
        out_i = np.full((qq.shape[0], k), np.iinfo(np.uint64).max, np.uint64)
        out_d = np.full((qq.shape[0], k), np.inf, f32)
        xf = self.x.astype(et)
        for qi in range(qq.shape[0]):
            ci, cd = [], []
            for p in probes[qi]:
                rows = perm[int(offs[p]):int(offs[p + 1])]
                if len(rows) == 0:
                    continue
                d = oracle.distance_batch(self.metric, qq[qi], xf[rows])
                hi, hd = oracle.heap_topk(d, self.rid[rows], k)
                ci.append(hi); cd.append(hd)
            if ci:
                ai, ad = oracle.sort_fetch(np.concatenate(ci), np.concatenate(cd), k)
                out_i[qi, :len(ai)] = ai
                out_d[qi, :len(ad)] = ad
        return torch.from_numpy(out_i.view(np.int64)), torch.from_numpy(out_d)


def install(monkeypatch):
    """Swap the device classes of lance_amd for the stand-ins (for one test)."""
    import lance_amd.engine as E
    import lance_amd.vector as V
    import lance_amd.dist as D
    eng = OracleEngine()
    for mod in (E, V, D):
        if hasattr(mod, "to_device"):
            monkeypatch.setattr(mod, "to_device", cpu_to_device)
    monkeypatch.setattr(E, "DeviceIndex", OracleDeviceIndex)
    monkeypatch.setattr(E, "DeviceFlatIndex", OracleDeviceFlatIndex)
    monkeypatch.setattr(V, "DeviceIndex", OracleDeviceIndex)
    monkeypatch.setattr(V, "DeviceFlatIndex", OracleDeviceFlatIndex)
    monkeypatch.setattr(V, "_engine", eng)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    return eng
