"""Host-side mirror of the reference's IVF_PQ build / query interface for the hot path.

Names, argument meaning and defaults follow the reference (citations relative to the
lancedb/lance tree):
  * create_index(..., "IVF_PQ", metric, num_partitions, num_sub_vectors, ivf_centroids=,
    pq_codebook=, sample_rate, max_iters)          python/python/lance/dataset.py:2517-2545
  * IvfBuildParams / PQBuildParams defaults        rust/lance-index/src/vector/ivf/builder.rs:62-78,
                                                   pq/builder.rs:48-58
  * build order: sample -> train IVF -> residuals -> train PQ -> transform all rows ->
    per-partition storage                          rust/lance/src/index/vector/builder.rs:236-254,377-466
  * nearest={"q", "k", "nprobes", "refine_factor"} python/src/dataset.rs:984-1095
  * KMeans(k, metric_type, max_iters, centroids).fit/.predict   python/python/lance/util.py:45-170

All computation happens in liblance_hip.so; this module only orchestrates.
"""
import time
from dataclasses import dataclass, field
from typing import Optional

import numpy as np
import torch

from ._lib import METRICS, NONE
from .engine import DeviceFlatIndex, DeviceIndex, Engine, to_device

_engine = None


def default_engine():
    global _engine
    if _engine is None:
        _engine = Engine()
    return _engine


def _normalize_metric_type(metric):
    m = str(metric).lower()
    if m == "euclidean":
        m = "l2"
    if m not in ("l2", "cosine", "dot"):
        raise ValueError(f"Metric {metric} not supported.")
    return m


class KMeans:
    """lance.util.KMeans (python/python/lance/util.py:45-170) on the MI355X engine."""

    def __init__(self, k, metric_type="l2", max_iters=50, centroids=None, seed=0, engine=None):
        self.k = k
        self._metric_type = _normalize_metric_type(metric_type)
        self.max_iters = max_iters
        self.seed = seed
        self._engine = engine or default_engine()
        self._centroids = None if centroids is None else to_device(np.asarray(centroids, np.float32))
        self.loss = None
        self.iters = None

    def __repr__(self):
        return f"lance_amd.KMeans(k={self.k}, metric_type={self._metric_type})"

    @property
    def centroids(self):
        return None if self._centroids is None else self._centroids.cpu().numpy()

    @staticmethod
    def _check(data):
        if isinstance(data, torch.Tensor):
            if data.dim() != 2 or data.dtype != torch.float32:
                raise ValueError("Data must be a 2-D float32 array")
            return data
        data = np.asarray(data)
        if data.ndim != 2:
            raise ValueError(f"Numpy array must be a 2-D array, got {data.ndim}-D")
        if data.dtype != np.float32:
            raise ValueError(f"Numpy array must be float32 type, got: {data.dtype}")
        return data

    def fit(self, data):
        x = to_device(self._check(data))
        metric = self._metric_type
        if metric == "cosine":  # python/src/utils.rs _KMeans::fit -> KMeans::new_with_params normalises for cosine
            x = self._engine.normalize(x)
            metric = "l2"
        # _KMeans uses KMeansParams::new(...) (redos 1, no balance) and sample_rate 256 (python/src/utils.rs:75-109)
        n = x.shape[0]
        if n > 256 * self.k:
            x = x[: 256 * self.k]
        self._centroids, self.loss, self.iters = self._engine.kmeans_train(
            x, self.k, max_iters=self.max_iters, init=self._centroids, seed=self.seed, metric=metric)

    def predict(self, data):
        if self._centroids is None:
            raise ValueError("KMeans model is not trained")
        x = to_device(self._check(data))
        metric = self._metric_type
        if metric == "cosine":
            x = self._engine.normalize(x)
            metric = "l2"
        ids, _ = self._engine.assign(x, self._centroids, metric)
        return ids.cpu().numpy().view(np.uint32)


@dataclass
class IvfPqParams:
    num_partitions: int = 256
    num_sub_vectors: int = 16
    num_bits: int = 8
    metric: str = "l2"
    max_iters: int = 50          # IvfBuildParams::max_iters / PQBuildParams::max_iters
    sample_rate: int = 256       # both default to 256
    seed: int = 42


@dataclass
class BuildStats:
    seconds: dict = field(default_factory=dict)
    ivf_iters: int = 0
    pq_iters: Optional[np.ndarray] = None
    ivf_loss: float = 0.0
    ivf_training: str = "single"      # multi-GPU builds: "replicated" | "sharded" | "hierarchical" (lance_amd/dist.py)
    ivf_hierarchical: Optional[dict] = None      # hierarchical training spread over ranks: rounds, splits applied / thrown away

    @property
    def total(self):
        return sum(self.seconds.values())


class IvfPqIndex:
    """An IVF_PQ index resident in HBM with the reference's query semantics."""

    def __init__(self, dev_index, params, stats=None, part_ids=None, codes=None):
        self._ix = dev_index
        self.params = params
        self.stats = stats
        self.part_ids = part_ids   # shuffle-buffer columns (device), kept for hand-off to Lance
        self.codes = codes

    @property
    def centroids(self):
        return self._ix.centroids.cpu().numpy()

    @property
    def codebook(self):
        """numpy (M, 256, d/M) -- the `pq_codebook` artefact (dataset.py:2928-2954)"""
        return self._ix.codebook.cpu().numpy()

    def info(self):
        return self._ix.info()

    def export_storage(self):
        return self._ix.export()

    def shuffle_buffers(self):
        """(row_id u64, __ivf_part_id u32, __pq_code u8[M]) as numpy -- the artefact the reference's
        `precomputed_shuffle_buffers` hand-off consumes (python/lance/vector.py:659-665)."""
        part = self.part_ids.cpu().numpy().view(np.uint32)
        keep = part != NONE
        rid = np.arange(part.size, dtype=np.uint64)[keep]
        return rid, part[keep], self.codes.cpu().numpy()[keep]

    def nearest(self, q, k=10, nprobes=1, refine_factor=None, prefilter=None, distance_range=None, minimum_nprobes=None,
                maximum_nprobes=None):
        """-> (row ids int64 [nq,k] (-1 = missing), distances f32 [nq,k]) as numpy.
        prefilter: boolean array over row ids (True = row may be returned) -- `nearest=..., filter=..., prefilter=True`
        of the reference (scanner.rs prefilter -> FlatIndex::search's RowIdMask branch, flat/index.rs:129-165).  The mask is
        tested inside the scan kernels (lance_hip_ivfpq_search_filtered); no filtered copy of the index is built.
        distance_range: (lower, upper), either may be None -- rows with lower <= d < upper only (Query::lower_bound /
        upper_bound, flat/index.rs:98-113).
        minimum_nprobes / maximum_nprobes: adaptive probing (Query::minimum_nprobes / maximum_nprobes; ANNIvfSubIndexExec
        initial_search + late_search, knn.rs:714-860): every query searches its `minimum_nprobes` nearest partitions; a query
        that found fewer than k rows (a selective prefilter, tiny partitions) goes on through further partitions, nearest
        first, until it has k rows or `maximum_nprobes` (default: all) partitions were searched.  `nprobes=n` alone means
        minimum = maximum = n, as in pylance (python/src/dataset.rs:984-1020).  The reference extends the search one
        partition at a time from concurrently running tasks, so how far it overshoots depends on thread timing; here the
        extension is deterministic: the number of partitions doubles until the query is satisfied.  late_search's shortcut
        for prefilters that select at most k rows (the unfound selected rows come back at distance +inf) is reproduced for
        searches without a refine factor."""
        rf = 0 if refine_factor is None else refine_factor
        nlist = self.params.num_partitions
        min_np = nprobes if minimum_nprobes is None else minimum_nprobes
        max_np = (min_np if minimum_nprobes is None else nlist) if maximum_nprobes is None else maximum_nprobes
        max_np = max(min(max_np, nlist), min(min_np, nlist))
        min_np = min(min_np, max_np)

        def run(qq, npb):
            if distance_range is not None:      # late_search extends range queries too (knn.rs:714-860)
                lo, hi = distance_range
                return self._ix.search_range(qq, k, npb, lo, hi, refine_factor=rf, allow=prefilter)
            if prefilter is None:
                return self._ix.search(qq, k, npb, rf)
            return self._ix.search_filtered(qq, k, npb, prefilter, rf)

        ids, dists = run(q, min_np)
        if max_np > min_np and prefilter is not None and not rf and distance_range is None:
            # late_search's shortcut (knn.rs:741-779): when the prefilter selects no more than k rows, a query that has not
            # found all of them yet gets the rest back with distance +inf instead of searching further partitions
            allow_np = np.ascontiguousarray(prefilter.cpu().numpy() if isinstance(prefilter, torch.Tensor) else prefilter, dtype=bool)
            mask_ids = np.flatnonzero(allow_np).astype(np.int64)
            if mask_ids.size <= k:
                ids_h, dists_h = ids.cpu().numpy().copy(), dists.cpu().numpy().copy()
                for qi in range(ids_h.shape[0]):
                    found = ids_h[qi][ids_h[qi] >= 0]
                    if found.size < k and found.size < mask_ids.size:
                        rest = np.setdiff1d(mask_ids, found)          # ascending row ids: SortExec's tie order at +inf
                        ids_h[qi, found.size:found.size + rest.size] = rest
                        dists_h[qi, found.size:found.size + rest.size] = np.inf
                return ids_h, dists_h
        if max_np > min_np:
            qt = q if isinstance(q, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(q))
            qt = qt.reshape(-1, self._ix.centroids.shape[1])
            npb = min_np
            while npb < max_np:
                starved = torch.nonzero((ids < 0).any(dim=1)).reshape(-1)     # fewer than k rows found so far
                if starved.numel() == 0:
                    break
                npb = min(max_np, max(npb * 2, npb + 1))
                si, sd = run(qt[starved.cpu()] if not qt.is_cuda else qt[starved], npb)
                ids[starved] = si
                dists[starved] = sd
        return ids.cpu().numpy(), dists.cpu().numpy()

    def _storage_rows(self):
        """(part ids int32 [n], row-major codes u8 [n, code bytes], row ids int64 [n] | None) on the device, in an order
        whose stable grouping by partition is the stored order"""
        if self.part_ids is not None and self.codes is not None:
            return self.part_ids, self.codes, None
        offs, codes_t, rid = self._ix.export()          # an index opened from files: undo the per-partition transpose,
        n, nlist = len(rid), len(offs) - 1              # on the device (torch is only moving bytes here)
        cb = codes_t.size // n if n else 1
        ct = to_device(codes_t)
        rm = torch.empty((n, cb), dtype=torch.uint8, device=ct.device)
        for p in range(nlist):
            a, b = int(offs[p]), int(offs[p + 1])
            if b > a:
                rm[a:b] = ct[a * cb:b * cb].view(cb, b - a).t()
        lens = to_device(np.diff(offs.astype(np.int64)))
        part = torch.repeat_interleave(torch.arange(nlist, dtype=torch.int32, device=ct.device), lens)
        return part, rm, to_device(rid)

    def prefiltered(self, allow):
        """The index restricted to the rows whose id is selected by `allow` (bool over row ids).  Under a prefilter the
        reference visits a partition's rows in storage order, skips the unselected ones and feeds the rest to the same
        k-heap (flat/index.rs:129-165); for 8-bit PQ `distance(id)` sums the same table entries in the same order as
        `distance_all` (pq/storage.rs:893-960), so searching a compacted copy of the storage -- unselected rows dropped,
        order kept -- gives bit-identical results (tests/test_oracle_golden.py::test_prefilter_equals_compaction).  The
        compacted copy is built by lance_hip_index_create, which already drops rows without a partition.  One O(n) pass
        per distinct filter; queries sharing a filter should share the returned index."""
        from .engine import DeviceIndex
        if self.params.num_bits == 4:
            # 4-bit PQ: the reference scores FILTERED rows with the unquantised f32 table (pq/storage.rs:893-921), a different arithmetic
            # from its unfiltered fast-scan -- a compacted copy would be searched with the wrong one.  The masked kernels implement it
            # (lance_hip_ivfpq_search_filtered, search.hip pq4_masked_row): hand back a view that carries the mask into every search.
            return _MaskedIndexView(self, allow)
        part, codes, rid = self._storage_rows()
        allow_t = to_device(np.ascontiguousarray(allow, dtype=bool)) if not isinstance(allow, torch.Tensor) else allow.to(part.device)
        ids = torch.arange(part.numel(), device=part.device) if rid is None else rid
        inside = ids < allow_t.numel()
        sel = torch.zeros_like(inside)
        sel[inside] = allow_t[ids[inside]]
        masked = torch.where(sel, part, torch.full_like(part, -1))          # -1 = LANCE_HIP_NONE: dropped by index_create
        ix = self._ix
        dtype = {torch.float16: "float16", torch.int8: "int8"}.get(ix.data_dtype, "float32")
        sub = DeviceIndex.create(ix.engine, ix.metric, ix.centroids, ix.codebook, masked, codes, rid, raw=ix._raw, dtype=dtype)
        return IvfPqIndex(sub, self.params, self.stats, masked, codes if rid is None else None)

    def search_device(self, q, k, nprobes, refine_factor=0, out=None, sync=True, engine=None):
        return self._ix.search(q, k, nprobes, refine_factor, out=out, sync=sync, engine=engine)

    def save(self, index_dir):
        """Writes `index.idx` + `auxiliary.idx` under index_dir -- the files IvfIndexBuilder::merge_partitions produces
        (rust/lance/src/index/vector/builder.rs:938-1079), loadable by `load_index` here (and laid out for the
        reference's IvfQuantizationStorage reader)."""
        self._ix.save(index_dir, None if self.stats is None else self.stats.ivf_loss)

    def to_arrow_artifacts(self, batch_size=10240):
        """-> (ivf_centroids RecordBatch, pq_codebook RecordBatch, iterator of shuffle-buffer RecordBatches): the three
        arguments `Dataset.create_index(..., ivf_centroids=, pq_codebook=, precomputed_shuffle_buffers=)` takes
        (python/python/lance/dataset.py:2780-2954, vector.py:659-665); see lance_amd/arrow_io.py."""
        from . import arrow_io
        rid, part, codes = self.shuffle_buffers()
        return (arrow_io.ivf_centroids_batch(self.centroids), arrow_io.pq_codebook_batch(self.codebook),
                arrow_io.shuffle_buffer_batches(rid, part, codes, batch_size))


class IvfFlatIndex:
    """IVF_FLAT: IVF partitions over the raw vectors (exact distances inside the probed partitions)."""

    def __init__(self, ix, params, stats, part_ids):
        self._ix = ix
        self.params = params
        self.stats = stats
        self.part_ids = part_ids

    @property
    def centroids(self):
        return self._ix.centroids.cpu().numpy()

    def search_device(self, q, k, nprobes):
        return self._ix.search(q, k, nprobes)

    def nearest(self, q, k=10, nprobes=1, prefilter=None):
        """prefilter: boolean array over row ids; the mask is tested inside the scan kernels (no copy of the index)"""
        ids, dists = self._ix.search(q, k, nprobes) if prefilter is None else self._ix.search(q, k, nprobes, allow=prefilter)
        return ids.cpu().numpy().view(np.uint64), dists.cpu().numpy()

    def prefiltered(self, allow):
        """A compacted copy of the index restricted to the selected rows (kept for callers that reuse one filter for many
        batches; `nearest(prefilter=)` does not need it).  FlatIndex::search scores each selected row with the same distance
        function as the unfiltered scan (flat/storage.rs:345-402), so the compacted copy is exact."""
        from .engine import DeviceFlatIndex
        if self.part_ids is None or getattr(self, "_x", None) is None:
            raise NotImplementedError("prefilter needs the index's vectors and partition ids (an index built by create_index)")
        part = self.part_ids
        allow_t = to_device(np.ascontiguousarray(allow, dtype=bool)) if not isinstance(allow, torch.Tensor) else allow.to(part.device)
        ids = torch.arange(part.numel(), device=part.device)
        inside = ids < allow_t.numel()
        sel = torch.zeros_like(inside)
        sel[inside] = allow_t[ids[inside]]
        masked = torch.where(sel, part, torch.full_like(part, -1))
        sub = DeviceFlatIndex.create(self._ix.engine, self._ix.metric, self._ix.centroids, self._x, masked)
        out = IvfFlatIndex(sub, self.params, self.stats, masked)
        out._x = self._x
        return out

    def save(self, index_dir):
        self._ix.save(index_dir, None if self.stats is None else self.stats.ivf_loss)


def load_index(index_dir, dtype=None, raw=None, engine=None):
    """Opens an index directory (`index.idx` + `auxiliary.idx`, written by the reference or by `save`) straight into
    HBM -> IvfPqIndex | IvfFlatIndex.  dtype: element type of the indexed column when it differs from the stored model
    tensors ("int8" columns keep an f32 model); raw: the column's vectors for refine, indexed by row id."""
    from . import index_file
    from .engine import DeviceFlatIndex, DeviceIndex
    eng = engine or default_engine()
    c = index_file.read_index_files(index_dir, with_rows=False)    # model + metadata only: the rows go files -> HBM natively
    params = IvfPqParams(num_partitions=c.centroids.shape[0], num_sub_vectors=c.num_sub_vectors, num_bits=c.nbits or 8,
                         metric=c.metric)
    stats = BuildStats(ivf_loss=c.loss if c.loss is not None else 0.0)
    if c.index_type == "IVF_PQ":
        return IvfPqIndex(DeviceIndex.load(eng, index_dir, dtype=dtype, raw=raw), params, stats)
    return IvfFlatIndex(DeviceFlatIndex.load(eng, index_dir, dtype=dtype), params, stats, None)


def _sample_rows(n, size, rng):
    """maybe_sample_training_data (rust/lance/src/index/vector/utils.rs:173): all rows when the
    table is small, else `size` distinct random rows (ascending)."""
    if n <= size:
        return None
    return np.sort(rng.choice(n, size=size, replace=False))


def train_ivf_centroids(x, params: IvfPqParams, engine=None, init=None):
    """build_ivf_model (rust/lance/src/index/vector/ivf.rs:1213-1272): sample num_partitions*sample_rate
    rows, normalise for cosine, drop non-finite rows, k-means with balance factor 1.0 (:1846-1871)."""
    eng = engine or default_engine()
    metric = _normalize_metric_type(params.metric)
    x = to_device(x)
    rng = np.random.default_rng(params.seed)
    idx = _sample_rows(x.shape[0], params.num_partitions * params.sample_rate, rng)
    sample = x if idx is None else x[torch.from_numpy(idx).to(x.device)]
    if metric == "cosine":
        sample = eng.normalize(sample)
    sample = sample[torch.isfinite(sample).all(dim=1)]
    kmetric = "l2" if metric == "cosine" else metric
    pool = _hier_engine_pool(eng) if (params.num_partitions > 256 and isinstance(sample, torch.Tensor) and sample.is_cuda) else None
    if pool and len(pool) > 1:
        # k > 256: the reference trains hierarchically (kmeans.rs:1027) -- thousands of small k-means, one after the other.  The splits the
        # reference is about to pop are computed side by side on several engine contexts of this GPU and applied in its order (the
        # multi-GPU trainer of lance_amd/dist.py on one rank): same centroids bit for bit, a fraction of the wall time.
        from . import dist as _ld
        f16 = sample.dtype == torch.float16
        cent = _ld.train_kmeans_hierarchical_sharded(eng, sample, params.num_partitions, max_iters=params.max_iters, balance_factor=1.0,
                                                     seed=params.seed, metric=kmetric, group=False, engines=pool)
        return (cent.to(torch.float16) if f16 else cent), 0.0, 0      # "Loss is not meaningful for hierarchical clustering" (kmeans.rs:1001)
    return eng.kmeans_train(sample, params.num_partitions, max_iters=params.max_iters, balance_factor=1.0, init=init,
                            seed=params.seed, metric=kmetric)


_HIER_POOLS = {}


def _hier_engine_pool(eng):
    """engine contexts (own HIP stream + scratch arena each) the hierarchical trainer's splits run on side by side; LANCE_HIP_HIER_CONTEXTS
    sets the count (default 8; 1 = the library's own sequential loop)"""
    import os
    want = int(os.environ.get("LANCE_HIP_HIER_CONTEXTS", "8"))
    if want <= 1 or not hasattr(eng, "device"):
        return None
    key = id(eng)
    if key not in _HIER_POOLS:
        from .engine import Engine
        _HIER_POOLS[key] = [eng] + [Engine(device=eng.device) for _ in range(want - 1)]
    return _HIER_POOLS[key]


def pq_sample_indices(n, params: IvfPqParams):
    """The rows the PQ codebook is trained on (maybe_sample_training_data, rust/lance/src/index/vector/utils.rs:173): a pure function of
    (n, params) -- create_index draws it on a host thread WHILE the IVF k-means runs on the device (2.7 ms of numpy at n = 1M that used to
    sit inside train_pq with the GPU idle)."""
    rng = np.random.default_rng(params.seed + 1)
    return _sample_rows(n, params.sample_rate * (1 << params.num_bits), rng)


_UNSET = object()


def train_pq_codebook(x, centroids, params: IvfPqParams, engine=None, sample_idx=_UNSET):
    """load_or_build_quantizer (rust/lance/src/index/vector/builder.rs:399-466): sample
    sample_rate * 2^nbits rows, normalise (cosine), drop non-finite, residual vs the IVF centroids
    (L2/cosine), then PQBuildParams::build (L2 k-means per sub-vector).  sample_idx: the result of pq_sample_indices when the
    caller already has it (None = all rows), or a concurrent.futures.Future of it."""
    eng = engine or default_engine()
    metric = _normalize_metric_type(params.metric)
    x = to_device(x)
    if sample_idx is _UNSET:
        idx = pq_sample_indices(x.shape[0], params)
    else:
        idx = sample_idx.result() if hasattr(sample_idx, "result") else sample_idx
    sample = x if idx is None else x[torch.from_numpy(idx).to(x.device)]
    if metric == "cosine":
        sample = eng.normalize(sample)
    sample = sample[torch.isfinite(sample).all(dim=1)]
    if metric in ("l2", "cosine"):
        part, _ = eng.assign(sample, centroids, "l2")
        sample = eng.residual(sample, centroids, part)
    return eng.pq_train(sample, params.num_sub_vectors, params.num_bits, params.max_iters, params.sample_rate, params.seed + 2)


def create_index(x, index_type="IVF_PQ", metric="l2", num_partitions=256, num_sub_vectors=16, num_bits=8, max_iters=50,
                 sample_rate=256, ivf_centroids=None, pq_codebook=None, seed=42, keep_raw=True, engine=None):
    """Dataset.create_index(column, "IVF_PQ", ...) for a vector matrix resident (or copied) in HBM."""
    # ---- argument rules of Dataset.create_index (python/python/lance/dataset.py:2708-2960), checked before any device work
    if not isinstance(metric, str):
        raise ValueError(f"Metric {metric} not supported.")
    metric_n = _normalize_metric_type(metric)
    itype = str(index_type).upper()
    if itype not in ("IVF_PQ", "IVF_FLAT"):
        raise NotImplementedError(f"index_type {index_type}: IVF_PQ and IVF_FLAT are on this engine's hot path")
    if isinstance(num_partitions, float):
        import warnings
        warnings.warn("num_partitions is float, converting to int")
        num_partitions = int(num_partitions)
    elif num_partitions is not None and not isinstance(num_partitions, (int, np.integer)):
        raise TypeError(f"num_partitions must be int, got {type(num_partitions)}")
    shape = tuple(x.shape)
    if len(shape) != 2:
        raise TypeError(f"Vector column must be a 2-D (rows, dimension) array, got shape {shape}")
    d = shape[1]
    if "PQ" in itype:
        if num_sub_vectors is None or num_partitions is None:
            raise ValueError("num_partitions and num_sub_vectors are required for IVF_PQ")
        if d % num_sub_vectors != 0:
            raise ValueError(f"dimension ({d}) must be divisible by num_sub_vectors ({num_sub_vectors})")
    if ivf_centroids is None and pq_codebook is not None:
        raise ValueError("ivf_centroids must be specified when pq_codebook is provided")
    as_np = lambda a: a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    if ivf_centroids is not None:
        ivf_centroids = as_np(ivf_centroids)
        if ivf_centroids.ndim != 2 or ivf_centroids.shape[0] != num_partitions:
            raise ValueError(f"Ivf centroids must be 2D array: (clusters, dim), got {ivf_centroids.shape}")
        if ivf_centroids.dtype not in (np.float16, np.float32, np.float64):
            raise TypeError("IVF centroids must be floating number" + f"got {ivf_centroids.dtype}")
    if pq_codebook is not None:
        pq_codebook = as_np(pq_codebook)
        if pq_codebook.ndim != 3 or pq_codebook.shape[0] != num_sub_vectors or pq_codebook.shape[1] != (1 << num_bits):
            raise ValueError(f"PQ codebook must be 3D array: (sub_vectors, {1 << num_bits}, dim), got {pq_codebook.shape}")
        if pq_codebook.dtype not in (np.float16, np.float32, np.float64):
            raise TypeError("PQ codebook must be floating number" + f"got {pq_codebook.dtype}")
    eng = engine or default_engine()
    params = IvfPqParams(num_partitions, num_sub_vectors, num_bits, metric_n, max_iters, sample_rate, seed)
    x = to_device(x)
    n, d = x.shape
    if x.dtype == torch.int8 and params.metric == "cosine":
        raise NotImplementedError("int8 vectors with the cosine metric are not supported by this engine (use l2 or dot)")
    stats = BuildStats()

    def timed(name, fn):
        torch.cuda.synchronize()
        t = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        stats.seconds[name] = time.perf_counter() - t
        return out

    pq_idx = _UNSET
    pq_pool = None
    if itype == "IVF_PQ" and pq_codebook is None and ivf_centroids is None:
        # the PQ training sample depends on (n, params) only: drawn on a host thread while the IVF k-means occupies the device
        from concurrent.futures import ThreadPoolExecutor
        pq_pool = ThreadPoolExecutor(max_workers=1)
        pq_idx = pq_pool.submit(pq_sample_indices, n, params)
    if ivf_centroids is not None:
        cent = to_device(np.asarray(ivf_centroids, np.float32))
        if cent.shape != (num_partitions, d):
            raise ValueError(f"IVF centroids length mismatch: {tuple(cent.shape)} != {(num_partitions, d)}")
    else:
        try:
            cent, stats.ivf_loss, stats.ivf_iters = timed("train_ivf", lambda: train_ivf_centroids(x, params, eng))
        except BaseException:
            if pq_pool is not None:
                pq_pool.shutdown(wait=True)
            raise
    if itype == "IVF_FLAT":
        if params.metric == "cosine":
            # IvfTransformer::new_flat (rust/lance-index/src/vector/ivf.rs:147-175): rows are normalised, assigned with L2 and
            # STORED normalised; the sub-index keeps the cosine distance function (ivf/v2.rs:405-411)
            if x.dtype not in (torch.float32, torch.float16):
                raise NotImplementedError("IVF_FLAT with the cosine metric needs float32 or float16 vectors (normalize_fsl accepts float arrays only)")
            xs = timed("normalize", lambda: eng.normalize(x))
            part, _ = timed("transform", lambda: eng.assign(xs, cent, "l2"))
        else:
            xs = x
            part, _ = timed("transform", lambda: eng.assign(x, cent, params.metric))
        fx = timed("build_partitions", lambda: DeviceFlatIndex.create(eng, params.metric, cent, xs, part))
        out = IvfFlatIndex(fx, params, stats, part)
        out._x = xs if keep_raw else None     # the stored rows (borrowed): needed to re-partition under a prefilter
        return out
    if num_bits not in (4, 8):
        raise ValueError(f"ProductQuantization: num_bits {num_bits} not supported")
    if pq_codebook is not None:
        cb = to_device(np.asarray(pq_codebook, np.float32).reshape(num_sub_vectors, 1 << num_bits, d // num_sub_vectors))
    else:
        try:
            cb, stats.pq_iters = timed("train_pq", lambda: train_pq_codebook(x, cent, params, eng, sample_idx=pq_idx))
        finally:
            if pq_pool is not None:
                pq_pool.shutdown(wait=True)
    part, codes, _ = timed("transform", lambda: eng.ivfpq_encode(x, cent, cb, params.metric, want_loss=False))
    ix = timed("build_partitions", lambda: DeviceIndex.create(eng, params.metric, cent, cb, part, codes, None,
                                                              raw=x if keep_raw else None,
                                                              dtype="int8" if x.dtype == torch.int8 else None))
    # Index::prewarm (ivf/v2.rs:349-352): the search-side constants (matrix-core scan tables, the lossless u8 refine copy of an
    # integer-valued f32 column) are built here, inside the build's clock, instead of inside the first search
    timed("prewarm", ix.prewarm)
    return IvfPqIndex(ix, params, stats, part, codes)


class _MaskedIndexView:
    """`IvfPqIndex.prefiltered(allow)` of a 4-bit index: the same index, every search under the row-id mask (no copy)."""

    def __init__(self, index, allow):
        self._index, self._allow = index, allow
        self.params, self.stats = index.params, index.stats

    def nearest(self, q, k=10, nprobes=1, refine_factor=None, distance_range=None, **kw):
        return self._index.nearest(q, k=k, nprobes=nprobes, refine_factor=refine_factor, prefilter=self._allow, distance_range=distance_range, **kw)


def validate_vector_index(index, vectors, refine_factor=5, sample_size=None, pass_threshold=1.0, seed=0):
    """lance.util.validate_vector_index (python/python/lance/util.py:171-220): in-sample queries with k=1, nprobes=1 and
    a refine factor must come back at distance ~0 (|d| < 1e-6) for at least `pass_threshold` of the non-NaN vectors,
    else ValueError with the reference's message.  One batched device search instead of one query per row."""
    vecs = np.asarray(vectors.detach().cpu().numpy() if isinstance(vectors, torch.Tensor) else vectors)
    if sample_size is not None and sample_size < len(vecs):
        vecs = vecs[np.sort(np.random.default_rng(seed).choice(len(vecs), size=sample_size, replace=False))]
    ok = ~np.isnan(vecs.astype(np.float32)).any(axis=1)
    total = int(ok.sum())
    passes = 0
    if total:
        _, dist = index.nearest(vecs[ok], k=1, nprobes=1, refine_factor=refine_factor)
        passes = int((np.abs(dist[:, 0]) < 1e-6).sum())
    if total and passes / total < pass_threshold:
        raise ValueError(f"Vector index failed sanity check, only {passes}/{total} passed")
    return passes, total


def flat_knn(x, q, k=10, metric="l2", engine=None, prefilter=None):
    """Exhaustive KNN (`use_index=False`): (row ids, distances) sorted by (distance, row id).
    prefilter: boolean array over rows; the scan then covers the selected rows only, as the reference's filtered
    scan feeds KNNVectorDistanceExec (scanner.rs:3386-3411) -- row ids stay those of the full table."""
    eng = engine or default_engine()
    if prefilter is not None:
        xt = to_device(x)
        keep = torch.nonzero(to_device(np.ascontiguousarray(prefilter, dtype=bool)) if not isinstance(prefilter, torch.Tensor)
                             else prefilter.to(xt.device)).reshape(-1)
        return eng.flat_topk(xt[keep].contiguous(), q, k, _normalize_metric_type(metric), row_ids=keep)
    ids, dists = eng.flat_topk(x, q, k, _normalize_metric_type(metric))
    return ids, dists
