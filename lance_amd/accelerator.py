"""The reference's accelerator operator module, on the MI355X engine.

pylance's `accelerator=` build path (python/python/lance/vector.py:134-755) is written against two small modules:
`lance.torch.distance` (`l2_distance`, `dot_distance`, `cosine_distance`: nearest centroid id + distance per row) and
`lance.torch.kmeans.KMeans` (`fit` / `transform` / `.centroids`).  This file exposes the same names, signatures, return
conventions (ids int64 with -1 for rows whose distances are NaN, distances float32, tensors on the device) and error
messages, so `train_ivf_centroids_on_accelerator`, `compute_partitions` and `compute_pq_codes` can run on this engine
by importing from here instead -- INTEGRATION.md section 2.

Differences that are deliberate: training follows the Rust k-means of the reference's CPU path (kmeans.rs:610-719:
distinct random rows as the init, loss-based convergence, empty-cluster splits) rather than the torch re-implementation
(python/python/lance/torch/kmeans.py:202-280), so an index built through this module equals one built by
`create_index` without an accelerator; `tolerance` is that loop's relative loss tolerance.  Distances are the exact
l2_scalar / dot_scalar values, not `cdist`-then-square.
"""
from typing import Optional, Tuple

import numpy as np
import torch

from .vector import _normalize_metric_type, default_engine

__all__ = ["l2_distance", "dot_distance", "cosine_distance", "KMeans"]


def _check_2d(x, y, xn="x", yn="y"):
    if len(x.shape) != 2 or len(y.shape) != 2:
        raise ValueError(f"x and y must be 2-D matrix, got: {xn}.shape={x.shape}, {yn}.shape={y.shape}")


def _assign(vectors, centroids, metric, engine=None):
    eng = engine or default_engine()
    ids, dists = eng.assign(vectors, centroids, metric)
    idx = ids.to(torch.int64)                      # LANCE_HIP_NONE (no finite distance) is -1 as int32 already
    dists = torch.where(idx < 0, torch.full_like(dists, float("nan")), dists)
    return idx, dists


def l2_distance(vectors: torch.Tensor, centroids: torch.Tensor, y2: Optional[torch.Tensor] = None,
                engine=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """lance.torch.distance.l2_distance (distance.py:204-238): (nearest centroid id, squared L2 distance) per row; rows
    with NaN distances get id -1.  y2 (pre-computed centroid norms) is accepted and ignored."""
    _check_2d(vectors, centroids)
    return _assign(vectors, centroids, "l2", engine)


def dot_distance(x: torch.Tensor, y: torch.Tensor, engine=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """lance.torch.distance.dot_distance (distance.py:241-266): argmin and value of `1 - x . y`."""
    _check_2d(x, y)
    return _assign(x, y, "dot", engine)


def cosine_distance(vectors: torch.Tensor, centroids: torch.Tensor, engine=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """lance.torch.distance.cosine_distance (distance.py:84-115).  Computed the way the reference's index does
    (normalise both sides, L2; cosine = L2 / 2 on unit vectors -- pq/storage.rs:931-944)."""
    if len(vectors.shape) != 2 or len(centroids.shape) != 2:
        raise ValueError(f"x and y must be 2-D matrix, got: vectors.shape={vectors.shape}, centroids.shape={centroids.shape}")
    eng = engine or default_engine()
    idx, d = _assign(eng.normalize(vectors), eng.normalize(centroids), "l2", eng)
    return idx, d / 2


class KMeans:
    """lance.torch.kmeans.KMeans (kmeans.py:26-330): same constructor, `fit`, `transform`, `centroids`."""

    def __init__(self, k: int, *, metric: str = "l2", init: str = "random", max_iters: int = 50, tolerance: float = 1e-4,
                 centroids: Optional[torch.Tensor] = None, seed: Optional[int] = None, device: Optional[str] = None, engine=None):
        self.k = k
        self.max_iters = max_iters
        self.metric = _normalize_metric_type(metric)
        if init != "random":
            raise ValueError(f"Only random initialization is supported, got: {init}")
        self.init = init
        self.tolerance = tolerance
        self.centroids = centroids
        self.seed = seed
        self.device = device
        self.total_distance = 0
        self._engine = engine

    def __repr__(self):
        return f"KMeans(k={self.k}, metric={self.metric}, device={self.device or 'rocm'})"

    def _eng(self):
        if self._engine is None:
            self._engine = default_engine()
        return self._engine

    @staticmethod
    def _to_tensor(data):
        if isinstance(data, torch.Tensor):
            return data
        if isinstance(data, np.ndarray):
            return torch.from_numpy(np.ascontiguousarray(data))
        if hasattr(data, "values") and hasattr(data, "type") and hasattr(data.type, "list_size"):   # pa.FixedSizeListArray
            return torch.from_numpy(np.ascontiguousarray(data.values.to_numpy(zero_copy_only=False).reshape(-1, data.type.list_size)))
        raise ValueError("KMeans::fit accepts pyarrow FixedSizeListArray" + f"np.ndarray or torch.Tensor, got: {type(data)}")

    def fit(self, data, column: Optional[str] = None) -> None:
        """Train.  An iterable of batches (tensors, or dicts holding `column`) is concatenated first: the training sample
        of the largest BASELINE configuration is 8.6 GB, a small part of one GPU's HBM."""
        if not isinstance(data, (torch.Tensor, np.ndarray)) and not hasattr(data, "type") and hasattr(data, "__iter__"):
            parts = []
            for batch in data:
                if isinstance(batch, dict):
                    if column is None:
                        raise ValueError("column must be given when the batches are dictionaries")
                    batch = batch[column]
                parts.append(self._to_tensor(batch).reshape(-1, self._to_tensor(batch).shape[-1]))
            data = torch.cat(parts)
        x = self._to_tensor(data)
        eng = self._eng()
        metric = self.metric
        if metric == "cosine":      # kmeans.py:288-289 normalises; the centroids then live on the unit-vector side
            x = eng.normalize(x)
            metric = "l2"
        self.centroids, self.total_distance, self.iters = eng.kmeans_train(
            x, self.k, max_iters=self.max_iters, tol=self.tolerance, init=self.centroids,
            seed=0 if self.seed is None else self.seed, metric=metric)

    def _transform(self, data, y2=None):
        eng = self._eng()
        if self.metric == "cosine":
            data = eng.normalize(data)
        if self.metric in ("l2", "cosine"):
            return l2_distance(data, self.centroids, engine=eng)
        return dot_distance(data, self.centroids, engine=eng)

    def transform(self, data) -> torch.Tensor:
        """Cluster id of each row (int64, -1 for rows without a finite distance)."""
        assert self.centroids is not None
        return self._transform(self._to_tensor(data))[0]
