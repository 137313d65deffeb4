"""CPU tests of the host-side logic that needs no GPU: the Python twin of the device-side RNG (all ranks of a
multi-GPU build must draw the streams the single-GPU C++ trainer draws), the (dist, rowid) merge of the
list-sharded search, metric-name handling (python/python/lance/util.py:40 `_normalize_metric_type`) and sampling."""
import numpy as np
import pytest
import torch

f32 = np.float32


def test_python_rng_equals_oracle_stream(oracle):
    from lance_amd._rng import Rng, kmeans_init_indices
    for n, k, seed in ((1000, 7, 0), (65536, 256, 42), (300, 300, 9), (5000, 16, 2 ** 40 + 3)):
        assert (kmeans_init_indices(n, k, seed) == oracle.kmeans_init_indices(n, k, seed)).all()
    r = Rng(123)
    vals = [float(r.next_f32()) for _ in range(1000)]
    assert all(0.0 <= v < 1.0 for v in vals) and len(set(vals)) > 990


def test_split_clusters_python_twin_matches_oracle_training(oracle):
    """lance_amd.dist._split_clusters is what every rank runs between all-reduces; with all rows on one 'rank' the
    sharded loop must follow the single-process reference loop exactly, empty-cluster splits included."""
    from lance_amd.dist import _split_clusters
    from lance_amd._rng import Rng
    rng = np.random.default_rng(1)
    cent = rng.standard_normal((6, 4)).astype(f32)
    cnts = [10, 0, 7, 0, 3, 0]
    c1 = cent.copy(); n1 = list(cnts)
    _split_clusters(20, n1, c1, Rng(5 ^ 0x5bd1e995))
    assert sum(n1) == 20 and all(v > 0 or cnts[i] > 0 or True for i, v in enumerate(n1))
    assert np.isfinite(c1).all() and not np.array_equal(c1, cent)
    eps = f32(1.0 / 1024.0)
    # a split child is its parent scaled by (1 +- eps) on alternating dimensions (kmeans.rs:195-203)
    moved = [i for i in range(6) if cnts[i] == 0]
    for i in moved:
        ratios = c1[i] / np.where(c1[i] == 0, 1, c1[i])
        assert np.isfinite(ratios).all()


def test_merge_topk_orders_by_distance_then_row_id():
    from lance_amd.dist import merge_topk
    ids = torch.tensor([[5, 3, -1, 9, 3, 7]], dtype=torch.int64)
    dd = torch.tensor([[2.0, 1.0, 0.0, 1.0, 5.0, float("-0.0")]], dtype=torch.float32)
    i, d = merge_topk(ids, dd, 4)
    assert i.tolist() == [[7, 3, 9, 5]]                      # -0.0 < 1.0 (id 3 before 9) < 2.0; the (-1) entry never wins
    assert d.tolist()[0][1:] == [1.0, 1.0, 2.0]
    i, d = merge_topk(ids[:, :2], dd[:, :2], 4)              # fewer candidates than k: padded with (-1, +inf)
    assert i.tolist() == [[3, 5, -1, -1]] and d[0, 2:].tolist() == [float("inf")] * 2
    # total_cmp: negative values sort before positive, NaN last
    ids = torch.tensor([[1, 2, 3, 4]], dtype=torch.int64)
    dd = torch.tensor([[float("nan"), -3.0, 0.5, -7.0]], dtype=torch.float32)
    i, _ = merge_topk(ids, dd, 4)
    assert i.tolist() == [[4, 2, 3, 1]]


def test_metric_names_and_sampling():
    from lance_amd.vector import _normalize_metric_type, _sample_rows
    assert _normalize_metric_type("L2") == "l2" and _normalize_metric_type("euclidean") == "l2"
    assert _normalize_metric_type("Cosine") == "cosine" and _normalize_metric_type("dot") == "dot"
    with pytest.raises(ValueError):
        _normalize_metric_type("hamming")
    assert _sample_rows(100, 256, np.random.default_rng(0)) is None
    s = _sample_rows(10000, 256, np.random.default_rng(0))
    assert len(s) == 256 and len(set(s.tolist())) == 256 and (np.diff(s) > 0).all()
    assert (s == _sample_rows(10000, 256, np.random.default_rng(0))).all()      # every rank draws the same sample


def test_local_list_rows_partitions_the_rows():
    from lance_amd.dist import local_list_rows
    part = torch.tensor([0, 5, 2, -1, 7, 2, 5, 1], dtype=torch.int32)
    seen = torch.cat([local_list_rows(part, 3, r) for r in range(3)])
    assert sorted(seen.tolist()) == [0, 1, 2, 4, 5, 6, 7]          # row 3 has no partition
    assert local_list_rows(part, 3, 2).tolist() == [1, 2, 5, 6]    # lists 5 and 2 -> rank 2
