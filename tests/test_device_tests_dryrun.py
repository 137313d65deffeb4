"""Runs the bodies of the `-m gpu` tests that have not met hardware yet with the device classes replaced by oracle-backed
stand-ins (tests/oracle_engine.py).  This does not test kernels; it catches plumbing mistakes -- argument order, shapes,
dtypes, attribute names, the tests' own expectations -- in the host layer those tests drive (lance_amd/vector.py load /
save / prefilter, dist.load_list_shard, accelerator.py) and in the tests themselves, here where no GPU exists.  The index
files they write and read go through the real native reader/writer."""
import pytest

import oracle_engine
import test_gpu_parity as G


@pytest.fixture
def fake(monkeypatch):
    return oracle_engine.install(monkeypatch)


def test_dryrun_accelerator_module(fake):
    G.test_accelerator_module_on_device(fake)


def test_dryrun_prefilter(fake, oracle):
    import lance_amd
    G.test_prefilter_matches_reference_branch(lance_amd, oracle)


def test_dryrun_reference_stored_artefacts(fake, oracle):
    G.test_gpu_reproduces_what_the_reference_stored(fake, oracle)


def test_dryrun_load_reference_index(fake, oracle):
    G.test_load_reference_written_index_and_search(fake, oracle)


@pytest.mark.parametrize("kind", ["f32", "f16", "4bit", "dot", "cosine"])
def test_dryrun_index_roundtrip(fake, oracle, tmp_path, kind):
    G.test_index_save_load_roundtrip.__wrapped__(fake, oracle, tmp_path, kind) if hasattr(G.test_index_save_load_roundtrip, "__wrapped__") \
        else G.test_index_save_load_roundtrip(fake, oracle, tmp_path, kind)


def test_dryrun_ivf_flat_roundtrip(fake, oracle, tmp_path):
    G.test_ivf_flat_save_load_roundtrip(fake, oracle, tmp_path)


def test_dryrun_legacy_index(fake, oracle, tmp_path):
    G.test_load_legacy_reference_index_c2_shape(fake, oracle, tmp_path)


def test_dryrun_load_list_shard(fake, oracle, tmp_path):
    G.test_load_list_shard_world1(fake, oracle, tmp_path)


@pytest.mark.parametrize("metric", ["l2", "dot"])
def test_dryrun_distance_range(fake, oracle, metric):
    G.test_distance_range_search(fake, oracle, metric)


@pytest.mark.parametrize("world", [2, 3])
def test_dryrun_load_lists_shards(fake, oracle, tmp_path, world):
    G.test_load_lists_shards_on_one_gpu(fake, oracle, tmp_path, world)


# ---- tests that HAVE run on hardware but go through host code edited since (create_index argument checks, nearest /
# ---- flat_knn signatures, IvfFlatIndex construction): a regression there would only show on the GPU box otherwise
def test_dryrun_python_api_end_to_end(fake, oracle):
    import lance_amd
    G.test_python_api_end_to_end(lance_amd, oracle)


@pytest.mark.parametrize("metric,d", [("l2", 128), ("dot", 40)])
def test_dryrun_ivf_flat_matches_oracle(fake, oracle, metric, d):
    G.test_ivf_flat_matches_oracle(fake, oracle, metric, d)


@pytest.mark.parametrize("d", [128, 40])
def test_dryrun_ivf_flat_cosine(fake, oracle, d, tmp_path):
    G.test_ivf_flat_cosine_matches_oracle(fake, oracle, d, tmp_path)


def test_dryrun_list_sharded_world1(fake, oracle):
    G.test_list_sharded_search_on_device_world1(fake, oracle)


def test_dryrun_smoke_flow(fake, oracle):
    """the calls __graft_entry__.smoke() makes, in the same order"""
    import numpy as np
    import lance_amd
    rng = np.random.default_rng(0)
    centers = rng.uniform(0, 128, (16, 32))
    x = np.clip(np.rint(centers[rng.integers(0, 16, 8000)] + rng.normal(0, 20, (8000, 32))), 0, 218).astype(np.float32)
    q = x[:64] + 1.0
    idx = lance_amd.create_index(x, "IVF_PQ", metric="l2", num_partitions=16, num_sub_vectors=4, max_iters=8)
    ids, dists = idx.nearest(q, k=10, nprobes=16)
    oi, od = oracle.build_index(x, idx.centroids, idx.codebook).search(q, 10, 16)
    assert (ids.view(np.uint64) == oi).all() and (dists.view(np.uint32) == od.view(np.uint32)).all()


def test_dryrun_adaptive_probing(fake, oracle):
    import lance_amd
    import test_gpu_pm_scan as P
    P.test_adaptive_probing_extends_starved_queries(lance_amd, oracle)


@pytest.mark.parametrize("metric", ["l2", "dot"])
def test_dryrun_4bit(fake, oracle, metric):
    """everything up to the single-partition entry point (which the stand-in engine does not model): build, search, the
    prefilter branch and prefilter + range"""
    try:
        G.test_4bit_pq_bit_exact(fake, oracle, metric)
    except AttributeError as e:
        assert "pq_scan_topk" in str(e)


# ---- Float16 columns under dot / cosine (tests/test_zz_gpu_f16_metrics.py): the host layer and the tests' own expectations
@pytest.mark.parametrize("d", [32, 56, 20])
def test_dryrun_f16_dot_assign(fake, oracle, d):
    import test_zz_gpu_f16_metrics as Z
    Z.test_f16_dot_assign_and_find_partitions(fake, oracle, d)


def test_dryrun_f16_normalize_and_kmeans(fake, oracle):
    import test_zz_gpu_f16_metrics as Z
    Z.test_f16_normalize_half_precision(fake, oracle)
    Z.test_f16_dot_kmeans_training(fake, oracle)


@pytest.mark.parametrize("metric", ["dot", "cosine"])
def test_dryrun_f16_index(fake, oracle, metric):
    import test_zz_gpu_f16_metrics as Z
    Z.test_f16_index_build_search_refine(fake, oracle, metric)
    Z.test_f16_python_api(fake, oracle, metric)


@pytest.mark.parametrize("metric", ["dot", "cosine"])
def test_dryrun_f16_flat_and_ivfflat(fake, oracle, metric):
    import test_zz_gpu_f16_metrics as Z
    Z.test_f16_flat_and_ivfflat(fake, oracle, metric, 40)


# ---- wide rows on the matrix cores (tests/test_zz_gpu_wide_mfma.py): the tests' own expectations
def test_dryrun_wide_rows(fake, oracle):
    import test_zz_gpu_wide_mfma as W
    W.test_wide_rows_assign_on_matrix_cores(fake, oracle, 144, "dot")
    W.test_wide_rows_assign_on_matrix_cores(fake, oracle, 200, "l2")
    W.test_wide_rows_kmeans_and_encode_chain(fake, oracle)


def test_dryrun_ivfflat_ties(fake, oracle):
    import test_zz_gpu_zz_ivfflat_ties as T
    T.test_ivf_flat_more_ties_than_the_pool_holds(oracle)
    T.test_ivf_flat_f16_cosine_norm_overflow_gives_nan_like_the_reference(oracle)
