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
