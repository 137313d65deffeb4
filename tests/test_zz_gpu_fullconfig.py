"""BASELINE configs 3 and 5 at their REAL index parameters (C3: d 1536 cosine, nlist 1024 hierarchical, M 96; C5: int8 rows,
nlist 65,536, M 32) end to end on the GPU -- training, assign, encode, storage layout, find_partitions, searches up to the
exhaustive probe (v2.rs:1354-1381: nprobes = nlist), refine, flat ground truth -- against results the CPU oracle produced once
in the build container (tests/golden/fullconfig.npz, written by tests/golden/make_fullconfig_golden.py; large arrays are
recorded as SHA-256 digests, search results verbatim).  Every surrogate is on its default setting: K-tiled / register-resident
MFMA assign, integer bound pass, u16 filter scan, exact re-evaluation."""
import os

import numpy as np
import pytest

from fullconfig_spec import C3, C4, C5, C5T, c3_data, c4_data, c5_data, digest, f32

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullconfig.npz")


def _np(t):
    return t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)


@pytest.fixture(scope="module")
def eng():
    import lance_amd
    return lance_amd.default_engine()


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def same(a, d):
    return (digest(_np(a)) == d).all()


def test_c3_dbpedia_parameters_100k_rows(eng, gold):
    from lance_amd.engine import DeviceIndex
    c = C3
    x, q = c3_data()
    xs = eng.normalize(x)
    cent, _, _ = eng.kmeans_train(xs, c["nlist"], max_iters=c["ivf_iters"], balance_factor=1.0, seed=c["seed"])
    assert cent.shape[0] == c["nlist"]
    assert same(_np(cent).astype(f32), gold["c3_centroids"]), "hierarchical IVF centroids differ from the oracle's"
    part, _ = eng.assign(xs, cent, "l2")
    res = eng.residual(xs, cent, part)
    cb, its = eng.pq_train(res[:65536], c["m"], max_iters=c["pq_iters"], seed=c["seed"] + 1)
    assert (_np(its).astype(np.uint32) == gold["c3_pq_iters"]).all()
    assert same(_np(cb).astype(f32), gold["c3_codebook"]), "PQ codebook differs from the oracle's"
    gpart, gcodes, _ = eng.ivfpq_encode(x, cent, cb, "cosine")
    assert same(_np(gpart).view(np.uint32), gold["c3_part_ids"]) and same(_np(gcodes), gold["c3_codes"])
    g = DeviceIndex.create(eng, "cosine", cent, cb, gpart, gcodes, None, raw=x)
    offs, _, _ = g.export()
    assert (offs == gold["c3_part_offsets"]).all()
    for (k, nprobes, rf) in c["searches"]:
        gi, gd = g.search(q, k, nprobes, rf)
        assert (_np(gi).view(np.uint64) == gold[f"c3_ids_{k}_{nprobes}_{rf}"]).all(), (k, nprobes, rf)
        assert (_np(gd).view(np.uint32) == gold[f"c3_dists_{k}_{nprobes}_{rf}"].view(np.uint32)).all(), (k, nprobes, rf)
    # a batch large enough for the partition-major path (nq * nprobes >= 4096): the same 200 queries tiled, same answers per copy
    qq = np.tile(q, (3, 1))
    gi, gd = g.search(qq, 10, 10, 10)
    want = np.tile(gold["c3_ids_10_10_10"], (3, 1))
    assert (_np(gi).view(np.uint64) == want).all()
    assert (_np(gd).view(np.uint32) == np.tile(gold["c3_dists_10_10_10"], (3, 1)).view(np.uint32)).all()
    gi, gd = eng.flat_topk(x, q[:50], 10, "cosine")
    assert (_np(gi).view(np.uint64) == gold["c3_flat_ids"]).all() and (_np(gd).view(np.uint32) == gold["c3_flat_dists"].view(np.uint32)).all()
    g.close()


def test_c5_bigann_parameters_nlist_65536_int8(eng, gold):
    import torch
    from lance_amd.engine import DeviceIndex
    c = C5
    xi, qi = c5_data()
    xt, qt = torch.from_numpy(xi), torch.from_numpy(qi)
    from lance_amd._rng import kmeans_init_indices
    assert (kmeans_init_indices(c["n"], c["nlist"], c["seed"]) == gold["c5_init_rows"]).all()      # the reservoir draw, host side of the library
    cent = xi[gold["c5_init_rows"].astype(np.int64)].astype(f32)       # kmeans_random_init rows as the coarse quantiser (see the spec)
    assert same(cent, gold["c5_centroids"])
    part, _ = eng.assign(xt, cent, "l2")
    res = eng.residual(xi.astype(f32), cent, part)
    cb, its = eng.pq_train(res[:65536], c["m"], max_iters=c["pq_iters"], seed=c["seed"] + 1)
    assert (_np(its).astype(np.uint32) == gold["c5_pq_iters"]).all()
    assert same(_np(cb).astype(f32), gold["c5_codebook"])
    gpart, gcodes, _ = eng.ivfpq_encode(xt, cent, cb, "l2")
    assert same(_np(gpart).view(np.uint32), gold["c5_part_ids"]) and same(_np(gcodes), gold["c5_codes"])
    g = DeviceIndex.create(eng, "l2", cent, cb, gpart, gcodes, None, raw=xt, dtype="int8")
    offs, _, _ = g.export()
    assert same(offs.astype(np.uint32), gold["c5_part_offsets_digest"])
    pi, pd = eng.find_partitions(qt[:200], cent, 64, "l2")
    assert (_np(pi).view(np.uint32) == gold["c5_probe_ids"]).all() and (_np(pd).view(np.uint32) == gold["c5_probe_dists"].view(np.uint32)).all()
    for (k, nprobes, rf) in c["searches"]:
        gi, gd = g.search(qt, k, nprobes, rf)
        assert (_np(gi).view(np.uint64) == gold[f"c5_ids_{k}_{nprobes}_{rf}"]).all(), (k, nprobes, rf)
        assert (_np(gd).view(np.uint32) == gold[f"c5_dists_{k}_{nprobes}_{rf}"].view(np.uint32)).all(), (k, nprobes, rf)
    g.close()


def test_c4_f16_rows_hierarchical_nlist_4096_one_million_rows(eng, gold):
    """BASELINE config 4 at its real index parameters on 1/100 of its rows: Float16 column, the hierarchical trainer in its
    Float16Type instantiation up to nlist 4096, M 16; searches at 1 / 10 / 50 probes and the exhaustive probe (nprobes = nlist)."""
    import torch
    from lance_amd.engine import DeviceIndex
    if "c4_centroids" not in gold:
        pytest.fail("tests/golden/fullconfig.npz carries no c4 record (python tests/golden/make_fullconfig_golden.py c4)")
    c = C4
    x, q = c4_data()
    xt = torch.from_numpy(x)
    cent, _, _ = eng.kmeans_train(xt, c["nlist"], max_iters=c["ivf_iters"], balance_factor=1.0, seed=c["seed"])
    assert cent.shape[0] == c["nlist"]
    cent_h = _np(cent).astype(np.float16)
    assert same(cent_h, gold["c4_centroids"]), "f16 hierarchical IVF centroids differ from the oracle's"
    part, _ = eng.assign(xt, cent_h, "l2")
    res = eng.residual(xt, cent_h, part)
    cb, its = eng.pq_train(res[:65536], c["m"], max_iters=c["pq_iters"], seed=c["seed"] + 1)
    assert (_np(its).astype(np.uint32) == gold["c4_pq_iters"]).all()
    cb_h = _np(cb).astype(np.float16)
    assert same(cb_h, gold["c4_codebook"]), "PQ codebook differs from the oracle's"
    gpart, gcodes, _ = eng.ivfpq_encode(xt, cent_h, cb_h, "l2")
    assert same(_np(gpart).view(np.uint32), gold["c4_part_ids"]) and same(_np(gcodes), gold["c4_codes"])
    g = DeviceIndex.create(eng, "l2", cent_h, cb_h, gpart, gcodes, None, raw=xt, dtype="float16")
    offs, _, _ = g.export()
    assert (offs == gold["c4_part_offsets"]).all()
    for (k, nprobes, rf) in c["searches"]:
        gi, gd = g.search(q, k, nprobes, rf)
        assert (_np(gi).view(np.uint64) == gold[f"c4_ids_{k}_{nprobes}_{rf}"]).all(), (k, nprobes, rf)
        assert (_np(gd).view(np.uint32) == gold[f"c4_dists_{k}_{nprobes}_{rf}"].view(np.uint32)).all(), (k, nprobes, rf)
    # the partition-major path (nq * nprobes >= 4096): the same queries tiled
    qq = np.tile(q, (3, 1))
    gi, gd = g.search(qq, 10, 10, 10)
    assert (_np(gi).view(np.uint64) == np.tile(gold["c4_ids_10_10_10"], (3, 1))).all()
    assert (_np(gd).view(np.uint32) == np.tile(gold["c4_dists_10_10_10"], (3, 1)).view(np.uint32)).all()
    g.close()


def test_c5_coarse_quantiser_trained_to_65536_by_the_engine(eng, gold):
    """Config 5's nlist built for real: the hierarchical trainer (the reference's route for k > 256, kmeans.rs:746-1003) run
    to 65,536 centroids on the device -- ~4,400 cluster splits, the last levels with a handful of rows per cluster -- equal to
    the oracle's run centroid for centroid, then the index and its searches."""
    import torch
    from lance_amd.engine import DeviceIndex
    if "c5t_centroids" not in gold:
        pytest.fail("tests/golden/fullconfig.npz carries no c5t record (python tests/golden/make_fullconfig_golden.py c5t)")
    c, ct = C5, C5T
    xi, qi = c5_data()
    xt, qt = torch.from_numpy(xi), torch.from_numpy(qi)
    cent, _, _ = eng.kmeans_train(xt, ct["nlist"], max_iters=ct["ivf_iters"], balance_factor=1.0, seed=ct["seed"])
    cent = _np(cent).astype(f32)
    assert cent.shape[0] == int(gold["c5t_ncent"][0])
    assert same(cent, gold["c5t_centroids"]), "hierarchical centroids (target 65,536) differ from the oracle's"
    part, _ = eng.assign(xt, cent, "l2")
    res = eng.residual(xi.astype(f32), cent, part)
    cb, _ = eng.pq_train(res[:65536], c["m"], max_iters=c["pq_iters"], seed=ct["seed"] + 1)
    assert same(_np(cb).astype(f32), gold["c5t_codebook"])
    gpart, gcodes, _ = eng.ivfpq_encode(xt, cent, cb, "l2")
    assert same(_np(gpart).view(np.uint32), gold["c5t_part_ids"]) and same(_np(gcodes), gold["c5t_codes"])
    g = DeviceIndex.create(eng, "l2", cent, cb, gpart, gcodes, None, raw=xt, dtype="int8")
    for (k, nprobes, rf) in ct["searches"]:
        gi, gd = g.search(qt, k, nprobes, rf)
        assert (_np(gi).view(np.uint64) == gold[f"c5t_ids_{k}_{nprobes}_{rf}"]).all(), (k, nprobes, rf)
        assert (_np(gd).view(np.uint32) == gold[f"c5t_dists_{k}_{nprobes}_{rf}"].view(np.uint32)).all(), (k, nprobes, rf)
    g.close()
