"""GPU parity of Float16 columns under the dot and cosine metrics.

A Float16 column takes half::f16's own arms of the reference's distance traits (lance-linalg, no `fp16kernels` feature):
  Dot       dot_scalar::<f16, f32, 32>            dot.rs:91-102,138-161   (32 lane accumulators, not f32's 16)
  Normalize norm_l2_impl::<f16, f32, 32>          norm_l2.rs:60-85
  Cosine    the trait default cosine_scalar       cosine.rs:36-45,171-179 (flat scan / refine / IVF_FLAT partitions)
  normalize_fsl::<Float16Type>                    kernels.rs:141-186      (half-precision arithmetic; cosine indices)
L2 stays l2_scalar::<f16, f32, 16>.  Every comparison is bit for bit against the oracle's f16 arms (tests/test_oracle_golden.py
pins those on the reference's own f16.c and on numpy restatements).  The file sorts last on purpose: newest device code last.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
f32 = np.float32


def _np(t):
    return t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)


def f16_data(n, d, seed, ncl=16, scale=2.0):
    rng = np.random.default_rng(seed)
    c = rng.standard_normal((ncl, d)) * scale
    return (c[rng.integers(0, ncl, n)] + rng.standard_normal((n, d)) * 0.7).astype(np.float16)


@pytest.fixture(scope="module")
def eng():
    import lance_amd
    return lance_amd.default_engine()


@pytest.mark.parametrize("d", [32, 128, 48, 56, 100, 20])
def test_f16_dot_assign_and_find_partitions(eng, oracle, d):
    """coarse quantiser of an f16 column under dot: MFMA surrogate + 32-lane exact re-check (n >= 2048, k >= 32, d % 16 == 0,
    d <= 128), the two-pass 32-lane kernel of wide.hip otherwise (any d: remainders of 16, 24 (two pieces), 4, and d < 32)"""
    x = f16_data(5000, d, 11 + d)
    c0 = f16_data(40, d, 12 + d)
    for n in (5000, 700):
        ids, dists = eng.assign(x[:n], c0, "dot")
        oi, od = oracle.assign(x[:n], c0, "dot")
        assert (_np(ids).view(np.uint32) == oi).all(), (d, n)
        assert (_np(dists).view(np.uint32) == od.view(np.uint32)).all(), (d, n)
    # few centroids (k < 32: exact kernels only) and a bias-free single row
    ids, dists = eng.assign(x[:3000], c0[:7], "dot")
    oi, od = oracle.assign(x[:3000], c0[:7], "dot")
    assert (_np(ids).view(np.uint32) == oi).all() and (_np(dists).view(np.uint32) == od.view(np.uint32)).all()
    q = f16_data(90, d, 13 + d)
    gp, gpd = eng.find_partitions(q, c0, 6, "dot")
    op, opd = oracle.find_partitions(q, c0, 6, "dot")
    assert (_np(gp).view(np.uint32) == op).all() and (_np(gpd).view(np.uint32) == opd.view(np.uint32)).all()
    # ... and the same values really differ from the 16-lane order somewhere (the test would be vacuous otherwise)
    if d >= 32:
        o16, d16 = oracle.assign(x.astype(f32), c0.astype(f32), "dot")
        assert (d16.view(np.uint32) != oracle.assign(x, c0, "dot")[1].view(np.uint32)).any()


def test_f16_normalize_half_precision(eng, oracle):
    for n, d in ((3000, 40), (1000, 96), (130, 200)):        # lane-per-row kernel (d < 64) and the tiled one (ragged last tile too)
        x = (f16_data(n, d, 3 + d) * 3).astype(np.float16)
        got = eng.normalize(x)
        assert got.dtype == __import__("torch").float16
        assert (_np(got).view(np.uint16) == oracle.normalize(x).view(np.uint16)).all(), (n, d)
    xf = (f16_data(777, 1536, 9) * 3).astype(f32)              # f32 rows, C3 dimension, rows not a multiple of 64
    assert (_np(eng.normalize(xf)).view(np.uint32) == oracle.normalize(xf).view(np.uint32)).all()


def test_f16_dot_kmeans_training(eng, oracle):
    x = f16_data(6000, 32, 21)
    cent, loss, iters = eng.kmeans_train(x, 24, max_iters=12, balance_factor=1.0, seed=5, metric="dot")
    oc, ol, oit, _ = oracle.kmeans_train(x, 24, max_iters=12, balance_factor=f32(1.0) / f32(6000), seed=5, metric="dot")
    assert iters == oit and loss == ol
    assert (_np(cent).view(np.uint16) == oc.view(np.uint16)).all()


@pytest.mark.parametrize("metric", ["dot", "cosine"])
def test_f16_index_build_search_refine(eng, oracle, metric):
    """IVF_PQ over an f16 column: cosine = half-precision normalize + the f16 L2 pipeline, refine with the scalar cosine on the
    original key; dot = 32-lane coarse quantiser, table entries on sub-vectors of 8 (same additions in both orders), refine
    with the 32-lane dot.  Small batches (query-major scan) and >= 4096 pairs (partition-major / exact pair scan)."""
    from lance_amd.engine import DeviceIndex
    n, d, nlist, m = 20000, 64, 32, 8
    x = f16_data(n, d, 31)
    q = f16_data(500, d, 32)
    km = "l2" if metric == "cosine" else metric
    tr = oracle.normalize(x[:4096]) if metric == "cosine" else x[:4096]
    cent, _, _, _ = oracle.kmeans_train(tr, nlist, max_iters=8, seed=1, metric=km)
    part, _ = oracle.assign(tr, cent, km)
    res = oracle.residual(tr, cent, part) if metric == "cosine" else tr
    cb, _ = oracle.pq_train(res, m, max_iters=6, seed=2)
    oidx = oracle.build_index(x, cent, cb, metric)
    gpart, gcodes, _ = eng.ivfpq_encode(x, cent, cb, metric)
    assert (_np(gpart).view(np.uint32) == oidx.part_ids).all()
    assert (_np(gcodes) == oidx.codes_rowmajor).all()
    gidx = DeviceIndex.create(eng, metric, cent, cb, gpart, gcodes, None, raw=x)
    xf = x.astype(f32)
    for nq, k, nprobes, rf in ((150, 10, nlist, 0), (150, 10, 6, 0), (150, 10, 6, 10), (500, 10, 10, 0), (500, 10, 10, 5)):
        gi, gd = gidx.search(q[:nq], k, nprobes, rf)
        oi, od = oidx.search(q[:nq], k, nprobes, refine=rf, raw=xf)
        assert (_np(gi).view(np.uint64) == oi).all(), (metric, nq, k, nprobes, rf)
        assert (_np(gd).view(np.uint32) == od.view(np.uint32)).all(), (metric, nq, k, nprobes, rf)


@pytest.mark.parametrize("metric", ["dot", "cosine"])
@pytest.mark.parametrize("d", [64, 40])
def test_f16_flat_and_ivfflat(eng, oracle, metric, d):
    """un-indexed KNN and IVF_FLAT partitions of an f16 column score rows with f16's own dot / cosine"""
    import lance_amd
    n = 9000
    x = f16_data(n, d, 41 + d)
    x[500:520] = x[3]                                  # ties broken by row id
    q = f16_data(70, d, 42 + d)
    q[:3] = x[3]
    for k in (10, 130):
        gi, gd = eng.flat_topk(x, q, k, metric)
        oi, od = oracle.flat_knn(x, q, k, metric)
        assert (_np(gi).view(np.uint64) == oi).all(), (metric, d, k)
        assert (_np(gd).view(np.uint32) == od.view(np.uint32)).all(), (metric, d, k)
    fx = lance_amd.create_index(x, "IVF_FLAT", metric=metric, num_partitions=12, max_iters=6, sample_rate=64)
    for k, nprobes in ((10, 4), (5, 12)):
        gi, gd = fx.nearest(q, k, nprobes)
        oi, od = oracle.ivfflat_search(x, fx.centroids, q, k, nprobes, metric)
        assert (gi == oi).all(), (metric, d, k, nprobes)
        assert (gd.view(np.uint32) == od.view(np.uint32)).all(), (metric, d, k, nprobes)


@pytest.mark.parametrize("metric", ["dot", "cosine"])
def test_f16_python_api(eng, oracle, metric):
    """create_index / nearest on float16 vectors with dot / cosine (the combinations refused before)"""
    import lance_amd
    x = f16_data(12000, 32, 51)
    q = f16_data(60, 32, 52)
    ix = lance_amd.create_index(x, "IVF_PQ", metric=metric, num_partitions=16, num_sub_vectors=4, max_iters=6, sample_rate=64)
    oidx = oracle.build_index(x, ix.centroids.astype(np.float16), ix.codebook.astype(np.float16), metric)
    for nprobes, rf in ((16, None), (5, 4)):
        gi, gd = ix.nearest(q, 10, nprobes, refine_factor=rf)
        oi, od = oidx.search(q, 10, nprobes, refine=rf or 0, raw=x.astype(f32) if rf else None)
        assert (gi.view(np.uint64) == oi).all(), (metric, nprobes, rf)
        assert (gd.view(np.uint32) == od.view(np.uint32)).all()


@pytest.mark.parametrize("d,m", [(96, 4), (128, 4), (80, 2)])
def test_f16_dot_index_with_sub_vectors_longer_than_16(eng, oracle, d, m):
    """Round 4: an f16 dot index whose PQ sub-vectors have more than 16 elements (24 / 32 / 40 here).  The table entries are
    dot_scalar::<f16, f32, 32> products (dot.rs:91-102,138-161) -- a different addition order from the 16-lane form beyond 16
    elements -- built by the query-major kernels' run-time-dimension table code (search.hip: lut_entry_rt); lance_hip_pq_encode
    under dot takes the 32-lane argmin as well."""
    from lance_amd.engine import DeviceIndex
    n, nlist = 12000, 16
    x = f16_data(n, d, 71 + d)
    q = f16_data(120, d, 72 + d)
    cent, _, _, _ = oracle.kmeans_train(x[:4096], nlist, max_iters=6, seed=1, metric="dot")
    cb, _ = oracle.pq_train(x[:4096], m, max_iters=5, seed=2)
    oidx = oracle.build_index(x, cent, cb, "dot")
    gpart, gcodes, _ = eng.ivfpq_encode(x, cent, cb, "dot")
    assert (_np(gpart).view(np.uint32) == oidx.part_ids).all()
    assert (_np(gcodes) == oidx.codes_rowmajor).all()
    gidx = DeviceIndex.create(eng, "dot", cent, cb, gpart, gcodes, None, raw=x)
    xf = x.astype(f32)
    for nq, k, nprobes, rf in ((120, 10, nlist, 0), (120, 10, 4, 0), (120, 5, 4, 8)):
        gi, gd = gidx.search(q[:nq], k, nprobes, rf)
        oi, od = oidx.search(q[:nq], k, nprobes, refine=rf, raw=xf)
        assert (_np(gi).view(np.uint64) == oi).all(), (d, m, nq, k, nprobes, rf)
        assert (_np(gd).view(np.uint32) == od.view(np.uint32)).all(), (d, m, nq, k, nprobes, rf)
    # the stand-alone encoder under dot: nearest codeword by dot distance of f16 sub-vectors
    gc = eng.pq_encode(x[:3000], cb, "dot")
    oc = oracle.pq_encode(x[:3000], cb, "dot")
    assert (_np(gc) == oc).all()
