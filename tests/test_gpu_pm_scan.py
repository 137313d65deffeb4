"""GPU parity of the PARTITION-MAJOR ADC scan (search_pm.hip) -- the kernel the bench times.

The engine takes the partition-major path when `nq * nprobes >= 4096` and the shape is one it supports
(M in {16, 32}, sub-dimension in {4, 8, 16}, 8-bit codes, k * refine <= 128).  Every case here is sized to take it
and ASSERTS that it did (the `ivfpq_scan_c1` timer counts main-pass launches), then compares ids and distances
with the CPU oracle bit for bit.  Covers every template instantiation of the scan: sub-dimension 4 / 8 / 16 x
L2 / dot x M 16 / 32, the cosine route (normalise + L2), f16 columns (`round_f16` residuals, the C4 shape at reduced N),
int8 columns (C5 shape: M = 32), refine on / off, the two-class flow (LANCE_HIP_PM_NOBOUND=1) and the overflow ->
exact-replay path (more rows tied at the bound inside one partition than a candidate buffer holds).

Reference behaviour matched: pq/distance.rs:109-144 (sequential-m ADC sum), flat/index.rs:94-126 (per-partition heap,
earlier row wins ties), scanner.rs:3440-3468 ((dist, rowid) merge), v2.rs:316-332 (residual query).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
f32 = np.float32


@pytest.fixture(scope="module")
def eng(engine):
    from lance_amd.engine import Engine
    e = Engine()
    yield e
    e.close()


def _np(t):
    return t.cpu().numpy()


def clustered(n, d, seed, ncl=48, lo=0.0, hi=128.0, sigma=20.0, integer=True):
    rng = np.random.default_rng(seed)
    centers = rng.uniform(lo, hi, (ncl, d))
    x = centers[rng.integers(0, ncl, n)] + rng.normal(0, sigma, (n, d))
    if integer:
        x = np.clip(np.rint(x), 0, 218)
    return x.astype(f32)


def _models(oracle, x, nlist, m, metric, seed):
    xs = oracle.normalize(x) if metric == "cosine" else x
    km = "l2" if metric == "cosine" else metric
    cent, _, _, _ = oracle.kmeans_train(xs[: nlist * 40], nlist, max_iters=4, seed=seed, metric=km)
    part, _ = oracle.assign(xs, cent, km)
    res = oracle.residual(xs, cent, np.where(part == oracle.NONE, 0, part)) if km == "l2" else xs
    cb, _ = oracle.pq_train(res[: 256 * 12], m, max_iters=3, seed=seed + 1)
    return cent, cb


class _pm_used:
    """context manager: the searches inside must have launched the partition-major main pass"""

    def __init__(self, eng):
        self.eng = eng

    def __enter__(self):
        self.eng.timing(True)
        self.before = self.eng.timing_query("ivfpq_scan_c1")[1]
        return self

    def __exit__(self, *a):
        self.eng.synchronize()
        after = self.eng.timing_query("ivfpq_scan_c1")[1]
        self.eng.timing(False)
        if a[0] is None:
            assert after > self.before, "the partition-major scan was not taken (nq * nprobes < 4096 or unsupported shape?)"


def _check(eng, oracle, gidx, oidx, qg, q, raw, cases):
    for k, nprobes, rf in cases:
        with _pm_used(eng):
            gi, gd = gidx.search(qg, k, nprobes, rf)
        oi, od = oidx.search(q, k, nprobes, refine=rf, raw=raw if rf else None)
        bad = np.nonzero((_np(gi).view(np.uint64) != oi).any(axis=1))[0]
        assert bad.size == 0, f"ids differ for {bad.size} queries (first {bad[:5]}) at k={k} nprobes={nprobes} refine={rf}"
        assert (_np(gd).view(np.uint32) == od.view(np.uint32)).all(), (k, nprobes, rf)


# (d, M): sub-dimension 4 / 8 / 16 with M = 16 (MU = 1) and M = 32 (MU = 2)
PM_SHAPES = [(64, 16), (128, 16), (256, 16), (128, 32), (256, 32), (512, 32)]


@pytest.mark.parametrize("metric", ["l2", "dot", "cosine"])
@pytest.mark.parametrize("d,m", PM_SHAPES)
def test_pm_scan_f32_every_instantiation(eng, oracle, d, m, metric):
    from lance_amd.engine import DeviceIndex
    n, nlist, nq = 16000, 32, 640
    x = clustered(n, d, 100 + d + m) + (1.0 if metric == "cosine" else 0.0)
    q = clustered(nq, d, 200 + d + m) + (1.0 if metric == "cosine" else 0.0)
    cent, cb = _models(oracle, x, nlist, m, metric, seed=d + m)
    oidx = oracle.build_index(x, cent, cb, metric)
    gpart, gcodes, _ = eng.ivfpq_encode(x, cent, cb, metric)
    assert (_np(gpart).view(np.uint32) == oidx.part_ids).all() and (_np(gcodes) == oidx.codes_rowmajor).all()
    gidx = DeviceIndex.create(eng, metric, cent, cb, gpart, gcodes, None, raw=x)
    # nq * nprobes >= 4096 in every case: 640 x 7 = 4480
    _check(eng, oracle, gidx, oidx, q, q, x, [(10, 8, 0), (10, 8, 10), (10, nlist, 0), (100, 7, 0), (1, 7, 1), (37, 9, 3)])
    gidx.close()


@pytest.mark.parametrize("d,m", [(128, 16), (128, 32)])
def test_pm_scan_f16_column_c4_shape(eng, oracle, d, m):
    """C4 shape at reduced N: f16 vectors, L2, nlist = 4096 (hierarchically trained in the reference; here the centroids are
    sampled rows, which exercises the same scan), M = 16; the residual query is rounded to f16 (`round_f16`)."""
    from lance_amd.engine import DeviceIndex
    rng = np.random.default_rng(9)
    n, nlist, nq = 40000, 4096, 512
    c = rng.standard_normal((64, d)) * 2
    x = (c[rng.integers(0, 64, n)] + rng.standard_normal((n, d)) * 0.7).astype(np.float16)
    q = (c[rng.integers(0, 64, nq)] + rng.standard_normal((nq, d)) * 0.7).astype(np.float16)
    cent = x[rng.choice(n, nlist, replace=False)].copy()
    part, _ = oracle.assign(x, cent)
    res = oracle.residual(x, cent, part)
    cb, _ = oracle.pq_train(res[:4096], m, max_iters=3, seed=2)
    oidx = oracle.build_index(x, cent, cb)
    gpart, gcodes, _ = eng.ivfpq_encode(x, cent, cb)
    assert (_np(gpart).view(np.uint32) == oidx.part_ids).all() and (_np(gcodes) == oidx.codes_rowmajor).all()
    gidx = DeviceIndex.create(eng, "l2", cent, cb, gpart, gcodes, None, raw=x)
    _check(eng, oracle, gidx, oidx, q, q, x.astype(f32), [(10, 10, 0), (10, 10, 10), (10, 64, 0), (50, 8, 2)])
    gidx.close()


@pytest.mark.parametrize("metric", ["l2", "dot"])
def test_pm_scan_int8_column_c5_shape(eng, oracle, metric):
    """C5 shape at reduced N: int8 vectors (model kept in f32), d = 128, M = 32 (sub-dimension 4, MU = 2)."""
    import torch
    from lance_amd.engine import DeviceIndex
    rng = np.random.default_rng(12)
    n, d, nlist, m, nq = 24000, 128, 64, 32, 600
    centers = rng.integers(-90, 90, (80, d))
    x8 = np.clip(centers[rng.integers(0, 80, n)] + rng.normal(0, 14, (n, d)), -128, 127).astype(np.int8)
    q8 = np.clip(centers[rng.integers(0, 80, nq)] + rng.normal(0, 14, (nq, d)), -128, 127).astype(np.int8)
    xf, qf = x8.astype(f32), q8.astype(f32)
    cent, cb = _models(oracle, xf, nlist, m, metric, seed=3)
    oidx = oracle.build_index(xf, cent, cb, metric=metric)
    part, codes, _ = eng.ivfpq_encode(torch.from_numpy(x8), cent, cb, metric)
    assert (_np(part).view(np.uint32) == oidx.part_ids).all() and (_np(codes) == oidx.codes_rowmajor).all()
    g = DeviceIndex.create(eng, metric, cent, cb, part, codes, None, raw=torch.from_numpy(x8), dtype="int8")
    _check(eng, oracle, g, oidx, torch.from_numpy(q8), qf, xf, [(10, 8, 0), (10, 8, 10), (10, nlist, 0), (64, 7, 2)])
    g.close()


@pytest.mark.parametrize("d,m,metric", [(128, 16, "l2"), (128, 32, "dot"), (256, 16, "l2")])
def test_pm_scan_two_class_flow(eng, oracle, d, m, metric):
    """LANCE_HIP_PM_NOBOUND=1: the earlier flow (class 0 = nearest partition scanned with candidate selection, class 1 = the
    rest) -- the <.., RPL=1, PM_CAP> class-0 and class-1 instantiations."""
    from lance_amd.engine import DeviceIndex
    n, nlist, nq = 20000, 24, 600
    x = clustered(n, d, 7 + d)
    q = clustered(nq, d, 8 + d)
    cent, cb = _models(oracle, x, nlist, m, metric, seed=5)
    oidx = oracle.build_index(x, cent, cb, metric)
    gpart, gcodes, _ = eng.ivfpq_encode(x, cent, cb, metric)
    gidx = DeviceIndex.create(eng, metric, cent, cb, gpart, gcodes, None, raw=x)
    os.environ["LANCE_HIP_PM_NOBOUND"] = "1"
    try:
        _check(eng, oracle, gidx, oidx, q, q, x, [(10, 8, 0), (10, 8, 10), (100, nlist, 0)])
    finally:
        del os.environ["LANCE_HIP_PM_NOBOUND"]
    gidx.close()


@pytest.mark.parametrize("ndup", [400, 1500])
def test_pm_scan_overflow_goes_to_exact_replay(eng, oracle, ndup):
    """More rows tied at the k-th distance inside ONE partition than the candidate buffers hold: the partition-major scan
    must flag those queries and the exact kernel must replay them through the BinaryHeap emulation (flat/index.rs:94-126:
    which of the tied rows survive depends on heap order).  `ndup` copies of one vector share a code, hence a distance."""
    from lance_amd.engine import DeviceIndex
    rng = np.random.default_rng(4)
    n, d, nlist, m, nq = 16000, 128, 16, 16, 512
    x = clustered(n, d, 21, ncl=16)
    hot = x[5].copy()
    pos = rng.choice(np.arange(100, n), ndup, replace=False)
    x[pos] = hot                                  # ndup identical rows, scattered over the input order
    q = clustered(nq, d, 22, ncl=16)
    q[: nq // 2] = hot + rng.integers(-1, 2, (nq // 2, d)).astype(f32)     # half the queries sit on the duplicated vector
    cent, cb = _models(oracle, x, nlist, m, "l2", seed=6)
    oidx = oracle.build_index(x, cent, cb)
    gpart, gcodes, _ = eng.ivfpq_encode(x, cent, cb)
    gidx = DeviceIndex.create(eng, "l2", cent, cb, gpart, gcodes, None, raw=x)
    replays = 0
    for k, nprobes, rf in ((10, 8, 0), (10, 8, 10), (100, nlist, 0)):
        with _pm_used(eng):
            gi, gd = gidx.search(q, k, nprobes, rf)
        replays += eng.search_stats()
        oi, od = oidx.search(q, k, nprobes, refine=rf, raw=x if rf else None)
        assert (_np(gi).view(np.uint64) == oi).all(), (k, nprobes, rf)
        assert (_np(gd).view(np.uint32) == od.view(np.uint32)).all()
    assert replays > 0, "no query was replayed by the exact kernel: the tie case was not exercised"
    gidx.close()


def test_pm_scan_random_shapes(eng, oracle):
    """The randomised differential run of tests/fuzz_parity.py restricted to shapes and batch sizes that take the
    partition-major path (>= 256 queries, nq * nprobes >= 4096)."""
    from lance_amd.engine import DeviceIndex
    rng = np.random.default_rng(1)
    for case in range(10):
        sd = int(rng.choice([4, 8, 16]))
        m = int(rng.choice([16, 32]))
        d = m * sd
        n = int(rng.integers(3000, 30000))
        nlist = int(rng.integers(1, 48))
        metric = str(rng.choice(["l2", "dot", "cosine"]))
        integer = bool(rng.integers(0, 2))
        if integer:
            x = rng.integers(0, 30, (n, d)).astype(f32) + (1.0 if metric == "cosine" else 0.0)
            q = rng.integers(0, 30, (512, d)).astype(f32) + (1.0 if metric == "cosine" else 0.0)
        else:
            x = (rng.standard_normal((n, d)) * 3 + (2.0 if metric == "cosine" else 0.0)).astype(f32)
            q = (rng.standard_normal((512, d)) * 3 + (2.0 if metric == "cosine" else 0.0)).astype(f32)
        xs = oracle.normalize(x) if metric == "cosine" else x
        km = "l2" if metric == "cosine" else metric
        cent, _, _, _ = oracle.kmeans_train(xs[: max(nlist * 32, nlist)], nlist, max_iters=4, seed=case, metric=km)
        part, _ = oracle.assign(xs, cent, km)
        res = oracle.residual(xs, cent, np.where(part == oracle.NONE, 0, part)) if km == "l2" else xs
        cb, _ = oracle.pq_train(res[: 256 * 8], m, max_iters=3, seed=case + 1)
        oidx = oracle.build_index(x, cent, cb, metric)
        gpart, gcodes, _ = eng.ivfpq_encode(x, cent, cb, metric)
        g = DeviceIndex.create(eng, metric, cent, cb, gpart, gcodes, None, raw=x)
        for _ in range(3):
            nprobes = int(rng.integers(1, nlist + 1))
            nq = max(256, -(-4096 // nprobes))
            qq = np.concatenate([q] * (-(-nq // 512)))[:nq]
            k = int(rng.integers(1, 60)); rf = int(rng.choice([0, 0, 1, 3]))
            if k * max(rf, 1) > 128:
                rf = 0
            cfg = dict(case=case, n=n, d=d, m=m, nlist=nlist, metric=metric, integer=integer, k=k, nprobes=nprobes, rf=rf, nq=nq)
            with _pm_used(eng):
                gi, gd = g.search(qq, k, nprobes, rf)
            oi, od = oidx.search(qq, k, nprobes, refine=rf, raw=x if rf else None)
            assert (_np(gi).view(np.uint64) == oi).all(), cfg
            assert (_np(gd).view(np.uint32) == od.view(np.uint32)).all(), cfg
        g.close()


@pytest.mark.parametrize("metric,d,m", [("l2", 128, 16), ("dot", 128, 16), ("cosine", 64, 16), ("l2", 128, 32)])
def test_prefilter_mask_fused_in_every_scan_kernel(eng, oracle, metric, d, m):
    """`nearest(..., prefilter=)` under a RowIdMask (flat/index.rs:129-165): the bitmap is tested inside the kernels -- bound
    pass, integer filter scan, exact pair scan (dot), rescan, query-major scan (small batches) -- and the result must equal
    the oracle's literal restatement of the reference branch (per-row distance(id) over the selected rows), for selective
    and for permissive filters, with and without refine."""
    from lance_amd.engine import DeviceIndex
    rng = np.random.default_rng(31)
    n, nlist, nq = 16000, 24, 640
    x = clustered(n, d, 300 + d) + (1.0 if metric == "cosine" else 0.0)
    q = clustered(nq, d, 301 + d) + (1.0 if metric == "cosine" else 0.0)
    cent, cb = _models(oracle, x, nlist, m, metric, seed=9)
    oidx = oracle.build_index(x, cent, cb, metric)
    gpart, gcodes, _ = eng.ivfpq_encode(x, cent, cb, metric)
    g = DeviceIndex.create(eng, metric, cent, cb, gpart, gcodes, None, raw=x)
    for frac in (0.02, 0.5, 0.97):
        allow = rng.random(n + 100) < frac         # longer than the table: ids beyond n never occur; shorter is tested below
        for k, nprobes, rf in ((10, 8, 0), (10, 8, 10), (40, nlist, 0)):
            with _pm_used(eng):
                gi, gd = g.search_filtered(q, k, nprobes, allow, rf)
            oi, od = oidx.search(q, k, nprobes, refine=rf, raw=x if rf else None, prefilter=allow[:n])
            assert (_np(gi).view(np.uint64) == oi).all(), (metric, frac, k, nprobes, rf)
            assert (_np(gd).view(np.uint32) == od.view(np.uint32)).all()
        # a small batch takes the query-major kernel
        gi, gd = g.search_filtered(q[:7], 10, 5, allow)
        oi, od = oidx.search(q[:7], 10, 5, prefilter=allow[:n])
        assert (_np(gi).view(np.uint64) == oi).all() and (_np(gd).view(np.uint32) == od.view(np.uint32)).all()
    short = rng.random(n // 2) < 0.5                # rows with id >= len(filter) are filtered out
    gi, gd = g.search_filtered(q, 10, 8, short)
    full = np.zeros(n, bool); full[: n // 2] = short
    oi, od = oidx.search(q, 10, 8, prefilter=full)
    assert (_np(gi).view(np.uint64) == oi).all() and (_np(gd).view(np.uint32) == od.view(np.uint32)).all()
    g.close()


def test_adaptive_probing_extends_starved_queries(engine, oracle):
    """minimum_nprobes < maximum_nprobes (knn.rs:714-860): a query that found fewer than k rows in its first partitions keeps
    going, nearest partitions first.  With a very selective prefilter every returned list must equal the oracle's search at
    the number of partitions that query ended with (doubling schedule), and must be complete (k rows) whenever the oracle
    finds k rows at maximum_nprobes."""
    import lance_amd
    from lance_amd.testing import sift_like as latent_sift
    x = latent_sift(30000, 64, 71, n_clusters=32)
    q = latent_sift(300, 64, 72, n_clusters=32)
    idx = lance_amd.create_index(x, "IVF_PQ", metric="l2", num_partitions=32, num_sub_vectors=16, max_iters=8)
    oidx = oracle.build_index(x, idx.centroids, idx.codebook)
    rng = np.random.default_rng(5)
    allow = rng.random(30000) < 0.004               # ~120 selected rows: 1 or 2 probes rarely hold 10 of them
    k = 10
    ids, dists = idx.nearest(q, k=k, prefilter=allow, minimum_nprobes=2, maximum_nprobes=32)
    per_np = {}
    npb = 2
    while True:
        per_np[npb] = oidx.search(q, k, npb, prefilter=allow)
        if npb >= 32:
            break
        npb = min(32, npb * 2)
    for i in range(q.shape[0]):
        for npb in sorted(per_np):
            oi, od = per_np[npb]
            if (oi[i] != np.uint64(0xFFFFFFFFFFFFFFFF)).all() or npb == 32:
                assert (ids[i].view(np.uint64) == oi[i]).all() and (dists[i].view(np.uint32) == od[i].view(np.uint32)).all(), (i, npb)
                break
    # nprobes alone = fixed probing (minimum = maximum), as pylance passes it
    a, _ = idx.nearest(q, k=k, nprobes=3, prefilter=allow)
    b, _ = oidx.search(q, k, 3, prefilter=allow)
    assert (a.view(np.uint64) == b).all()
    # late_search's shortcut (knn.rs:741-779): a prefilter selecting <= k rows -> whatever the first partitions did not
    # find comes back at +inf, ascending row ids, and no further partition is searched
    few = np.zeros(30000, bool)
    few[rng.choice(30000, size=7, replace=False)] = True
    c, cd = idx.nearest(q, k=k, prefilter=few, minimum_nprobes=2, maximum_nprobes=32)
    oi, od = oidx.search(q, k, 2, prefilter=few)
    sel = np.flatnonzero(few).astype(np.uint64)
    for i in range(q.shape[0]):
        found = oi[i][oi[i] != np.uint64(0xFFFFFFFFFFFFFFFF)]
        rest = np.setdiff1d(sel, found)
        want = np.concatenate([found, rest])
        assert (c[i].view(np.uint64)[:want.size] == want).all(), i
        assert (c[i][want.size:] == -1).all()
        assert (cd[i][:found.size].view(np.uint32) == od[i][:found.size].view(np.uint32)).all() and np.isinf(cd[i][found.size:want.size]).all()


# ---- M = 48 / 64 / 96: the table tiled over the sub-quantisers (search_qt.hip), 32-bit sums, class B through the rescan kernel
TILED_SHAPES = [(96 * 16, 96), (96 * 8, 96), (96 * 4, 96), (64 * 16, 64), (64 * 4, 64), (48 * 8, 48), (48 * 16, 48)]


@pytest.mark.parametrize("metric", ["l2", "cosine"])
@pytest.mark.parametrize("d,m", TILED_SHAPES)
def test_pm_scan_tiled_table_m48_m64_m96(eng, oracle, d, m, metric):
    """BASELINE config 3's PQ shape (M = 96, sub-dimension 16) and its neighbours at reduced N.  nq * nprobes >= 2048 takes the
    partition-major path for these shapes.  Partition sizes straddle the 2048-row block of the tiled kernel (one huge
    partition: the table is rebuilt per row block) and k * refine (tiny partitions: no bound -> class B -> exact rescan)."""
    from lance_amd.engine import DeviceIndex
    rng = np.random.default_rng(d + m)
    n, nlist, nq = 18000, 24, 300
    ncl = 12
    centers = rng.standard_normal((ncl, d)).astype(f32) * 2
    w = np.array([30] + [3] * (ncl - 1), dtype=np.float64); w /= w.sum()        # one dominant cluster -> one partition of > 2048 rows
    x = (centers[rng.choice(ncl, n, p=w)] + rng.standard_normal((n, d)).astype(f32) * 0.7 + (3.0 if metric == "cosine" else 0.0)).astype(f32)
    q = (centers[rng.choice(ncl, nq, p=w)] + rng.standard_normal((nq, d)).astype(f32) * 0.7 + (3.0 if metric == "cosine" else 0.0)).astype(f32)
    cent, cb = _models(oracle, x, nlist, m, metric, seed=d + m)
    oidx = oracle.build_index(x, cent, cb, metric)
    sizes = np.diff(oidx.part_offsets.astype(np.int64))
    gpart, gcodes, _ = eng.ivfpq_encode(x, cent, cb, metric)
    assert (_np(gpart).view(np.uint32) == oidx.part_ids).all() and (_np(gcodes) == oidx.codes_rowmajor).all()
    gidx = DeviceIndex.create(eng, metric, cent, cb, gpart, gcodes, None, raw=x)
    # 300 x 7 = 2100 pairs; k * refine = 120 exceeds the small partitions' row counts for some queries (class B)
    _check(eng, oracle, gidx, oidx, q, q, x, [(10, 8, 0), (10, 8, 10), (10, nlist, 0), (100, 7, 0), (1, 7, 1), (40, 9, 3)])
    assert sizes.max() > 1024 and sizes.min() < 100, sizes    # two rows per lane (most cases: several 2048-row blocks) and tiny partitions were really there
    gidx.close()


def test_pm_scan_tiled_table_prefilter_and_class_b(eng, oracle):
    """M = 96: a selective prefilter (tested inside the tiled bound and scan kernels) and an index whose partitions all hold
    fewer rows than k * refine, so that EVERY query is class B (no bound -> rescan kernel -> merge)."""
    from lance_amd.engine import DeviceIndex
    from lance_amd.vector import IvfPqIndex, IvfPqParams
    rng = np.random.default_rng(5)
    n, d, m, nlist, nq = 6000, 96 * 4, 96, 40, 256
    x = clustered(n, d, 71, ncl=20, integer=False)
    q = clustered(nq, d, 72, ncl=20, integer=False)
    cent, cb = _models(oracle, x, nlist, m, "l2", seed=9)
    oidx = oracle.build_index(x, cent, cb, "l2")
    gpart, gcodes, _ = eng.ivfpq_encode(x, cent, cb, "l2")
    gidx = DeviceIndex.create(eng, "l2", cent, cb, gpart, gcodes, None, raw=x)
    allow = rng.random(n) < 0.2
    vi = IvfPqIndex(gidx, IvfPqParams(nlist, m, 8, "l2"), None, gpart, gcodes)
    with _pm_used(eng):
        gi, gd = vi.nearest(q, 10, 10, prefilter=allow)
    oi, od = oidx.search(q, 10, 10, prefilter=allow)
    assert (gi.view(np.uint64) == oi).all() and (gd.view(np.uint32) == od.view(np.uint32)).all()
    gidx.close()
    # 6000 rows over 400 partitions: ~15 rows each, k * refine = 100 -> nobody gets a bound
    nlist2 = 400
    cent2 = x[rng.choice(n, nlist2, replace=False)].copy()
    part2, _ = oracle.assign(x, cent2)
    res2 = oracle.residual(x, cent2, part2)
    cb2, _ = oracle.pq_train(res2[:3072], m, max_iters=3, seed=4)
    oidx2 = oracle.build_index(x, cent2, cb2, "l2")
    gpart2, gcodes2, _ = eng.ivfpq_encode(x, cent2, cb2, "l2")
    gidx2 = DeviceIndex.create(eng, "l2", cent2, cb2, gpart2, gcodes2, None, raw=x)
    _check(eng, oracle, gidx2, oidx2, q, q, x, [(10, 12, 10), (100, 9, 0), (10, 40, 0)])
    gidx2.close()


@pytest.mark.parametrize("d,m", [(64, 16), (128, 32), (96 * 8, 96)])
def test_pm_scan_loose_bounds_overflow_segments_are_rescanned(eng, oracle, d, m):
    """A query whose nearest partition holds barely k * refine rows gets a LOOSE bound; in the dominant cluster's partitions
    (thousands of rows) more than 256 rows then pass the integer filter -> the segment overflows -> ivfpq_qrescan_kernel
    redoes that (query, probe) with the exact table, one workgroup per overflowed segment, several segments of one query
    appending to its pool concurrently (round 3).  LANCE_HIP_Q_STATS=1 prints the overflow counts (scripts/gpu/r03v.sh)."""
    from lance_amd.engine import DeviceIndex
    rng = np.random.default_rng(700 + d + m)
    n, nlist, nq = 24000, 30, 700
    ncl = 16
    centers = rng.standard_normal((ncl, d)).astype(f32) * 0.4
    w = np.array([60.0] + [1.0] * (ncl - 1)); w /= w.sum()      # ~80 % of the rows in one TIGHT cluster, the others ~300 loose rows each
    cl = rng.choice(ncl, n, p=w)
    x = (centers[cl] + rng.standard_normal((n, d)).astype(f32) * np.where(cl == 0, 0.35, 0.8)[:, None].astype(f32)).astype(f32)
    wq = np.array([1.0] + [4.0] * (ncl - 1)); wq /= wq.sum()    # most queries near the loose clusters: their k * refine-th neighbour
    q = (centers[rng.choice(ncl, nq, p=wq)] + rng.standard_normal((nq, d)).astype(f32) * 0.8).astype(f32)   # is farther than the whole tight cluster
    cent, cb = _models(oracle, x, nlist, m, "l2", seed=d)
    oidx = oracle.build_index(x, cent, cb, "l2")
    sizes = np.diff(oidx.part_offsets.astype(np.int64))
    gpart, gcodes, _ = eng.ivfpq_encode(x, cent, cb, "l2")
    gidx = DeviceIndex.create(eng, "l2", cent, cb, gpart, gcodes, None, raw=x)
    _check(eng, oracle, gidx, oidx, q, q, x, [(10, 8, 10), (10, 12, 0), (25, 30, 4), (100, 8, 0)])
    assert sizes.max() > 1500, sizes
    gidx.close()
