"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, bit for bit.

Mirrors the reference's own tests for this path (naive-argmin equality kmeans.rs:1398-1422,
NaN rows :1447-1486, encode == naive pq.rs:628-665, ADC == LUT sum pq.rs:580-625, recall with
nprobes = nlist v2.rs:1354-1381) but asserts EXACT equality of ids / codes / distances, which
the reference never does between its CPU and accelerator paths.
"""
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


def sift_like(n, d, seed, ncl=32):
    rng = np.random.default_rng(seed)
    centers = rng.uniform(0, 128, (ncl, d))
    x = centers[rng.integers(0, ncl, n)] + rng.normal(0, 24, (n, d))
    return np.clip(np.rint(x), 0, 218).astype(f32)


@pytest.mark.parametrize("d,k", [(128, 256), (128, 18), (8, 256), (16, 256), (4, 256), (32, 300), (64, 33), (96, 64),
                                 (100, 40), (20, 7), (1536, 12)])
@pytest.mark.parametrize("metric", ["l2", "dot"])
def test_assign_matches_oracle(eng, oracle, d, k, metric):
    rng = np.random.default_rng(d * 1000 + k)
    n = 3000 if d <= 128 else 300
    x = rng.standard_normal((n, d)).astype(f32) * 3
    c = rng.standard_normal((k, d)).astype(f32) * 3
    ids, dists = eng.assign(x, c, metric)
    oi, od = oracle.assign(x, c, metric)
    assert (_np(ids).view(np.uint32) == oi).all()
    assert (_np(dists).view(np.uint32) == od.view(np.uint32)).all()


def test_assign_nan_inf_bias_ties(eng, oracle):
    rng = np.random.default_rng(3)
    x = sift_like(2000, 128, 1)
    c = sift_like(64, 128, 2)
    c[5] = c[3]                      # duplicate centroid: first index must win
    x[10] = np.nan                   # all-NaN row -> None (kmeans.rs:1447-1486)
    x[11, 7] = np.nan                # partially NaN -> every distance NaN -> None
    x[12, 3] = np.inf
    bias = rng.random(64).astype(f32) * 1000
    for b in (None, bias):
        ids, dists = eng.assign(x, c, "l2", bias=b)
        oi, od = oracle.assign(x, c, "l2", bias=b)
        assert (_np(ids).view(np.uint32) == oi).all()
        ok = oi != oracle.NONE
        assert (_np(dists)[ok].view(np.uint32) == od[ok].view(np.uint32)).all()
    assert oi[10] == oracle.NONE and oi[11] == oracle.NONE


@pytest.mark.parametrize("n,d,k,bf", [(4096, 32, 16, 0.0), (6000, 128, 64, 1.0), (3000, 8, 256, 0.0)])
def test_kmeans_train_bit_exact(eng, oracle, n, d, k, bf):
    x = sift_like(n, d, n + d)
    cent, loss, iters = eng.kmeans_train(x, k, max_iters=30, balance_factor=bf, seed=7)
    # train_kmeans (kmeans.rs:1344) scales the balance factor by 1/n before the loop
    oc, ol, oit, _ = oracle.kmeans_train(x, k, max_iters=30, balance_factor=f32(bf) / f32(n), seed=7)
    assert iters == oit
    assert (_np(cent).view(np.uint32) == oc.view(np.uint32)).all()
    assert loss == ol


@pytest.mark.parametrize("n,d,k", [(20000, 16, 300), (9000, 32, 257)])
def test_hierarchical_kmeans_bit_exact(eng, oracle, n, d, k):
    # k > 256 -> train_hierarchical_kmeans (kmeans.rs:746-1003, dispatcher :1027)
    x = sift_like(n, d, n + k)
    cent, loss, _ = eng.kmeans_train(x, k, max_iters=12, balance_factor=1.0, seed=5)
    oc = oracle.kmeans_train_hierarchical(x, k, max_iters=12, balance_factor_scaled=f32(1.0) / f32(n), seed=5)
    assert cent.shape[0] == oc.shape[0] == k and loss == 0.0
    assert (_np(cent).view(np.uint32) == oc.view(np.uint32)).all()


def test_kmeans_empty_cluster_split(eng, oracle):
    # many duplicates -> empty clusters -> split_clusters (kmeans.rs:174-207) on both sides
    rng = np.random.default_rng(5)
    base = rng.standard_normal((6, 16)).astype(f32)
    x = base[rng.integers(0, 6, 2000)]
    x[:40] += rng.standard_normal((40, 16)).astype(f32) * 0.01
    init = x[:12].copy()
    cent, loss, iters = eng.kmeans_train(x, 12, max_iters=10, init=init, seed=3)
    oc, ol, oit, _ = oracle.kmeans_train(x, 12, max_iters=10, init=init, seed=3)
    assert iters == oit and loss == ol
    assert (_np(cent).view(np.uint32) == oc.view(np.uint32)).all()


def test_pq_train_and_encode_bit_exact(eng, oracle):
    n, d, m = 5000, 64, 8
    x = sift_like(n, d, 11)
    cent = sift_like(16, d, 12)
    part, _ = oracle.assign(x, cent)
    res = oracle.residual(x, cent, part)
    assert (_np(eng.residual(x, cent, part)) == res).all()
    cb, iters = eng.pq_train(res, m, max_iters=12, seed=21)
    ocb, oit = oracle.pq_train(res, m, max_iters=12, seed=21)
    assert (iters == oit.astype(np.uint32)).all()
    assert (_np(cb).view(np.uint32) == ocb.view(np.uint32)).all()
    codes = eng.pq_encode(res, cb)
    assert (_np(codes) == oracle.pq_encode(res, ocb)).all()


@pytest.mark.parametrize("metric", ["l2", "cosine", "dot"])
def test_ivfpq_encode_matches_transform_chain(eng, oracle, metric):
    n, d, nlist, m = 4000, 32, 24, 4
    rng = np.random.default_rng(4)
    x = rng.standard_normal((n, d)).astype(f32)
    x[17] = np.nan; x[99, 3] = np.inf; x[100] = 0.0      # non-finite rows are dropped (utils.rs:263-286)
    cent = rng.standard_normal((nlist, d)).astype(f32)
    if metric == "cosine":
        cent = oracle.normalize(cent)
    cb = rng.standard_normal((m, 256, d // m)).astype(f32) * 0.5
    part, codes, _ = eng.ivfpq_encode(x, cent, cb, metric)
    oidx = oracle.build_index(x, cent, cb, metric)
    part = _np(part).view(np.uint32)
    keep = oracle.is_finite(oracle.normalize(x) if metric == "cosine" else x)
    assert (part[~keep] == oracle.NONE).all()
    assert (part[keep] == oidx.part_ids).all()
    valid = keep & (part != oracle.NONE)
    assert (_np(codes)[valid] == oidx.codes_rowmajor[oidx.part_ids != oracle.NONE]).all()


def test_find_partitions(eng, oracle):
    rng = np.random.default_rng(8)
    cent = sift_like(256, 128, 5)
    cent[9] = cent[200]     # tie -> (dist, id) order
    q = sift_like(300, 128, 6)
    for nprobes in (1, 10, 256):
        ids, d = eng.find_partitions(q, cent, nprobes)
        oi, od = oracle.find_partitions(q, cent, nprobes)
        assert (_np(ids).view(np.uint32) == oi).all()
        assert (_np(d).view(np.uint32) == od.view(np.uint32)).all()


def _build_pair(eng, oracle, x, nlist, m, metric="l2", seed=1):
    from lance_amd.engine import DeviceIndex
    d = x.shape[1]
    xs = oracle.normalize(x) if metric == "cosine" else x
    kmetric = "l2" if metric == "cosine" else metric
    cent, _, _, _ = oracle.kmeans_train(xs[: nlist * 64], nlist, max_iters=8, seed=seed, metric=kmetric)
    part, _ = oracle.assign(xs, cent, kmetric)
    res = oracle.residual(xs, cent, part) if kmetric == "l2" else xs
    cb, _ = oracle.pq_train(res[: 256 * 32], m, max_iters=6, seed=seed + 1)
    oidx = oracle.build_index(x, cent, cb, metric)
    gpart, gcodes, _ = eng.ivfpq_encode(x, cent, cb, metric)
    gidx = DeviceIndex.create(eng, metric, cent, cb, gpart, gcodes, None, raw=x)
    return oidx, gidx


@pytest.mark.parametrize("metric", ["l2", "cosine", "dot"])
def test_ivfpq_search_ids_bit_exact(eng, oracle, metric):
    n, d, nlist, m = 20000, 64, 32, 8
    x = sift_like(n, d, 31) if metric != "cosine" else sift_like(n, d, 31) + 1.0
    q = sift_like(200, d, 32) + (1.0 if metric == "cosine" else 0.0)
    oidx, gidx = _build_pair(eng, oracle, x, nlist, m, metric)
    # storage layout round trip equals the oracle's canonical layout
    offs, codes_t, rid = gidx.export()
    assert (offs == oidx.part_offsets).all() and (rid == oidx.row_ids).all() and (codes_t == oidx.codes_t).all()
    for k, nprobes in ((10, nlist), (10, 5), (1, 1), (100, nlist), (37, 3)):
        gi, gd = gidx.search(q, k, nprobes)
        oi, od = oidx.search(q, k, nprobes)
        assert (_np(gi).view(np.uint64) == oi).all(), (metric, k, nprobes)
        assert (_np(gd).view(np.uint32) == od.view(np.uint32)).all()


def test_ivfpq_search_refine_and_from_storage(eng, oracle):
    from lance_amd.engine import DeviceIndex
    n, d, nlist, m = 30000, 128, 64, 16
    x = sift_like(n, d, 41)
    q = sift_like(150, d, 42)
    oidx, gidx = _build_pair(eng, oracle, x, nlist, m)
    for k, nprobes, rf in ((10, nlist, 10), (10, 8, 5), (5, 2, 1)):
        gi, gd = gidx.search(q, k, nprobes, rf)
        oi, od = oidx.search(q, k, nprobes, refine=rf, raw=x)
        assert (_np(gi).view(np.uint64) == oi).all(), (k, nprobes, rf)
        assert (_np(gd).view(np.uint32) == od.view(np.uint32)).all()
    # load the reference storage layout (transposed codes) straight into HBM
    g2 = DeviceIndex.from_storage(eng, "l2", oidx.centroids, oidx.codebook, oidx.part_offsets, oidx.codes_t, oidx.row_ids,
                                  transposed=True)
    gi, gd = g2.search(q, 10, 7)
    oi, od = oidx.search(q, 10, 7)
    assert (_np(gi).view(np.uint64) == oi).all() and (_np(gd).view(np.uint32) == od.view(np.uint32)).all()
    # single query / tiny batches take the probe-split path
    for nq in (1, 3):
        gi, gd = gidx.search(q[:nq], 10, nlist)
        oi, od = oidx.search(q[:nq], 10, nlist)
        assert (_np(gi).view(np.uint64) == oi).all() and (_np(gd).view(np.uint32) == od.view(np.uint32)).all()


def test_ivfpq_search_edge_cases(eng, oracle):
    from lance_amd.engine import DeviceIndex
    rng = np.random.default_rng(2)
    n, d, nlist, m = 600, 16, 8, 4
    x = rng.standard_normal((n, d)).astype(f32)
    cent = rng.standard_normal((nlist, d)).astype(f32) * 2
    cent[6] = 100.0  # an empty partition
    cb = rng.standard_normal((m, 256, d // m)).astype(f32)
    oidx = oracle.build_index(x, cent, cb)
    part, codes, _ = eng.ivfpq_encode(x, cent, cb)
    gidx = DeviceIndex.create(eng, "l2", cent, cb, part, codes)
    q = rng.standard_normal((20, d)).astype(f32)
    # k larger than the rows in the probed partitions -> missing results padded
    gi, gd = gidx.search(q, 100, 1)
    oi, od = oidx.search(q, 100, 1)
    assert (_np(gi).view(np.uint64) == oi).all()
    assert (_np(gd).view(np.uint32) == od.view(np.uint32)).all()
    # empty query batch
    gi, _ = gidx.search(np.zeros((0, d), f32), 5, 2)
    assert gi.shape == (0, 5)


def test_pq_scan_topk_single_partition(eng, oracle):
    rng = np.random.default_rng(6)
    d, m, n_p = 32, 8, 3000
    cb = rng.standard_normal((m, 256, d // m)).astype(f32)
    codes = rng.integers(0, 256, (n_p, m), dtype=np.uint8)
    rid = rng.permutation(10 * n_p)[:n_p].astype(np.uint64)
    qr = rng.standard_normal(d).astype(f32)
    lut = oracle.build_lut(qr, cb)
    dist = oracle.pq_scan(lut, oracle.transpose(codes))
    gi, gd = eng.pq_scan_topk(qr, cb, oracle.transpose(codes), rid, 20)
    hi, hd = oracle.heap_topk(dist, rid, 20)
    ei, ed = oracle.sort_fetch(hi, hd, 20)
    assert (_np(gi).view(np.uint64) == ei).all() and (_np(gd).view(np.uint32) == ed.view(np.uint32)).all()
    lo, up = float(np.sort(dist)[50]), float(np.sort(dist)[400])
    gi, gd = eng.pq_scan_topk(qr, cb, oracle.transpose(codes), rid, 30, lower=lo, upper=up)
    hi, hd = oracle.heap_topk(dist, rid, 30, lower=lo, upper=up)
    ei, ed = oracle.sort_fetch(hi, hd, 30)
    assert (_np(gi).view(np.uint64) == ei).all() and (_np(gd).view(np.uint32) == ed.view(np.uint32)).all()


@pytest.mark.parametrize("d", [128, 32, 100, 7, 20, 200, 384])
@pytest.mark.parametrize("metric", ["l2", "dot"])
def test_flat_knn_bit_exact_with_ties(eng, oracle, d, metric):
    x = sift_like(20000, d, 51)          # integer-valued -> many exact distance ties
    x[100:110] = x[5]                    # duplicates
    q = sift_like(300, d, 52)
    rid = np.random.default_rng(1).permutation(10 ** 6)[:20000].astype(np.uint64)
    for k in (1, 10, 50):
        gi, gd = eng.flat_topk(x, q, k, metric, row_ids=rid)
        oi, od = oracle.flat_knn(x, q, k, metric, row_ids=rid)
        assert (_np(gi).view(np.uint64) == oi).all(), (d, metric, k)
        assert (_np(gd).view(np.uint32) == od.view(np.uint32)).all()


@pytest.mark.parametrize("d", [8, 16, 20, 32, 48, 128, 200, 1536])
def test_flat_cosine_bit_exact(eng, oracle, d):
    rng = np.random.default_rng(d)
    x = rng.standard_normal((5000, d)).astype(f32) * 2
    q = rng.standard_normal((70, d)).astype(f32)
    gi, gd = eng.flat_topk(x, q, 10, "cosine")
    oi, od = oracle.flat_knn(x, q, 10, "cosine")
    assert (_np(gi).view(np.uint64) == oi).all()
    assert (_np(gd).view(np.uint32) == od.view(np.uint32)).all()


def test_flat_large_k_and_adversarial_order(eng, oracle):
    """k > 128 keeps the lanes-own-queries kernel; rows sorted farthest-first force the candidate-pool
    overflow repair of the v2 path (every epoch passes the stale threshold)."""
    rng = np.random.default_rng(5)
    x = rng.standard_normal((30000, 64)).astype(f32)
    q = rng.standard_normal((9, 64)).astype(f32)
    gi, gd = eng.flat_topk(x, q, 200)
    oi, od = oracle.flat_knn(x, q, 200)
    assert (_np(gi).view(np.uint64) == oi).all() and (_np(gd).view(np.uint32) == od.view(np.uint32)).all()
    order = np.argsort(-((x - q[0]) ** 2).sum(1), kind="stable")
    xs = np.ascontiguousarray(x[order])
    for d_sel in (slice(None), slice(0, 40)):         # fixed-D path (64) and the any-dimension path (40)
        xa, qa = np.ascontiguousarray(xs[:, d_sel]), np.ascontiguousarray(q[:, d_sel])
        gi, gd = eng.flat_topk(xa, qa, 10)
        oi, od = oracle.flat_knn(xa, qa, 10)
        assert (_np(gi).view(np.uint64) == oi).all() and (_np(gd).view(np.uint32) == od.view(np.uint32)).all()


def test_cosine_index_refine_bit_exact(eng, oracle):
    # dbpedia-style: cosine index = normalise + L2 residual PQ; refine re-ranks with the flat cosine
    # kernel on the RAW vectors and the ORIGINAL query (scanner.rs:2884-2904)
    n, d, nlist, m = 20000, 96, 32, 12
    x = sift_like(n, d, 71) + 1.0
    q = sift_like(120, d, 72) + 1.0
    oidx, gidx = _build_pair(eng, oracle, x, nlist, m, "cosine")
    for k, nprobes, rf in ((10, nlist, 10), (10, 4, 3), (5, 1, 1)):
        gi, gd = gidx.search(q, k, nprobes, rf)
        oi, od = oidx.search(q, k, nprobes, refine=rf, raw=x)
        assert (_np(gi).view(np.uint64) == oi).all(), (k, nprobes, rf)
        assert (_np(gd).view(np.uint32) == od.view(np.uint32)).all()


def test_full_size_properties_sift1m(engine, oracle):
    """BASELINE config 2 at full size (1M x 128, IVF256, PQ16) through size-independent properties:
    partition offsets cover every row exactly once, results are sorted by (dist, rowid), searching is
    idempotent, self-queries with refine come back at distance 0 (lance/util.py:171-220 validate_vector_index),
    and a slice of the batch equals the oracle bit for bit."""
    import torch
    import lance_amd
    from lance_amd.testing import sift_like as latent_sift
    x = latent_sift(1_000_000, 128, 1234, device="cuda")
    idx = lance_amd.create_index(x, "IVF_PQ", num_partitions=256, num_sub_vectors=16)
    offs, codes_t, rid = idx.export_storage()
    assert offs[0] == 0 and offs[-1] == 1_000_000 and (np.diff(offs.astype(np.int64)) >= 0).all()
    assert (np.sort(rid) == np.arange(1_000_000, dtype=np.uint64)).all()
    q = latent_sift(2000, 128, 4321, device="cuda")
    ids1, d1 = idx.search_device(q, 10, 10, 10)
    ids2, d2 = idx.search_device(q, 10, 10, 10)
    assert torch.equal(ids1, ids2) and torch.equal(d1, d2)
    dd = d1.cpu().numpy(); ii = ids1.cpu().numpy().view(np.uint64)
    assert (np.diff(dd, axis=1) >= 0).all()
    tie = np.diff(dd, axis=1) == 0
    assert (np.diff(ii.astype(np.int64), axis=1)[tie] > 0).all()
    # self queries: the row itself is returned first with distance exactly 0 after refine
    probe = torch.arange(0, 1_000_000, 997, device="cuda")[:1000]
    si, sd = idx.search_device(x[probe], 1, 256, 10)
    hit = (sd[:, 0] == 0).float().mean().item()
    assert hit >= 0.99, hit
    # oracle on the same artefacts, 64 queries, un-refined and refined, exhaustive probes included
    oidx = oracle.IvfPqIndex("l2", idx.centroids, idx.codebook, offs, codes_t, rid)
    qh = q[:64].cpu().numpy(); xh = x.cpu().numpy()
    for nprobes, rf in ((256, 0), (256, 10), (10, 10)):     # the bit-exact id check at nprobes = nlist
        gi, gd = idx.search_device(q[:64], 10, nprobes, rf)
        oi, od = oidx.search(qh, 10, nprobes, refine=rf, raw=xh)
        assert (gi.cpu().numpy().view(np.uint64) == oi).all()
        assert (gd.cpu().numpy().view(np.uint32) == od.view(np.uint32)).all()
    # The HEADLINE kernel on the headline index: the slices above stay under the matrix-core scan's batch threshold
    # (96 pairs per partition: 2000 x 10 and 64 x 256 pairs both fall short of 96 x 256), so they ran the integer scan.  128 queries x
    # every partition (v2.rs:1354-1381: nprobes = nlist, un-refined and refined) and one bench-sized 10,000 x 10 batch go through
    # ivfpq_mscan_kernel -- asserted on the context's stage counter, which also counts replayed graphs -- and must equal the oracle.
    eng = lance_amd.default_engine()
    ms_runs = lambda: eng.timing_query("count:ivfpq_mscan")[1]
    q128 = latent_sift(128, 128, 987, device="cuda"); q128h = q128.cpu().numpy()
    for nprobes, rf in ((256, 0), (256, 10)):
        before = ms_runs()
        gi, gd = idx.search_device(q128, 10, nprobes, rf)
        assert ms_runs() > before, "128 x 256 pairs did not take the matrix-core scan"
        oi, od = oidx.search(q128h, 10, nprobes, refine=rf, raw=xh)
        bad = np.nonzero((gi.cpu().numpy().view(np.uint64) != oi).any(axis=1))[0]
        assert bad.size == 0, (nprobes, rf, bad[:8])
        assert (gd.cpu().numpy().view(np.uint32) == od.view(np.uint32)).all()
    q10k = latent_sift(10_000, 128, 4321 + 7, device="cuda"); q10kh = q10k.cpu().numpy()
    oi, od = oidx.search(q10kh, 10, 10, refine=10, raw=xh)
    out = (torch.empty((10_000, 10), dtype=torch.int64, device="cuda"), torch.empty((10_000, 10), dtype=torch.float32, device="cuda"))
    for rep in range(3):                                    # plain call, captured call, replayed graph: the serving pattern
        out[0].fill_(-7); out[1].fill_(-7.0)
        before = ms_runs()
        idx.search_device(q10k, 10, 10, 10, out=out)
        assert ms_runs() > before, f"10,000 x 10 pairs did not take the matrix-core scan (rep {rep})"
        bad = np.nonzero((out[0].cpu().numpy().view(np.uint64) != oi).any(axis=1))[0]
        assert bad.size == 0, (rep, bad.size, bad[:8])
        assert (out[1].cpu().numpy().view(np.uint32) == od.view(np.uint32)).all(), rep
    # flat ground truth at full size equals the oracle for a few queries
    gi, gd = lance_amd.flat_knn(x, q[:8], 10)
    oi, od = oracle.flat_knn(xh, qh[:8], 10)
    assert (gi.cpu().numpy().view(np.uint64) == oi).all() and (gd.cpu().numpy().view(np.uint32) == od.view(np.uint32)).all()


def test_dbpedia_shape_cosine_m96(eng, oracle):
    """BASELINE config 3 shape at reduced N: d=1536 f32 cosine, M=96 (sub-dim 16, 96 KiB LUT), hierarchical
    IVF training (nlist > 256), refine with the flat cosine kernel."""
    from lance_amd.engine import DeviceIndex
    rng = np.random.default_rng(77)
    n, d, nlist, m = 6000, 1536, 260, 96
    centers = rng.standard_normal((40, d)).astype(f32)
    x = centers[rng.integers(0, 40, n)] + rng.standard_normal((n, d)).astype(f32) * 0.5
    x = (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(f32)      # ada-002 embeddings are unit norm
    q = centers[rng.integers(0, 40, 40)] + rng.standard_normal((40, d)).astype(f32) * 0.5
    xs = oracle.normalize(x)
    # IVF: k=260 > 256 -> hierarchical on both sides, bit-exact
    cent, loss, _ = eng.kmeans_train(xs, nlist, max_iters=6, balance_factor=1.0, seed=4)
    oc = oracle.kmeans_train_hierarchical(xs, nlist, max_iters=6, balance_factor_scaled=f32(1.0) / f32(n), seed=4)
    assert (_np(cent).view(np.uint32) == oc.view(np.uint32)).all()
    part, _ = oracle.assign(xs, oc)
    res = oracle.residual(xs, oc, part)
    cb, it = eng.pq_train(res, m, max_iters=4, seed=8)
    ocb, _ = oracle.pq_train(res, m, max_iters=4, seed=8)
    assert (_np(cb).view(np.uint32) == ocb.view(np.uint32)).all()
    oidx = oracle.build_index(x, oc, ocb, "cosine")
    gpart, gcodes, _ = eng.ivfpq_encode(x, oc, ocb, "cosine")
    assert (_np(gpart).view(np.uint32) == oidx.part_ids).all() and (_np(gcodes) == oidx.codes_rowmajor).all()
    gidx = DeviceIndex.create(eng, "cosine", oc, ocb, gpart, gcodes, None, raw=x)
    for k, nprobes, rf in ((10, nlist, 0), (10, 20, 10), (10, 3, 0)):
        gi, gd = gidx.search(q, k, nprobes, rf)
        oi, od = oidx.search(q, k, nprobes, refine=rf, raw=x)
        assert (_np(gi).view(np.uint64) == oi).all(), (k, nprobes, rf)
        assert (_np(gd).view(np.uint32) == od.view(np.uint32)).all()


@pytest.mark.parametrize("metric", ["l2", "dot"])
def test_4bit_pq_bit_exact(eng, oracle, metric):
    """num_bits = 4 (a18): k=16 sub-quantisers, nibble packing (pq.rs:168-172), and the quantised fast-scan
    compute_pq_distance_4bit (pq/distance.rs:147-284) with its exact head/tail rows."""
    from lance_amd.engine import DeviceIndex
    n, d, nlist, m = 20000, 64, 16, 16
    x = sift_like(n, d, 91)
    q = sift_like(100, d, 92)
    kmetric = metric
    cent, _, _, _ = oracle.kmeans_train(x[:2048], nlist, max_iters=6, seed=1, metric=kmetric)
    part, _ = oracle.assign(x, cent, kmetric)
    res = oracle.residual(x, cent, part) if metric == "l2" else x
    cb, it = eng.pq_train(res[:8192], m, nbits=4, max_iters=8, seed=2)
    ocb, oit = oracle.pq_train(res[:8192], m, nbits=4, max_iters=8, seed=2)
    assert cb.shape == (m, 16, d // m) and (it == oit.astype(np.uint32)).all()
    assert (_np(cb).view(np.uint32) == ocb.view(np.uint32)).all()
    codes = eng.pq_encode(res, ocb, metric)
    assert codes.shape == (n, m // 2)
    assert (_np(codes) == oracle.pq_encode(res, ocb, metric, nbits=4)).all()
    oidx = oracle.build_index(x, cent, ocb, metric, nbits=4)
    gpart, gcodes, _ = eng.ivfpq_encode(x, cent, ocb, metric)
    gidx = DeviceIndex.create(eng, metric, cent, ocb, gpart, gcodes, None, raw=x)
    offs, codes_t, rid = gidx.export()
    assert (offs == oidx.part_offsets).all() and (codes_t == oidx.codes_t).all()
    for k, nprobes, rf in ((10, nlist, 0), (10, 3, 0), (10, 4, 5), (100, nlist, 0), (1, 1, 0)):
        gi, gd = gidx.search(q, k, nprobes, rf)
        oi, od = oidx.search(q, k, nprobes, refine=rf, raw=x)
        assert (_np(gi).view(np.uint64) == oi).all(), (metric, k, nprobes, rf)
        assert (_np(gd).view(np.uint32) == od.view(np.uint32)).all()
    # under a prefilter the reference scores the selected rows with distance(id): the unquantised table, byte-wise terms
    # (pq/storage.rs:893-921) -- a third arithmetic, fused into the 4-bit scan and its exact replay
    rng = np.random.default_rng(17)
    for frac, k, nprobes, rf in ((0.5, 10, nlist, 0), (0.05, 10, 4, 0), (0.3, 100, nlist, 0), (0.5, 10, 4, 5), (0.002, 10, nlist, 0)):
        allow = rng.random(n) < frac
        gi, gd = gidx.search_filtered(q, k, nprobes, allow, rf)
        oi, od = oidx.search(q, k, nprobes, refine=rf, raw=x, prefilter=allow)
        assert (_np(gi).view(np.uint64) == oi).all(), (metric, "prefilter", frac, k, nprobes, rf)
        assert (_np(gd).view(np.uint32) == od.view(np.uint32)).all()
    allow = rng.random(n) < 0.5                      # ... and together with a distance range
    _, ud = oidx.search(q, 40, nlist, prefilter=allow)
    fin = ud[np.isfinite(ud)]
    lo, hi = float(np.quantile(fin, 0.2)), float(np.quantile(fin, 0.7))
    gi, gd = gidx.search_range(q, 10, 4, lo, hi, allow=allow)
    oi, od = oidx.search(q, 10, 4, prefilter=allow, lower=lo, upper=hi)
    assert (_np(gi).view(np.uint64) == oi).all() and (_np(gd).view(np.uint32) == od.view(np.uint32)).all()
    # single partition entry point, small partition (all rows exact) and a 16-multiple boundary
    for n_p in (150, 1008, 1013):
        ct = oracle.transpose(oracle.pq_encode(res[:n_p], ocb, metric, nbits=4))
        ridp = np.arange(n_p, dtype=np.uint64) * 3
        qr = res[5]
        lut = oracle.build_lut(qr, ocb, metric, nbits=4)
        dist = oracle.pq_scan4(lut, ct, 10, metric)
        hi, hd = oracle.heap_topk(dist, ridp, 10)
        ei, ed = oracle.sort_fetch(hi, hd, 10)
        gi, gd = eng.pq_scan_topk(qr, ocb, ct, ridp, 10, metric)
        assert (_np(gi).view(np.uint64) == ei).all() and (_np(gd).view(np.uint32) == ed.view(np.uint32)).all()


def f16_data(n, d, seed, ncl=16):
    # small-magnitude f16-exact values (like C4's f16 vectors)
    rng = np.random.default_rng(seed)
    c = rng.standard_normal((ncl, d)) * 2
    return (c[rng.integers(0, ncl, n)] + rng.standard_normal((n, d)) * 0.7).astype(np.float16)


def test_f16_assign_kmeans_pq_bit_exact(eng, oracle):
    """Float16Type instantiation: distances widen every element (l2.rs:128-159), the M-step sums / scales /
    splits in half::f16 arithmetic (kmeans.rs:380,405-418), residuals are f16 subtractions."""
    x = f16_data(6000, 32, 1)
    c0 = f16_data(24, 32, 2)
    ids, dists = eng.assign(x, c0)
    oi, od = oracle.assign(x, c0)
    assert (_np(ids).view(np.uint32) == oi).all() and (_np(dists).view(np.uint32) == od.view(np.uint32)).all()
    cent, loss, iters = eng.kmeans_train(x, 24, max_iters=15, balance_factor=1.0, seed=3)
    oc, ol, oit, _ = oracle.kmeans_train(x, 24, max_iters=15, balance_factor=f32(1.0) / f32(6000), seed=3)
    assert cent.dtype == __import__("torch").float16 and iters == oit and loss == ol
    assert (_np(cent).view(np.uint16) == oc.view(np.uint16)).all()
    part, _ = oracle.assign(x, oc)
    res = oracle.residual(x, oc, part)
    gres = eng.residual(x, oc, part)
    assert (_np(gres).view(np.uint16) == res.view(np.uint16)).all()
    cb, it = eng.pq_train(res, 4, max_iters=10, seed=9)
    ocb, oit2 = oracle.pq_train(res, 4, max_iters=10, seed=9)
    assert (it == oit2.astype(np.uint32)).all()
    assert (_np(cb).view(np.uint16) == ocb.view(np.uint16)).all()
    assert (_np(eng.pq_encode(res, cb)) == oracle.pq_encode(res.astype(f32), ocb.astype(f32))).all()


def test_f16_index_build_and_search_bit_exact(eng, oracle):
    from lance_amd.engine import DeviceIndex
    n, d, nlist, m = 20000, 64, 32, 8
    x = f16_data(n, d, 5)
    q = f16_data(150, d, 6)
    cent, _, _, _ = oracle.kmeans_train(x[:4096], nlist, max_iters=8, seed=1)
    part, _ = oracle.assign(x, cent)
    res = oracle.residual(x, cent, part)
    cb, _ = oracle.pq_train(res[:8192], m, max_iters=6, seed=2)
    oidx = oracle.build_index(x, cent, cb)
    gpart, gcodes, _ = eng.ivfpq_encode(x, cent, cb)
    assert (_np(gpart).view(np.uint32) == oidx.part_ids).all()
    assert (_np(gcodes) == oidx.codes_rowmajor).all()
    gidx = DeviceIndex.create(eng, "l2", cent, cb, gpart, gcodes, None, raw=x)
    xf = x.astype(f32)
    for k, nprobes, rf in ((10, nlist, 0), (10, 6, 0), (10, 6, 10), (100, nlist, 0)):
        gi, gd = gidx.search(q, k, nprobes, rf)
        oi, od = oidx.search(q, k, nprobes, refine=rf, raw=xf)
        assert (_np(gi).view(np.uint64) == oi).all(), (k, nprobes, rf)
        assert (_np(gd).view(np.uint32) == od.view(np.uint32)).all()
    gi, gd = eng.flat_topk(x, q[:40], 10)
    oi, od = oracle.flat_knn(xf, q[:40].astype(f32), 10)
    assert (_np(gi).view(np.uint64) == oi).all() and (_np(gd).view(np.uint32) == od.view(np.uint32)).all()
    gp, gpd = eng.find_partitions(q, cent, 5)
    op, opd = oracle.find_partitions(q.astype(f32), cent.astype(f32), 5)
    assert (_np(gp).view(np.uint32) == op).all() and (_np(gpd).view(np.uint32) == opd.view(np.uint32)).all()


def test_python_api_end_to_end(engine, oracle):
    """create_index / nearest / KMeans mirror the reference API; results equal the oracle run on the
    engine's own trained artefacts (recall check as in v2.rs:1354-1381)."""
    import lance_amd
    from lance_amd.testing import sift_like as latent_sift
    x = latent_sift(30000, 64, 61, n_clusters=32)
    q = latent_sift(100, 64, 62, n_clusters=32)
    idx = lance_amd.create_index(x, "IVF_PQ", metric="l2", num_partitions=32, num_sub_vectors=8, max_iters=10)
    assert idx.info()["n"] == 30000
    oidx = oracle.build_index(x, idx.centroids, idx.codebook)
    ids, dists = idx.nearest(q, k=10, nprobes=32)
    oi, od = oidx.search(q, 10, 32)
    assert (ids.view(np.uint64) == oi).all() and (dists.view(np.uint32) == od.view(np.uint32)).all()
    ids_r, _ = idx.nearest(q, k=10, nprobes=32, refine_factor=10)
    gt, _ = oracle.flat_knn(x, q, 10)
    recall = np.mean([len(set(a.tolist()) & set(b.tolist())) / 10 for a, b in zip(ids_r.view(np.uint64), gt)])
    assert recall >= 0.9
    km = lance_amd.KMeans(8, max_iters=10)
    km.fit(x[:4000])
    assert km.centroids.shape == (8, 64)
    assert (km.predict(x[:100]) == oracle.assign(x[:100], km.centroids)[0]).all()


def test_int8_vectors_widen_to_f32(eng, oracle):
    """Int8 columns (BigANN-style): the reference converts the vectors to f32 and keeps an f32
    model (kmeans.rs:1216-1224, l2.rs:253-260); the int8 element type of the C ABI must equal the oracle
    run on the widened data, bit for bit, through train -> encode -> search -> refine and the flat scan."""
    import torch
    from lance_amd.engine import DeviceIndex
    rng = np.random.default_rng(8)
    n, d, nlist, m = 12000, 64, 16, 8
    centers = rng.integers(-90, 90, (40, d))
    x8 = np.clip(centers[rng.integers(0, 40, n)] + rng.normal(0, 12, (n, d)), -128, 127).astype(np.int8)
    q8 = np.clip(centers[rng.integers(0, 40, 64)] + rng.normal(0, 12, (64, d)), -128, 127).astype(np.int8)
    xf, qf = x8.astype(f32), q8.astype(f32)
    init = xf[oracle.kmeans_init_indices(n, nlist, 3)]
    cent, loss, iters = eng.kmeans_train(torch.from_numpy(x8), nlist, max_iters=10, balance_factor=1.0, init=init, seed=1)
    # the engine applies the reference's k*512 row cap (kmeans.rs:623-627) itself; the oracle trains on the rows it is given
    oc, ol, oit, _ = oracle.kmeans_train(xf[: nlist * 512], nlist, max_iters=10, balance_factor=f32(1.0) / f32(n), init=init, seed=1)
    assert cent.dtype == torch.float32 and (_np(cent).view(np.uint32) == oc.view(np.uint32)).all() and loss == ol and iters == oit
    ids, dists = eng.assign(torch.from_numpy(x8), oc)
    oi, od = oracle.assign(xf, oc)
    assert (_np(ids).view(np.uint32) == oi).all() and (_np(dists).view(np.uint32) == od.view(np.uint32)).all()
    res = eng.residual(torch.from_numpy(x8), oc, oi)
    ores = oracle.residual(xf, oc, oi)
    assert (_np(res).view(np.uint32) == ores.view(np.uint32)).all()
    cb, _ = eng.pq_train(res, m, max_iters=8, seed=2)
    ocb, _ = oracle.pq_train(ores, m, max_iters=8, seed=2)
    assert (_np(cb).view(np.uint32) == ocb.view(np.uint32)).all()
    for metric in ("l2", "dot"):
        part, codes, _ = eng.ivfpq_encode(torch.from_numpy(x8), oc, ocb, metric)
        oidx = oracle.build_index(xf, oc, ocb, metric=metric)
        assert (_np(part).view(np.uint32) == oidx.part_ids).all() and (_np(codes) == oidx.codes_rowmajor).all()
        g = DeviceIndex.create(eng, metric, oc, ocb, part, codes, None, raw=torch.from_numpy(x8), dtype="int8")
        for nprobes, rf in ((4, 0), (nlist, 0), (4, 5)):
            gi, gd = g.search(torch.from_numpy(q8), 10, nprobes, rf)
            oi2, od2 = oidx.search(qf, 10, nprobes, refine=rf, raw=xf if rf else None)
            assert (_np(gi).view(np.uint64) == oi2).all(), (metric, nprobes, rf)
            assert (_np(gd).view(np.uint32) == od2.view(np.uint32)).all()
        pid, pd = eng.find_partitions(torch.from_numpy(q8), oc, 5, metric)
        opid, opd = oracle.find_partitions(qf, oc, 5, metric)
        assert (_np(pid).view(np.uint32) == opid).all() and (_np(pd).view(np.uint32) == opd.view(np.uint32)).all()
        gi, gd = eng.flat_topk(torch.from_numpy(x8), torch.from_numpy(q8), 10, metric)
        oi3, od3 = oracle.flat_knn(xf, qf, 10, metric)
        assert (_np(gi).view(np.uint64) == oi3).all() and (_np(gd).view(np.uint32) == od3.view(np.uint32)).all()
    gi, gd = eng.flat_topk(torch.from_numpy(x8), torch.from_numpy(q8), 10, "cosine")
    oi3, od3 = oracle.flat_knn(xf, qf, 10, "cosine")
    assert (_np(gi).view(np.uint64) == oi3).all() and (_np(gd).view(np.uint32) == od3.view(np.uint32)).all()


def test_f16_mstep_overflow_terminates_like_oracle(eng, oracle):
    """KMeansAlgoFloat<Float16Type> sums the members of a cluster in f16 (kmeans.rs:403-406): SIFT-range values
    overflow to inf, every later distance is NaN and the reference's split_clusters rejection loop would spin
    forever once no cluster has two members.  Engine and oracle must both terminate, and agree."""
    import torch
    rng = np.random.default_rng(0)
    x = rng.integers(150, 218, (4096, 16)).astype(np.float16)     # 512 members x ~184 > 65504 in every cluster
    cent, loss, iters = eng.kmeans_train(torch.from_numpy(x), 8, max_iters=5, seed=1)
    oc, ol, oit, _ = oracle.kmeans_train(x, 8, max_iters=5, seed=1)
    g = _np(cent).astype(f32); o = oc.astype(f32)
    assert iters == oit
    assert (np.isnan(g) == np.isnan(o)).all()
    assert (g[~np.isnan(g)].view(np.uint32) == o[~np.isnan(o)].view(np.uint32)).all()


@pytest.mark.parametrize("metric", ["l2", "dot"])
@pytest.mark.parametrize("d", [32, 128, 40])
def test_ivf_flat_matches_oracle(eng, oracle, metric, d):
    """IVF_FLAT: exact distances inside the probed partitions, FlatIndex heap per partition + SortExec."""
    from lance_amd.engine import DeviceFlatIndex
    n, nlist = 20000, 24
    x = sift_like(n, d, 90 + d)
    x[50:60] = x[7]                      # duplicates: exact ties
    q = sift_like(48, d, 91 + d)
    cent, _, _, _ = oracle.kmeans_train(x[:4096], nlist, max_iters=6, seed=2)
    part, _ = eng.assign(x, cent, metric)
    g = DeviceFlatIndex.create(eng, metric, cent, x, part)
    for k, nprobes in ((10, 5), (1, 3), (50, nlist)):     # the last one hits a boundary tie inside one partition (heap replay)
        gi, gd = g.search(q, k, nprobes)
        oi, od = oracle.ivfflat_search(x, cent, q, k, nprobes, metric)
        assert (_np(gi).view(np.uint64) == oi).all(), (metric, d, k, nprobes)
        assert (_np(gd).view(np.uint32) == od.view(np.uint32)).all()
    if metric == "l2" and d == 128:      # exhaustive probing == flat KNN
        gi, gd = g.search(q, 10, nlist)
        fi, fd = eng.flat_topk(x, q, 10, metric)
        assert (gi == fi).all() and (_np(gd).view(np.uint32) == _np(fd).view(np.uint32)).all()
        import lance_amd
        ix = lance_amd.create_index(x, "IVF_FLAT", metric="l2", num_partitions=nlist, sample_rate=64)
        ids, dd = ix.search_device(q, 10, nlist)
        assert (ids == fi).all()


@pytest.mark.parametrize("d", [128, 40])
def test_ivf_flat_cosine_matches_oracle(eng, oracle, d, tmp_path):
    """IVF_FLAT with the cosine metric (VERDICT missing #5): rows normalised + L2 coarse quantiser at build
    (ivf.rs:147-175), normalised query key (knn.rs:498), cosine_distance inside the partitions; save / load round trip."""
    import lance_amd
    n, nlist = 12000, 16
    x = sift_like(n, d, 190 + d)
    x[50:60] = x[7]                      # duplicates: exact ties
    x[70] = 3.0 * x[71]                  # same direction, different length: a tie only after normalisation
    q = sift_like(40, d, 191 + d)
    cent, _, _, _ = oracle.kmeans_train(oracle.normalize(x[:4096]), nlist, max_iters=6, seed=2)
    ix = lance_amd.create_index(x, "IVF_FLAT", metric="cosine", num_partitions=nlist, ivf_centroids=cent)
    for k, nprobes in ((10, 4), (1, 2), (40, nlist)):
        gi, gd = ix.nearest(q, k=k, nprobes=nprobes)
        oi, od = oracle.ivfflat_search(x, cent, q, k, nprobes, "cosine")
        assert (gi.view(np.uint64) == oi).all(), (d, k, nprobes)
        assert (gd.view(np.uint32) == od.view(np.uint32)).all()
    ix.save(tmp_path / "ivfflat_cos")
    iy = lance_amd.load_index(tmp_path / "ivfflat_cos")
    assert iy.params.metric == "cosine"
    gi2, gd2 = iy.nearest(q, k=10, nprobes=4)
    oi, od = oracle.ivfflat_search(x, cent, q, 10, 4, "cosine")
    assert (gi2.view(np.uint64) == oi).all() and (gd2.view(np.uint32) == od.view(np.uint32)).all()


def test_list_sharded_search_on_device_world1(eng, oracle):
    """lance_amd.dist.search_list_sharded with the real DeviceIndex as the local searcher (world_size 1: the collective
    code runs, the shard is the whole index) must equal the plain search; the 2-rank logic is covered on CPU by
    tests/test_dist_gloo.py."""
    import os
    import torch
    import torch.distributed as dist
    import lance_amd
    from lance_amd.dist import create_list_shard, search_list_sharded
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29777")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        x = sift_like(30000, 64, 301)
        q = sift_like(200, 64, 302)
        idx = lance_amd.create_index(x, "IVF_PQ", metric="l2", num_partitions=32, num_sub_vectors=8, sample_rate=64, engine=eng)
        shard, l2g = create_list_shard(eng, "l2", idx._ix.centroids, idx._ix.codebook, idx.part_ids, idx.codes, raw=torch.from_numpy(x))

        def local_search(qq, kk, nprobes, rf):
            i, d = shard.search(qq, kk, nprobes, rf)
            return i.cpu(), d.cpu()

        for k, nprobes, rf in ((10, 4, 0), (10, 32, 0), (10, 4, 5)):
            gi, gd = search_list_sharded(local_search, l2g.cpu(), torch.from_numpy(q), k, nprobes, rf)
            ri, rd = idx.search_device(q, k, nprobes, rf)
            assert (gi == ri.cpu()).all(), (k, nprobes, rf)
            assert (gd.numpy().view(np.uint32) == rd.cpu().numpy().view(np.uint32)).all()
    finally:
        if created:
            dist.destroy_process_group()


def test_sharded_build_world1_is_bit_identical_to_single_gpu(eng, oracle):
    """lance_amd.dist.create_index_sharded with replicated IVF training, model-parallel PQ and a row-sharded transform
    (RCCL, world_size 1 here) must produce the single-GPU index bit for bit; the row-sharded k-means variant agrees
    to f32 round-off."""
    import os
    import torch
    import torch.distributed as dist
    import lance_amd
    from lance_amd.dist import create_index_sharded
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = "29791"
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        x = torch.from_numpy(sift_like(40000, 64, 411)).cuda()
        a = lance_amd.create_index(x, "IVF_PQ", metric="l2", num_partitions=32, num_sub_vectors=8, sample_rate=64, engine=eng)
        b = create_index_sharded(x, metric="l2", num_partitions=32, num_sub_vectors=8, sample_rate=64, engine=eng)
        assert b.stats.ivf_training == "replicated"
        assert (a.centroids.view(np.uint32) == b.centroids.view(np.uint32)).all()
        assert (a.codebook.view(np.uint32) == b.codebook.view(np.uint32)).all()
        assert torch.equal(a.part_ids, b.part_ids) and torch.equal(a.codes, b.codes)
        c = create_index_sharded(x, metric="l2", num_partitions=32, num_sub_vectors=8, sample_rate=64, engine=eng, ivf_training="sharded")
        assert np.allclose(a.centroids, c.centroids, rtol=1e-4, atol=1e-3)
    finally:
        if created:
            dist.destroy_process_group()


def test_query_chunking_flat_and_ivf_flat(eng, oracle):
    """Both pool-based paths process queries in chunks of 2048: more queries than one chunk, with explicit row ids."""
    from lance_amd.engine import DeviceFlatIndex
    rng = np.random.default_rng(12)
    n, d, nq = 6000, 32, 2500
    x = rng.integers(0, 40, (n, d)).astype(f32)
    q = rng.integers(0, 40, (nq, d)).astype(f32)
    rid = rng.permutation(10 ** 7)[:n].astype(np.uint64)
    gi, gd = eng.flat_topk(x, q, 5, "l2", row_ids=rid)
    oi, od = oracle.flat_knn(x, q, 5, "l2", row_ids=rid)
    assert (_np(gi).view(np.uint64) == oi).all() and (_np(gd).view(np.uint32) == od.view(np.uint32)).all()
    xh = x.astype(np.float16)
    gi, gd = eng.flat_topk(xh, q.astype(np.float16), 5, "l2")
    oi, od = oracle.flat_knn(xh.astype(f32), q, 5, "l2")          # integer values: exact in f16, widening is exact
    assert (_np(gi).view(np.uint64) == oi).all() and (_np(gd).view(np.uint32) == od.view(np.uint32)).all()
    cent = x[:10].copy()
    part, _ = eng.assign(x, cent)
    g = DeviceFlatIndex.create(eng, "l2", cent, x, part, row_ids=rid)
    gi, gd = g.search(q, 5, 10)                                  # all partitions probed == flat KNN (ties by row id)
    fi, fd = oracle.flat_knn(x, q, 5, "l2", row_ids=rid)
    same = _np(gi).view(np.uint64) == fi
    # a boundary tie inside one partition follows the reference heap, not the row id: distances must agree everywhere
    assert (_np(gd).view(np.uint32) == fd.view(np.uint32)).all() and same.mean() > 0.99


def test_many_partitions_nlist_20000(eng, oracle):
    """nlist beyond the LDS-sorted range (C5 shape: nlist 65,536): the two-digit stable partition sort (index layout and
    the 2 x nlist virtual partitions of the partition-major scan) and the pool-based probe selection."""
    from lance_amd.engine import DeviceIndex, DeviceFlatIndex
    rng = np.random.default_rng(21)
    n, d, nlist, m = 100000, 32, 20000, 8
    x = sift_like(n, d, 501)
    q = sift_like(320, d, 502)
    cent = x[rng.permutation(n)[:nlist]].copy() + f32(0.25)          # distinct, well spread centroids
    part, _ = oracle.assign(x, cent, "l2")
    cb, _ = oracle.pq_train(oracle.residual(x[:8192], cent, part[:8192]), m, max_iters=4, seed=3)
    oidx = oracle.build_index(x, cent, cb, "l2")
    gpart, gcodes, _ = eng.ivfpq_encode(x, cent, cb, "l2")
    assert (_np(gpart).view(np.uint32) == oidx.part_ids).all() and (_np(gcodes) == oidx.codes_rowmajor).all()
    g = DeviceIndex.create(eng, "l2", cent, cb, gpart, gcodes, None, raw=x)
    offs, codes_t, rid = g.export()
    assert (offs == oidx.part_offsets).all() and (rid == oidx.row_ids).all() and (codes_t == oidx.codes_t).all()
    pid, pd = eng.find_partitions(q, cent, 20)
    opid, opd = oracle.find_partitions(q, cent, 20)
    assert (_np(pid).view(np.uint32) == opid).all() and (_np(pd).view(np.uint32) == opd.view(np.uint32)).all()
    for k, nprobes, rf in ((10, 20, 0), (10, 64, 5), (5, 3, 0)):       # 320 x 20 pairs -> partition-major path
        gi, gd = g.search(q, k, nprobes, rf)
        oi, od = oidx.search(q, k, nprobes, refine=rf, raw=x if rf else None)
        assert (_np(gi).view(np.uint64) == oi).all(), (k, nprobes, rf)
        assert (_np(gd).view(np.uint32) == od.view(np.uint32)).all()
    fx = DeviceFlatIndex.create(eng, "l2", cent, x, gpart)
    gi, gd = fx.search(q[:32], 10, 30)
    oi, od = oracle.ivfflat_search(x, cent, q[:32], 10, 30, "l2")
    assert (_np(gi).view(np.uint64) == oi).all() and (_np(gd).view(np.uint32) == od.view(np.uint32)).all()


# ---- written after the round's GPU budget was spent: ordered from thin wrappers over proven entry points to new
# ---- native code (index files <-> HBM) ------------------------------------------------
def test_accelerator_module_on_device(eng):
    """lance_amd.accelerator (lance.torch.distance / lance.torch.kmeans names) on the device against the outputs of the
    reference's own module (tests/golden/ref_torch_assign.npz); the same check runs on CPU with an oracle-backed engine
    in tests/test_golden_fixtures.py."""
    import os
    from lance_amd import accelerator
    from test_golden_fixtures import GOLD, _check_accelerator_module
    _check_accelerator_module(accelerator, eng, np.load(os.path.join(GOLD, "ref_torch_assign.npz")))


def test_prefilter_matches_reference_branch(engine, oracle):
    """`nearest=..., filter=..., prefilter=True`: FlatIndex::search's RowIdMask branch (flat/index.rs:129-165) restated in
    the oracle (orc_ivfpq_search_filtered) against the device path (compacted storage, lance_amd/vector.py prefiltered);
    IVF_FLAT and flat KNN under the same masks."""
    import lance_amd
    x = sift_like(20000, 64, 131)
    q = sift_like(50, 64, 132)
    rng = np.random.default_rng(7)
    ix = lance_amd.create_index(x, "IVF_PQ", metric="l2", num_partitions=16, num_sub_vectors=8, max_iters=6, sample_rate=64)
    oidx = oracle.build_index(x, ix.centroids, ix.codebook)
    fx = lance_amd.create_index(x, "IVF_FLAT", metric="l2", num_partitions=16, max_iters=6, sample_rate=64)
    for frac in (0.5, 0.02, 1.0):
        allow = rng.random(x.shape[0]) < frac
        keep = np.nonzero(allow)[0]
        for k, nprobes, rf in ((10, 5, None), (10, 16, 4)):
            gi, gd = ix.nearest(q, k, nprobes, refine_factor=rf, prefilter=allow)
            oi, od = oidx.search(q, k, nprobes, refine=rf or 0, raw=x if rf else None, prefilter=allow)
            assert (gi.view(np.uint64) == oi).all(), (frac, k, nprobes, rf)
            assert (gd.view(np.uint32) == od.view(np.uint32)).all()
        gi, gd = fx.nearest(q, 10, 6, prefilter=allow)
        oi, od = oracle.ivfflat_search(x[keep], fx.centroids, q, 10, 6, "l2", row_ids=keep.astype(np.uint64))
        assert (gi == oi).all() and (gd.view(np.uint32) == od.view(np.uint32)).all()
        fi, fd = lance_amd.flat_knn(x, q, 10, "l2", prefilter=allow)
        oi, od = oracle.flat_knn(x[keep], q, 10, "l2", row_ids=keep.astype(np.uint64))
        assert (_np(fi).view(np.uint64) == oi).all() and (_np(fd).view(np.uint32) == od.view(np.uint32)).all()


def test_gpu_reproduces_what_the_reference_stored(eng, oracle):
    """The HIP path against the reference's OWN outputs (no oracle in between): index files written by Lance 0.27.1 /
    0.21.0 (tests/golden/ref_index.npz; see tests/test_index_files.py for how they are parsed and pinned on the CPU side).
    encode: assign + residual + PQ codes == the stored `__pq_code` bytes; loss == the recorded k-means loss (f64, to the
    bit); k-means over the 256 rows the reference trained on == its stored IVF centroid, bit for bit."""
    import os
    from lance_amd import index_file as IF
    from test_index_files import _legacy_index
    from ref_fixtures import ref_index_dir
    gold = ref_index_dir()
    c = IF.read_index_files(os.path.join(gold, "v0.27.1_pq_in_schema"))
    x = IF.read_column(os.path.join(gold, "v0.27.1_pq_in_schema", "data.lance"), "vec", np.float32, 32)
    x = x[c.row_ids.astype(np.int64)]
    part, codes, loss = eng.ivfpq_encode(x, c.centroids, c.codebook, "l2")
    assert (_np(part).view(np.uint32) == c.part_ids()).all()
    assert (_np(codes) == c.codes_row_major()).all()
    assert loss == c.loss
    base = os.path.join(gold, "v0.21.0_legacy")
    cent, cb, lengths, raw = _legacy_index(os.path.join(base, "index_256.idx"))
    x2 = IF.read_column(os.path.join(base, "data_256.lance"), "vector", np.float32, 16)
    gc, _, _ = eng.kmeans_train(x2, 1, max_iters=50, seed=9)
    assert (_np(gc).view(np.uint32) == cent.view(np.uint32)).all()
    rid = np.frombuffer(raw[256:256 + 8 * 256], np.uint64)
    rows = (rid & np.uint64(0xFFFFFFFF)).astype(np.int64)
    _, codes2, _ = eng.ivfpq_encode(x2[rows], cent, cb, "l2")
    assert (_np(codes2)[:, 0] == np.frombuffer(raw[:256], np.uint8)).all()
    # Lance 0.8.14, IVF4 / PQ16 over 128-d vectors (the BASELINE C2 shape): stored partition and 16 code bytes of 3000 rows
    z = np.load(os.path.join(gold, "v0.8.14_ivf4_pq16.npz"))
    for k in (0, 1):
        xs = z["x"][z[f"rows{k}"]]
        part, codes, _ = eng.ivfpq_encode(xs, z[f"centroids{k}"], z[f"codebook{k}"], "l2")
        assert (_np(part).view(np.uint32) == z[f"part{k}"]).all()
        assert (_np(codes) == z[f"codes{k}"]).all()


@pytest.mark.parametrize("metric", ["l2", "dot"])
def test_distance_range_search(eng, oracle, metric):
    """`nearest={..., "distance_range": (lower, upper)}`: inside each probed partition only rows with lower <= d < upper
    enter the k-heap (flat/index.rs:98-113; v2.rs distance-range tests).  Batched form of what lance_hip_pq_scan_topk
    does for one partition; large batches leave the partition-major path for the query-major kernel."""
    from lance_amd.vector import IvfPqIndex, IvfPqParams
    n, d, nlist, m = 20000, 64, 24, 8
    x = sift_like(n, d, 141)
    oidx, gidx = _build_pair(eng, oracle, x, nlist, m, metric)
    ix = IvfPqIndex(gidx, IvfPqParams(nlist, m, 8, metric))
    for nq in (40, 600):                               # 600 x 8 probes >= 4096 pairs: would take the PM path without a range
        q = sift_like(nq, d, 142 + nq)
        _, ud = oidx.search(q, 60, 8)
        fin = ud[np.isfinite(ud)]
        lo, hi = float(np.quantile(fin, 0.2)), float(np.quantile(fin, 0.6))
        for k, nprobes, rng_ in ((10, 8, (lo, hi)), (10, 8, (None, hi)), (25, 3, (lo, None)), (5, nlist, (hi, hi))):
            gi, gd = ix.nearest(q, k, nprobes, distance_range=rng_)
            oi, od = oidx.search(q, k, nprobes, lower=rng_[0] if rng_[0] is not None else np.finfo(f32).min,
                                 upper=rng_[1] if rng_[1] is not None else np.finfo(f32).max)
            assert (gi.view(np.uint64) == oi).all(), (metric, nq, k, nprobes, rng_)
            assert (gd.view(np.uint32) == od.view(np.uint32)).all()
        # with refine: ADC range in the partitions, exact re-rank of the k * rf candidates, exact distances filtered by the
        # same range before the fetch (scanner.rs:3334-3377) -- exact L2 / dot distances live on another scale than the
        # ADC ones, so take the bounds from the exact neighbourhood
        _, ed = oracle.flat_knn(x, q, 40, metric)
        elo, ehi = float(np.quantile(ed, 0.1)), float(np.quantile(ed, 0.9))
        for k, nprobes, rf, rng_ in ((10, 8, 4, (elo, ehi)), (5, nlist, 10, (None, ehi))):
            gi, gd = ix.nearest(q, k, nprobes, refine_factor=rf, distance_range=rng_)
            oi, od = oidx.search(q, k, nprobes, refine=rf, raw=x, lower=rng_[0] if rng_[0] is not None else np.finfo(f32).min,
                                 upper=rng_[1] if rng_[1] is not None else np.finfo(f32).max)
            assert (gi.view(np.uint64) == oi).all(), (metric, nq, k, nprobes, rf, rng_)
            assert (gd.view(np.uint32) == od.view(np.uint32)).all()
        # range + row-id prefilter, both tested inside the scan (flat/index.rs:131-149); no filtered copy of the index
        allow = np.random.default_rng(nq).random(n) < 0.4
        for k, nprobes, rf, rng_ in ((10, 8, 0, (lo, hi)), (10, nlist, 0, (None, hi)), (5, 8, 4, (elo, ehi))):
            gi, gd = ix.nearest(q, k, nprobes, refine_factor=rf or None, distance_range=rng_, prefilter=allow)
            oi, od = oidx.search(q, k, nprobes, refine=rf, raw=x if rf else None, prefilter=allow,
                                 lower=rng_[0] if rng_[0] is not None else np.finfo(f32).min,
                                 upper=rng_[1] if rng_[1] is not None else np.finfo(f32).max)
            assert (gi.view(np.uint64) == oi).all(), (metric, nq, k, nprobes, rf, rng_, "prefilter")
            assert (gd.view(np.uint32) == od.view(np.uint32)).all()


def test_load_reference_written_index_and_search(eng, oracle):
    """An index directory written by real Lance (tests/golden/ref_index.npz, 512 x 32, IVF1,PQ4) goes files -> HBM through
    lance_hip_index_load and answers queries exactly as the oracle does on the same stored model (the oracle's encode of
    the fixture's raw vectors equals the stored codes: tests/test_index_files.py)."""
    import os
    import lance_amd
    from lance_amd import index_file as IF
    from ref_fixtures import ref_index_dir
    ref = os.path.join(ref_index_dir(), "v0.27.1_pq_in_schema")
    c = IF.read_index_files(ref)
    x = IF.read_column(os.path.join(ref, "data.lance"), "vec", np.float32, 32)
    ix = lance_amd.load_index(ref, raw=x, engine=eng)
    assert ix.info() == {"n": 512, "nlist": 1, "m": 4, "d": 32}
    offs, codes_t, rid = ix.export_storage()
    assert (offs == c.part_offsets).all() and (codes_t == c.codes).all() and (rid == c.row_ids).all()
    oidx = oracle.build_index(x, c.centroids, c.codebook)
    assert (oidx.codes_t == c.codes).all()
    rng = np.random.default_rng(5)
    q = np.concatenate([x[:40], rng.random((60, 32)).astype(f32)])
    for k, rf in ((10, 0), (1, 0), (100, 0), (10, 5)):
        gi, gd = ix.search_device(q, k, 1, rf)
        oi, od = oidx.search(q, k, 1, refine=rf, raw=x) if rf else oidx.search(q, k, 1)
        assert (_np(gi).view(np.uint64) == oi).all(), (k, rf)
        assert (_np(gd).view(np.uint32) == od.view(np.uint32)).all()


@pytest.mark.parametrize("kind", ["f32", "f16", "4bit", "dot", "cosine"])
def test_index_save_load_roundtrip(eng, oracle, tmp_path, kind):
    """HBM -> files (lance_hip_index_save) -> HBM (lance_hip_index_load): identical storage and identical answers; the
    files equal what the host-side writer produces from the exported arrays, and the oracle agrees with the re-loaded
    index.  The indices are built exactly as in the tests above (oracle-trained model, lance_hip_ivfpq_encode)."""
    from lance_amd import index_file as IF
    from lance_amd.engine import DeviceIndex
    n, d, nlist = 12000, 64, 20
    metric = kind if kind in ("dot", "cosine") else "l2"
    nbits, m = (4, 16) if kind == "4bit" else (8, 8)
    if kind == "f16":
        x, q = f16_data(n, d, 77), f16_data(64, d, 78)
    else:
        x, q = sift_like(n, d, 77) + (1.0 if kind == "cosine" else 0.0), sift_like(64, d, 78) + (1.0 if kind == "cosine" else 0.0)
    xs = oracle.normalize(x) if metric == "cosine" else x
    kmetric = "l2" if metric == "cosine" else metric
    cent, _, _, _ = oracle.kmeans_train(xs[:4096], nlist, max_iters=6, seed=1, metric=kmetric)
    part, _ = oracle.assign(xs, cent, kmetric)
    res = oracle.residual(xs, cent, part) if kmetric == "l2" else xs
    cb, _ = oracle.pq_train(res[:8192], m, nbits=nbits, max_iters=6, seed=2)
    gpart, gcodes, _ = eng.ivfpq_encode(x, cent, cb, metric)
    ix = DeviceIndex.create(eng, metric, cent, cb, gpart, gcodes)
    ix.save(tmp_path / "a", loss=1234.5678)
    c = IF.read_index_files(tmp_path / "a")
    offs, codes_t, rid = ix.export()
    assert (c.metric, c.dtype, c.nbits, c.num_sub_vectors, c.loss) == (metric, "float16" if kind == "f16" else "float32", nbits, m, 1234.5678)
    assert (c.part_offsets == offs).all() and (c.codes == codes_t).all() and (c.row_ids == rid).all()
    assert (c.centroids == np.asarray(cent, f32)).all() and (c.codebook == np.asarray(cb, f32)).all()
    IF.write_index_files(tmp_path / "b", c)            # host writer on the same contents: same bytes
    for name in ("index.idx", "auxiliary.idx"):
        assert (tmp_path / "a" / name).read_bytes() == (tmp_path / "b" / name).read_bytes(), name
    ix2 = DeviceIndex.load(eng, tmp_path / "a")
    o2, c2, r2 = ix2.export()
    assert (o2 == offs).all() and (c2 == codes_t).all() and (r2 == rid).all()
    oidx = oracle.build_index(x, cent, cb, metric, nbits=nbits)
    for k, nprobes in ((10, nlist), (10, 3)):
        a = ix.search(q, k, nprobes)
        b = ix2.search(q, k, nprobes)
        assert (a[0] == b[0]).all() and (_np(a[1]).view(np.uint32) == _np(b[1]).view(np.uint32)).all()
        oi, od = oidx.search(q, k, nprobes)
        assert (_np(b[0]).view(np.uint64) == oi).all() and (_np(b[1]).view(np.uint32) == od.view(np.uint32)).all()


def test_ivf_flat_save_load_roundtrip(eng, oracle, tmp_path):
    import lance_amd
    from lance_amd import index_file as IF
    x = sift_like(9000, 48, 81)
    q = sift_like(40, 48, 82)
    ix = lance_amd.create_index(x, "IVF_FLAT", metric="l2", num_partitions=12, sample_rate=64)
    ix.save(tmp_path / "f")
    c = IF.read_index_files(tmp_path / "f")
    assert c.index_type == "IVF_FLAT" and c.vectors.shape == (9000, 48)
    assert (c.vectors == x[c.row_ids.astype(np.int64)]).all()
    ix2 = lance_amd.load_index(tmp_path / "f", engine=eng)
    for k, nprobes in ((10, 4), (5, 12)):
        a = ix.search_device(q, k, nprobes); b = ix2.search_device(q, k, nprobes)
        assert (a[0] == b[0]).all() and (_np(a[1]).view(np.uint32) == _np(b[1]).view(np.uint32)).all()


def test_load_legacy_reference_index_c2_shape(eng, oracle, tmp_path):
    """A legacy-format IVF_PQ index written by Lance 0.8.14 (2000 x 128, 4 partitions, 16 sub-vectors: the BASELINE C2
    shape) goes file -> HBM (row-major codes, row addresses as ids) and is searched exactly as the oracle searches the
    same stored model and codes."""
    import os
    import shutil
    import lance_amd
    from lance_amd import index_file as IF
    from ref_fixtures import ref_index_dir
    gold = ref_index_dir()
    d = tmp_path / "legacy"
    d.mkdir()
    shutil.copyfile(os.path.join(gold, "v0.8.14_legacy", "index_2000.idx"), d / "index.idx")
    c = IF.read_index_files(d)
    rm = c.codes_row_major()
    codes_t = np.concatenate([rm[c.part_offsets[p]:c.part_offsets[p + 1]].T.reshape(-1) for p in range(4)])
    oidx = oracle.IvfPqIndex("l2", c.centroids, c.codebook, c.part_offsets, codes_t, c.row_ids)
    ix = lance_amd.load_index(d, engine=eng)
    assert ix.info() == {"n": 2000, "nlist": 4, "m": 16, "d": 128}
    offs, ct, rid = ix.export_storage()
    assert (offs == c.part_offsets).all() and (ct == codes_t).all() and (rid == c.row_ids).all()
    z = np.load(os.path.join(gold, "v0.8.14_ivf4_pq16.npz"))
    rng = np.random.default_rng(17)
    q = np.concatenate([z["x"][:50], z["x"][50:100] + rng.normal(0, 0.05, (50, 128)).astype(f32)]).astype(f32)
    for k, nprobes in ((10, 4), (10, 1), (100, 2), (1, 4)):
        gi, gd = ix.search_device(q, k, nprobes)
        oi, od = oidx.search(q, k, nprobes)
        assert (_np(gi).view(np.uint64) == oi).all(), (k, nprobes)
        assert (_np(gd).view(np.uint32) == od.view(np.uint32)).all()


def test_load_list_shard_world1(eng, oracle, tmp_path):
    """lance_amd.dist.load_list_shard (index files -> this rank's lists in HBM) with world_size 1: the shard is the whole
    index, so search_list_sharded must equal the search of lance_amd.load_index on the same directory.  The multi-rank
    placement is covered on CPU (tests/test_dist_gloo.py::test_list_sharded_search_from_reference_index_files)."""
    import os
    import shutil
    import torch
    import torch.distributed as dist
    import lance_amd
    from lance_amd.dist import load_list_shard, search_list_sharded
    from ref_fixtures import ref_index_dir
    gold = ref_index_dir()
    d = tmp_path / "legacy"
    d.mkdir()
    shutil.copyfile(os.path.join(gold, "v0.8.14_legacy", "index_2000.idx"), d / "index.idx")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29778")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        whole = lance_amd.load_index(d, engine=eng)
        shard, l2g = load_list_shard(eng, d)
        q = np.load(os.path.join(gold, "v0.8.14_ivf4_pq16.npz"))["x"][:80]

        def local_search(qq, kk, nprobes, rf):
            i, dd = shard.search(qq, kk, nprobes, rf)
            return i.cpu(), dd.cpu()

        for k, nprobes in ((10, 4), (10, 2), (50, 1)):
            gi, gd = search_list_sharded(local_search, l2g, torch.from_numpy(q), k, nprobes)
            ri, rd = whole.search_device(q, k, nprobes)
            assert (gi == ri.cpu()).all(), (k, nprobes)
            assert (gd.numpy().view(np.uint32) == rd.cpu().numpy().view(np.uint32)).all()
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_load_lists_shards_on_one_gpu(eng, oracle, tmp_path, world):
    """lance_hip_index_load_lists: every shard (lists p % world == r) of the Lance-0.8.14 index loaded on this one GPU; the
    shards' answers merged by (dist, rowid) must be the whole index's answer.  No process group: the placement and the
    native shard loader are what is under test."""
    import os
    import shutil
    import torch
    import lance_amd
    from lance_amd.dist import merge_topk
    from lance_amd.engine import DeviceIndex
    from ref_fixtures import ref_index_dir
    d = tmp_path / "legacy"
    d.mkdir()
    shutil.copyfile(os.path.join(ref_index_dir(), "v0.8.14_legacy", "index_2000.idx"), d / "index.idx")
    whole = lance_amd.load_index(d, engine=eng)
    shards = [DeviceIndex.load(eng, d, lists=(world, r)) for r in range(world)]
    assert sum(s.info()["n"] for s in shards) == 2000
    for r, s in enumerate(shards):
        offs, _, _ = s.export()
        lens = np.diff(offs.astype(np.int64))
        assert all(lens[p] == 0 for p in range(4) if p % world != r) and lens.sum() == s.info()["n"]
    q = np.load(os.path.join(ref_index_dir(), "v0.8.14_ivf4_pq16.npz"))["x"][:80]
    for k, nprobes in ((10, 4), (10, 2), (40, 3)):
        parts = [s.search(q, k, nprobes) for s in shards]
        gi, gd = merge_topk(torch.cat([p[0] for p in parts], 1), torch.cat([p[1] for p in parts], 1), k)
        ri, rd = whole.search_device(q, k, nprobes)
        assert (gi == ri).all(), (world, k, nprobes)
        assert (_np(gd).view(np.uint32) == _np(rd).view(np.uint32)).all()


def test_rowsharded_build_and_list_shards_world1(eng, oracle):
    """lance_amd.dist.create_index_rowsharded (device-resident sharded Lloyd loop: enqueue-only E-step / all-reduce / update),
    replica_index, list_shard_index (all_to_all by list owner) and search_list_sharded with the device (dist, rowid) merge --
    RCCL with one rank here; the 2- and 3-rank logic runs on CPU in tests/test_dist_gloo.py.  Everything searched must equal the
    oracle on the trained model, and the device merge must equal the torch-sort merge."""
    import os
    import torch
    import torch.distributed as dist
    from lance_amd import dist as ld
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = "29793"
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        xh = sift_like(40000, 64, 511)
        qh = sift_like(300, 64, 512)
        x = torch.from_numpy(xh).cuda()
        for mode in ("sharded", "replicated"):
            b = ld.create_index_rowsharded(x, metric="l2", num_partitions=32, num_sub_vectors=16, sample_rate=64, engine=eng, ivf_training=mode)
            assert b.stats.ivf_iters >= 1 and b.n_total == 40000 and b.row0 == 0
            cent, cb = b.centroids.cpu().numpy(), b.codebook.cpu().numpy()
            oidx = oracle.build_index(xh, cent, cb)
            assert (b.part_local.cpu().numpy().view(np.uint32) == oidx.part_ids).all() and (b.codes_local.cpu().numpy() == oidx.codes_rowmajor).all()
            rep, _ = ld.replica_index(b, x, engine=eng)
            shard, l2g = ld.list_shard_index(b, x, engine=eng)
            for k, nprobes, rf in ((10, 6, 0), (10, 32, 0), (10, 6, 5)):
                oi, od = oidx.search(qh, k, nprobes, refine=rf, raw=xh if rf else None)
                gi, gd = rep.search_device(qh, k, nprobes, rf)
                assert (_np(gi).view(np.uint64) == oi).all() and (_np(gd).view(np.uint32) == od.view(np.uint32)).all(), (mode, "replica", k, nprobes, rf)
                fn = lambda qq, kk, npb, r: shard.search(qq, kk, npb, r)
                li, ldist = ld.search_list_sharded(fn, l2g, torch.from_numpy(qh).cuda(), k, nprobes, rf, engine=eng)      # device merge
                ti, td = ld.search_list_sharded(fn, l2g, torch.from_numpy(qh).cuda(), k, nprobes, rf)                 # torch-sort merge
                assert (_np(li).view(np.uint64) == oi).all() and (_np(ldist).view(np.uint32) == od.view(np.uint32)).all(), (mode, "lists", k, nprobes, rf)
                assert torch.equal(li, ti) and torch.equal(ldist, td)
                # round 6: the local half as ONE scan (lance_hip_ivfpq_search_candidates: PQ order + exact distances, aligned)
                scans = eng.timing_query("count:refine")[1]
                ci, cdist = ld.search_list_sharded(fn, l2g, torch.from_numpy(qh).cuda(), k, nprobes, rf, engine=eng,
                                                   local_candidates=lambda qq, ke, npb: shard.search_candidates(qq, ke, npb))
                assert torch.equal(ci, li) and torch.equal(cdist, ldist), (mode, "one-scan candidates", k, nprobes, rf)
                if rf:
                    assert eng.timing_query("count:refine")[1] == scans + 1      # one pass of the refine kernels, no second search
                    # the candidate call itself: ids / PQ distances = search(k = keff, no refine); exact = the oracle's re-ranked distances
                    i3, p3, e3 = shard.search_candidates(qh, k * rf, nprobes)
                    i0, p0 = shard.search(qh, k * rf, nprobes, 0)
                    assert torch.equal(i3, i0) and torch.equal(p3, p0)
                    oi2, od2 = oidx.search(qh, k * rf, nprobes, refine=1, raw=xh)       # same candidate set, exact distances, re-ranked
                    gl = torch.where(i3 < 0, i3, l2g.to(i3.device)[i3.clamp(min=0)]) if l2g.numel() else i3
                    got = {(int(a), int(b)) for a, b in zip(_np(gl)[0], _np(e3)[0].view(np.uint32)) if a >= 0}
                    want = {(int(a), int(b)) for a, b in zip(oi2[0].astype(np.int64), od2[0].view(np.uint32)) if a != np.iinfo(np.uint64).max}
                    assert got == want
            shard.close()
        # the sharded loop on one rank adds the rows in the single-GPU order: same centroids as the single-GPU trainer
        samp = x[:16384]
        c1, l1, i1 = eng.kmeans_train(samp, 32, max_iters=20, balance_factor=1.0, seed=3)
        c2, l2, i2 = ld.train_kmeans_sharded(eng, samp, 32, 16384, max_iters=20, balance_factor=1.0, init=None, seed=3)
        init = samp[torch.from_numpy(oracle.kmeans_init_indices(16384, 32, 3).astype(np.int64)).cuda()]
        c3, l3, i3 = eng.kmeans_train(samp, 32, max_iters=20, balance_factor=1.0, init=init, seed=3)
        assert i2 >= 1 and np.allclose(_np(c2), _np(c3), rtol=1e-5, atol=1e-4)
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("metric", ["l2", "dot"])
@pytest.mark.parametrize("d", [32, 128, 96])
def test_flat_batch_on_matrix_cores_bit_exact(eng, oracle, d, metric):
    """Query batches (>= 128 queries, d % 16 == 0) run the flat scan's later epochs through the bf16x3 MFMA surrogate + exact
    re-check (flat_mfma.hip).  Ids and distances must equal the oracle's flat scan, ties (duplicate rows, integer data)
    included, and the run with LANCE_HIP_NO_MFMA_FLAT (exact VALU kernel) must agree bit for bit."""
    rng = np.random.default_rng(d)
    n, nq = 60000, 300
    x = sift_like(n, d, 900 + d)
    x[1000:1040] = x[7]                              # 40 identical rows: ties broken by row id
    q = sift_like(nq, d, 901 + d)
    q[:5] = x[7]
    if metric == "dot":
        x = x / 64.0; q = q / 64.0
    for k in (10, 100):
        gi, gd = eng.flat_topk(x, q, k, metric)
        oi, od = oracle.flat_knn(x, q, k, metric)
        assert (_np(gi).view(np.uint64) == oi).all(), (d, metric, k)
        assert (_np(gd).view(np.uint32) == od.view(np.uint32)).all()
    # real-valued data too (no exact ties, near ties everywhere)
    xr = (rng.standard_normal((n, d)) * 3).astype(f32); qr = (rng.standard_normal((nq, d)) * 3).astype(f32)
    gi, gd = eng.flat_topk(xr, qr, 10, metric)
    oi, od = oracle.flat_knn(xr, qr, 10, metric)
    assert (_np(gi).view(np.uint64) == oi).all() and (_np(gd).view(np.uint32) == od.view(np.uint32)).all()


@pytest.mark.parametrize("kind,metric", [("f16", "l2"), ("int8", "l2"), ("int8", "dot")])
@pytest.mark.parametrize("d", [64, 128])
def test_flat_scan_reads_f16_int8_rows_natively(eng, oracle, monkeypatch, kind, metric, d):
    """f16 / int8 columns: the fixed-dimension flat kernels (exact filter for small batches, MFMA filter + exact re-check for
    batches of >= 128 queries) read the rows in the column's own element type and widen per element in registers
    (l2.rs:128-159, :253-260) -- no f32 copy of the column.  Same bits as the oracle and as the widened route."""
    import torch
    rng = np.random.default_rng(3 * d + len(kind))
    n = 50000
    if kind == "f16":
        xs = (rng.standard_normal((n, d)) * 4).astype(np.float16); qs = (rng.standard_normal((260, d)) * 4).astype(np.float16)
    else:
        xs = rng.integers(-100, 100, (n, d)).astype(np.int8); qs = rng.integers(-100, 100, (260, d)).astype(np.int8)
    xs[2000:2030] = xs[11]                           # ties broken by row id
    xf, qf = xs.astype(f32), qs.astype(f32)
    for nq in (7, 260):
        gi, gd = eng.flat_topk(torch.from_numpy(xs), torch.from_numpy(qs[:nq]), 10, metric)
        oi, od = oracle.flat_knn(xf, qf[:nq], 10, metric)
        assert (_np(gi).view(np.uint64) == oi).all(), (kind, metric, d, nq)
        assert (_np(gd).view(np.uint32) == od.view(np.uint32)).all()
        monkeypatch.setenv("LANCE_HIP_NO_NATIVE_FLAT", "1")
        wi, wd = eng.flat_topk(torch.from_numpy(xs), torch.from_numpy(qs[:nq]), 10, metric)
        monkeypatch.delenv("LANCE_HIP_NO_NATIVE_FLAT")
        assert (gi == wi).all() and (_np(gd).view(np.uint32) == _np(wd).view(np.uint32)).all()
