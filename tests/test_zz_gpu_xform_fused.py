"""GPU parity of the single-pass transform (lance_amd/csrc/xform_fused.hip): coarse assign + exact re-check + residual + PQ encode
in one kernel, against the oracle's chain (IvfTransformer, ivf.rs:188-236: compute_partitions kmeans.rs:1187-1246 ->
do_compute_residual residual.rs:58-102 -> ProductQuantizer::transform pq.rs:116-191), bit for bit: partition ids, PQ codes and
the f64 loss (sum of the assignment distances, kmeans.rs:1276-1290).

Cases: every element type (f32 / f16 / int8) x metric (l2 / dot / cosine), sub-dimension 8 and 4, dimensions that leave the
last MFMA k-step ragged in M (48, 80, 112), row counts that are not a multiple of the 128-row workgroup, non-finite and zero
rows, duplicated centroids (two, three, and five of a kind: the last forces the recompute list) and duplicated codewords (the
fix list).  Every case asserts that the fused kernel served the call.  The file sorts late on purpose: newest device code last."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
f32 = np.float32


def _np(t):
    return t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)


@pytest.fixture(scope="module")
def eng():
    import lance_amd
    return lance_amd.default_engine()


def clustered(n, d, seed, ncl=40, scale=40.0, noise=12.0, integer=True):
    rng = np.random.default_rng(seed)
    c = rng.uniform(0, scale * 3, (ncl, d))
    x = c[rng.integers(0, ncl, n)] + rng.normal(0, noise, (n, d))
    return np.rint(x) if integer else x


def oracle_chain(oracle, x, cent, cb, metric):
    """-> (part ids [n] with NONE for dropped rows, codes [n][m] (rows without a partition: unspecified), loss)"""
    f16 = x.dtype == np.float16
    xs = x
    if metric == "cosine":
        xs = oracle.normalize(x)
    keep = oracle.is_finite(xs)
    sm = "l2" if metric == "cosine" else metric
    part = np.full(x.shape[0], oracle.NONE, np.uint32)
    dist = np.full(x.shape[0], np.inf, f32)
    xk = np.ascontiguousarray(xs[keep])
    pk, dk = oracle.assign(xk, cent.astype(np.float16) if f16 else cent, sm)
    part[keep] = pk; dist[keep] = dk
    res = oracle.residual(xk, cent, np.where(pk == oracle.NONE, 0, pk)) if sm == "l2" else xk
    codes = np.zeros((x.shape[0], cb.shape[0]), np.uint8)
    codes[keep] = oracle.pq_encode(res, cb, "l2")
    losses = np.zeros(cent.shape[0], np.float64)
    for r in np.nonzero(part != oracle.NONE)[0]:
        losses[part[r]] += np.float64(dist[r])
    tot = np.float64(0.0)
    for v in losses:
        tot = tot + v
    return part, codes, float(tot)


def run_case(eng, oracle, x, cent, cb, metric, stage="xform_fused"):
    before = eng.timing_query("count:" + stage)[1]
    part, codes, loss = eng.ivfpq_encode(x, cent, cb, metric)
    import os
    assert os.environ.get("XF_ALLOW_OLD") or eng.timing_query("count:" + stage)[1] == before + 1, f"the {stage} kernel did not serve this call"
    part = _np(part).view(np.uint32); codes = _np(codes)
    opart, ocodes, oloss = oracle_chain(oracle, x, cent, cb, metric)
    bad = np.nonzero(part != opart)[0]
    assert bad.size == 0, (metric, x.dtype, "partition ids differ at rows", bad[:10], part[bad[:10]], opart[bad[:10]])
    ok = opart != oracle.NONE
    badc = np.nonzero((codes[ok] != ocodes[ok]).any(axis=1))[0]
    assert badc.size == 0, (metric, x.dtype, "codes differ at", badc[:10])
    assert loss == oloss, (loss, oloss)


def spoil(x):
    x = x.copy()
    if x.dtype != np.int8:
        x[17] = np.nan; x[99, 3] = np.inf; x[x.shape[0] - 7, 0] = -np.inf
    x[100] = 0
    return x


def dup_centroids(cent):
    cent = cent.copy()
    k = cent.shape[0]
    cent[9] = cent[k - 3]                       # two of a kind: the smaller index wins
    cent[4] = cent[11]; cent[20] = cent[11]     # three of a kind
    for i in (1, 6, 13, 27):                    # five of a kind: more candidates than the kernel keeps -> recompute list
        cent[i] = cent[30]
    return cent


def dup_codewords(cb):
    cb = cb.copy()
    cb[min(1, cb.shape[0] - 1), 7] = cb[min(1, cb.shape[0] - 1), 100]
    cb[-1, 255] = cb[-1, 0]
    return cb


@pytest.mark.parametrize("metric", ["l2", "dot", "cosine"])
@pytest.mark.parametrize("d,m,nlist,n", [(128, 16, 256, 5037), (64, 8, 40, 3000), (128, 32, 100, 2177), (48, 6, 33, 2100), (16, 2, 32, 2048),
                                         (80, 20, 64, 2300), (112, 14, 70, 2600), (32, 8, 1000, 4100),
                                         (128, 8, 64, 2500), (48, 3, 40, 2200), (16, 1, 32, 2100)])      # sub-dimension 16
def test_f32_rows(eng, oracle, metric, d, m, nlist, n):
    x = spoil(clustered(n, d, 7 + d + m).astype(f32))
    if metric == "cosine":
        x[100] = 1.0          # a zero row has no direction: normalize gives NaN and the row is dropped -- keep one ordinary row instead
        x[101] = 0.0
    rng = np.random.default_rng(d * 3 + m)
    xs = oracle.normalize(x) if metric == "cosine" else x
    fin = oracle.is_finite(xs)
    cent = dup_centroids(np.ascontiguousarray(xs[fin][rng.choice(int(fin.sum()), nlist, replace=False)]))
    part, _ = oracle.assign(np.ascontiguousarray(xs[fin]), cent, "l2" if metric == "cosine" else metric)
    res = oracle.residual(np.ascontiguousarray(xs[fin]), cent, np.where(part == oracle.NONE, 0, part)) if metric != "dot" else xs[fin]
    sd = d // m
    cb = np.stack([res[rng.choice(res.shape[0], 256, replace=False)][:, i * sd:(i + 1) * sd] for i in range(m)]).astype(f32)
    run_case(eng, oracle, x, cent, dup_codewords(cb), metric)


@pytest.mark.parametrize("metric", ["l2", "cosine"])        # f16 under dot with d > 16 keeps the 32-lane kernels (not this path)
@pytest.mark.parametrize("d,m,nlist,n", [(128, 16, 256, 4133), (64, 16, 50, 2500), (96, 12, 64, 2200)])
def test_f16_rows(eng, oracle, metric, d, m, nlist, n):
    x = spoil((clustered(n, d, 70 + d, scale=1.5, noise=0.6, integer=False)).astype(np.float16))
    if metric == "cosine":
        x[100] = 1.0
    rng = np.random.default_rng(d + m)
    xs = oracle.normalize(x) if metric == "cosine" else x
    fin = oracle.is_finite(xs)
    cent = dup_centroids(np.ascontiguousarray(xs[fin][rng.choice(int(fin.sum()), nlist, replace=False)]))
    part, _ = oracle.assign(np.ascontiguousarray(xs[fin]), cent, "l2")
    res = oracle.residual(np.ascontiguousarray(xs[fin]), cent, np.where(part == oracle.NONE, 0, part))
    sd = d // m
    cb = np.stack([res[rng.choice(res.shape[0], 256, replace=False)][:, i * sd:(i + 1) * sd] for i in range(m)]).astype(np.float16)
    run_case(eng, oracle, x, cent, dup_codewords(cb), metric)


def test_f16_rows_dot_d16(eng, oracle):
    """d = 16: the 32-lane and the 16-lane dot orders coincide, so an f16 column under dot takes the fused kernel"""
    n, d, m, nlist = 2400, 16, 4, 40
    x = spoil((clustered(n, d, 5, scale=1.5, noise=0.6, integer=False)).astype(np.float16))
    rng = np.random.default_rng(3)
    cent = dup_centroids(np.ascontiguousarray(x[200:200 + nlist]))
    cb = np.stack([x[rng.choice(np.arange(200, n), 256, replace=False)][:, i * 4:(i + 1) * 4] for i in range(m)]).astype(np.float16)
    run_case(eng, oracle, x, cent, dup_codewords(cb), "dot")


@pytest.mark.parametrize("metric", ["l2", "dot"])
@pytest.mark.parametrize("d,m,nlist,n", [(128, 32, 300, 4099), (128, 16, 64, 2304), (32, 8, 48, 2050)])
def test_int8_rows(eng, oracle, metric, d, m, nlist, n):
    rng = np.random.default_rng(d + m + nlist)
    x = spoil(np.clip(clustered(n, d, 90 + d, scale=30.0, noise=14.0) - 40, -128, 127).astype(np.int8))
    cent = dup_centroids(x[rng.choice(n, nlist, replace=False)].astype(f32) + rng.normal(0, 0.25, (nlist, d)).astype(f32))
    part, _ = oracle.assign(x.astype(f32), cent, metric)
    res = oracle.residual(x.astype(f32), cent, np.where(part == oracle.NONE, 0, part)) if metric == "l2" else x.astype(f32)
    sd = d // m
    cb = np.stack([res[rng.choice(n, 256, replace=False)][:, i * sd:(i + 1) * sd] for i in range(m)]).astype(f32)
    run_case(eng, oracle, x, cent, dup_codewords(cb), metric)


def test_large_values_and_tiny_values(eng, oracle):
    """magnitudes at both ends of what bf16 fragments carry: 1e18-scale rows (squares near the top of f32) and 1e-20-scale rows
    (squares underflow): the margins turn infinite / zero and the items must fall to the exact kernels, not to a wrong code"""
    n, d, m, nlist = 2200, 32, 4, 32
    rng = np.random.default_rng(12)
    for scale in (1e18, 1e-20):
        x = (rng.standard_normal((n, d)) * scale).astype(f32)
        cent = np.ascontiguousarray(x[:nlist])
        cb = np.stack([x[rng.choice(n, 256, replace=False)][:, i * 8:(i + 1) * 8] for i in range(m)]).astype(f32)
        print("scale", scale)
        run_case(eng, oracle, x, cent, cb, "l2")


@pytest.mark.parametrize("metric", ["l2", "dot", "cosine"])
@pytest.mark.parametrize("d,m,nlist,n", [(256, 16, 64, 2300), (1536, 96, 100, 2177), (192, 12, 40, 2100), (144, 9, 33, 2050), (256, 32, 64, 2400),
                                         (384, 96, 48, 2060)])
def test_long_rows_take_the_tail_kernel(eng, oracle, metric, d, m, nlist, n):
    """d > 128: the K-tiled coarse quantiser (mfma_assign.hip), then ONE kernel per 128-column block for residual + PQ encode
    (xf_tail_kernel): sub-dimension 16 (C3's shape: 1536 / 96), a last block of fewer than 128 columns (192, 144), sub-dimension 8 and 4"""
    x = spoil(clustered(n, d, 17 + d + m, scale=6.0, noise=2.5, integer=False).astype(f32))
    if metric == "cosine":
        x[100] = 1.0
    rng = np.random.default_rng(d + 5 * m)
    xs = oracle.normalize(x) if metric == "cosine" else x
    fin = oracle.is_finite(xs)
    cent = dup_centroids(np.ascontiguousarray(xs[fin][rng.choice(int(fin.sum()), nlist, replace=False)]))
    part, _ = oracle.assign(np.ascontiguousarray(xs[fin]), cent, "l2" if metric == "cosine" else metric)
    res = oracle.residual(np.ascontiguousarray(xs[fin]), cent, np.where(part == oracle.NONE, 0, part)) if metric != "dot" else xs[fin]
    sd = d // m
    cb = np.stack([res[rng.choice(res.shape[0], 256, replace=False)][:, i * sd:(i + 1) * sd] for i in range(m)]).astype(f32)
    run_case(eng, oracle, x, cent, dup_codewords(cb), metric, stage="xform_tail")


def test_long_f16_rows_take_the_tail_kernel(eng, oracle):
    n, d, m, nlist = 2300, 256, 16, 48
    x = spoil((clustered(n, d, 3, scale=1.5, noise=0.6, integer=False)).astype(np.float16))
    rng = np.random.default_rng(9)
    cent = dup_centroids(np.ascontiguousarray(x[200:200 + nlist]))
    part, _ = oracle.assign(np.ascontiguousarray(x[200:]), cent, "l2")
    res = oracle.residual(np.ascontiguousarray(x[200:]), cent, np.where(part == oracle.NONE, 0, part))
    cb = np.stack([res[rng.choice(res.shape[0], 256, replace=False)][:, i * 16:(i + 1) * 16] for i in range(m)]).astype(np.float16)
    run_case(eng, oracle, x, cent, dup_codewords(cb), "l2", stage="xform_tail")


def test_unit_rows_take_the_single_f16_product_in_the_coarse_quantiser(eng, oracle):
    """cosine, d > 128: the normalise kernel leaves the unit rows as a binary16 plane and the K-tiled coarse quantiser multiplies it by
    a binary16 centroid plane -- one v_mfma_f32_32x32x16_f16 instead of three bf16 products, margin E = 2^-9.5 (|x|^2 + max|c|^2)
    (mfma_assign.hip: ma_top3_wide_kernel<.., F16>).  Partition ids, codes and the loss stay the oracle's bit for bit, with duplicated
    centroids (two / three / five of a kind), near-duplicates inside the wider margin, zero rows (no direction: dropped) and a caller's
    centroids that do NOT fit the plane's scale (the bf16 route takes those)."""
    n, d, m, nlist = 3000, 384, 24, 96
    x = spoil(clustered(n, d, 77, scale=6.0, noise=2.5, integer=False).astype(f32))
    x[100] = 1.0
    x[101] = 0.0
    rng = np.random.default_rng(5)
    xs = oracle.normalize(x)
    fin = oracle.is_finite(xs)
    cent = dup_centroids(np.ascontiguousarray(xs[fin][rng.choice(int(fin.sum()), nlist, replace=False)]))
    cent[40] = cent[41] * f32(1.0005)                 # inside the f16 margin, outside the bf16 x 3 one: exact check decides
    cent[50] = cent[51] + f32(1e-4)
    part, _ = oracle.assign(np.ascontiguousarray(xs[fin]), cent, "l2")
    res = oracle.residual(np.ascontiguousarray(xs[fin]), cent, np.where(part == oracle.NONE, 0, part))
    cb = np.stack([res[rng.choice(res.shape[0], 256, replace=False)][:, i * 16:(i + 1) * 16] for i in range(m)]).astype(f32)
    before = eng.timing_query("count:ma_wide_f16")[1]
    run_case(eng, oracle, x, cent, dup_codewords(cb), "cosine", stage="xform_tail")
    assert eng.timing_query("count:ma_wide_f16")[1] == before + 1, "the f16 coarse sweep did not serve the call"
    big = cent * f32(60.0)                            # components beyond 2: outside the plane's scale -> the three-term bf16 sweep
    run_case(eng, oracle, x, big, dup_codewords(cb), "cosine", stage="xform_tail")
    assert eng.timing_query("count:ma_wide_f16")[1] == before + 1


# ---- the codebook training's E-step on the same PQ phase (xf_pqtrain_kernel; pq/builder.rs:89-157 -> kmeans.rs:317-369) ----------------
@pytest.mark.parametrize("d,m,n,integer", [(128, 16, 6037, False), (128, 32, 4100, False), (128, 8, 3000, False), (256, 16, 2500, False),
                                           (1536, 96, 2177, False), (128, 16, 5000, True), (256, 64, 2300, True)])
def test_pq_training_takes_the_pq_phase_kernel(eng, oracle, d, m, n, integer):
    """trained codebook, iteration counts bit-equal to the oracle's; integer-valued residuals bring exact ties (the undecided lists);
    row counts that leave the last 128-row tile ragged; problems that converge at different iterations (the `active` flags)"""
    x = clustered(n, d, 700 + d + m, integer=integer).astype(f32)
    if not integer:
        x = (x * f32(0.37)).astype(f32)
    cent = clustered(24, d, 701 + d, integer=integer).astype(f32)
    part, _ = oracle.assign(x, cent)
    res = oracle.residual(x, cent, part)
    res[5] = 0
    res[n - 3] = res[n - 9]
    before = eng.timing_query("count:xf_pqtrain")[1]
    cb, iters = eng.pq_train(res, m, max_iters=7, seed=31)
    assert eng.timing_query("count:xf_pqtrain")[1] > before, "xf_pqtrain_kernel did not serve the training"
    ocb, oit = oracle.pq_train(res, m, max_iters=7, seed=31)
    assert (_np(iters) == oit.astype(np.uint32)).all(), (_np(iters), oit)
    assert (_np(cb).view(np.uint32) == ocb.view(np.uint32)).all()


def test_pq_training_kernel_long_run_converges_like_the_oracle(eng, oracle):
    """enough iterations for some sub-quantisers to converge before others: converged problems are skipped, not recomputed"""
    n, d, m = 4096, 128, 16
    rng = np.random.default_rng(77)
    res = rng.normal(0, 1, (n, d)).astype(f32)
    res[:, :32] = np.rint(res[:, :32] * 2)          # four sub-quantisers over a tiny alphabet: they converge within a few iterations
    before = eng.timing_query("count:xf_pqtrain")[1]
    cb, iters = eng.pq_train(res, m, max_iters=30, seed=5)
    assert eng.timing_query("count:xf_pqtrain")[1] > before
    ocb, oit = oracle.pq_train(res, m, max_iters=30, seed=5)
    assert (_np(iters) == oit.astype(np.uint32)).all(), (_np(iters), oit)
    assert len(set(oit.tolist())) > 1, "the case is meant to have problems that stop at different iterations"
    assert (_np(cb).view(np.uint32) == ocb.view(np.uint32)).all()


# ---- f32 assign (k-means E-step, lance_hip_assign) through phases 1-3 of the transform kernel (xf_kernel<.., ASSIGN>) ----------------------
@pytest.mark.parametrize("metric", ["l2", "dot"])
@pytest.mark.parametrize("n,d,k", [(5037, 128, 256), (3000, 64, 40), (2177, 16, 32), (2500, 112, 300), (4100, 48, 1000)])
def test_f32_assign_takes_the_transform_kernels_first_half(eng, oracle, metric, n, d, k):
    """ids and distances bit-equal to the oracle's argmin (kmeans.rs:317-369, argmin_value_float), with and without the k-means balance
    bias; non-finite and zero rows; two, three and five centroids of a kind (the last forces the recompute list)"""
    x = spoil(clustered(n, d, 900 + d + k).astype(f32))
    cent = clustered(k, d, 901 + d + k).astype(f32)
    if k >= 31:
        cent = dup_centroids(cent)
    rng = np.random.default_rng(d + k)
    bias = (rng.random(k) * (2000.0 if metric == "l2" else 50.0)).astype(f32)
    for b in (None, bias):
        before = eng.timing_query("count:xf_assign")[1]
        ids, dists = eng.assign(x, cent, metric, bias=b)
        assert eng.timing_query("count:xf_assign")[1] == before + 1, "the transform kernel's first half did not serve this assign call"
        oi, od = oracle.assign(x, cent, metric, bias=b)
        ids = _np(ids).view(np.uint32)
        bad = np.nonzero(ids != oi)[0]
        assert bad.size == 0, (metric, b is not None, bad[:10], ids[bad[:10]], oi[bad[:10]])
        ok = oi != oracle.NONE
        assert (_np(dists)[ok].view(np.uint32) == od[ok].view(np.uint32)).all()


@pytest.mark.parametrize("metric", ["l2", "dot"])
def test_kmeans_training_with_a_balance_factor_runs_on_it(eng, oracle, metric):
    n, d, k = 6000, 128, 64
    x = clustered(n, d, 33, integer=False).astype(f32)
    before = eng.timing_query("count:xf_assign")[1]
    cent, loss, iters = eng.kmeans_train(x, k, max_iters=25, balance_factor=1.0, seed=7, metric=metric)
    assert eng.timing_query("count:xf_assign")[1] >= before + 2
    oc, ol, oit, _ = oracle.kmeans_train(x, k, max_iters=25, balance_factor=f32(1.0) / f32(n), seed=7, metric=metric)
    assert iters == oit and loss == ol
    assert (_np(cent).view(np.uint32) == oc.view(np.uint32)).all()
