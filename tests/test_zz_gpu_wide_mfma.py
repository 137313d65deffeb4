"""GPU parity of the coarse quantiser on rows of more than 128 elements (C3: 1536-dimensional embeddings): the K-tiled bf16x3
MFMA surrogate (ma_top3_wide_kernel) + exact re-check must return the reference's argmin -- ids and distances bit for bit
(kmeans.rs:317-369, kernels.rs:79-111) -- through assign, k-means training (bias, convergence flags) and the encode chain
(cosine normalisation, KeepFiniteVectors).  Sorted last: newest device code last."""
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


def mixture(n, d, k, seed, scale=1.0):
    rng = np.random.default_rng(seed)
    c = (rng.standard_normal((k, d)) * scale).astype(f32)
    x = (c[rng.integers(0, k, n)] + rng.standard_normal((n, d)).astype(f32) * 0.6 * scale).astype(f32)
    return x, c


@pytest.mark.parametrize("d,metric", [(256, "l2"), (1536, "l2"), (200, "l2"), (144, "dot"), (1536, "dot")])
def test_wide_rows_assign_on_matrix_cores(eng, oracle, d, metric):
    n, k = 3000, 100                     # k is not a multiple of the 128-centroid super-tile: masked tail
    x, c = mixture(n, d, k, 7 + d)
    c[7] = c[3]                          # duplicate centroids: an exact tie, the smaller index wins
    x[11] = c[50]                        # a row that IS a centroid
    for rows, cents in ((x, c), (x[:2100], c[:64]), (x, np.concatenate([c, c * 1.001, c * 0.999]))):
        ids, dists = eng.assign(rows, cents, metric)
        oi, od = oracle.assign(rows, cents, metric)
        assert (_np(ids).view(np.uint32) == oi).all(), (d, metric, cents.shape)
        assert (_np(dists).view(np.uint32) == od.view(np.uint32)).all(), (d, metric, cents.shape)
    bias = np.random.default_rng(1).random(k).astype(f32) * (0.5 if metric == "dot" else 20.0)
    ids, dists = eng.assign(x, c, metric, bias=bias)
    oi, od = oracle.assign(x, c, metric, bias=bias)
    assert (_np(ids).view(np.uint32) == oi).all() and (_np(dists).view(np.uint32) == od.view(np.uint32)).all()
    # integer-valued rows (SIFT-like magnitudes): many near and exact ties between candidates
    xi = np.rint(x * 20).astype(f32); ci = np.rint(c * 20).astype(f32)
    ids, dists = eng.assign(xi, ci, metric)
    oi, od = oracle.assign(xi, ci, metric)
    assert (_np(ids).view(np.uint32) == oi).all() and (_np(dists).view(np.uint32) == od.view(np.uint32)).all()


def test_wide_rows_kmeans_and_encode_chain(eng, oracle):
    d, k = 256, 64
    x, _ = mixture(4096, d, k, 3)
    cent, loss, iters = eng.kmeans_train(x, k, max_iters=10, balance_factor=1.0, seed=4)
    oc, ol, oit, _ = oracle.kmeans_train(x, k, max_iters=10, balance_factor=f32(1.0) / f32(4096), seed=4)
    assert iters == oit and loss == ol
    assert (_np(cent).view(np.uint32) == oc.view(np.uint32)).all()
    # encode chain under cosine: normalise -> keep finite -> assign (MFMA, wide) -> residual -> PQ
    xs = x.copy()
    xs[17, 5] = np.inf                   # KeepFiniteVectors: no partition
    part, _ = oracle.assign(oracle.normalize(x), oc)
    res = oracle.residual(oracle.normalize(x), oc, part)
    cb, _ = oracle.pq_train(res, 16, max_iters=4, seed=2)
    oidx = oracle.build_index(xs, oc, cb, "cosine")
    gpart, gcodes, _ = eng.ivfpq_encode(xs, oc, cb, "cosine")
    keep = np.ones(len(xs), bool); keep[17] = False
    assert _np(gpart).view(np.uint32)[17] == oracle.NONE
    assert (_np(gpart).view(np.uint32)[keep] == oidx.part_ids).all()
    assert (_np(gcodes)[keep] == oidx.codes_rowmajor).all()
