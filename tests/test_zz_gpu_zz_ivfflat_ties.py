"""IVF_FLAT when more rows tie at a query's bound than its candidate pool holds (2048): thousands of duplicate vectors, or
distances that are all NaN (a Float16 cosine index whose row norms overflow half precision: normalize_fsl::<Float16Type>
turns such rows into zeros, kernels.rs:141-186, and cosine against a zero row is 0 / 0).  The threshold cannot separate the
ties, so those queries are replayed by the heap-emulating exact kernel (FlatIndex::search's BinaryHeap, flat/index.rs:94-126);
found by tests/fuzz_parity.py.  Sorted last: newest device code last."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
f32 = np.float32


def test_ivf_flat_more_ties_than_the_pool_holds(oracle):
    import lance_amd
    rng = np.random.default_rng(5)
    n, d = 9000, 32
    x = rng.integers(0, 30, (n, d)).astype(f32)
    x[1000:5000] = x[7]                       # 4000 identical rows: one distance value, far more than a pool of 2048
    q = rng.integers(0, 30, (12, d)).astype(f32)
    q[:4] = x[7]                              # distance 0 to all of them
    q[4:8] = x[7] + 1.0                       # and a common non-zero distance
    fx = lance_amd.create_index(x, "IVF_FLAT", metric="l2", num_partitions=6, max_iters=5, sample_rate=64)
    for k, nprobes in ((10, 6), (100, 3), (5, 1)):
        gi, gd = fx.nearest(q, k, nprobes)
        oi, od = oracle.ivfflat_search(x, fx.centroids, q, k, nprobes, "l2")
        assert (gi == oi).all(), (k, nprobes)
        assert (gd.view(np.uint32) == od.view(np.uint32)).all(), (k, nprobes)
    allow = np.ones(n, bool); allow[1000:1200] = False            # the same under a prefilter
    gi, gd = fx.nearest(q, 10, 6, prefilter=allow)
    keep = np.nonzero(allow)[0]
    oi, od = oracle.ivfflat_search(x[keep], fx.centroids, q, 10, 6, "l2", row_ids=keep.astype(np.uint64))
    assert (gi == oi).all() and (gd.view(np.uint32) == od.view(np.uint32)).all()


def test_ivf_flat_f16_cosine_norm_overflow_gives_nan_like_the_reference(oracle):
    import lance_amd
    rng = np.random.default_rng(6)
    n, d = 5000, 128
    x = rng.integers(1, 31, (n, d)).astype(np.float16)            # |x|^2 of ~ 128 x 320 (more in the doubled half): beyond f16's 65504
    x[:, :64] *= 2
    q = rng.integers(1, 31, (6, d)).astype(np.float16)
    assert not np.isfinite((x.astype(f32) ** 2).sum(1).astype(np.float16)).all()
    fx = lance_amd.create_index(x, "IVF_FLAT", metric="cosine", num_partitions=4, max_iters=4, sample_rate=64)
    gi, gd = fx.nearest(q, 10, 4)                                  # used to fail with "candidate pool did not converge"
    oi, od = oracle.ivfflat_search(x, fx.centroids, q, 10, 4, "cosine")
    # every distance is NaN on both sides; which ten rows come back depends on the NaN's sign under f32::total_cmp (0 / 0 is
    # -NaN on x86, +NaN on aarch64: the reference itself is platform-dependent here), so only the shape of the answer is pinned
    assert np.isnan(od).all() and np.isnan(gd).all()
    assert all(len(set(r.tolist())) == 10 and (r < n).all() for r in gi)
