"""Limits the reference does not have: k * refine_factor beyond 2048 on IVF_PQ, k beyond 128 on IVF_FLAT, more than
256 probes when the index has more than 8192 partitions.  Past the fast kernels' selection width every query goes through the
heap-emulating exact kernels, whose answers must still equal the oracle's bit for bit."""
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


def _sift_like(n, d, seed, ncl=32):
    rng = np.random.default_rng(seed)
    centers = rng.uniform(0, 128, (ncl, d))
    x = centers[rng.integers(0, ncl, n)] + rng.normal(0, 24, (n, d))
    return np.clip(np.rint(x), 0, 218).astype(f32)


def test_ivfpq_large_k_times_refine(eng, oracle):
    from lance_amd.engine import DeviceIndex
    n, d, nlist, m = 24_000, 64, 16, 8
    x = _sift_like(n, d, 3)
    q = _sift_like(12, d, 4)
    cent, _, _, _ = oracle.kmeans_train(x[:4096], nlist, max_iters=6, seed=1)
    part, _ = oracle.assign(x, cent)
    cb, _ = oracle.pq_train(oracle.residual(x, cent, part)[:8192], m, max_iters=6, seed=2)
    oidx = oracle.build_index(x, cent, cb)
    gpart, gcodes, _ = eng.ivfpq_encode(x, cent, cb, "l2")
    g = DeviceIndex.create(eng, "l2", cent, cb, gpart, gcodes, None, raw=x)
    for k, nprobes, rf in ((300, nlist, 10), (1000, 4, 0), (2500, nlist, 2), (5000, nlist, 0)):
        gi, gd = g.search(q, k, nprobes, rf)
        oi, od = oidx.search(q, k, nprobes, refine=rf, raw=x if rf else None)
        assert (_np(gi).view(np.uint64) == oi).all(), (k, nprobes, rf)
        assert (_np(gd).view(np.uint32) == od.view(np.uint32)).all(), (k, nprobes, rf)
    g.close()


@pytest.mark.parametrize("metric", ["l2", "dot"])
def test_ivf_flat_large_k(eng, oracle, metric):
    from lance_amd.engine import DeviceFlatIndex
    n, d, nlist = 16_000, 64, 12
    x = _sift_like(n, d, 7)
    x[50:60] = x[7]
    q = _sift_like(10, d, 8)
    cent, _, _, _ = oracle.kmeans_train(x[:4096], nlist, max_iters=6, seed=2)
    part, _ = eng.assign(x, cent, metric)
    g = DeviceFlatIndex.create(eng, metric, cent, x, part)
    for k, nprobes in ((129, 3), (500, nlist), (3000, nlist)):
        gi, gd = g.search(q, k, nprobes)
        oi, od = oracle.ivfflat_search(x, cent, q, k, nprobes, metric)
        assert (_np(gi).view(np.uint64) == oi).all(), (metric, k, nprobes)
        assert (_np(gd).view(np.uint32) == od.view(np.uint32)).all()


def test_many_probes_on_many_partitions(eng, oracle):
    rng = np.random.default_rng(9)
    d, nlist = 32, 10_000
    cent = np.rint(rng.uniform(0, 100, (nlist, d))).astype(f32)
    q = np.rint(rng.uniform(0, 100, (20, d))).astype(f32)
    for nprobes in (257, 1000, 2048):
        gi, gd = eng.find_partitions(q, cent, nprobes)
        oi, od = oracle.find_partitions(q, cent, nprobes)
        assert (_np(gi).view(np.uint32) == oi).all(), nprobes
        assert (_np(gd).view(np.uint32) == od.view(np.uint32)).all(), nprobes
