"""The bound pass on the matrix cores (lance_amd/csrc/search_ms.hip: ms_bound_kernel).

Every query needs an upper bound T of its final k*refine-th ADC distance before the filter scan (search_q.hip header).  For batches the
matrix-core scan serves, T now comes from the same matrix product: a 512-bin histogram of dist~ = |c^|^2 - 2 r.c^ + |r|^2 over the
query's nearest partition (f16 residuals and the f16 codebook resident in LDS), T = upper bin edge + the filter's error bound E.
T only has to be AN upper bound -- the survivors are still decided by the reference-order LUT arithmetic (pq/distance.rs:109-144) --
so ids and distances must stay bit-equal to the oracle whichever pass produced it.  Every case asserts which pass ran (`ivfpq_msbound`).
The integer pass keeps its coverage through the batches below the matrix-core threshold (tests/test_gpu_pm_scan.py) and a child
process with LANCE_HIP_NO_MSBOUND=1.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from test_gpu_pm_scan import _models, _np, clustered

pytestmark = pytest.mark.gpu
f32 = np.float32


@pytest.fixture(scope="module")
def eng(engine):
    from lance_amd.engine import Engine
    e = Engine()
    yield e
    e.close()


class _mb_used:
    def __init__(self, eng, expect=True):
        self.eng, self.expect = eng, expect

    def __enter__(self):
        self.eng.timing(True)
        self.before = self.eng.timing_query("ivfpq_msbound")[1]
        return self

    def __exit__(self, *a):
        self.eng.synchronize()
        after = self.eng.timing_query("ivfpq_msbound")[1]
        self.eng.timing(False)
        if a[0] is None:
            assert (after > self.before) == self.expect, "matrix-core bound pass " + ("not taken" if self.expect else "taken unexpectedly")


def _equal(gi, gd, oi, od, what):
    bad = np.nonzero((_np(gi).view(np.uint64) != oi).any(axis=1))[0]
    assert bad.size == 0, f"{what}: ids differ for {bad.size} queries (first {bad[:5]})"
    assert (_np(gd).view(np.uint32) == od.view(np.uint32)).all(), f"{what}: distance bits differ"


@pytest.mark.parametrize("metric", ["l2", "cosine"])
@pytest.mark.parametrize("d,m", [(64, 16), (128, 16), (128, 32)])
def test_msbound_every_instantiation(eng, oracle, d, m, metric):
    from lance_amd.engine import DeviceIndex
    n, nlist, nq = 20000, 24, 700
    x = clustered(n, d, 310 + d + m) + (1.0 if metric == "cosine" else 0.0)
    q = clustered(nq, d, 410 + d + m) + (1.0 if metric == "cosine" else 0.0)
    cent, cb = _models(oracle, x, nlist, m, metric, seed=d + m + 2)
    oidx = oracle.build_index(x, cent, cb, metric)
    gpart, gcodes, _ = eng.ivfpq_encode(x, cent, cb, metric)
    gidx = DeviceIndex.create(eng, metric, cent, cb, gpart, gcodes, None, raw=x)
    # ~29 queries per nearest partition on average, far more for the popular ones: blocks of one and of two query tiles, several blocks
    # per partition; k * refine from 1 (the first non-empty bin) to 128, the largest the batched kernels take
    for k, nprobes, rf in [(10, 8, 0), (10, 8, 10), (10, nlist, 0), (1, 7, 1), (64, 7, 2), (37, 9, 3)]:
        with _mb_used(eng):
            gi, gd = gidx.search(q, k, nprobes, rf)
        oi, od = oidx.search(q, k, nprobes, refine=rf, raw=x if rf else None)
        _equal(gi, gd, oi, od, f"{metric} d={d} m={m} k={k} nprobes={nprobes} refine={rf}")
    gidx.close()


def test_msbound_skewed_partitions_far_queries_and_prefilter(eng, oracle):
    """Non-integer rows of small magnitude, partitions of 1 .. a few thousand rows (fewer rows than k*refine: no bound), 200 queries
    packed onto ONE partition (four blocks of 64), queries far from every centroid (residuals that overflow binary16 after scaling:
    no bound from this pass), and a prefilter that leaves some nearest partitions with fewer allowed rows than k*refine."""
    from lance_amd.engine import DeviceIndex
    rng = np.random.default_rng(5)
    d, m, nlist = 128, 16, 40
    sizes = np.concatenate([[1, 2, 5, 31, 33, 64, 100], rng.integers(200, 3000, nlist - 7)])
    centers = rng.normal(0, 1.0, (nlist, d))
    x = np.concatenate([centers[i] + rng.normal(0, 0.05, (int(s), d)) for i, s in enumerate(sizes)]).astype(f32)
    n = x.shape[0]
    q = np.concatenate([
        centers[rng.integers(0, nlist, 1500)] + rng.normal(0, 0.05, (1500, d)),
        centers[20] + rng.normal(0, 0.02, (200, d)),              # one partition's nearest-query list: 200+ queries
        rng.normal(0, 1.0, (60, d)) * 3e4,                        # |r| sigma beyond binary16
        centers[rng.integers(0, 7, 40)] + rng.normal(0, 0.05, (40, d)),   # nearest partition too small for a bound
    ]).astype(f32)
    cent, cb = _models(oracle, x, nlist, m, "l2", seed=13)
    oidx = oracle.build_index(x, cent, cb, "l2")
    gpart, gcodes, _ = eng.ivfpq_encode(x, cent, cb, "l2")
    gidx = DeviceIndex.create(eng, "l2", cent, cb, gpart, gcodes, None, raw=x)
    for k, nprobes, rf in [(10, 6, 0), (10, 6, 5), (50, 5, 2)]:
        with _mb_used(eng):
            gi, gd = gidx.search(q, k, nprobes, rf)
        oi, od = oidx.search(q, k, nprobes, refine=rf, raw=x if rf else None)
        _equal(gi, gd, oi, od, f"k={k} nprobes={nprobes} refine={rf}")
    allow = rng.random(n) < 0.03
    with _mb_used(eng):
        gi, gd = gidx.search_filtered(q, 10, 6, allow)
    oi, od = oidx.search(q, 10, 6, prefilter=allow)
    _equal(gi, gd, oi, od, "prefilter")
    gidx.close()


def test_msbound_f16_column(eng, oracle):
    """Float16 rows: the residual is rounded to binary16 before anything else (`round_f16`), as the exact path does."""
    from lance_amd.engine import DeviceIndex
    rng = np.random.default_rng(23)
    n, d, m, nlist, nq = 30000, 128, 16, 32, 3200      # 3200 x 10 pairs >= 96 x 32
    c = rng.standard_normal((64, d)) * 2
    x = (c[rng.integers(0, 64, n)] + rng.standard_normal((n, d)) * 0.7).astype(np.float16)
    q = (c[rng.integers(0, 64, nq)] + rng.standard_normal((nq, d)) * 0.7).astype(np.float16)
    cent = x[rng.choice(n, nlist, replace=False)].copy()
    part, _ = oracle.assign(x, cent)
    res = oracle.residual(x, cent, part)
    cb, _ = oracle.pq_train(res[:4096], m, max_iters=3, seed=2)
    oidx = oracle.build_index(x, cent, cb)
    gpart, gcodes, _ = eng.ivfpq_encode(x, cent, cb)
    gidx = DeviceIndex.create(eng, "l2", cent, cb, gpart, gcodes, None, raw=x)
    for k, nprobes, rf in [(10, 10, 0), (10, 10, 10)]:
        with _mb_used(eng):
            gi, gd = gidx.search(q, k, nprobes, rf)
        oi, od = oidx.search(q, k, nprobes, refine=rf, raw=x.astype(f32) if rf else None)
        _equal(gi, gd, oi, od, f"f16 k={k} refine={rf}")
    gidx.close()


def test_small_batches_keep_the_integer_bound(eng, oracle):
    from lance_amd.engine import DeviceIndex
    n, d, m, nlist, nq = 20000, 128, 16, 24, 100      # 100 x 8 pairs < 96 x 24: below the matrix-core threshold
    x = clustered(n, d, 71)
    q = clustered(nq, d, 72)
    cent, cb = _models(oracle, x, nlist, m, "l2", seed=4)
    oidx = oracle.build_index(x, cent, cb, "l2")
    gpart, gcodes, _ = eng.ivfpq_encode(x, cent, cb, "l2")
    gidx = DeviceIndex.create(eng, "l2", cent, cb, gpart, gcodes, None, raw=x)
    with _mb_used(eng, False):
        gi, gd = gidx.search(q, 10, 8, 10)
    oi, od = oidx.search(q, 10, 8, refine=10, raw=x)
    _equal(gi, gd, oi, od, "small batch")
    gidx.close()


def test_integer_bound_keeps_its_coverage_under_the_matrix_core_scan():
    """LANCE_HIP_NO_MSBOUND=1 (read once per process): the matrix-core scan fed by the integer bound pass, as in round 4."""
    env = dict(os.environ, LANCE_HIP_NO_MSBOUND="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.join(os.path.dirname(__file__), "test_zz_gpu_mscan.py"),
                        "-k", "every_instantiation or many_ties"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
