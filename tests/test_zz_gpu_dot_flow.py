"""The quantised flow of the DOT metric (round 6): dist = 1 - q . c^ with c^ the stored row's reconstruction (no residual:
pq/distance.rs:60-92 `build_distance_table_dot`, pq/storage.rs:949-957 the -(M - 1) offset).  The integer tables of search_q.hip need
entries >= 0, so a dot batch used to take the exact pair scan (3.47 ms per 10,000-query batch at the C2 shape against 0.66 for L2,
gpurun r06zq).  Now the matrix-core bound pass and scan (search_ms.hip) serve it: operand q / 2 against the same f16 codebook plane
(-2 sigma c), row term zero, limits and integer sums relative to a per-query base 1 - |q| cmax (Cauchy-Schwarz), so T may have either
sign; class-B queries go to the exact pair kernel's dot instantiation, survivors are re-evaluated by the merge / rescan kernels' dot
instantiations in the reference's arithmetic -- ids and distances must stay bit-equal to the oracle.

Every case is sized for the matrix-core scan (nq * nprobes >= 96 * nlist) and ASSERTS that it and the matrix-core bound pass ran.
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


class _dot_flow_used:
    def __init__(self, eng, expect=True):
        self.eng, self.expect = eng, expect

    def __enter__(self):
        self.eng.timing(True)
        self.before = (self.eng.timing_query("ivfpq_mscan")[1], self.eng.timing_query("ivfpq_msbound")[1])
        return self

    def __exit__(self, *a):
        self.eng.synchronize()
        after = (self.eng.timing_query("ivfpq_mscan")[1], self.eng.timing_query("ivfpq_msbound")[1])
        self.eng.timing(False)
        if a[0] is None:
            took = after[0] > self.before[0] and after[1] > self.before[1]
            assert took == self.expect, "dot batch: matrix-core bound pass + scan " + ("not taken" if self.expect else "taken unexpectedly")


def _check(eng, gidx, oidx, q, raw, cases, allow=None, n=None):
    for k, nprobes, rf in cases:
        with _dot_flow_used(eng):
            if allow is None:
                gi, gd = gidx.search(q, k, nprobes, rf)
            else:
                gi, gd = gidx.search_filtered(q, k, nprobes, allow, rf)
        oi, od = oidx.search(q, k, nprobes, refine=rf, raw=raw if rf else None, **({} if allow is None else {"prefilter": allow[:n]}))
        bad = np.nonzero((_np(gi).view(np.uint64) != oi).any(axis=1))[0]
        assert bad.size == 0, f"ids differ for {bad.size} queries (first {bad[:5]}) at k={k} nprobes={nprobes} refine={rf}"
        assert (_np(gd).view(np.uint32) == od.view(np.uint32)).all(), (k, nprobes, rf)


@pytest.mark.parametrize("d,m", [(64, 16), (128, 16), (128, 32)])
def test_dot_flow_every_instantiation(eng, oracle, d, m):
    """SIFT-like rows (all components >= 0: every distance is a large negative number, T < 0 for every query)."""
    from lance_amd.engine import DeviceIndex
    n, nlist, nq = 20000, 24, 700
    x = clustered(n, d, 500 + d + m)
    q = clustered(nq, d, 600 + d + m)
    cent, cb = _models(oracle, x, nlist, m, "dot", seed=d + m + 2)
    oidx = oracle.build_index(x, cent, cb, "dot")
    gpart, gcodes, _ = eng.ivfpq_encode(x, cent, cb, "dot")
    gidx = DeviceIndex.create(eng, "dot", cent, cb, gpart, gcodes, None, raw=x)
    _check(eng, gidx, oidx, q, x, [(10, 8, 0), (10, 8, 10), (10, nlist, 0), (100, 7, 0), (1, 7, 1), (37, 9, 3), (128, 6, 0)])
    gidx.close()


def test_dot_flow_signed_data_tiny_partitions_far_and_zero_queries(eng, oracle):
    """Zero-mean Gaussian rows of small magnitude: dot products of both signs, distances around 1 (T > 0), sigma far from 1,
    partitions of 1 .. 300 rows, queries scaled by 6 and all-zero queries (every row ties at distance 1: segments overflow)."""
    from lance_amd.engine import DeviceIndex
    rng = np.random.default_rng(78)
    d, m, nlist, nq = 128, 16, 40, 900
    sizes = np.concatenate([[1, 2, 3, 255, 256, 257, 300], rng.integers(20, 900, nlist - 7)])
    centers = rng.standard_normal((nlist, d)) * 0.05
    x = np.concatenate([centers[i] + rng.standard_normal((s, d)) * 0.02 for i, s in enumerate(sizes)]).astype(f32)
    q = (centers[rng.integers(0, nlist, nq)] + rng.standard_normal((nq, d)) * 0.02).astype(f32)
    q[::7] *= 6.0
    q[3::50] = 0.0
    q[5::60] *= -1.0
    cent = centers.astype(f32)
    part, _ = oracle.assign(x, cent, "dot")
    cb, _ = oracle.pq_train(x[rng.choice(len(x), 4096, replace=False)], m, max_iters=3, seed=5)
    oidx = oracle.build_index(x, cent, cb, "dot")
    gpart, gcodes, _ = eng.ivfpq_encode(x, cent, cb, "dot")
    assert (_np(gpart).view(np.uint32) == oidx.part_ids).all() and (_np(gcodes) == oidx.codes_rowmajor).all()
    gidx = DeviceIndex.create(eng, "dot", cent, cb, gpart, gcodes, None, raw=x)
    _check(eng, gidx, oidx, q, x, [(10, 10, 0), (10, 10, 10), (5, nlist, 0), (60, 12, 2)])
    gidx.close()


def test_dot_flow_large_magnitudes_and_many_ties(eng, oracle):
    """Rows duplicated many times (hundreds tie at the bound: overflowed segments -> exact rescan, (dist, rowid) order of the ties) and
    components up to a few thousand (products ~1e7: the -(M - 1) offset disappears in the rounding, as in the reference)."""
    from lance_amd.engine import DeviceIndex
    rng = np.random.default_rng(6)
    d, m, nlist, nq = 128, 16, 16, 600
    base = clustered(300, d, 9) * 16.0 - 900.0
    x = base[rng.integers(0, 300, 24000)]
    q = base[rng.integers(0, 300, nq)] + rng.integers(0, 2, (nq, d)).astype(f32)
    cent, cb = _models(oracle, x, nlist, m, "dot", seed=4)
    oidx = oracle.build_index(x, cent, cb, "dot")
    gpart, gcodes, _ = eng.ivfpq_encode(x, cent, cb, "dot")
    gidx = DeviceIndex.create(eng, "dot", cent, cb, gpart, gcodes, None, raw=x)
    _check(eng, gidx, oidx, q, x, [(10, 8, 0), (40, nlist, 0), (10, 8, 4)])
    gidx.close()


def test_dot_flow_prefilter_and_f16_column(eng, oracle):
    from lance_amd.engine import DeviceIndex
    rng = np.random.default_rng(33)
    n, d, m, nlist, nq = 16000, 128, 16, 24, 640
    x = clustered(n, d, 301, integer=False) - 40.0
    q = clustered(nq, d, 302, integer=False) - 40.0
    cent, cb = _models(oracle, x, nlist, m, "dot", seed=9)
    oidx = oracle.build_index(x, cent, cb, "dot")
    gpart, gcodes, _ = eng.ivfpq_encode(x, cent, cb, "dot")
    g = DeviceIndex.create(eng, "dot", cent, cb, gpart, gcodes, None, raw=x)
    for frac in (0.02, 0.5, 0.97):
        allow = rng.random(n + 100) < frac
        _check(eng, g, oidx, q, x, [(10, 8, 0), (10, 8, 10), (40, nlist, 0)], allow=allow, n=n)
    g.close()
    # f16 column: the query is NOT rounded for dot (no residual), the rows' codes come from the f16 rows
    xh = (x * 0.05).astype(np.float16)
    qh = (q * 0.05).astype(np.float16)
    cent2 = xh[rng.choice(n, nlist, replace=False)].copy()
    cb2, _ = oracle.pq_train(xh[:4096], m, max_iters=3, seed=2)      # f16 in -> f16 codebook
    oidx2 = oracle.build_index(xh, cent2, cb2, "dot")
    gpart2, gcodes2, _ = eng.ivfpq_encode(xh, cent2, cb2, "dot")
    g2 = DeviceIndex.create(eng, "dot", cent2, cb2, gpart2, gcodes2, None, raw=xh)
    _check(eng, g2, oidx2, qh, xh.astype(f32), [(10, 10, 0), (10, 10, 10), (50, 8, 2)])
    g2.close()


def test_dot_flow_not_taken_for_small_batches(eng, oracle):
    from lance_amd.engine import DeviceIndex
    n, d, m, nlist = 12000, 128, 16, 512
    x = clustered(n, d, 31)
    q = clustered(600, d, 32)
    cent = x[np.random.default_rng(1).choice(n, nlist, replace=False)].copy()
    cb, _ = oracle.pq_train(x[:3072], m, max_iters=2, seed=3)
    oidx = oracle.build_index(x, cent, cb, "dot")
    gpart, gcodes, _ = eng.ivfpq_encode(x, cent, cb, "dot")
    gidx = DeviceIndex.create(eng, "dot", cent, cb, gpart, gcodes, None, raw=x)
    with _dot_flow_used(eng, expect=False):            # 600 x 8 = 4800 pairs < 96 x 512: the exact pair scan serves it
        gi, gd = gidx.search(q, 10, 8, 0)
    oi, od = oidx.search(q, 10, 8)
    assert (_np(gi).view(np.uint64) == oi).all() and (_np(gd).view(np.uint32) == od.view(np.uint32)).all()
    gidx.close()


def test_exact_pair_scan_keeps_its_dot_coverage():
    """LANCE_HIP_NO_DOT_FLOW=1: dot batches back on the exact pair scan (the route of every dot batch the matrix-core passes do not take)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LANCE_HIP_NO_DOT_FLOW="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_pm_scan.py"), "-m", "gpu", "-q", "-x",
                        "-k", "dot", "-p", "no:cacheprovider"], cwd=root, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_skew_child_uneven_lists_keep_the_exact_pair_scan(eng, oracle):
    """Runs in the child below with LANCE_HIP_DOT_FLOW_SKEW=8 (an A/B guard, off by default): an index whose largest list holds more than 8 x the
    mean keeps the exact pair scan for dot batches (search_ms.hip: mscan_dot_ready), an even one takes the flow; both equal the oracle."""
    if os.environ.get("LANCE_TEST_SKEW_CHILD") != "1":
        pytest.skip("child of test_skew_guard_switch")
    from lance_amd.engine import DeviceIndex
    rng = np.random.default_rng(12)
    d, m, nlist, nq = 128, 16, 16, 600
    for uneven in (True, False):
        if uneven:                                      # rows with components >= 0 and one long centroid: every row's largest dot product
            x = clustered(16000, d, 91)
            q = clustered(nq, d, 92)
            cent = x[rng.choice(len(x), nlist, replace=False)].copy()
            cent[0] *= 3.0
            cb, _ = oracle.pq_train(x[:4096], m, max_iters=3, seed=1)
        else:                                           # centred rows, dot k-means: even lists
            x = clustered(16000, d, 91) - f32(64.0)
            q = clustered(nq, d, 92) - f32(64.0)
            cent, cb = _models(oracle, x, nlist, m, "dot", seed=3)
        oidx = oracle.build_index(x, cent, cb, "dot")
        sizes = np.diff(oidx.part_offsets)
        assert (sizes.max() * nlist > 8 * len(x)) == uneven, sizes
        gpart, gcodes, _ = eng.ivfpq_encode(x, cent, cb, "dot")
        gidx = DeviceIndex.create(eng, "dot", cent, cb, gpart, gcodes, None, raw=x)
        with _dot_flow_used(eng, expect=not uneven):
            gi, gd = gidx.search(q, 10, 8, 4)
        oi, od = oidx.search(q, 10, 8, refine=4, raw=x)
        assert (_np(gi).view(np.uint64) == oi).all() and (_np(gd).view(np.uint32) == od.view(np.uint32)).all()
        gidx.close()


def test_skew_guard_switch():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LANCE_HIP_DOT_FLOW_SKEW="8", LANCE_TEST_SKEW_CHILD="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-k", "skew_child", "-p", "no:cacheprovider"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "1 passed" in r.stdout


@pytest.mark.parametrize("dtype", ["float32", "float16"])
def test_dot_tiny_magnitudes_tie_at_one(eng, oracle, dtype):
    """Rows and centroids of magnitude 1e-3: every dot product is ~1e-5, the reference's distances 1 - x.c are a handful of f32 values just
    under / over 1 and centroids (rows) whose products differ by less than an ulp of 1 TIE there -- the first index (the smaller row id) wins.
    The matrix-core surrogates rank by the product itself; their margins carry 2^-22 of absolute slack under dot so that such rows reach the
    exact evaluation (found by tests/fuzz_dot_flow.py: 9 of 26,482 rows went to the other of two tied lists).  assign, find_partitions, the
    batched flat filters (d = 128: flat_mfma.hip; d = 256: flat_mfma_wide.hip) against the oracle."""
    rng = np.random.default_rng(25)
    for d, nq in ((128, 700), (256, 700)):
        x = (rng.standard_normal((30000, d)) * 1.1e-3).astype(dtype)
        q = (rng.standard_normal((nq, d)) * 1.1e-3).astype(dtype)
        cent = x[rng.choice(len(x), 64, replace=False)].copy()
        gi, gd = eng.assign(x, cent, "dot")
        oi, od = oracle.assign(x, cent, "dot")
        assert (_np(gi).view(np.uint32) == oi).all(), f"assign: {int((_np(gi).view(np.uint32) != oi).sum())} rows differ (d={d} {dtype})"
        assert (_np(gd).view(np.uint32) == od.view(np.uint32)).all()
        gp, gpd = eng.find_partitions(q, cent, 10, "dot")
        op, opd = oracle.find_partitions(q, cent, 10, "dot")
        assert (_np(gpd).view(np.uint32) == opd.view(np.uint32)).all(), f"find_partitions distances (d={d} {dtype})"
        # equal distances may come in any order of partitions (the reference's partial sort is unstable): compare the SETS per distance value
        same = np.sort(_np(gp).astype(np.int64) + (_np(gpd).view(np.uint32).astype(np.int64) << 20), axis=1) == \
               np.sort(op.astype(np.int64) + (opd.view(np.uint32).astype(np.int64) << 20), axis=1)
        assert same.all(), f"find_partitions ids (d={d} {dtype})"
        fi, fd = eng.flat_topk(x, q, 10, "dot")
        ofi, ofd = oracle.flat_knn(x, q, 10, "dot")
        assert (_np(fi).view(np.uint64) == ofi).all(), f"flat ids: {int((_np(fi).view(np.uint64) != ofi).any(axis=1).sum())} queries differ (d={d} {dtype})"
        assert (_np(fd).view(np.uint32) == ofd.view(np.uint32)).all()
