"""The ADC filter scan on the matrix cores (lance_amd/csrc/search_ms.hip): dist = |c^|^2 - 2 r.c^ + |r|^2 with c^ the row's
reconstruction, evaluated per partition as a [queries x d] x [d x rows] f16 product (v_mfma_f32_32x32x16_f16) whose accumulator starts
at |c^|^2 - limit (limit = T + E - |r|^2); one compare with zero per (row, query).  It only FILTERS -- the survivors are re-evaluated in the reference's
arithmetic (pq/distance.rs:109-144, sequential-m sum) by the merge kernel -- so ids and distances must stay bit-equal to the oracle.

Taken for 8-bit PQ, L2 / cosine, d = 64 (M 16) / 128 (M 16 / 32) once `nq * nprobes >= 96 * nlist` (a partition sees three tiles of queries on average); every case below is sized for it
and ASSERTS it ran (the `ivfpq_mscan` timer).  The integer scan it replaces for these shapes (search_q.hip) keeps its coverage through
a child process with LANCE_HIP_NO_MSCAN=1.
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


class _ms_used:
    def __init__(self, eng, expect=True):
        self.eng, self.expect = eng, expect

    def __enter__(self):
        self.eng.timing(True)
        self.before = self.eng.timing_query("ivfpq_mscan")[1]
        return self

    def __exit__(self, *a):
        self.eng.synchronize()
        after = self.eng.timing_query("ivfpq_mscan")[1]
        self.eng.timing(False)
        if a[0] is None:
            assert (after > self.before) == self.expect, "matrix-core scan " + ("not taken" if self.expect else "taken unexpectedly")


def _check(eng, gidx, oidx, qg, q, raw, cases, allow=None):
    for k, nprobes, rf in cases:
        with _ms_used(eng):
            gi, gd = gidx.search(qg, k, nprobes, rf)
        oi, od = oidx.search(q, k, nprobes, refine=rf, raw=raw if rf else None)
        bad = np.nonzero((_np(gi).view(np.uint64) != oi).any(axis=1))[0]
        assert bad.size == 0, f"ids differ for {bad.size} queries (first {bad[:5]}) at k={k} nprobes={nprobes} refine={rf}"
        assert (_np(gd).view(np.uint32) == od.view(np.uint32)).all(), (k, nprobes, rf)


@pytest.mark.parametrize("metric", ["l2", "cosine"])
@pytest.mark.parametrize("d,m", [(64, 16), (128, 16), (128, 32)])
def test_mscan_every_instantiation(eng, oracle, d, m, metric):
    from lance_amd.engine import DeviceIndex
    n, nlist, nq = 20000, 24, 700
    x = clustered(n, d, 300 + d + m) + (1.0 if metric == "cosine" else 0.0)
    q = clustered(nq, d, 400 + d + m) + (1.0 if metric == "cosine" else 0.0)
    cent, cb = _models(oracle, x, nlist, m, metric, seed=d + m + 1)
    oidx = oracle.build_index(x, cent, cb, metric)
    gpart, gcodes, _ = eng.ivfpq_encode(x, cent, cb, metric)
    gidx = DeviceIndex.create(eng, metric, cent, cb, gpart, gcodes, None, raw=x)
    # nprobes = nlist: 700 pairs per partition = two resident pair blocks of 352; k = 100: class B appears in the small partitions
    _check(eng, gidx, oidx, q, q, x, [(10, 8, 0), (10, 8, 10), (10, nlist, 0), (100, 7, 0), (1, 7, 1), (37, 9, 3), (128, 6, 0)])
    gidx.close()


def test_mscan_non_integer_data_and_tiny_partitions(eng, oracle):
    """Gaussian rows of small magnitude (sigma = 2^13 / max codeword far from 1), unit-scale residuals, partitions of 1 .. 300 rows
    (row padding inside a 32-row chunk, slices of a single row), queries far from every centroid (large |r|^2 / T: pairs whose
    slack exceeds the cap go to the exact rescan)."""
    from lance_amd.engine import DeviceIndex
    rng = np.random.default_rng(77)
    d, m, nlist, nq = 128, 16, 40, 900
    sizes = np.concatenate([[1, 2, 3, 255, 256, 257, 300], rng.integers(20, 900, nlist - 7)])
    centers = rng.standard_normal((nlist, d)) * 0.05
    x = np.concatenate([centers[i] + rng.standard_normal((s, d)) * 0.004 for i, s in enumerate(sizes)]).astype(f32)
    q = (centers[rng.integers(0, nlist, nq)] + rng.standard_normal((nq, d)) * 0.004).astype(f32)
    q[::7] *= 6.0                                      # far queries
    q[3::50] = 0.0
    cent = centers.astype(f32)
    part, _ = oracle.assign(x, cent)
    res = oracle.residual(x, cent, part)
    cb, _ = oracle.pq_train(res[rng.choice(len(x), 4096, replace=False)], m, max_iters=3, seed=5)
    oidx = oracle.build_index(x, cent, cb)
    gpart, gcodes, _ = eng.ivfpq_encode(x, cent, cb)
    assert (_np(gpart).view(np.uint32) == oidx.part_ids).all() and (_np(gcodes) == oidx.codes_rowmajor).all()
    gidx = DeviceIndex.create(eng, "l2", cent, cb, gpart, gcodes, None, raw=x)
    _check(eng, gidx, oidx, q, q, x, [(10, 10, 0), (10, 10, 10), (5, nlist, 0), (60, 12, 2)])
    gidx.close()


def test_mscan_f16_column(eng, oracle):
    """f16 rows: the residual query is rounded to f16 before anything else (`round_f16`)."""
    from lance_amd.engine import DeviceIndex
    rng = np.random.default_rng(19)
    n, d, m, nlist, nq = 30000, 128, 16, 64, 800
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
    _check(eng, gidx, oidx, q, q, x.astype(f32), [(10, 10, 0), (10, 10, 10), (10, 64, 0), (50, 8, 2)])
    gidx.close()


def test_mscan_many_ties(eng, oracle):
    """Rows duplicated many times: hundreds of rows tie at the bound inside one partition -- segments overflow (more than 256
    survivors) and go through the exact rescan; the (dist, rowid) order of the ties must still be the oracle's."""
    from lance_amd.engine import DeviceIndex
    rng = np.random.default_rng(5)
    d, m, nlist, nq = 128, 16, 16, 600
    base = clustered(300, d, 8)
    x = base[rng.integers(0, 300, 24000)]
    q = base[rng.integers(0, 300, nq)] + rng.integers(0, 2, (nq, d)).astype(f32)
    cent, cb = _models(oracle, x, nlist, m, "l2", seed=4)
    oidx = oracle.build_index(x, cent, cb, "l2")
    gpart, gcodes, _ = eng.ivfpq_encode(x, cent, cb, "l2")
    gidx = DeviceIndex.create(eng, "l2", cent, cb, gpart, gcodes, None, raw=x)
    _check(eng, gidx, oidx, q, q, x, [(10, 8, 0), (40, nlist, 0), (10, 8, 4)])
    gidx.close()


def test_mscan_not_taken_for_small_batches_or_dot(eng, oracle):
    from lance_amd.engine import DeviceIndex
    n, d, m, nlist = 12000, 128, 16, 512
    x = clustered(n, d, 31)
    q = clustered(600, d, 32)
    cent = x[np.random.default_rng(1).choice(n, nlist, replace=False)].copy()
    part, _ = oracle.assign(x, cent)
    cb, _ = oracle.pq_train(oracle.residual(x, cent, part)[:3072], m, max_iters=2, seed=3)
    oidx = oracle.build_index(x, cent, cb)
    gpart, gcodes, _ = eng.ivfpq_encode(x, cent, cb)
    gidx = DeviceIndex.create(eng, "l2", cent, cb, gpart, gcodes, None, raw=x)
    with _ms_used(eng, expect=False):                  # 600 x 8 = 4800 pairs < 96 x 512: the integer scan serves it
        gi, gd = gidx.search(q, 10, 8, 0)
    oi, od = oidx.search(q, 10, 8)
    assert (_np(gi).view(np.uint64) == oi).all() and (_np(gd).view(np.uint32) == od.view(np.uint32)).all()
    gidx.close()


def test_large_batches_go_through_in_slices(eng, oracle, monkeypatch):
    """DeviceIndex.search splits a synchronous batch whose (query, probe) pairs exceed MAX_PAIRS_PER_CALL into slices of queries (the
    library's batched kernels refuse batches whose survivor segments pass 2 GiB of scratch): same answers, every slice on the batched path."""
    from lance_amd.engine import DeviceIndex
    n, d, m, nlist, nq = 12000, 128, 16, 12, 1500
    x = clustered(n, d, 71)
    q = clustered(nq, d, 72)
    cent, cb = _models(oracle, x, nlist, m, "l2", seed=9)
    oidx = oracle.build_index(x, cent, cb, "l2")
    gpart, gcodes, _ = eng.ivfpq_encode(x, cent, cb, "l2")
    gidx = DeviceIndex.create(eng, "l2", cent, cb, gpart, gcodes, None, raw=x)
    monkeypatch.setattr(DeviceIndex, "MAX_PAIRS_PER_CALL", 4800)      # 1500 x 8 = 12,000 pairs -> three slices of 600 queries
    _check(eng, gidx, oidx, q, q, x, [(10, 8, 0), (10, 8, 5)])
    gidx.close()


def test_integer_scan_keeps_its_coverage_without_mscan():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LANCE_HIP_NO_MSCAN="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_pm_scan.py"), "-m", "gpu", "-q", "-x",
                        "-k", "every_instantiation or two_class or overflow or random_shapes or prefilter", "-p", "no:cacheprovider"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
