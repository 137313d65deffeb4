"""Exhaustive KNN for one to four queries (lance_amd/csrc/flat_small.hip, round 4): a single streaming pass -- 16-lane groups own
rows (lane i = lane accumulator i of l2_scalar / dot_scalar), workgroups keep a running threshold, a merge kernel takes the
(distance, row id) order.  Row ids and distances must equal the oracle's flat_knn bit for bit
(KNNVectorDistanceExec + SortExec, knn.rs:218-246, scanner.rs:3386-3406), for every column type the kernel reads natively."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
f32 = np.float32


@pytest.fixture(scope="module")
def eng(engine):
    from lance_amd.engine import Engine
    e = Engine()
    yield e
    e.close()


def _check(eng, oracle, x, q, k, metric, row_ids=None, tag=None):
    gi, gd = eng.flat_topk(x, q, k, metric, row_ids=row_ids)
    xo = x.numpy() if isinstance(x, torch.Tensor) else x
    qo = q.numpy() if isinstance(q, torch.Tensor) else q
    oi, od = oracle.flat_knn(xo if xo.dtype == np.float16 else xo.astype(f32), qo.astype(np.float16) if xo.dtype == np.float16 else qo.astype(f32), k, metric,
                             row_ids=row_ids)
    assert (gi.cpu().numpy().view(np.uint64) == oi).all(), tag
    assert (gd.cpu().numpy().view(np.uint32) == od.view(np.uint32)).all(), tag


@pytest.mark.parametrize("metric", ["l2", "dot"])
def test_small_batches_match_the_oracle(eng, oracle, metric):
    rng = np.random.default_rng(17)
    for d, n in ((128, 60_000), (100, 20_000), (20, 9_000), (1536, 6_000), (16, 5_000)):
        centers = rng.uniform(0, 128, (32, d))
        x = np.clip(np.rint(centers[rng.integers(0, 32, n)] + rng.normal(0, 24, (n, d))), 0, 218).astype(f32)
        x[100] = x[7]; x[4000] = x[7]                 # exact ties: the smaller row id first
        for nq in (1, 2, 3, 4):
            q = x[rng.integers(0, n, nq)] + rng.integers(0, 2, (nq, d)).astype(f32)
            q[0] = x[7]
            for k in (1, 10, 128):
                _check(eng, oracle, x, q, k, metric, tag=(d, n, nq, k))


def test_native_f16_and_int8_columns(eng, oracle):
    rng = np.random.default_rng(18)
    n, d = 30_000, 128
    xi = rng.integers(-128, 128, (n, d)).astype(np.int8)
    qi = rng.integers(-128, 128, (3, d)).astype(np.int8)
    for metric in ("l2", "dot"):
        _check(eng, oracle, torch.from_numpy(xi), torch.from_numpy(qi), 10, metric, tag=("int8", metric))
    xh = (rng.integers(0, 219, (n, d)) / 256.0).astype(np.float16)
    qh = (rng.integers(0, 219, (2, d)) / 256.0).astype(np.float16)
    _check(eng, oracle, torch.from_numpy(xh), torch.from_numpy(qh), 10, "l2", tag="f16-l2")
    _check(eng, oracle, torch.from_numpy(xh), torch.from_numpy(qh), 10, "dot", tag="f16-dot (32-lane order: batch path)")


def test_row_ids_nan_rows_and_mass_duplicates(eng, oracle):
    rng = np.random.default_rng(19)
    n, d = 40_000, 64
    x = np.rint(rng.uniform(0, 60, (n, d))).astype(f32)
    rid = rng.permutation(n).astype(np.uint64) * 3 + 5          # ids unrelated to the storage order
    x[50] = np.nan                                               # a NaN row sorts last (total_cmp)
    q = x[[11, 12]] + 1.0
    _check(eng, oracle, x, q, 10, "l2", row_ids=rid, tag="row ids")
    # 6000 identical rows: more ties at the threshold than a workgroup's list holds -> the flag sends the call to the batch path
    x[10_000:16_000] = x[9]
    q2 = x[[9]].copy()
    _check(eng, oracle, x, q2, 10, "l2", tag="mass duplicates")
    _check(eng, oracle, x, q2, 128, "l2", row_ids=rid, tag="mass duplicates, row ids")
    # fewer rows than k in the table's tail slices, and k larger than the table's distinct distances
    _check(eng, oracle, x[:4100], q, 128, "l2", tag="short table")


def test_merge_refill_and_flag_word_reset(eng, oracle):
    """Round 6: the merge kernel's two parallel passes hand over to its step loop when more pairs lie at or under the k-th smallest lane
    minimum than the list holds (here: tens of thousands of identical rows spread over every workgroup's slice, k = 128, so nearly all
    G x k pairs tie), the step loop raises the flag, the call takes the batch path -- and the NEXT single-query call must find the
    device flag word cleared (the merge kernel clears it behind itself: there is no fill kernel in front of the scan any more)."""
    rng = np.random.default_rng(23)
    n, d = 60_000, 16
    x = np.rint(rng.uniform(0, 40, (n, d))).astype(f32)
    dup = rng.permutation(n)[:40_000]
    x[dup] = x[3]
    q = x[[3]].copy()
    _check(eng, oracle, x, q, 128, "l2", tag="ties in every slice, k = 128")
    _check(eng, oracle, x, q + 0.5, 10, "l2", tag="after an overflowed call")
    y = np.rint(rng.uniform(0, 40, (n, d))).astype(f32)
    for k in (1, 10, 100):
        _check(eng, oracle, y, y[[17]] + 0.25, k, "dot", tag=("clean table after an overflowed call", k))


def test_two_to_four_queries_on_the_single_pass_kernel():
    """By default one or two queries take flat_small.hip (the batch path is faster from three queries on); LANCE_HIP_FLAT_SMALL_MAXQ=4
    (read once per process) sends two to four there as well: the cases above again in a child process with the switch set."""
    import os
    import subprocess
    import sys
    if os.environ.get("LANCE_HIP_FLAT_SMALL_MAXQ"):
        pytest.skip("already inside the child run")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider",
                        "-k", "small_batches or native or row_ids or refill"], cwd=root, env=dict(os.environ, LANCE_HIP_FLAT_SMALL_MAXQ="4"),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
