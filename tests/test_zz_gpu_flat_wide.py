"""Batched flat KNN over LONG f32 rows on the matrix cores (lance_amd/csrc/flat_mfma_wide.hip): a K-tiled bf16 product filters the
(query, row) pairs, the pairs that can beat a query's threshold are recomputed in the reference's arithmetic (l2.rs:57-91 / dot.rs:52-89
16-lane order, cosine.rs:143-175 cosine_fast) -- so row ids and distances must equal the oracle's flat_knn bit for bit
(KNNVectorDistanceExec + SortExec, knn.rs:218-246, scanner.rs:3386-3406), for L2, dot and cosine, any d % 16 == 0.

Every case asserts whether the matrix-core filter ran (`flat_mfma_wide` stage counter); its error margin is wide by design
(one bf16 product: 0.84 % of |x||q|), so the cases put many rows close to each query's k-th distance."""
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


def _ran(eng):
    return eng.timing_query("count:flat_mfma_wide")[1]


def _check(eng, oracle, x, q, k, metric, row_ids=None, tag=None, expect=True):
    before = _ran(eng)
    gi, gd = eng.flat_topk(x, q, k, metric, row_ids=row_ids)
    assert (_ran(eng) > before) == expect, (tag, "matrix-core filter " + ("not taken" if expect else "taken unexpectedly"))
    oi, od = oracle.flat_knn(x, q, k, metric, row_ids=row_ids)
    gi = gi.cpu().numpy().view(np.uint64); gd = gd.cpu().numpy()
    bad = np.nonzero((gi != oi).any(axis=1))[0]
    assert bad.size == 0, (tag, f"ids differ for {bad.size} of {q.shape[0]} queries, first {bad[:5]}")
    assert (gd.view(np.uint32) == od.view(np.uint32)).all(), (tag, "distance bits differ")


def _data(rng, n, nq, d, unit=False, integer=False):
    centers = rng.normal(0, 1.0, (24, d))
    x = centers[rng.integers(0, 24, n)] + rng.normal(0, 0.35, (n, d))
    q = centers[rng.integers(0, 24, nq)] + rng.normal(0, 0.35, (nq, d))
    if integer:
        x, q = np.rint(x * 20), np.rint(q * 20)
    if unit:
        x /= np.linalg.norm(x, axis=1, keepdims=True); q /= np.linalg.norm(q, axis=1, keepdims=True)
    return x.astype(f32), q.astype(f32)


@pytest.mark.parametrize("metric", ["l2", "dot", "cosine"])
@pytest.mark.parametrize("d,n", [(256, 40_000), (1536, 24_000), (144, 30_000), (960, 16_000)])
def test_long_rows_match_the_oracle(eng, oracle, d, n, metric):
    rng = np.random.default_rng(1000 + d)
    x, q = _data(rng, n, 300, d, unit=(metric == "cosine" and d == 1536))
    x[100] = x[7]; x[5000] = x[7]; q[0] = x[7]              # exact ties: the smaller row id first
    for k in (10, 1, 100):
        _check(eng, oracle, x, q, k, metric, tag=(d, metric, k))


def test_cosine_takes_it_at_every_dimension(eng, oracle):
    """Cosine has no fixed-dimension kernel (cosine_fast's own lane layout), so its query batches take the K-tiled filter for short rows
    too; the rows' bf16 plane comes out of the row-norm pass."""
    rng = np.random.default_rng(21)
    for d in (32, 64, 128):
        x, q = _data(rng, 30_000, 256, d)
        _check(eng, oracle, x, q, 10, "cosine", tag=("cosine", d))


def test_integer_rows_mass_ties_and_row_ids(eng, oracle):
    """Integer-valued rows: many exactly equal distances around every threshold; 3000 copies of one row; row ids unrelated to the storage order."""
    rng = np.random.default_rng(7)
    n, d = 30_000, 256
    x, q = _data(rng, n, 200, d, integer=True)
    x[10_000:13_000] = x[9]
    q[:20] = x[9] + rng.integers(0, 2, (20, d))
    rid = rng.permutation(n).astype(np.uint64) * 3 + 5
    for metric in ("l2", "dot", "cosine"):
        _check(eng, oracle, x, q, 10, metric, row_ids=rid, tag=("ties", metric))
    _check(eng, oracle, x, q, 128, "l2", tag="ties k=128")


def test_degenerate_rows(eng, oracle):
    """A row of huge components (|x|^2 overflows f32: L2 distance inf, cosine norm inf), a row scaled by 1e-20 (its norm underflows), and
    zero rows: the filter must hand them to the exact arithmetic, whatever it decides.  Under cosine a zero row's distance is 0 / 0 -- a
    NaN whose SIGN is the platform's (x86: -NaN, first under f32::total_cmp; this GPU and aarch64: +NaN, last; the reference itself
    differs between platforms here, tests/test_zz_gpu_zz_ivfflat_ties.py), so with zero rows the cosine answers are compared with the
    oracle's after its NaN entries are dropped, and against the exact kernel's own answer (profiles/r05i_nan_debug.txt)."""
    rng = np.random.default_rng(9)
    n, d = 20_000, 256
    x, q = _data(rng, n, 160, d)
    x[31] = 3e19                                   # |x|^2 = inf
    x[32] *= f32(1e-20)
    for metric in ("cosine", "l2", "dot"):
        _check(eng, oracle, x, q, 10, metric, tag=("degenerate rows", metric))
    x[17] = 0.0; x[9000] = 0.0                     # row 17 is seen by the exact first epoch, row 9000 by the matrix-core filter
    for metric in ("l2", "dot"):
        _check(eng, oracle, x, q, 10, metric, tag=("zero rows", metric))
    before = _ran(eng)
    gi, gd = eng.flat_topk(x, q, 10, "cosine")
    assert _ran(eng) > before
    gi = gi.cpu().numpy().view(np.uint64); gd = gd.cpu().numpy()
    oi, od = oracle.flat_knn(x, q, 12, "cosine")
    for r in range(q.shape[0]):
        keep = ~np.isnan(od[r])
        assert (gi[r] == oi[r][keep][:10]).all() and (gd[r].view(np.uint32) == od[r][keep][:10].view(np.uint32)).all(), ("zero rows, cosine", r)


def test_degenerate_queries(eng, oracle):
    """A zero query and a query scaled by 1e-20: every row is a candidate of the zero query under cosine (0 / 0), the queue overflows and the
    repair loop of flat.hip decides with the exact kernel.  Its ten distances are NaN on both sides (sign: see above); every other query
    must be bit-equal."""
    rng = np.random.default_rng(10)
    n, d = 12_000, 256
    x, q = _data(rng, n, 160, d)
    q[5] = 0.0
    q[6] *= f32(1e-20)
    for metric in ("l2", "dot"):
        _check(eng, oracle, x, q, 10, metric, tag=("degenerate queries", metric))
    before = _ran(eng)
    gi, gd = eng.flat_topk(x, q, 10, "cosine")
    assert _ran(eng) > before
    gi = gi.cpu().numpy().view(np.uint64); gd = gd.cpu().numpy()
    oi, od = oracle.flat_knn(x, q, 10, "cosine")
    rest = np.arange(q.shape[0]) != 5
    assert (gi[rest] == oi[rest]).all() and (gd[rest].view(np.uint32) == od[rest].view(np.uint32)).all()
    assert np.isnan(gd[5]).all() and np.isnan(od[5]).all() and len(set(gi[5].tolist())) == 10 and (gi[5] < n).all()


def test_small_batches_and_short_rows_keep_their_kernels(eng, oracle):
    rng = np.random.default_rng(11)
    x, q = _data(rng, 20_000, 100, 256)
    _check(eng, oracle, x, q, 10, "l2", tag="nq < 128", expect=False)
    x, q = _data(rng, 20_000, 300, 128)
    _check(eng, oracle, x, q, 10, "l2", tag="d = 128: the register-resident filter", expect=False)


def test_many_query_chunks(eng, oracle):
    """More queries than one pass takes (2048): the rows' bf16 plane is made once, the last chunk is short."""
    rng = np.random.default_rng(13)
    x, q = _data(rng, 12_000, 2048 + 200, 192)
    _check(eng, oracle, x, q, 10, "cosine", tag="chunks")


def test_aligned_rounding_errors_do_not_drop_neighbours(eng, oracle):
    """The filter's worst case (tests/test_flat_wide_spec.py): every element half an ulp below a bfloat16 rounding boundary, all positive,
    rows nearly parallel to their query -- each product errs by -2^-7 of itself and nothing cancels, so x~.q~ is 0.78 % short of x.q.  The
    near neighbours arrive in LATER epochs than a first set that already gave the query a tight threshold: a margin of one operand's
    roundoff (0.45 %, the kernel's first version) drops them; the margin of the product's roundoff (0.84 %) must not."""
    rng = np.random.default_rng(17)
    d, nq = 1024, 128
    base = 1.0 + (2.0 ** -8) * (1 - 2.0 ** -10)

    def variant(v, m):
        x = v.copy()
        x[rng.choice(d, m, replace=False)] *= 2.0
        return x

    qs = [base * rng.choice([1.0, 2.0], d) for _ in range(nq)]
    first = [variant(qv, int(rng.integers(6, 9))) for qv in qs for _ in range(12)]                # 1536 rows: the first epoch's threshold
    near = [variant(qv, int(rng.integers(1, 6))) for qv in qs for _ in range(12)]                 # arrive in the second / third epoch
    filler = [base * rng.choice([1.0, 2.0], d) for _ in range(30_000)]
    x = np.asarray(first + filler[:600] + filler[600:20_000] + near + filler[20_000:], f32)
    q = np.asarray(qs, f32)
    assert (x[:2048].shape[0] == 2048) and len(first) == 1536
    for metric in ("cosine", "l2", "dot"):
        _check(eng, oracle, x, q, 10, metric, tag=("aligned roundoff", metric))
