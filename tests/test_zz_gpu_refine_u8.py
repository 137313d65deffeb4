"""Refine from the lossless u8 copy of an integer-valued f32 column (lance_amd/csrc/search.hip raw_compact_prepare, index.h raw_u8).

The refine step (scanner.rs:2884-2904 `take` + flat_knn :3336-3412) reads k * refine_factor raw rows per query at random: at C2 that is
512 MB per 10,000-query batch and the kernel sits at the HBM rate.  When EVERY element of the f32 column is bit-for-bit the widening of
a byte (SIFT descriptors are) the index keeps a u8 copy and the refine kernel reads a quarter of the bytes -- the same f32 values after
`v_cvt_f32_ubyte`, the same operation order, so ids AND distance bits must equal the oracle's, which reads the f32 column.

Asserted here: which source the refine read (the `refine_u8` stage counter), for columns that are representable and for the four ways a
column is not (a fraction, a value above 255, a negative value, a subnormal), that -0.0 is taken as the byte 0 (numpy's rint / clip leave
it behind; neither metric can tell the sign of a zero element: search.hip), that set_raw drops the copy, and that prewarm builds it."""
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


class _u8_used:
    def __init__(self, eng, expect):
        self.eng, self.expect = eng, expect

    def __enter__(self):
        self.eng.timing(True)
        self.before = self.eng.timing_query("refine_u8")[1]
        return self

    def __exit__(self, *a):
        self.eng.synchronize()
        after = self.eng.timing_query("refine_u8")[1]
        self.eng.timing(False)
        if a[0] is None:
            assert (after > self.before) == self.expect, "u8 refine source " + ("not taken" if self.expect else "taken unexpectedly")


def _equal(gi, gd, oi, od, what):
    bad = np.nonzero((_np(gi).view(np.uint64) != oi).any(axis=1))[0]
    assert bad.size == 0, f"{what}: ids differ for {bad.size} queries (first {bad[:5]})"
    assert (_np(gd).view(np.uint32) == od.view(np.uint32)).all(), f"{what}: distance bits differ"


@pytest.mark.parametrize("metric", ["l2", "dot"])
@pytest.mark.parametrize("d,m", [(128, 16), (64, 16), (48, 12), (272, 17)])
def test_refine_from_u8_copy_is_bit_equal(eng, oracle, metric, d, m):
    from lance_amd.engine import DeviceIndex
    n, nlist, nq = 6000, 12, 300
    x = clustered(n, d, 900 + d, hi=255.0)
    x[0, :] = 0.0
    x[1, :] = 255.0            # both ends of the byte range
    q = clustered(nq, d, 901 + d, integer=False)      # queries need not be integers
    cent, cb = _models(oracle, x, nlist, m, metric, seed=d + 3)
    oidx = oracle.build_index(x, cent, cb, metric)
    gpart, gcodes, _ = eng.ivfpq_encode(x, cent, cb, metric)
    gidx = DeviceIndex.create(eng, metric, cent, cb, gpart, gcodes, None, raw=x)
    for k, nprobes, rf in [(10, 4, 10), (10, nlist, 3), (1, 3, 1), (100, 5, 2)]:
        with _u8_used(eng, True):
            gi, gd = gidx.search(q, k, nprobes, rf)
        oi, od = oidx.search(q, k, nprobes, refine=rf, raw=x)
        _equal(gi, gd, oi, od, f"{metric} d={d} k={k} nprobes={nprobes} refine={rf}")
    gidx.close()


@pytest.mark.parametrize("spoil", ["fraction", "above", "negative", "subnormal"])
def test_column_that_is_not_representable_stays_f32(eng, oracle, spoil):
    from lance_amd.engine import DeviceIndex
    n, d, m, nlist, nq = 5000, 64, 16, 8, 200
    x = clustered(n, d, 77)
    x[n - 1, d - 1] = {"fraction": 17.5, "above": 256.0, "negative": -1.0, "subnormal": 1e-40}[spoil]      # ONE element, in the last lane's last chunk
    q = clustered(nq, d, 78)
    cent, cb = _models(oracle, x, nlist, m, "l2", seed=5)
    oidx = oracle.build_index(x, cent, cb, "l2")
    gpart, gcodes, _ = eng.ivfpq_encode(x, cent, cb, "l2")
    gidx = DeviceIndex.create(eng, "l2", cent, cb, gpart, gcodes, None, raw=x)
    with _u8_used(eng, False):
        gi, gd = gidx.search(q, 10, 4, 5)
    oi, od = oidx.search(q, 10, 4, refine=5, raw=x)
    _equal(gi, gd, oi, od, spoil)
    gidx.close()


@pytest.mark.parametrize("metric", ["l2", "dot"])
def test_minus_zero_elements_are_the_byte_zero(eng, oracle, metric):
    from lance_amd.engine import DeviceIndex
    n, d, m, nlist, nq = 5000, 64, 16, 8, 200
    x = clustered(n, d, 79)
    x[x < 40] = -0.0            # thousands of negative zeros
    assert np.signbit(x).any()
    q = clustered(nq, d, 80, integer=False) - f32(60.0)      # negative, zero-crossing query components
    q[0, :8] = 0.0; q[1, :8] = -0.0
    cent, cb = _models(oracle, x, nlist, m, metric, seed=6)
    oidx = oracle.build_index(x, cent, cb, metric)
    gpart, gcodes, _ = eng.ivfpq_encode(x, cent, cb, metric)
    gidx = DeviceIndex.create(eng, metric, cent, cb, gpart, gcodes, None, raw=x)
    with _u8_used(eng, True):
        gi, gd = gidx.search(q, 10, 4, 5)
    oi, od = oidx.search(q, 10, 4, refine=5, raw=x)
    _equal(gi, gd, oi, od, "minus zero " + metric)
    gidx.close()


def test_set_raw_drops_the_copy_and_prewarm_builds_it(eng, oracle):
    from lance_amd.engine import DeviceIndex
    n, d, m, nlist, nq = 5000, 128, 16, 8, 200
    x = clustered(n, d, 31)
    q = clustered(nq, d, 32)
    cent, cb = _models(oracle, x, nlist, m, "l2", seed=9)
    oidx = oracle.build_index(x, cent, cb, "l2")
    gpart, gcodes, _ = eng.ivfpq_encode(x, cent, cb, "l2")
    gidx = DeviceIndex.create(eng, "l2", cent, cb, gpart, gcodes, None, raw=x)
    gidx.prewarm()
    with _u8_used(eng, True):
        gi, gd = gidx.search(q, 10, 4, 5)
    oi, od = oidx.search(q, 10, 4, refine=5, raw=x)
    _equal(gi, gd, oi, od, "prewarmed")
    # a different column behind the same row ids: fractional values -> the old copy must not be read
    x2 = x + f32(0.25)
    gidx.set_raw(x2)
    with _u8_used(eng, False):
        gi, gd = gidx.search(q, 10, 4, 5)
    oi, od = oidx.search(q, 10, 4, refine=5, raw=x2)
    _equal(gi, gd, oi, od, "after set_raw(fractional)")
    # and back to an integer column with other contents
    x3 = np.clip(x + f32(3.0), 0, 255).astype(f32)
    gidx.set_raw(x3)
    with _u8_used(eng, True):
        gi, gd = gidx.search(q, 10, 4, 5)
    oi, od = oidx.search(q, 10, 4, refine=5, raw=x3)
    _equal(gi, gd, oi, od, "after set_raw(integer)")
    gidx.close()


def test_u8_refine_through_captured_graphs(eng, oracle):
    """The serving pattern: same buffers, new query contents, the call replayed as a HIP graph -- the first (plain) call builds the copy,
    the captured calls read it."""
    import torch
    from lance_amd.engine import DeviceIndex
    n, d, m, nlist, nq = 8000, 128, 16, 8, 256
    x = clustered(n, d, 41)
    cent, cb = _models(oracle, x, nlist, m, "l2", seed=11)
    oidx = oracle.build_index(x, cent, cb, "l2")
    gpart, gcodes, _ = eng.ivfpq_encode(x, cent, cb, "l2")
    gidx = DeviceIndex.create(eng, "l2", cent, cb, gpart, gcodes, None, raw=x)
    qbuf = torch.empty((nq, d), dtype=torch.float32, device="cuda")
    out = (torch.empty((nq, 10), dtype=torch.int64, device="cuda"), torch.empty((nq, 10), dtype=torch.float32, device="cuda"))
    stage = lambda: eng.timing_query("count:refine_u8")[1]        # stage counters also count the stages of a replayed graph
    replays = lambda: eng.timing_query("count:graph_replay")[1]
    r0 = replays()
    for rep in range(6):
        q = clustered(nq, d, 500 + rep)
        qbuf.copy_(torch.from_numpy(q))
        out[0].fill_(-7); out[1].fill_(-7.0)
        s0 = stage()
        gi, gd = gidx.search(qbuf, 10, 4, 10, out=out)
        torch.cuda.synchronize()
        assert stage() > s0, f"rep {rep}: the refine did not read the u8 copy"
        oi, od = oidx.search(q, 10, 4, refine=10, raw=x)
        _equal(gi, gd, oi, od, f"rep {rep}")
    assert replays() - r0 >= 3, "the repeated call was not replayed as a graph"
    gidx.close()


def test_set_raw_with_the_same_pointer_invalidates_captured_graphs(eng, oracle):
    """A caller re-attaches the SAME device buffer after changing its contents: the captured searches of the first attachment hold the
    address of a u8 copy that set_raw has freed -- they must not be replayed (the graph key carries the attachment's generation)."""
    import torch
    from lance_amd.engine import DeviceIndex
    n, d, m, nlist, nq = 8000, 128, 16, 8, 256
    x = clustered(n, d, 51)
    cent, cb = _models(oracle, x, nlist, m, "l2", seed=12)
    oidx = oracle.build_index(x, cent, cb, "l2")
    gpart, gcodes, _ = eng.ivfpq_encode(x, cent, cb, "l2")
    xt = torch.from_numpy(x).cuda()
    gidx = DeviceIndex.create(eng, "l2", cent, cb, gpart, gcodes, None, raw=xt)
    q = clustered(nq, d, 52)
    qbuf = torch.from_numpy(q).cuda()
    out = (torch.empty((nq, 10), dtype=torch.int64, device="cuda"), torch.empty((nq, 10), dtype=torch.float32, device="cuda"))
    replays = lambda: eng.timing_query("count:graph_replay")[1]
    for contents in (x, np.clip(x + f32(2.0), 0, 255).astype(f32), x + f32(0.5)):      # u8, u8 (other values), not representable
        xt.copy_(torch.from_numpy(contents))
        gidx.set_raw(xt)                                       # the same tensor, hence the same pointer
        oi, od = oidx.search(q, 10, 4, refine=10, raw=contents)
        r0 = replays()
        for rep in range(4):
            out[0].fill_(-7); out[1].fill_(-7.0)
            gi, gd = gidx.search(qbuf, 10, 4, 10, out=out)
            torch.cuda.synchronize()
            _equal(gi, gd, oi, od, f"rep {rep}")
        assert replays() - r0 >= 2, "the repeated call was not replayed as a graph"
    gidx.close()
