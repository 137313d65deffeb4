"""CPU check of the claim the lossless u8 refine source rests on (lance_amd/csrc/search.hip: raw_to_u8_kernel): an element -0.0 of the
f32 column may be stored as the byte 0, i.e. read back as +0.0, without changing a single bit of the refine's distances -- squared L2
forms d = x - y and squares it, dot adds x * y to an accumulator that is never -0.0.  Checked with the reference arithmetic itself (the
oracle's l2_scalar / dot_scalar restatements, l2.rs:57-91, dot.rs:52-89) on columns full of negative zeros, against queries with zeros of
both signs, negative and fractional components, and for the byte widening as a whole (a column of integers 0..255)."""
import numpy as np
import pytest

f32 = np.float32


@pytest.mark.parametrize("d", [16, 48, 128, 272])
def test_sign_of_a_zero_element_never_reaches_the_distance(oracle, d):
    rng = np.random.default_rng(d)
    for trial in range(200):
        y = rng.integers(0, 256, d).astype(f32)
        y[rng.random(d) < 0.4] = 0.0
        y_neg = y.copy()
        y_neg[y_neg == 0.0] = -0.0                      # the column as the caller holds it
        assert np.signbit(y_neg).any() or (y != 0).all()
        q = rng.normal(0, 40, d).astype(f32)
        q[rng.random(d) < 0.3] = 0.0
        q[rng.random(d) < 0.2] = -0.0
        if trial % 5 == 0:
            q[:] = 0.0 if trial % 10 == 0 else -0.0     # all-zero queries of either sign: every product / difference is a zero
        y_u8 = y_neg.astype(np.uint8).astype(f32)       # what the refine kernel widens: +0.0 where the column has -0.0
        assert not np.signbit(y_u8).any() and (y_u8 == y_neg).all()
        for fn in (oracle.l2, oracle.dot):
            a, b = f32(fn(q, y_neg)), f32(fn(q, y_u8))
            assert a.view(np.uint32) == b.view(np.uint32), (fn.__name__, d, trial, a, b)


def test_a_column_with_anything_else_is_not_representable():
    """The kernel's test is `(float)(uint8)v == v` per element: fractions, values outside [0, 255], NaN and infinities fail it; -0.0 passes."""
    v = np.array([0.0, -0.0, 1.0, 255.0, 0.5, 255.5, 256.0, -1.0, 1e-40, np.nan, np.inf, -np.inf], f32)
    with np.errstate(invalid="ignore", over="ignore"):
        b = np.where((v >= 0) & (v < 256), v, 0).astype(np.uint8)
    ok = b.astype(f32) == v
    assert ok.tolist() == [True, True, True, True, False, False, False, False, False, False, False, False]
