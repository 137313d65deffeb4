"""CPU check of the long-row flat filter's margin (lance_amd/csrc/flat_mfma_wide.hip): one bf16 product per (row, query) pair, f32 accumulation
over d <= 4096 elements.  Its header claims |x~.q~ - x.q| < EW |x||q| with EW = 0.0084 (bfloat16's unit roundoff is 2^-8 PER OPERAND: 2^-7 for the
product; the first version of the kernel carried 0.0045 and this test is what caught it) and derives the three tests from it
(L2: |x|^2 (1 - EW) + |q|^2 (1 - EW) - 2 x~.q~ <= T; dot: 1 - x~.q~ - EW (|x|^2 + |q|^2) <= T; cosine: (1 - EWC - T) |q||x| - x~.q~ <= 0).
Here the rounding is restated in numpy (round-to-nearest-even to bfloat16, products exact in f64, accumulated in f32 in two orders) and the
claim is checked on adversarially aligned data as well as random data: no pair whose TRUE distance is <= T may fail its test."""
import numpy as np
import pytest

f32, f64 = np.float32, np.float64
EW, EWC = f32(0.0084), f32(0.0085)


def bf16_rne(a):
    u = np.ascontiguousarray(a, f32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return (r & 0xFFFFFFFF).astype(np.uint32).view(f32)


def acc_f32(xb, qb, order):
    prod = (xb.astype(f64) * qb.astype(f64)[None, :])          # 8-bit x 8-bit significands: exact
    if order == "seq":
        acc = np.zeros(xb.shape[0], f32)
        for k in range(prod.shape[1]):
            acc = (acc.astype(f64) + prod[:, k]).astype(f32)
        return acc
    p = prod.astype(f32)                                       # exact as f32 too (16 significant bits)
    while p.shape[1] > 1:
        if p.shape[1] % 2:
            p = np.concatenate([p, np.zeros((p.shape[0], 1), f32)], axis=1)
        p = (p[:, 0::2] + p[:, 1::2]).astype(f32)
    return p[:, 0]


def data(kind, n, d, rng):
    if kind == "gauss":
        x = rng.standard_normal((n, d)); q = rng.standard_normal(d)
    elif kind == "unit-aligned":                                # rows nearly parallel to the query: the product error adds up coherently
        q = rng.standard_normal(d); q /= np.linalg.norm(q)
        x = q[None, :] + 0.05 * rng.standard_normal((n, d))
    elif kind == "worst-mantissa":                              # every element just below a bf16 rounding boundary, rows PARALLEL to the query, all
        base = 1.0 + (2.0 ** -8) * (1 - 2.0 ** -10)             # positive: each product errs by -(2^-7) of itself and nothing cancels.  (bfloat16 has 8
        q = base * np.exp2(rng.integers(-3, 4, d))              # significant bits: half an ulp of 1.0 is 2^-8)
        x = q[None, :] * np.exp2(rng.integers(-2, 3, (n, 1)))
    else:                                                       # integer SIFT-like
        x = rng.integers(0, 219, (n, d)).astype(f64); q = rng.integers(0, 219, d).astype(f64)
    return x.astype(f32), q.astype(f32)


@pytest.mark.parametrize("kind", ["gauss", "unit-aligned", "worst-mantissa", "sift"])
@pytest.mark.parametrize("d", [144, 1536, 4096])
def test_one_bf16_product_stays_inside_the_margin(kind, d):
    rng = np.random.default_rng(d + len(kind))
    x, q = data(kind, 400 if d < 4096 else 150, d, rng)
    xb, qb = bf16_rne(x), bf16_rne(q)
    true_dot = x.astype(f64) @ q.astype(f64)
    nx, nq = np.linalg.norm(x.astype(f64), axis=1), np.linalg.norm(q.astype(f64))
    for order in ("seq", "tree"):
        approx = acc_f32(xb, qb, order).astype(f64)
        err = np.abs(approx - true_dot)
        assert (err <= float(EW) * nx * nq).all(), (kind, d, order, float((err / (nx * nq)).max()))
        assert (err / (nx * nq)).max() <= 2.0 ** -7 * 1.07      # the header's accounting: 2^-7 + 2^-16 + the accumulation term
        if kind == "worst-mantissa":
            assert (err / (nx * nq)).max() > 0.0045             # ... and the aligned case really gets there: one operand's roundoff is not enough
        # the three tests, at the tightest legal threshold (T = the pair's own true distance): a pair with distance <= T must pass
        n2x, n2q = nx * nx, nq * nq
        l2_true = n2x + n2q - 2 * true_dot
        assert (n2x * (1 - float(EW)) + n2q * (1 - float(EW)) - 2 * approx <= l2_true + 1e-9 * (n2x + n2q)).all(), (kind, d, "l2")
        dot_true = 1 - true_dot
        assert (1 - approx - float(EW) * (n2x + n2q) <= dot_true + 1e-9 * (n2x + n2q)).all(), (kind, d, "dot")
        cos_true = 1 - true_dot / (nx * nq)
        assert (((1 - float(EWC) - cos_true) * nq * nx - approx) <= 1e-9 * nx * nq).all(), (kind, d, "cosine")
