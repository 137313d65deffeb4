"""CPU check of the matrix-core ADC filter's bound (lance_amd/csrc/search_ms.hip) through its executable specification
(scripts/sim/ms_filter_spec.py: binary16 operands, exact products, f32 accumulation in two orders, the pre-pass's slack E in f32):
no row whose reference ADC distance (oracle table, sequential-m sum -- pq/distance.rs:109-144) is <= T may fail the test
`|c^|^2 - limit - 2 r.c^ <= 0`, and a passing row's integer sum must lie within the per-pair slack the merge kernel's cut carries.
The GPU side of the same statement is tests/test_zz_gpu_mscan.py (bit-equal search results)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts", "sim"))
f32 = np.float32


@pytest.fixture(scope="module")
def spec(oracle):
    import ms_filter_spec
    return ms_filter_spec


@pytest.mark.parametrize("d,m", [(128, 16), (128, 32), (64, 16)])
def test_filter_never_drops_a_row_under_the_bound(spec, d, m):
    x = spec.sift_like(12000, d, 11 + d + m)
    q = spec.sift_like(120, d, 12 + d + m)
    t = spec.run(f"d={d} M={m}", x, q, m, 8, keff=60, nprobes=3, max_pairs=45, seed=d + m, verbose=False)
    assert t["pairs"] >= 40 and t["must"] > 1000
    assert t["violations"] == 0 and t["sum_violations"] == 0
    assert t["survivors"] <= 1.3 * t["must"]                 # it is a filter: the slack lets ~10 % extra rows through, not multiples
    assert t["worst_sum_err"] <= 1.0


def test_unit_vectors_and_far_queries(spec):
    rng = np.random.default_rng(5)
    x = rng.standard_normal((12000, 128)).astype(f32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    q = rng.standard_normal((100, 128)).astype(f32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    t = spec.run("unit", x.astype(f32), q.astype(f32), 16, 8, keff=60, nprobes=3, max_pairs=30, seed=1, verbose=False)
    assert t["violations"] == 0 and t["sum_violations"] == 0 and t["must"] > 500
    # queries far outside the data: |r|^2 >> T, the slack exceeds 5 % of T -> those pairs are handed to the exact rescan, none is filtered wrongly
    # (x 8: the scaled residual leaves binary16; x 4: it fits, with |r|^2 several times T)
    xi = spec.sift_like(12000, 128, 21)
    qf = spec.sift_like(100, 128, 22)
    qf[0::3] *= f32(8.0)
    qf[1::3] *= f32(4.0)
    with np.errstate(over="ignore"):
        t = spec.run("far", xi, qf, 16, 8, keff=60, nprobes=3, max_pairs=45, seed=2, verbose=False)
    assert t["violations"] == 0 and t["sum_violations"] == 0
    assert t["handed"] > 0 and t["must"] > 500
