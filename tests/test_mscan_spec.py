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


@pytest.mark.parametrize("d,m,keff", [(128, 16, 100), (128, 32, 60), (64, 16, 10)])
def test_bound_pass_never_undershoots_the_keff_th_distance(spec, d, m, keff):
    """ms_bound_kernel (round 5): T = (upper edge of the histogram bin where the count of dist~ reaches keff) + 1.1 E must be >= the keff-th
    smallest REFERENCE distance of the query's nearest partition, for either accumulation order of the f16 products -- and stay a bound
    worth having (within 2 % of it)."""
    x = spec.sift_like(12000, d, 31 + d + m)
    q = spec.sift_like(150, d, 32 + d + m)
    t = spec.run_bound(f"d={d} M={m}", x, q, m, 8, keff=keff, max_queries=60, seed=d + m, verbose=False)
    assert t["queries"] >= 50 and t["no_bound"] == 0
    assert t["violations"] == 0
    assert 1.0 <= t["mean_ratio"] <= 1.02 and t["worst_ratio"] <= 1.03


def test_bound_pass_gives_no_bound_to_residuals_beyond_binary16(spec):
    xi = spec.sift_like(12000, 128, 41)
    qf = spec.sift_like(60, 128, 42) * f32(8.0)      # sigma |r| leaves binary16: the kernel must decline (class B: exact pair kernel), not guess
    with np.errstate(over="ignore"):
        t = spec.run_bound("far", xi, qf, 16, 8, keff=60, max_queries=40, seed=3, verbose=False)
    assert t["queries"] >= 30 and t["violations"] == 0 and t["no_bound"] == t["queries"]


@pytest.mark.parametrize("d,m,keff", [(128, 16, 60), (128, 32, 60), (64, 16, 10)])
def test_dot_flow_bound_and_filter(spec, d, m, keff):
    """Round 6, the dot metric on the same kernels (operand q / 2 against the CENTRED codebook plane, row term zero, limits and sums relative to
    base = 1 - q . mu - |q| cmax): the histogram bound is never below the nearest partition's keff-th reference distance
    (pq/distance.rs:60-92 table, sequential sum, minus M - 1), no row under the bound fails the test in any probed partition, and every passing
    row's integer sum is within the slack units of (dist - base) s -- for SIFT-like rows (all distances large and negative)."""
    x = spec.sift_like(12000, d, 41 + d + m)
    q = spec.sift_like(120, d, 42 + d + m)
    t = spec.run_dot(f"d={d} M={m}", x, q, m, 8, keff=keff, nprobes=3, max_pairs=36, seed=d + m, verbose=False)
    assert t["queries"] >= 10 and t["no_bound"] == 0 and t["must"] > 300
    assert t["bound_violations"] == 0 and t["violations"] == 0 and t["sum_violations"] == 0
    assert t["survivors"] <= 1.3 * t["must"] and t["worst_sum_err"] <= 1.0
    assert t["mean_excess"] < 0.01              # T sits within 1 % of G above the true keff-th distance


def test_dot_flow_unit_vectors_and_signed_rows(spec):
    rng = np.random.default_rng(6)
    x = rng.standard_normal((12000, 128)).astype(f32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    q = rng.standard_normal((100, 128)).astype(f32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    t = spec.run_dot("unit", x.astype(f32), q.astype(f32), 16, 8, keff=60, nprobes=3, max_pairs=30, seed=1, verbose=False)
    assert t["bound_violations"] == 0 and t["violations"] == 0 and t["sum_violations"] == 0 and t["must"] > 300
    xi = spec.sift_like(12000, 128, 23) - f32(64.0)      # dot products of both signs, distances on both sides of zero
    qi = spec.sift_like(100, 128, 24) - f32(64.0)
    qi[0::3] *= f32(8.0)
    t = spec.run_dot("centred", xi, qi, 16, 8, keff=60, nprobes=3, max_pairs=36, seed=2, verbose=False)
    assert t["bound_violations"] == 0 and t["violations"] == 0 and t["sum_violations"] == 0 and t["must"] > 300
