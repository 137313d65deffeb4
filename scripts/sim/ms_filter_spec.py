"""Executable specification (numpy, binary16 / f32-faithful) of the matrix-core ADC filter (lance_amd/csrc/search_ms.hip), with the slack
its header derives checked against the reference arithmetic.  Runs on the CPU; `tests/test_mscan_spec.py` runs a small instance.

For a (query, probed partition) pair with residual r = q - centroid (v2.rs:316-332) and a stored row with reconstruction c^ (its M
codewords concatenated):

    dist_ref      = the reference's f32 ADC distance: table entries (orc_build_lut_f32: l2_scalar per sub-vector), summed
                    sequentially over m (pq/distance.rs:109-144)
    sigma         = power of two with the largest |2 sigma c| in [2^13, 2^14)                      (per index)
    cbh           = binary16(-2 sigma c)                  rh  = binary16(sigma r)                  (per index / per pair)
    cn2           = f32(sigma^2 * sum_m |c_m(code_m)|^2)  (sequential f32 sum of per-codeword f32 norms, as ms_row_norm_kernel)
    lim           = f32((T (1 + 2^-17) + E) - |r|^2) * sigma^2                                     (ms_prep_kernel)
    acc           = f32(-lim) + sum_k binary16 x binary16 products accumulated in f32              (the MFMA; any accumulation order)
    pass         <=>  acc <= -cn2        (round 5; round 4 started the accumulator at f32(cn2 - lim) and compared with zero)
    val           = f32(acc + cn2)       (the survivor's value: relative to the limit)
    E             = 1.05 [2^-9 1.02 |r| (|r| + sqrt T) + 2^-13 (|r|^2 + T)] + E_abs               (the header's slack)
    S             = clamp(rint(val * (s / sigma^2) + (T (1 + 2^-17) + E) s), 0, 65535),  s = 30000 / T    (the merge kernel's sum)

Checked: (i) SOUNDNESS -- every row with dist_ref <= T passes; (ii) the sum of every passing row satisfies |S - dist_ref * s| <= the
per-pair slack units ceil(E s 1.1 + 3) the merge kernel's cut carries; (iii) selectivity -- survivors per row with dist_ref <= T.
Pairs the pre-pass hands to the exact rescan (residual beyond binary16, slack above 5 % of T) are counted, not filtered.
The accumulation is done in two orders (ascending k, and pairwise tree) -- the bound must hold for any.

Round 5: `run_bound` does the same for the bound pass on the matrix cores (ms_bound_kernel): histogram of dist~ over the query's nearest
partition, T = (upper edge of the keff-th bin) + 1.1 E -- checked: T is never below the keff-th smallest reference distance.

    python scripts/sim/ms_filter_spec.py
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle  # noqa: E402

f16, f32, f64 = np.float16, np.float32, np.float64
MS_SE = f32(30000.0)
MS_SLACK_CAP = f32(1500.0)
lib = oracle.lib()


def P(a):
    return a.ctypes.data_as(C.c_void_p)


def ref_lut(rq, cb, m, d):
    lut = np.empty((m, 256), f32)
    lib.orc_build_lut_f32(C.c_int(0), P(rq), C.c_size_t(d), P(cb), C.c_size_t(m), C.c_uint32(8), P(lut))
    return lut


def ref_adc(lut, codes):
    """sequential-m f32 sum of table entries (pq/distance.rs:128-141)"""
    acc = np.zeros(codes.shape[0], f32)
    for mm in range(lut.shape[0]):
        acc = (acc + lut[mm, codes[:, mm]]).astype(f32)
    return acc


def prep(r, T, sigma, d):
    """ms_prep_kernel, lane-0 arithmetic in f32: returns (ok, lim * sigma^2, y, z', slack units, rh)"""
    r = r.astype(f32)
    n2 = f32(0.0)
    # lane-strided partial sums then a xor-shuffle tree: any order serves a bound; here: 64 strided partials, pairwise tree
    part = np.zeros(64, f32)
    for e in range(d):
        part[e % 64] = f32(part[e % 64] + f32(r[e] * r[e]))
    w = part.copy()
    o = 32
    while o > 0:
        w = (w + np.roll(w, -o)).astype(f32)      # every lane ends with the full sum (xor butterfly == this for the sum)
        o //= 2
    n2 = f32(w[0])
    vmax = f32(np.max(np.abs(r)))
    s = f32(MS_SE / T)
    rn = f32(np.sqrt(n2, dtype=f32) * f32(1.000001))
    st = f32(np.sqrt(T, dtype=f32) * f32(1.000001))
    sqd = f32(np.sqrt(f32(d), dtype=f32) + f32(1.0))
    e_abs = f32(f32(6.1035156e-5) * sqd * (f32(3.0) * rn + f32(2.0) * st) / sigma + f32(d) * f32(3.7252903e-9) / (sigma * sigma))
    E = f32(f32(1.05) * (f32(1.9921875e-3) * rn * (rn + st) + f32(1.2207031e-4) * (n2 + T)) + e_abs)
    eu = f32(E * s * f32(1.1) + f32(3.0))
    lim = f32(f32(T * f32(1.0000077) + E) - n2)
    sig2 = f32(sigma * sigma)
    ok = bool(T > 0 and np.isfinite(T) and s > 0 and np.isfinite(s) and np.isfinite(n2) and vmax * sigma < 60000.0 and eu <= MS_SLACK_CAP
              and np.isfinite(lim * sig2) and n2 * s < 1e30)
    rh = (r * sigma).astype(f32).astype(f16)
    return ok, f32(lim * sig2), f32(s / sig2), f32(f32(T * f32(1.0000077) + E) * s), int(np.ceil(eu)), rh, E


def mfma_acc(init, rows_h, q_h, order):
    """acc[row] = f32 init + sum_k f16 x f16 products (exact in f32), accumulated in f32 in the given order"""
    prod = rows_h.astype(f32) * q_h.astype(f32)[None, :]      # exact: 11-bit x 11-bit significands
    if order == "seq":
        acc = init.astype(f32).copy()
        for k in range(prod.shape[1]):
            acc = (acc + prod[:, k]).astype(f32)
        return acc
    # pairwise tree over k, then one add to the start value
    p = prod
    while p.shape[1] > 1:
        if p.shape[1] % 2:
            p = np.concatenate([p, np.zeros((p.shape[0], 1), f32)], axis=1)
        p = (p[:, 0::2] + p[:, 1::2]).astype(f32)
    return (init.astype(f32) + p[:, 0]).astype(f32)


def run(name, x, q, m, nlist, keff=100, nprobes=4, max_pairs=200, seed=0, verbose=True):
    n, d = x.shape
    sd = d // m
    cent, _, _, _ = oracle.kmeans_train(x[: min(n, nlist * 256)], nlist, max_iters=6, seed=1)
    part, _ = oracle.assign(x, cent)
    res = oracle.residual(x, cent, part)
    cb, _ = oracle.pq_train(res[: min(n, 65536)], m, max_iters=5, seed=2)
    cb = np.ascontiguousarray(np.asarray(cb, f32).reshape(m, 256, sd))
    codes = np.asarray(oracle.pq_encode(res, cb, "l2")).reshape(n, m)
    # index constants (mscan_prepare / ms_codebook_kernel / ms_row_norm_kernel)
    cbmax = float(np.max(np.abs(cb)))
    e = int(np.frexp(cbmax)[1])
    sigma = f32(np.ldexp(1.0, 13 - e))
    cbh = (cb * f32(-2.0) * sigma).astype(f32).astype(f16)      # [m][256][sd]
    cbn2 = np.zeros((m, 256), f32)
    for u in range(sd):
        cbn2 = (cbn2 + (cb[:, :, u] * cb[:, :, u]).astype(f32)).astype(f32)
    row_cn2 = np.zeros(n, f32)
    for mm in range(m):
        row_cn2 = (row_cn2 + cbn2[mm, codes[:, mm]]).astype(f32)
    row_cn2 = (row_cn2 * f32(sigma * sigma)).astype(f32)
    rows_h = cbh[np.arange(m)[None, :], codes].reshape(n, d)      # the gathered reconstruction, binary16

    probes, _ = oracle.find_partitions(q, cent, nprobes)
    rng = np.random.default_rng(seed)
    tot = dict(pairs=0, handed=0, must=0, violations=0, survivors=0, sum_violations=0, worst_sum_err=0.0, worst_margin=np.inf)
    qsel = rng.permutation(len(q))
    for qi in qsel:
        if tot["pairs"] >= max_pairs:
            break
        # the bound T: keff-th smallest reference distance in the NEAREST partition (what the bound pass upper-bounds)
        p0 = int(probes[qi, 0])
        rows0 = np.nonzero(part == p0)[0]
        if len(rows0) < keff:
            continue
        r0 = (q[qi] - cent[p0]).astype(f32)
        d0 = ref_adc(ref_lut(r0, cb, m, d), codes[rows0])
        T = f32(np.partition(d0, keff - 1)[keff - 1] * f32(1.0 + 1e-3))      # any upper bound of it is legal
        for pr in probes[qi]:
            rows = np.nonzero(part == int(pr))[0]
            if len(rows) == 0:
                continue
            r = (q[qi] - cent[int(pr)]).astype(f32)
            tot["pairs"] += 1
            ok, lim_s, y, zp, units, rh, E = prep(r, T, sigma, d)
            if not ok:
                tot["handed"] += 1
                continue
            dref = ref_adc(ref_lut(r, cb, m, d), codes[rows])
            init = np.full(len(rows), -lim_s, f32)
            for order in ("seq", "tree"):
                acc0 = mfma_acc(init, rows_h[rows], rh, order)
                passed = acc0 <= -row_cn2[rows]
                acc = (acc0 + row_cn2[rows]).astype(f32)      # the survivor's value (and, for the margin statistic, the distance to the limit)
                must = dref <= T
                tot["violations"] += int(np.sum(must & ~passed))
                if order == "seq":
                    tot["must"] += int(must.sum())
                    tot["survivors"] += int(passed.sum())
                    if must.any():
                        tot["worst_margin"] = min(tot["worst_margin"], float(np.min(-acc[must]) / (float(sigma) ** 2) / max(float(E), 1e-30)))
                S = np.clip(np.rint((acc[passed].astype(f64) * f64(y) + f64(zp)).astype(f32)), 0, 65535)
                s = f64(MS_SE) / f64(T)
                err = np.abs(S.astype(f64) - dref[passed].astype(f64) * s)
                if err.size:
                    tot["worst_sum_err"] = max(tot["worst_sum_err"], float(np.max(err / units)))
                    tot["sum_violations"] += int(np.sum(err > units))
    if verbose:
        print(f"{name}: d={d} M={m} sigma=2^{int(np.log2(sigma))} pairs={tot['pairs']} handed_to_rescan={tot['handed']} rows_with_dist<=T={tot['must']} "
              f"violations={tot['violations']} survivors={tot['survivors']} ({tot['survivors'] / max(tot['must'], 1):.2f} x) "
              f"smallest (limit - acc) / E among must-pass rows={tot['worst_margin']:.3f} "
              f"largest |S - dist s| / slack units={tot['worst_sum_err']:.3f} sum_violations={tot['sum_violations']}")
    return tot


def bound_T(r, rows_h, cn2_rows, sigma, cb_mean, nu, d, keff, order="seq"):
    """ms_bound_kernel for one (query, nearest partition): returns (T, Ta, E) or None when the pass gives the query no bound.
    r: f32 residual; rows_h: [rows][d] binary16 reconstruction operand (-2 sigma c); cn2_rows: f32 sigma^2 |c^|^2 per row;
    cb_mean: [d] mean codeword of every sub-quantiser dimension; nu: sum over m of the mean |c|^2 (lance_hip_index::cb_mean)."""
    r = r.astype(f32)
    n2 = f32(0.0)
    rmu = f32(0.0)
    for e in range(d):      # (the kernel sums lane pairs then a xor tree: any order serves -- the scale only sets tightness, n2 enters E through its own slack)
        n2 = f32(n2 + f32(r[e] * r[e]))
        rmu = f32(rmu + f32(r[e] * cb_mean[e]))
    vmax = f32(np.max(np.abs(r)))
    mean = f32(f32(n2 - f32(2.0) * rmu) + nu)
    if not (np.isfinite(n2) and vmax * sigma < 60000.0 and mean > 0 and np.isfinite(mean)):
        return None
    sb = f32(f32(496.0) / mean)
    sig2 = f32(sigma * sigma)
    a = f32(sb / sig2)
    b = f32(n2 * sb)
    if not (sb > 0 and np.isfinite(sb) and a > 0 and np.isfinite(a) and n2 * sb < 1e30):
        return None
    rh = (r * sigma).astype(f32).astype(f16)
    acc = mfma_acc(np.zeros(rows_h.shape[0], f32), rows_h, rh, order)
    t = ((acc + cn2_rows).astype(f32).astype(f64) * f64(a) + f64(b)).astype(f32)      # one FMA: a single rounding of the exact product-sum
    inb = t < f32(512.0)
    bins = np.maximum(t[inb].astype(np.int64), 0)      # (int) truncation; negative values land in bin 0
    hist = np.bincount(bins, minlength=512)
    cum = np.cumsum(hist)
    hit = np.nonzero(cum >= keff)[0]
    if hit.size == 0:
        return None
    binb = int(hit[0])
    Ta = f32(f32(f32(binb + 1) / sb) * f32(1.000001))
    rn = f32(np.sqrt(n2, dtype=f32) * f32(1.000001))
    st = f32(np.sqrt(Ta, dtype=f32) * f32(1.000001))
    sqd = f32(np.sqrt(f32(d), dtype=f32) + f32(1.0))
    e_abs = f32(f32(6.1035156e-5) * sqd * (f32(3.0) * rn + f32(2.0) * st) / sigma + f32(d) * f32(3.7252903e-9) / (sigma * sigma))
    E = f32(f32(1.05) * (f32(1.9921875e-3) * rn * (rn + st) + f32(1.2207031e-4) * (n2 + Ta)) + e_abs)
    T = f32(f32(Ta + f32(1.1) * E) * f32(1.0000153))
    return T, Ta, E


def run_bound(name, x, q, m, nlist, keff=100, max_queries=150, seed=0, verbose=True):
    """The bound pass on the matrix cores (ms_bound_kernel): T must be >= the keff-th smallest REFERENCE distance of the query's nearest
    partition (then it bounds the final keff-th distance over all probed partitions too); reports how loose it is."""
    n, d = x.shape
    sd = d // m
    cent, _, _, _ = oracle.kmeans_train(x[: min(n, nlist * 256)], nlist, max_iters=6, seed=1)
    part, _ = oracle.assign(x, cent)
    res = oracle.residual(x, cent, part)
    cb, _ = oracle.pq_train(res[: min(n, 65536)], m, max_iters=5, seed=2)
    cb = np.ascontiguousarray(np.asarray(cb, f32).reshape(m, 256, sd))
    codes = np.asarray(oracle.pq_encode(res, cb, "l2")).reshape(n, m)
    cbmax = float(np.max(np.abs(cb)))
    sigma = f32(np.ldexp(1.0, 13 - int(np.frexp(cbmax)[1])))
    cbh = (cb * f32(-2.0) * sigma).astype(f32).astype(f16)
    cbn2 = np.zeros((m, 256), f32)
    for u in range(sd):
        cbn2 = (cbn2 + (cb[:, :, u] * cb[:, :, u]).astype(f32)).astype(f32)
    row_cn2 = np.zeros(n, f32)
    for mm in range(m):
        row_cn2 = (row_cn2 + cbn2[mm, codes[:, mm]]).astype(f32)
    row_cn2 = (row_cn2 * f32(sigma * sigma)).astype(f32)
    rows_h = cbh[np.arange(m)[None, :], codes].reshape(n, d)
    cb_mean = cb.mean(axis=1).astype(f32).reshape(d)                    # q_codebook_mean_kernel: mean codeword per sub-quantiser dimension
    nu = f32(np.sum((cb.astype(f64) ** 2).sum(axis=2).mean(axis=1)))   # sum over m of the mean |c|^2
    probes, _ = oracle.find_partitions(q, cent, 1)
    rng = np.random.default_rng(seed)
    tot = dict(queries=0, no_bound=0, violations=0, worst_ratio=0.0, mean_ratio=0.0)
    for qi in rng.permutation(len(q))[:max_queries]:
        p0 = int(probes[qi, 0])
        rows = np.nonzero(part == p0)[0]
        if len(rows) < keff:
            continue
        r = (q[qi] - cent[p0]).astype(f32)
        dref = ref_adc(ref_lut(r, cb, m, d), codes[rows])
        true_k = float(np.partition(dref, keff - 1)[keff - 1])
        tot["queries"] += 1
        for order in ("seq", "tree"):
            out = bound_T(r, rows_h[rows], row_cn2[rows], sigma, cb_mean, nu, d, keff, order)
            if out is None:
                tot["no_bound"] += order == "seq"
                continue
            T = float(out[0])
            tot["violations"] += int(not (T >= true_k))
            if order == "seq" and true_k > 0:
                tot["worst_ratio"] = max(tot["worst_ratio"], T / true_k)
                tot["mean_ratio"] += T / true_k
    if tot["queries"] > tot["no_bound"]:
        tot["mean_ratio"] /= (tot["queries"] - tot["no_bound"])
    if verbose:
        print(f"{name} [bound pass]: queries={tot['queries']} without a bound={tot['no_bound']} T below the true keff-th distance={tot['violations']} "
              f"T / true: mean {tot['mean_ratio']:.4f} worst {tot['worst_ratio']:.4f}")
    return tot


# ---- dot metric (round 6) -------------------------------------------------------------------------------------------------------------
# dist = 1 - q . c^ (pq/distance.rs:60-92, pq/storage.rs:949-957: sum of M entries 1 - q_m . c_m, minus M - 1; no residual).  The kernels
# evaluate (1 - q . mu) - q . c' with c' = c^ - mu the reconstruction from the CENTRED codebook (mu: mean codeword of every sub-quantiser):
#     cbh  = binary16(-2 sigma (c - mu))     rh = binary16(sigma q / 2)     row term = 0
#     G    = |q| cmax  (cmax >= |c'| of every stored row),   Gf = |q| cmax_full  (cmax_full >= |c^| of every stored row)
#     E    = 1.05 [2^-10 1.02 G + 2^-16 (G + 1 + |T| + |q . mu| + Gf) + 2^-24 (M + 2)^2] + E_abs
#     lim  = ((T - 1) + q . mu) + E        pass <=> f32(-lim sigma^2) + sum of products <= -0
#     S    = clamp(rint(val * (s / sigma^2) + (lim + G) s), 0, 65535),   s = 30000 / (((T - 1) + q . mu + G) + E)     (sum of dist - base, base = 1 - q . mu - G)
# bound pass: bins of (acc / sigma^2 + G) * 496 / G; T = ((1 - q . mu) - G + Ta) + 1.1 E + 2^-16 (|1 - q . mu| + G + Ta).


def ref_lut_dot(qv, cb, m, d):
    lut = np.empty((m, 256), f32)
    lib.orc_build_lut_f32(C.c_int(2), P(qv), C.c_size_t(d), P(cb), C.c_size_t(m), C.c_uint32(8), P(lut))
    return lut


def ref_adc_dot(lut, codes):
    acc = ref_adc(lut, codes)
    return (acc - f32(lut.shape[0] - 1)).astype(f32)


def _dot_E(G, Gf, tmag, qmu, qn, cmax, sigma, d, m):
    sqd = f32(np.sqrt(f32(d), dtype=f32) + f32(1.0))
    e_abs = f32(f32(6.1035156e-5) * sqd * (qn + f32(2.0) * cmax) / sigma + f32(d) * f32(3.7252903e-9) / (sigma * sigma))
    return f32(f32(1.05) * (f32(9.9609375e-4) * G + f32(1.5258789e-5) * (G + f32(1.0) + tmag + abs(qmu) + Gf) + f32(5.9604645e-8) * f32((m + 2) * (m + 2))) + e_abs)


def run_dot(name, x, q, m, nlist, keff=100, nprobes=4, max_pairs=200, seed=0, verbose=True):
    """Filter AND bound pass of the dot flow: (i) the bound T from the nearest partition's histogram is >= its keff-th smallest reference
    distance; (ii) with that T no row of any probed partition whose reference distance is <= T fails the test; (iii) a passing row's integer
    sum is within the slack units of (dist_ref - base) s.  Both accumulation orders."""
    n, d = x.shape
    sd = d // m
    cent, _, _, _ = oracle.kmeans_train(x[: min(n, nlist * 256)], nlist, max_iters=6, seed=1, metric="dot")
    part, _ = oracle.assign(x, cent, "dot")
    cb, _ = oracle.pq_train(x[: min(n, 65536)], m, max_iters=5, seed=2)
    cb = np.ascontiguousarray(np.asarray(cb, f32).reshape(m, 256, sd))
    codes = np.asarray(oracle.pq_encode(x, cb, "dot")).reshape(n, m)
    mu3 = np.zeros((m, sd), f32)                    # q_codebook_mean_kernel: f32 sums of 256 codewords, times 1 / 256
    for c in range(256):
        mu3 = (mu3 + cb[:, c, :]).astype(f32)
    mu3 = (mu3 * f32(1.0 / 256.0)).astype(f32)
    mu = mu3.reshape(d)
    cbc = (cb - mu3[:, None, :]).astype(f32)        # the centring in f32 (ms_codebook_kernel)
    cbmax = float(np.max(np.abs(cbc)))
    sigma = f32(np.ldexp(1.0, 13 - int(np.frexp(cbmax)[1])))
    cbh = (cbc * f32(-2.0) * sigma).astype(f32).astype(f16)
    rows_h = cbh[np.arange(m)[None, :], codes].reshape(n, d)
    recc = cbc[np.arange(m)[None, :], codes].reshape(n, d).astype(f64)
    cmax = f32(np.sqrt((recc ** 2).sum(axis=1).max()) * 1.0001)          # (the kernel: f32 row norms, max, sqrt, x 1.0001)
    cmax_full = f32(np.sqrt(sum(float((cb[mm].astype(f64) ** 2).sum(axis=1).max()) for mm in range(m))) * 1.00001)
    sig2 = f32(sigma * sigma)
    probes, _ = oracle.find_partitions(q, cent, nprobes, "dot")
    rng = np.random.default_rng(seed)
    tot = dict(pairs=0, handed=0, must=0, violations=0, survivors=0, sum_violations=0, worst_sum_err=0.0, queries=0, no_bound=0,
               bound_violations=0, mean_excess=0.0)
    for qi in rng.permutation(len(q)):
        if tot["pairs"] >= max_pairs:
            break
        qv = q[qi].astype(f32)
        n2 = f32(0.0); qmu = f32(0.0)
        for e in range(d):
            n2 = f32(n2 + f32(qv[e] * qv[e])); qmu = f32(qmu + f32(qv[e] * mu[e]))
        qn = f32(np.sqrt(n2, dtype=f32) * f32(1.000001))
        G = f32(qn * cmax * f32(1.000001)); Gf = f32(qn * cmax_full * f32(1.000001))
        rh = (qv * f32(0.5) * sigma).astype(f32).astype(f16)
        lut = ref_lut_dot(qv, cb, m, d)
        p0 = int(probes[qi, 0])
        rows0 = np.nonzero(part == p0)[0]
        if len(rows0) < keff or not G > 0:
            continue
        tot["queries"] += 1
        dref0 = ref_adc_dot(lut, codes[rows0])
        true_k = float(np.partition(dref0, keff - 1)[keff - 1])
        Ts = []
        for order in ("seq", "tree"):      # ---- bound pass
            sb = f32(f32(496.0) / G)
            a = f32(sb / sig2); b = f32(G * sb)
            acc = mfma_acc(np.zeros(len(rows0), f32), rows_h[rows0], rh, order)
            t = (acc.astype(f64) * f64(a) + f64(b)).astype(f32)
            inb = t < f32(512.0)
            hist = np.bincount(np.maximum(t[inb].astype(np.int64), 0), minlength=512)
            hit = np.nonzero(np.cumsum(hist) >= keff)[0]
            if hit.size == 0:
                Ts.append(None); continue
            Ta = f32(f32(f32(int(hit[0]) + 1) / sb) * f32(1.000001))
            tmag = f32(abs(f32(1.0) - qmu) + G + Ta)
            E = _dot_E(G, Gf, tmag, qmu, qn, cmax, sigma, d, m)
            T = f32(f32(f32(f32(1.0) - qmu) - G) + Ta) + f32(f32(1.1) * E + f32(1.5258789e-5) * tmag)
            T = f32(T)
            tot["bound_violations"] += int(not (float(T) >= true_k))
            Ts.append(T)
        if Ts[0] is None:
            tot["no_bound"] += 1
            continue
        T = Ts[0]
        tot["mean_excess"] += (float(T) - true_k) / max(float(G), 1e-30)
        # ---- filter with that T over every probed partition
        E = _dot_E(G, Gf, abs(T), qmu, qn, cmax, sigma, d, m)
        Tq = f32(f32(T - f32(1.0)) + qmu)
        Tp = f32(f32(Tq + G) + E)
        s = f32(MS_SE / Tp)
        eu = f32(E * s * f32(1.1) + f32(3.0))
        lim = f32(Tq + E)
        zsum = f32(f32(lim + G) * s)
        vmax = f32(np.max(np.abs(qv)))
        ok = bool(np.isfinite(T) and Tp > 0 and np.isfinite(Tp) and s > 0 and np.isfinite(s) and np.isfinite(Gf) and vmax * sigma < 60000.0
                  and eu <= MS_SLACK_CAP and np.isfinite(lim * sig2) and G * s < 1e30 and abs(zsum) < 1e30)
        for pr in probes[qi]:
            rows = np.nonzero(part == int(pr))[0]
            if len(rows) == 0:
                continue
            tot["pairs"] += 1
            if not ok:
                tot["handed"] += 1
                continue
            dref = ref_adc_dot(lut, codes[rows])
            must = dref <= T
            units = int(np.ceil(eu))
            base = f64(1.0) - f64(qmu) - f64(G)
            for order in ("seq", "tree"):
                acc = mfma_acc(np.full(len(rows), -f32(lim * sig2), f32), rows_h[rows], rh, order)
                passed = acc <= f32(-0.0)
                tot["violations"] += int(np.sum(must & ~passed))
                if order == "seq":
                    tot["must"] += int(must.sum()); tot["survivors"] += int(passed.sum())
                S = np.clip(np.rint((acc[passed].astype(f64) * f64(f32(s / sig2)) + f64(zsum)).astype(f32)), 0, 65535)
                err = np.abs(S.astype(f64) - np.maximum((dref[passed].astype(f64) - base) * f64(s), 0.0))
                if err.size:
                    tot["worst_sum_err"] = max(tot["worst_sum_err"], float(np.max(err / units)))
                    tot["sum_violations"] += int(np.sum(err > units))
    if tot["queries"] > tot["no_bound"]:
        tot["mean_excess"] /= (tot["queries"] - tot["no_bound"])
    if verbose:
        print(f"{name} [dot]: d={d} M={m} sigma=2^{int(np.log2(sigma))} cmax={float(cmax):.4g} (full {float(cmax_full):.4g}) queries={tot['queries']} "
              f"without a bound={tot['no_bound']} T below the true keff-th distance={tot['bound_violations']} mean (T - true) / G={tot['mean_excess']:.5f} | "
              f"pairs={tot['pairs']} handed_to_rescan={tot['handed']} rows_with_dist<=T={tot['must']} violations={tot['violations']} "
              f"survivors={tot['survivors']} ({tot['survivors'] / max(tot['must'], 1):.2f} x) largest |S - (dist - base) s| / slack units="
              f"{tot['worst_sum_err']:.3f} sum_violations={tot['sum_violations']}")
    return tot


def sift_like(n, d, seed):
    rng = np.random.default_rng(seed)
    centers = rng.uniform(0, 128, (64, d))
    x = centers[rng.integers(0, 64, n)] + rng.normal(0, 22, (n, d))
    return np.clip(np.rint(x), 0, 218).astype(f32)


def main():
    x = sift_like(40000, 128, 1)
    q = sift_like(300, 128, 2)
    run("sift-like integer rows", x, q, 16, 16)
    run("sift-like, M=32 (sub-dimension 4)", x, q, 32, 16)
    rng = np.random.default_rng(3)
    xu = rng.standard_normal((40000, 128)).astype(f32)
    xu /= np.linalg.norm(xu, axis=1, keepdims=True)
    qu = rng.standard_normal((300, 128)).astype(f32)
    qu /= np.linalg.norm(qu, axis=1, keepdims=True)
    run("unit vectors (sigma = 2^14)", xu.astype(f32), qu.astype(f32), 16, 16)
    run("rows far from the origin (|r|^2 >> T for far probes)", x + f32(3000.0), q + f32(3000.0), 16, 16)
    x64 = sift_like(30000, 64, 5)
    run("d=64 M=16", x64, sift_like(200, 64, 6), 16, 12)
    run_bound("sift-like integer rows", x, q, 16, 16)
    run_bound("sift-like, M=32", x, q, 32, 16)
    run_bound("unit vectors", xu.astype(f32), qu.astype(f32), 16, 16)
    run_bound("rows far from the origin", x + f32(3000.0), q + f32(3000.0), 16, 16)
    run_bound("d=64 M=16", x64, sift_like(200, 64, 6), 16, 12, keff=10)
    run_dot("sift-like integer rows", x, q, 16, 16)
    run_dot("sift-like, M=32", x, q, 32, 16)
    run_dot("unit vectors", xu.astype(f32), qu.astype(f32), 16, 16)
    run_dot("centred rows (dot products of both signs)", x - f32(64.0), q - f32(64.0), 16, 16)
    run_dot("d=64 M=16", x64, sift_like(200, 64, 6), 16, 12, keff=10)


if __name__ == "__main__":
    main()
