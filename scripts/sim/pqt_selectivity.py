"""CPU simulation (numpy) of the round-4 lever named in DESIGN.md section 8: a per-QUERY table + a per-ROW bias instead of a
table per (query, partition).

    ||(q - cen_p)_m - c||^2 = ||q_m - c||^2 + 2 cen_pm . c + (||cen_pm||^2 - 2 cen_pm . q_m)
    dist(q, row) = sum_m A_q[m][code_m] + beta_row + kappa_qp,      beta_row = sum_m 2 cen_pm . c_m(code),   kappa_qp = ||cen_p||^2 - 2 cen_p . q

Checks (1) the identity against the residual tables in f64, (2) what the filter `sum_m floor(A_q * s_q) <= s_q * (T - kappa - beta_row)`
lets through compared with the per-pair integer table (full-range u16 entries, 32-bit sums: the tiled kernels' encoding) and with
the exact count, (3) how many integer levels T is worth under the per-query scale, (4) the rounding slack
gamma * s_q * (||q||^2 + ||cen_p||^2 + sum_m max_c ||c||^2) in units.

    python scripts/sim/pqt_selectivity.py sift     # C2-like: 128-d SIFT-like rows, M = 16
    python scripts/sim/pqt_selectivity.py unit     # C3-like at reduced d: normalised 384-d rows, M = 24 (sub-dimension 16)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle  # noqa: E402

f32, f64 = np.float32, np.float64
kind = sys.argv[1] if len(sys.argv) > 1 else "unit"
rng = np.random.default_rng(3)
if kind == "sift":
    import torch
    from lance_amd.testing import sift_like
    n, d, m, nlist = 120_000, 128, 16, 30
    x = sift_like(n, d, seed=1, device=torch.device("cpu")).numpy().astype(f32)
    q = sift_like(64, d, seed=2, device=torch.device("cpu")).numpy().astype(f32)
else:
    n, d, m, nlist = 60_000, 384, 24, 60
    centers = rng.standard_normal((256, d)).astype(f32)
    x = centers[rng.integers(0, 256, n)] + rng.standard_normal((n, d), dtype=f32) * f32(0.5)
    x = (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(f32)
    q = centers[rng.integers(0, 256, 64)] + rng.standard_normal((64, d), dtype=f32) * f32(0.5)
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(f32)
sd = d // m
cent, _, _, _ = oracle.kmeans_train(x[: nlist * 256], nlist, max_iters=10, seed=1)
part, _ = oracle.assign(x, cent)
res = oracle.residual(x, cent, part)
cb, _ = oracle.pq_train(res[:65536], m, max_iters=8, seed=2)         # [m][256][sd]
codes = oracle.pq_encode(res, cb)
cb = np.asarray(cb, f64).reshape(m, 256, sd)
cent64 = cent.astype(f64)
ar = np.arange(m)[None, :]
# index constants: beta per row, its minimum per partition, sum over m of the largest ||c||^2
cen_sub = cent64.reshape(nlist, m, 1, sd)
B = 2.0 * (cen_sub * cb[None]).sum(-1)                                   # [nlist][m][256]
beta = B[part[:, None], ar, codes].sum(1)                                # [n]
beta_min = np.array([beta[part == p].min() if (part == p).any() else 0.0 for p in range(nlist)])
cmax = (cb ** 2).sum(-1).max(1).sum()
print(f"{kind}: n={n} d={d} M={m} sd={sd} nlist={nlist} rows/partition={n / nlist:.0f}")
keff, nprobes, SE = 100, 10, 61440
gamma = 2.2 * (sd + 2) * 2.0 ** -24
tot, worst_id, levels, slack = {}, 0.0, [], []


def count(name, v):
    tot[name] = tot.get(name, 0) + int(v)


for qi in range(q.shape[0]):
    q64 = q[qi].astype(f64)
    pi, _ = oracle.find_partitions(q[qi:qi + 1], cent, nprobes)
    A = ((q64.reshape(m, 1, sd) - cb) ** 2).sum(-1)                      # per-QUERY table [m][256]
    T, theta, per = None, 0.0, []
    for rank, p in enumerate(pi[0]):
        rows = np.nonzero(part == p)[0]
        if rows.size == 0:
            continue
        rq = q64 - cent64[p]
        lut = ((rq.reshape(m, 1, sd) - cb) ** 2).sum(-1)                 # per-(query, partition) table
        c = codes[rows]
        dist = lut[ar, c].sum(1)
        kappa = (cent64[p] ** 2).sum() - 2.0 * (cent64[p] * q64).sum()
        alt = A[ar, c].sum(1) + beta[rows] + kappa
        worst_id = max(worst_id, float(np.abs(alt - dist).max() / max(dist.max(), 1e-30)))
        if rank == 0:
            if rows.size < keff:
                break
            T = np.partition(dist, keff - 1)[keff - 1] * 1.03
        theta = max(theta, T - kappa - beta_min[p])
        per.append((p, rows, c, dist, lut, kappa))
    if T is None:
        continue
    s_q = SE / theta                                                     # one scale per query: every passing row's entries fit
    levels.append(T * s_q)
    for p, rows, c, dist, lut, kappa in per:
        count("rows scanned", rows.size)
        count("exact: dist <= T", (dist <= T).sum())
        e = np.minimum(np.floor(lut * (SE / T)), 65535)
        count("per-pair table (u16 full range, today's tiled kernels)", (e[ar, c].sum(1) <= SE + 4).sum())
        eq = np.minimum(np.floor(A * s_q), 65535)
        lim = np.floor(s_q * (T - kappa - beta[rows]))
        sl = gamma * s_q * ((q64 ** 2).sum() + (cent64[p] ** 2).sum() + cmax)
        slack.append(sl)
        count("per-query table + row bias", (eq[ar, c].sum(1) <= lim + 4 + np.ceil(sl)).sum())
nq = q.shape[0]
ex = tot["exact: dist <= T"] / nq
for k, v in tot.items():
    print(f"{k:58s} {v / nq:10.1f} per query   x{v / nq / ex:6.2f}")
print(f"identity: max |A + beta + kappa - dist| / max dist = {worst_id:.2e} (f64)")
print(f"integer levels per T under the per-query scale: median {np.median(levels):.0f}, min {np.min(levels):.0f}  (per-pair scale: {SE})")
print(f"rounding slack: median {np.median(slack):.2f} units, max {np.max(slack):.2f}")
