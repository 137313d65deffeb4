"""world_size-2 tests of the multi-GPU k-means host loop (lance_amd/dist.py) on CPU with gloo.

The device steps (estep_partial / finalize) are replaced by a stand-in engine built on the
oracle -- tests may use the oracle, the product may not -- so what is exercised here is the
collective logic: sharding, the fused all-reduce buffers, identical convergence decisions and
shared-seed empty-cluster splits on every rank.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f32 = np.float32


class OracleEngine:
    """CPU stand-in with the Engine.kmeans_estep_partial / kmeans_finalize contract."""

    def kmeans_estep_partial(self, x, centroids, metric="l2", bias=None):
        import oracle
        x = np.asarray(x, f32); c = np.asarray(centroids, f32)
        k, d = c.shape
        ids, dists = oracle.assign(x, c, metric, None if bias is None else np.asarray(bias, f32))
        sums = np.zeros((k, d), f32); counts = np.zeros(k, f32)
        losses = np.zeros(k, np.float64); radius = np.zeros(k, f32)
        for r in range(x.shape[0]):
            i = ids[r]
            if i == oracle.NONE:
                continue
            sums[i] = (sums[i] + x[r]).astype(f32)
            counts[i] += 1
            losses[i] += np.float64(dists[r])
            radius[i] = max(radius[i], dists[r])
        buf = torch.from_numpy(np.concatenate([sums.ravel(), counts]))
        return buf, torch.from_numpy(losses), torch.from_numpy(radius)

    def kmeans_finalize(self, buf, k, d):
        b = buf.numpy()
        sums = b[: k * d].reshape(k, d).copy(); counts = b[k * d:]
        for c in range(k):
            if counts[c] > 0:
                sums[c] = (sums[c] * (f32(1.0) / f32(counts[c]))).astype(f32)
        return torch.from_numpy(sums)


def _data(case):
    rng = np.random.default_rng(17)
    if case == "blobs":
        x = rng.standard_normal((2001, 16)).astype(f32)
        x[:700] += 5; x[700:1300] -= 4
        return x, 8, 0.0
    if case == "balanced":
        x = rng.standard_normal((1500, 8)).astype(f32) * 3
        return x, 6, 1.0
    base = rng.standard_normal((4, 8)).astype(f32)          # duplicates -> empty clusters -> split
    x = base[rng.integers(0, 4, 1200)]
    x[:30] += rng.standard_normal((30, 8)).astype(f32) * 0.01
    return x, 9, 0.0


def _worker(rank, world, port, case, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lance_amd.dist import train_kmeans_sharded
    x, k, bf = _data(case)
    n = x.shape[0]
    per = (n + world - 1) // world
    xl = torch.from_numpy(x[rank * per: min(n, (rank + 1) * per)])
    init = x[:k].copy()
    cent, loss, iters = train_kmeans_sharded(OracleEngine(), xl, k, n, max_iters=15, balance_factor=bf, init=init, seed=5)
    cent2, _, _ = train_kmeans_sharded(OracleEngine(), xl, k, n, max_iters=3, init=None, seed=9)  # rank-0 draw + broadcast
    q.put((rank, cent.numpy(), loss, iters, cent2.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", ["blobs", "balanced", "dups"])
def test_sharded_kmeans_two_ranks(case):
    import oracle
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + {"blobs": 0, "balanced": 1, "dups": 2}[case]
    procs = [ctx.Process(target=_worker, args=(r, 2, port, case, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, c0, l0, i0, d0), (_, c1, l1, i1, d1) = res
    # every rank holds the same model, bit for bit, and took the same decisions
    assert (c0.view(np.uint32) == c1.view(np.uint32)).all() and l0 == l1 and i0 == i1
    assert (d0.view(np.uint32) == d1.view(np.uint32)).all()
    # and it agrees with the single-process reference loop up to f32 summation order
    x, k, bf = _data(case)
    oc, ol, oit, _ = oracle.kmeans_train(x, k, max_iters=15, balance_factor=f32(bf) / f32(x.shape[0]), init=x[:k].copy(), seed=5)
    if case != "dups":          # split perturbations amplify round-off; only the structure is compared there
        assert i0 == oit
        assert np.allclose(c0, oc, rtol=1e-4, atol=1e-4)
        assert abs(l0 - ol) <= 1e-5 * abs(ol)
    else:
        assert np.isfinite(c0).all() and i0 >= 1


# ---- search with IVF lists sharded over the ranks (lance_amd/dist.py: search_list_sharded) -----------------
def _shard_case():
    import oracle
    rng = np.random.default_rng(23)
    n, d, nlist, m = 6000, 32, 12, 8
    centers = rng.integers(0, 120, (30, d))
    x = np.clip(np.rint(centers[rng.integers(0, 30, n)] + rng.normal(0, 14, (n, d))), 0, 200).astype(f32)   # integer-valued: ties
    x[40:48] = x[3]
    q = np.clip(np.rint(centers[rng.integers(0, 30, 40)] + rng.normal(0, 14, (40, d))), 0, 200).astype(f32)
    cent, _, _, _ = oracle.kmeans_train(x[:3000], nlist, max_iters=6, seed=4)
    part, _ = oracle.assign(x, cent, "l2")
    cb, _ = oracle.pq_train(oracle.residual(x[:3000], cent, part[:3000]), m, max_iters=6, seed=6)
    return x, q, cent, cb, part


def _search_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from lance_amd.dist import local_list_rows, search_list_sharded
    x, q, cent, cb, part = _shard_case()
    rows = local_list_rows(part.astype(np.int64), world, rank).numpy()
    xl = x[rows]
    oidx = oracle.build_index(xl, cent, cb, metric="l2")          # local row ids = 0..n_local-1, ascending with the global ids

    def local_search(qq, kk, nprobes, rf):
        ids, dd = oidx.search(np.asarray(qq, f32), kk, nprobes, refine=rf, raw=xl if rf else None)
        ids = ids.astype(np.int64)                                # UINT64_MAX (none) -> -1
        return torch.from_numpy(ids), torch.from_numpy(dd)

    res = {}
    for k, nprobes, rf in ((10, 3, 0), (10, 12, 0), (5, 4, 4), (10, 12, 3)):
        gi, gd = search_list_sharded(local_search, torch.from_numpy(rows), torch.from_numpy(q), k, nprobes, rf)
        res[(k, nprobes, rf)] = (gi.numpy(), gd.numpy())
    out.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_list_sharded_search_two_ranks_equals_single_index():
    import oracle
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_search_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=180) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    x, q, cent, cb, part = _shard_case()
    full = oracle.build_index(x, cent, cb, metric="l2")
    for key, (gi0, gd0) in res[0][1].items():
        gi1, gd1 = res[1][1][key]
        assert (gi0 == gi1).all() and (gd0.view(np.uint32) == gd1.view(np.uint32)).all()      # every rank holds the same answer
        k, nprobes, rf = key
        oi, od = full.search(q, k, nprobes, refine=rf, raw=x if rf else None)
        assert (gi0.astype(np.uint64) == oi).all(), key
        assert (gd0.view(np.uint32) == od.view(np.uint32)).all(), key
