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

    def kmeans_split(self, x, rows, k, max_iters=50, tol=1e-4, balance_factor_scaled=0.0, seed=0, metric="l2"):
        import oracle
        xn = x.numpy() if isinstance(x, torch.Tensor) else np.asarray(x)
        return oracle.kmeans_split(np.ascontiguousarray(xn), rows, k, max_iters=max_iters, tol=tol, balance_factor_scaled=balance_factor_scaled,
                                   seed=seed, metric=metric)

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


# ---- block gathers of the multi-GPU build (model-parallel PQ codebook slices, row-sharded transform output) ----------
def _gather_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lance_amd.dist import all_gather_blocks, block_ranges
    res = {}
    for total in (16, 7, 2, 1000, 1001, world - 1 if world > 1 else 1):      # even, uneven, fewer items than ranks
        _, ranges = block_ranges(total, world)
        lo, hi = ranges[rank]
        full2 = torch.arange(total * 3, dtype=torch.float32).reshape(total, 3) * 0.5
        full1 = (torch.arange(total, dtype=torch.int32) * 7) % 11
        res[total] = (all_gather_blocks(full2[lo:hi].clone(), total).numpy(), all_gather_blocks(full1[lo:hi].clone(), total).numpy())
    out.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2, 3])
def test_all_gather_blocks_reassembles_the_array(world):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 33500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = [out.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, r in res:
        for total, (a2, a1) in r.items():
            assert (a2 == (np.arange(total * 3, dtype=f32).reshape(total, 3) * 0.5)).all(), (world, rank, total)
            assert (a1 == (np.arange(total, dtype=np.int32) * 7) % 11).all()


def test_block_ranges_cover_everything():
    from lance_amd.dist import block_ranges
    for total in (0, 1, 5, 16, 1_000_000):
        for world in (1, 2, 3, 8):
            per, ranges = block_ranges(total, world)
            assert ranges[0][0] == 0 and ranges[-1][1] == total and len(ranges) == world
            assert all(a <= b and b - a <= per for a, b in ranges)
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))


# ---- the whole multi-rank build on CPU: create_index_sharded with an oracle-backed stand-in engine -----------------
class OracleBuildEngine:
    """Engine stand-in for create_index_sharded (CPU tensors in, CPU tensors out), every step computed by the oracle."""

    def _np(self, t):
        return t.numpy() if isinstance(t, torch.Tensor) else np.asarray(t)

    def normalize(self, x):
        import oracle
        return torch.from_numpy(oracle.normalize(self._np(x)))

    def kmeans_train(self, x, k, max_iters=50, tol=1e-4, balance_factor=0.0, init=None, seed=0, metric="l2", hierarchical_k=16):
        import oracle
        xn = self._np(x)
        n = xn.shape[0]
        c, loss, iters, _ = oracle.kmeans_train(xn[: min(n, k * 512)], k, max_iters=max_iters, tol=tol,
                                                balance_factor=f32(balance_factor) / f32(n), seed=seed, metric=metric)
        return torch.from_numpy(c), loss, iters

    def assign(self, x, cent, metric="l2", bias=None):
        import oracle
        ids, d = oracle.assign(self._np(x), self._np(cent), metric)
        return torch.from_numpy(ids.view(np.int32).copy()), torch.from_numpy(d)

    def residual(self, x, cent, part):
        import oracle
        return torch.from_numpy(oracle.residual(self._np(x), self._np(cent), self._np(part).view(np.uint32)))

    def pq_train(self, r, m, nbits=8, max_iters=50, sample_rate=256, seed=0):
        import oracle
        cb, it = oracle.pq_train(np.ascontiguousarray(self._np(r)), m, nbits=nbits, max_iters=max_iters, sample_rate=sample_rate, seed=seed)
        return torch.from_numpy(cb), it.astype(np.uint32)

    def ivfpq_encode(self, x, cent, cb, metric="l2", want_loss=True):
        import oracle
        xn = np.ascontiguousarray(self._np(x))
        nb = 4 if cb.shape[1] == 16 else 8
        oi = oracle.build_index(xn, self._np(cent), self._np(cb), metric, nbits=nb)
        return torch.from_numpy(oi.part_ids.view(np.int32).copy()), torch.from_numpy(oi.codes_rowmajor.copy()), 0.0


def _build_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lance_amd.dist import create_index_sharded
    rng = np.random.default_rng(5)
    res = {}
    for metric, n, nbits in (("l2", 3001, 8), ("dot", 2000, 8), ("cosine", 2500, 4)):
        x = torch.from_numpy((rng.standard_normal((n, 24)) * 2 + 1).astype(f32))
        ix = create_index_sharded(x, metric=metric, num_partitions=6, num_sub_vectors=6, num_bits=nbits, max_iters=6, sample_rate=64,
                                  seed=11, engine=OracleBuildEngine(), index_factory=lambda *a, **k: None)
        res[metric] = (ix.part_ids.numpy(), ix.codes.numpy(), ix.stats.ivf_training, np.asarray(ix.stats.pq_iters))
    out.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_create_index_sharded_is_independent_of_the_rank_count():
    """Replicated IVF training + model-parallel PQ (6 sub-quantisers over 1 / 2 / 3 ranks, uneven for none of them but the
    row split of 3001 rows is) + row-sharded transform: every rank of every world size must end with the same index."""
    ctx = mp.get_context("spawn")
    results = {}
    for world in (1, 2, 3):
        out = ctx.Queue()
        port = 35500 + (os.getpid() % 2000) + world
        procs = [ctx.Process(target=_build_worker, args=(r, world, port, out)) for r in range(world)]
        for p in procs:
            p.start()
        got = [out.get(timeout=300) for _ in range(world)]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        results[world] = dict(got)
    ref = results[1][0]
    for world, per_rank in results.items():
        for rank, res in per_rank.items():
            for metric, (part, codes, mode, its) in res.items():
                assert mode == "replicated"
                assert (part == ref[metric][0]).all() and (codes == ref[metric][1]).all(), (world, rank, metric)
                assert (its == ref[metric][3]).all()


# ---- list-sharded search straight from index files (lance_amd/dist.py: load_list_shard) ------------------------------
def _file_shard_worker(rank, world, port, index_dir, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import lance_amd.engine as E
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_engine import OracleDeviceIndex
    E.DeviceIndex = OracleDeviceIndex                      # load(lists=) -> lance_amd.dist.shard_index_contents -> oracle
    from lance_amd.dist import load_list_shard, search_list_sharded
    ix, l2g = load_list_shard(None, index_dir)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from ref_fixtures import ref_index_dir
    z = np.load(os.path.join(ref_index_dir(), "v0.8.14_ivf4_pq16.npz"))
    q = torch.from_numpy(np.ascontiguousarray(z["x"][:64]))
    res = {"rows": int(ix.o.row_ids.size)}
    for k, nprobes in ((10, 4), (10, 2), (50, 3)):
        gi, gd = search_list_sharded(lambda qq, kk, npb, rf: ix.search(qq, kk, npb, rf), l2g, q, k, nprobes)
        res[(k, nprobes)] = (gi.numpy(), gd.numpy())
    out.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_list_sharded_search_from_reference_index_files(world, tmp_path):
    """Each rank opens the index Lance 0.8.14 wrote (4 lists, legacy layout), keeps lists p % world == rank, and the
    all-gather + (dist, rowid) merge returns what a search of the whole index returns -- ids are row addresses
    (fragment << 32 | offset), not row numbers, so no local->global map is involved."""
    import shutil
    import oracle
    from lance_amd import index_file as IF
    d = tmp_path / "idx"
    d.mkdir()
    from ref_fixtures import ref_index_dir
    shutil.copyfile(os.path.join(ref_index_dir(), "v0.8.14_legacy", "index_2000.idx"), d / "index.idx")
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 33500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_file_shard_worker, args=(r, world, port, str(d), out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sum(r[1]["rows"] for r in res) == 2000
    c = IF.read_index_files(d)
    rm = c.codes_row_major()
    codes_t = np.concatenate([rm[c.part_offsets[p]:c.part_offsets[p + 1]].T.reshape(-1) for p in range(4)])
    full = oracle.IvfPqIndex("l2", c.centroids, c.codebook, c.part_offsets, codes_t, c.row_ids)
    z = np.load(os.path.join(ref_index_dir(), "v0.8.14_ivf4_pq16.npz"))
    for key in ((10, 4), (10, 2), (50, 3)):
        oi, od = full.search(z["x"][:64], key[0], key[1])
        for r in res:
            gi, gd = r[1][key]
            assert (gi.astype(np.uint64) == oi).all(), (world, key, r[0])
            assert (gd.view(np.uint32) == od.view(np.uint32)).all()


# ---- row-sharded build: every rank holds only its block of rows (lance_amd/dist.py: create_index_rowsharded) --------
def _rowshard_worker(rank, world, port, mode, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from oracle_engine import OracleDeviceIndex, OracleEngine as FullOracleEngine
    from lance_amd.dist import (block_ranges, create_index_rowsharded, list_shard_index, replica_index, search_list_sharded)

    class Eng(FullOracleEngine, OracleEngine):      # the host-loop contract (estep_partial / finalize) + the build steps
        pass

    rng = np.random.default_rng(3)
    n, d, nlist, m = 3001, 32, 8, 4
    x = (rng.standard_normal((n, d)) * 2 + 1).astype(f32)
    q = (rng.standard_normal((40, d)) * 2 + 1).astype(f32)
    _, ranges = block_ranges(n, world)
    lo, hi = ranges[rank]
    xl = torch.from_numpy(x[lo:hi].copy())
    eng = Eng()
    b = create_index_rowsharded(xl, metric="l2", num_partitions=nlist, num_sub_vectors=m, max_iters=5, sample_rate=32, seed=7,
                                engine=eng, ivf_training=mode)
    assert b.row0 == lo and b.n_total == n and b.part_local.shape[0] == hi - lo
    cent, cb = b.centroids.numpy(), b.codebook.numpy()
    # the single-index oracle on the same model: what every distributed search below must return
    oidx = oracle.build_index(x, cent, cb)
    assert (b.part_local.numpy().view(np.uint32) == oidx.part_ids[lo:hi]).all()
    assert (b.codes_local.numpy() == oidx.codes_rowmajor[lo:hi]).all()
    factory = lambda e, metric, c, cbk, part, codes, rid, raw=None: OracleDeviceIndex.create(e, metric, c, cbk, part, codes, rid, raw=raw)
    rep, raw = replica_index(b, xl, engine=eng, index_factory=factory)
    assert raw.shape[0] == n and (raw.numpy() == x).all()
    res = {"cent": cent, "cb": cb}
    for k, nprobes, rf in ((10, 3, 0), (5, nlist, 0), (10, 4, 3)):
        oi, od = oidx.search(q, k, nprobes, refine=rf, raw=x if rf else None)
        gi, gd = rep._ix.search(q, k, nprobes, rf)
        assert (gi.numpy().view(np.uint64) == oi).all() and (gd.numpy().view(np.uint32) == od.view(np.uint32)).all(), ("replica", k, nprobes, rf)
    shard, l2g = list_shard_index(b, xl, engine=eng, index_factory=factory)
    assert (np.diff(l2g.numpy()) > 0).all()          # local order == global order
    for k, nprobes, rf in ((10, 3, 0), (5, nlist, 0), (10, 4, 3)):
        oi, od = oidx.search(q, k, nprobes, refine=rf, raw=x if rf else None)
        gi, gd = search_list_sharded(lambda qq, kk, npb, r: shard.search(qq, kk, npb, r), l2g, torch.from_numpy(q), k, nprobes, rf)
        assert (gi.numpy().view(np.uint64) == oi).all() and (gd.numpy().view(np.uint32) == od.view(np.uint32)).all(), ("lists", k, nprobes, rf)
    out.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,mode", [(2, "sharded"), (3, "replicated"), (2, "replicated")])
def test_rowsharded_build_replica_and_list_shards(world, mode):
    """No rank ever sees another rank's vectors during the build (sampling, training, encode are local or collective); the
    replica (all-gather of codes) and the list shards (all_to_all by list owner) both answer like one index built by the
    oracle from the same model; every rank ends with the same model."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 36100 + (os.getpid() % 1500) + 3 * world + (0 if mode == "sharded" else 1)
    procs = [ctx.Process(target=_rowshard_worker, args=(r, world, port, mode, out)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(out.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(1, world):
        assert (got[r]["cent"].view(np.uint32) == got[0]["cent"].view(np.uint32)).all()
        assert (got[r]["cb"].view(np.uint32) == got[0]["cb"].view(np.uint32)).all()


# ---- hierarchical k-means with its splits spread over the ranks (lance_amd/dist.py: train_kmeans_hierarchical_sharded) ----------------
def _hier_data(f16=False):
    rng = np.random.default_rng(23)
    c = rng.standard_normal((40, 8)) * 4
    x = (c[rng.integers(0, 40, 9000)] + rng.standard_normal((9000, 8))).astype(f32)
    x[100:140] = x[100]                     # forty identical rows: a cluster that cannot be split (the `finalized` arm)
    return x.astype(np.float16) if f16 else x


def _hier_worker(rank, world, port, f16, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    try:
        from lance_amd import dist as ld
        x = _hier_data(f16)
        st = {}
        c = ld.train_kmeans_hierarchical_sharded(OracleEngine(), torch.from_numpy(x), 700, max_iters=12, balance_factor=1.0, seed=5, stats=st)
        out[rank] = (c.numpy().copy(), st)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("f16", [False, True])
def test_hierarchical_kmeans_spread_over_ranks_is_the_single_trainer_bit_for_bit(f16):
    """k = 700 > 256: the reference trains hierarchically (kmeans.rs:746-1003).  The multi-GPU trainer computes the splits of the largest
    clusters on different ranks and applies them in the reference's order: centroids equal to the oracle's single-process trainer bit
    for bit -- one process with speculation windows 1 (the plain loop), 4 and 9, and gloo worlds of 2 and 3 (sharded
    training for nlist > 4096 / north_star config 5; a row-sharded Lloyd loop would only agree to round-off)."""
    import oracle
    sys.path.insert(0, ROOT)
    from lance_amd import dist as ld
    x = _hier_data(f16)
    want = oracle.kmeans_train_hierarchical(x, 700, max_iters=12, balance_factor_scaled=f32(1.0) / f32(x.shape[0]), seed=5)
    assert want.shape[0] > 600
    wasted = 0
    for window in (1, 4, 9):
        st = {}
        got = ld.train_kmeans_hierarchical_sharded(OracleEngine(), torch.from_numpy(x), 700, max_iters=12, balance_factor=1.0, seed=5, window=window,
                                                   stats=st).numpy()
        assert got.shape == want.shape and (got.view(np.uint32) == np.ascontiguousarray(want, f32).view(np.uint32)).all(), window
        assert st["splits_thrown_away"] == 0 if window == 1 else True
        wasted += st["splits_thrown_away"]
        assert st["rounds"] <= st["splits_applied"]
    for world in (2, 3):
        mgr = mp.Manager()
        out = mgr.dict()
        mp.spawn(_hier_worker, args=(world, 29640 + world + (10 if f16 else 0), f16, out), nprocs=world, join=True)
        for r in range(world):
            c, st = out[r]
            assert c.shape == want.shape and (c.view(np.uint32) == np.ascontiguousarray(want, f32).view(np.uint32)).all(), (world, r)
            assert st["world"] == world and st["rounds"] < st["splits_applied"]      # several splits per round: the ranks did work side by side
