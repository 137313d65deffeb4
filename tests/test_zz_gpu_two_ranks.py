"""First-contact insurance for the N-rank path on a box with ONE GPU.

No multi-GPU node has ever been available to this repository, and RCCL refuses two ranks on one device -- so until the driver's scaling
run happens, the multi-GPU code had only met world size 1 on hardware and world size 2 on the CPU stand-in engine
(tests/test_dist_gloo.py).  These tests close what can be closed without a second GPU: TWO processes share the one MI355X, every kernel,
buffer layout, update step and exchange of the sharded build / list-sharded search runs for real, and only the transport differs
(gloo through host memory instead of RCCL over xGMI):

  * `lance_hip_kmeans_train_sharded` (lance_amd/csrc/comm.cpp), the Lloyd loop with its collectives behind the C ABI, at world size 2
    through `lance_hip_comm_from_callback` -- the same three exchanges per iteration it issues through ncclAllReduce;
  * `bench.py --gpus 2` end to end (LANCE_BENCH_ONE_GPU=1 LANCE_BENCH_BACKEND=gloo): row-sharded build in both IVF training modes, the
    replica all-gather, the list-sharded search with its all_to_all / all-gather / device merge, the max-over-ranks timing and the one
    JSON line of rank 0.

Reference: rust/lance-index/src/vector/kmeans.rs:610-719 (train_kmeans; the rayon reduction is where the exchange sits),
python/python/lance/torch/kmeans.py (the accelerator seam's sharded trainer).
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
f32 = np.float32
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _data():
    rng = np.random.default_rng(61)
    n, d, k = 40_001, 64, 32          # an odd row count: the ranks' shards differ in length
    centers = rng.uniform(0, 128, (40, d))
    x = np.clip(np.rint(centers[rng.integers(0, 40, n)] + rng.normal(0, 20, (n, d))), 0, 218).astype(f32)
    init = x[rng.permutation(n)[:k]].copy()
    return x, init, k


def _rank_main(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    import ctypes as C

    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lance_amd._lib import check
    from lance_amd.engine import Engine
    torch.cuda.set_device(0)
    eng = Engine(device=0)
    x, init, k = _data()
    n = x.shape[0]
    per = (n + world - 1) // world
    xl = x[rank * per: min(n, (rank + 1) * per)]
    calls = {"n": 0}

    def allreduce(buf, count, dtype, op, stream):
        host = np.empty(count, np.float64 if dtype == 1 else np.float32)
        check(eng.lib.lance_hip_memcpy_d2h(eng.h, host.ctypes.data_as(C.c_void_p), C.c_void_p(buf), host.nbytes))   # orders itself behind the stream
        t = torch.from_numpy(host)
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM)
        check(eng.lib.lance_hip_memcpy_h2d(eng.h, C.c_void_p(buf), host.ctypes.data_as(C.c_void_p), host.nbytes))
        calls["n"] += 1
        return 0

    comm = eng.comm_from_callback(allreduce, world, rank)
    try:
        cent, loss, iters = eng.kmeans_train_sharded(comm, xl, init, n, max_iters=20, balance_factor=1.0, seed=5)
    finally:
        eng.comm_destroy(comm)
    np.savez(out_path % rank, cent=cent.cpu().numpy(), loss=loss, iters=iters, calls=calls["n"])
    dist.barrier()
    dist.destroy_process_group()
    eng.close()


def test_sharded_trainer_behind_the_c_abi_two_ranks_one_gpu(engine, oracle, tmp_path):
    world, port = 2, _free_port()
    out = str(tmp_path / "rank%d.npz")
    code = ("import sys; sys.path.insert(0, %r); import tests.test_zz_gpu_two_ranks as t; t._rank_main(int(sys.argv[1]), %d, %d, %r)"
            % (ROOT, world, port, out))
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r)], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for pp in procs:
                pp.kill()
            raise
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-3000:] for l in logs)
    r0, r1 = np.load(out % 0), np.load(out % 1)
    # both ranks end with the same model (every rank applies the same reduced sums), after the same number of iterations
    assert (r0["cent"].view(np.uint32) == r1["cent"].view(np.uint32)).all()
    assert r0["loss"] == r1["loss"] and r0["iters"] == r1["iters"]
    # one status word before the first exchange, then three exchanges (sums | counts, losses, radii) per ENQUEUED iteration: the host
    # looks at the convergence state every 8 iterations, the device-side gate idles the iterations enqueued past convergence
    enq = min(20, (int(r0["iters"]) + 7) // 8 * 8)
    assert int(r0["calls"]) == 1 + 3 * enq, (int(r0["calls"]), int(r0["iters"]))
    # against the single-process loop: f32 round-off (the sums arrive in rank order, not row order) -- the tolerance
    # tests/test_dist_gloo.py::test_sharded_kmeans_two_ranks states
    x, init, k = _data()
    oc, ol, oit, _ = oracle.kmeans_train(x, k, max_iters=20, balance_factor=f32(1.0) / f32(x.shape[0]), init=init, seed=5)
    assert int(r0["iters"]) == oit
    assert np.allclose(r0["cent"], oc, rtol=1e-4, atol=1e-3)
    assert abs(float(r0["loss"]) - ol) <= 1e-5 * abs(ol)


def test_bench_two_ranks_on_one_gpu_end_to_end():
    """`python bench.py --gpus 2` as the driver will start it on a multi-GPU node, except that both ranks sit on device 0 and the
    collectives travel through host memory.  Checks the contract of the line and that the list-sharded search equals the replica's."""
    env = dict(os.environ, LANCE_BENCH_ONE_GPU="1", LANCE_BENCH_BACKEND="gloo", MASTER_PORT=str(_free_port()))
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--n", "200000",
                        "--nq", "3000", "--no-pmc", "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["scaling"] == "weak" and j["value"] > 0
    mg = j["multi_gpu"]
    assert mg["rccl_ranks"] == 2 and "gloo" in mg["transport"]
    assert mg["list_sharded_equals_replica"] is True
    assert mg["build_sec_ivf_sharded_allreduce"] > 0 and mg["build_sec_ivf_replicated"] > 0
    assert mg["list_sharded_qps_strong_scaling"] > 0 and j["strong_scaling_list_sharded_qps"] == mg["list_sharded_qps_strong_scaling"]
    assert mg["rows_per_rank"] == 100000
    assert j["recall_at_10"] > 0.85


def test_bench_eight_ranks_on_one_gpu_uneven_blocks_and_lists():
    """`python bench.py --gpus 8` -- the driver's scaling command -- with all eight ranks on the one GPU (gloo through host memory): 200,003
    rows (blocks of 25,001 and a short last one), 250 IVF lists (250 % 8 = 2: the list -> rank map is uneven).  The line's contract, and
    the list-sharded search -- its local half now ONE scan per rank (lance_hip_ivfpq_search_candidates) -- equal to the replica's."""
    env = dict(os.environ, LANCE_BENCH_ONE_GPU="1", LANCE_BENCH_BACKEND="gloo", MASTER_PORT=str(_free_port()))
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--n", "200003", "--nlist", "250",
                        "--nq", "2000", "--no-pmc", "--no-cpu-baseline", "--no-grid"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 8 and j["scaling"] == "weak" and j["value"] > 0 and j["config"]["nlist"] == 250
    mg = j["multi_gpu"]
    assert mg["rccl_ranks"] == 8 and mg["rows_per_rank"] == 25001
    assert mg["list_sharded_equals_replica"] is True
    assert mg["list_sharded_qps_strong_scaling"] > 0
    assert "round-off" in mg["ivf_training_sharded_is"]
    assert j["recall_at_10"] > 0.8


def test_hierarchical_training_spread_over_ranks_equals_the_single_gpu_trainer(engine, oracle):
    """lance_hip_kmeans_split as the unit of work of lance_amd.dist.train_kmeans_hierarchical_sharded (one process here: windows of 1 and
    6 speculated splits): centroids equal to lance_hip_kmeans_train's hierarchical path and to the oracle's, bit for bit -- f32, f16
    (half-precision M-step) and int8 samples."""
    import torch
    from lance_amd import dist as ld
    eng = engine.default_engine()
    rng = np.random.default_rng(41)
    c = rng.standard_normal((60, 32)) * 30
    base = np.clip(np.rint(c[rng.integers(0, 60, 30000)] + rng.standard_normal((30000, 32)) * 9), -127, 127)
    for x in (base.astype(f32), (base / 64).astype(np.float16), base.astype(np.int8)):
        xt = torch.from_numpy(x).cuda()
        single, _, _ = eng.kmeans_train(xt, 600, max_iters=10, balance_factor=1.0, seed=3)
        xo = x.astype(f32) if x.dtype == np.int8 else x
        want = oracle.kmeans_train_hierarchical(xo, 600, max_iters=10, balance_factor_scaled=f32(1.0) / f32(x.shape[0]), seed=3)
        assert (single.float().cpu().numpy().view(np.uint32) == np.ascontiguousarray(want, f32).view(np.uint32)).all()
        for window in (1, 6):
            st = {}
            got = ld.train_kmeans_hierarchical_sharded(eng, xt, 600, max_iters=10, balance_factor=1.0, seed=3, window=window, stats=st)
            assert got.shape[0] == want.shape[0]
            assert (got.cpu().numpy().view(np.uint32) == np.ascontiguousarray(want, f32).view(np.uint32)).all(), (x.dtype, window, st)
