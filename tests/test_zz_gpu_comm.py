"""The sharded Lloyd loop with its collectives behind the C ABI (lance_amd/csrc/comm.cpp; include/lance_hip.h
`lance_hip_comm_*`, `lance_hip_kmeans_train_sharded`): the entry points a host without torch.distributed binds.  One GPU box =
world size 1: a communicator is created through the library (ncclGetUniqueId -> ncclCommInitRank from the dlopen'ed librccl), every
Lloyd iteration runs its three ncclAllReduce calls on the context's stream, and the result must equal the run without a
communicator bit for bit (a one-rank all-reduce is the identity) and the single-process reference loop to f32 round-off -- the
tolerance tests/test_dist_gloo.py::test_sharded_kmeans_two_ranks states for two ranks."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
f32 = np.float32


def test_sharded_trainer_through_the_c_abi_world1(engine, oracle):
    from lance_amd.engine import Engine
    eng = Engine()
    rng = np.random.default_rng(61)
    n, d, k = 30_000, 64, 32
    centers = rng.uniform(0, 128, (40, d))
    x = np.clip(np.rint(centers[rng.integers(0, 40, n)] + rng.normal(0, 20, (n, d))), 0, 218).astype(f32)
    init = x[rng.permutation(n)[:k]].copy()
    uid = eng.comm_unique_id()
    assert len(uid) == 128
    comm = eng.comm_create(uid, 1, 0)
    try:
        c1, l1, i1 = eng.kmeans_train_sharded(comm, x, init, n, max_iters=20, balance_factor=1.0, seed=5)
        c0, l0, i0 = eng.kmeans_train_sharded(None, x, init, n, max_iters=20, balance_factor=1.0, seed=5)
    finally:
        eng.comm_destroy(comm)
    c1 = c1.cpu().numpy(); c0 = c0.cpu().numpy()
    assert (c1.view(np.uint32) == c0.view(np.uint32)).all() and l1 == l0 and i1 == i0
    oc, ol, oit, _ = oracle.kmeans_train(x, k, max_iters=20, balance_factor=f32(1.0) / f32(n), init=init, seed=5)
    assert i1 == oit
    assert np.allclose(c1, oc, rtol=1e-4, atol=1e-3)
    assert abs(l1 - ol) <= 1e-5 * abs(ol)
    eng.close()
