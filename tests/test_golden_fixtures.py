"""Committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py).

ref_torch_assign.npz holds outputs of the REFERENCE's Python accelerator kernels
(python/python/lance/torch/distance.py) on integer-valued inputs, where f32 arithmetic is exact and therefore
order-independent: the oracle (CPU suite) and the HIP path (GPU suite) must reproduce them bit for bit.
e2e_small.npz freezes an oracle-produced IVF_PQ build + search.
"""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
f32 = np.float32


@pytest.fixture(scope="module")
def ref():
    return np.load(os.path.join(GOLD, "ref_torch_assign.npz"))


@pytest.fixture(scope="module")
def e2e():
    return np.load(os.path.join(GOLD, "e2e_small.npz"))


def _bits(a):
    return np.ascontiguousarray(a, f32).view(np.uint32)


# ---------------------------------------------------------------- CPU: oracle vs fixtures
def test_oracle_assign_equals_reference_torch_kernels(oracle, ref):
    ids, dists = oracle.assign(ref["int_x"], ref["int_c"], "l2")
    assert (ids == ref["int_l2_ids"]).all()
    assert (_bits(dists) == _bits(ref["int_l2_min"])).all()
    for i in range(0, 40):
        row = oracle.distance_batch("l2", ref["int_x"][i], ref["int_c"])
        assert (_bits(row) == _bits(ref["int_l2_matrix"][i])).all()
    ids, dists = oracle.assign(ref["int_dot_x"], ref["int_c"], "dot")
    assert (ids == ref["int_dot_ids"]).all()
    assert (_bits(dists) == _bits(ref["int_dot_min"])).all()


def test_oracle_assign_close_to_reference_torch_on_floats(oracle, ref):
    ids, dists = oracle.assign(ref["f_x"], ref["f_c"], "l2")
    # the torch kernel goes through cdist (sqrt) and squares back: 1e-4 relative, ids equal unless the two best are that close
    np.testing.assert_allclose(dists, ref["f_l2_min"], rtol=1e-4)
    diff = ids != ref["f_l2_ids"]
    if diff.any():
        for i in np.nonzero(diff)[0]:
            row = np.sort(oracle.distance_batch("l2", ref["f_x"][i], ref["f_c"]))
            assert row[1] - row[0] <= 1e-4 * row[0]


@pytest.mark.parametrize("metric", ["l2", "dot"])
def test_oracle_reproduces_frozen_e2e(oracle, e2e, metric):
    x, q = e2e["x"], e2e["q"]
    n = x.shape[0]
    cent, loss, iters, _ = oracle.kmeans_train(x, 16, metric="l2", max_iters=20, tol=1e-4, balance_factor=1.0 / n,
                                               init=e2e[f"{metric}_init"], seed=5)
    assert (_bits(cent) == _bits(e2e[f"{metric}_centroids"])).all()
    assert loss == float(e2e[f"{metric}_ivf_loss"]) and iters == int(e2e[f"{metric}_ivf_iters"])
    idx = oracle.build_index(x, e2e[f"{metric}_centroids"], e2e[f"{metric}_codebook"], metric=metric)
    assert (idx.codes_t == e2e[f"{metric}_codes_t"]).all() and (idx.row_ids == e2e[f"{metric}_row_ids"]).all()
    ids, d = idx.search(q, 10, 4, refine=0)
    assert (ids == e2e[f"{metric}_ids_np4"]).all() and (_bits(d) == _bits(e2e[f"{metric}_dists_np4"])).all()
    ids, d = idx.search(q, 10, 4, refine=5, raw=x)
    assert (ids == e2e[f"{metric}_ids_np4_rf5"]).all() and (_bits(d) == _bits(e2e[f"{metric}_dists_np4_rf5"])).all()


# ---------------------------------------------------------------- GPU: HIP path vs fixtures
@pytest.fixture(scope="module")
def eng(engine):
    from lance_amd.engine import Engine
    e = Engine()
    yield e
    e.close()


@pytest.mark.gpu
def test_hip_assign_equals_reference_torch_kernels(eng, ref):
    ids, dists = eng.assign(ref["int_x"], ref["int_c"], "l2")
    assert (ids.cpu().numpy().view(np.uint32) == ref["int_l2_ids"]).all()
    assert (_bits(dists.cpu().numpy()) == _bits(ref["int_l2_min"])).all()
    ids, dists = eng.assign(ref["int_dot_x"], ref["int_c"], "dot")
    assert (ids.cpu().numpy().view(np.uint32) == ref["int_dot_ids"]).all()
    assert (_bits(dists.cpu().numpy()) == _bits(ref["int_dot_min"])).all()
    # full distance rows through find_partitions with nprobes = k: sorted (dist, id) of the reference matrix
    k = ref["int_c"].shape[0]
    pid, pd = eng.find_partitions(ref["int_x"][:64], ref["int_c"], k)
    mat = ref["int_l2_matrix"][:64]
    order = np.lexsort((np.broadcast_to(np.arange(k), mat.shape), mat), axis=1)
    assert (pid.cpu().numpy().view(np.uint32) == order).all()
    assert (_bits(pd.cpu().numpy()) == _bits(np.take_along_axis(mat, order, 1))).all()
    ids, dists = eng.assign(ref["f_x"], ref["f_c"], "l2")
    np.testing.assert_allclose(dists.cpu().numpy(), ref["f_l2_min"], rtol=1e-4)


def _check_accelerator_module(acc, engine, ref):
    """lance_amd.accelerator (the lance.torch.distance / lance.torch.kmeans names) against the outputs of the reference's
    own module stored in ref_torch_assign.npz: same ids (int64), same distances, same NaN convention, same messages."""
    import torch
    idx, d = acc.l2_distance(torch.from_numpy(ref["int_x"]), torch.from_numpy(ref["int_c"]), engine=engine)
    assert idx.dtype == torch.int64 and d.dtype == torch.float32 and idx.shape == (599,)
    assert (idx.cpu().numpy() == ref["int_l2_ids"].astype(np.int64)).all() and (_bits(d.cpu().numpy()) == _bits(ref["int_l2_min"])).all()
    idx, d = acc.dot_distance(torch.from_numpy(ref["int_dot_x"]), torch.from_numpy(ref["int_c"]), engine=engine)
    assert (idx.cpu().numpy() == ref["int_dot_ids"].astype(np.int64)).all() and (_bits(d.cpu().numpy()) == _bits(ref["int_dot_min"])).all()
    x = ref["int_x"][:8].copy()
    x[2, 5] = np.nan                                              # distance.py:199,265: NaN distance -> id -1
    idx, d = acc.l2_distance(torch.from_numpy(x), torch.from_numpy(ref["int_c"]), engine=engine)
    assert idx[2].item() == -1 and np.isnan(d.cpu().numpy()[2]) and (idx.cpu().numpy()[[0, 1, 3]] == ref["int_l2_ids"][[0, 1, 3]]).all()
    with pytest.raises(ValueError, match="x and y must be 2-D matrix"):
        acc.l2_distance(torch.zeros(4), torch.zeros(2, 4), engine=engine)
    ci, cd = acc.cosine_distance(torch.from_numpy(ref["f_x"]), torch.from_numpy(ref["f_c"]), engine=engine)
    xn = ref["f_x"] / np.linalg.norm(ref["f_x"], axis=1, keepdims=True)
    cn = ref["f_c"] / np.linalg.norm(ref["f_c"], axis=1, keepdims=True)
    want = 1.0 - xn @ cn.T
    assert np.allclose(cd.cpu().numpy(), want.min(1), atol=1e-5) and (ci.cpu().numpy() == want.argmin(1)).mean() > 0.99
    km = acc.KMeans(8, metric="l2", max_iters=10, seed=3, engine=engine)
    km.fit(ref["f_x"])
    assert tuple(km.centroids.shape) == (8, 40) and km.total_distance > 0
    part = km.transform(ref["f_x"])
    assert part.dtype == torch.int64 and (part.cpu().numpy() == acc.l2_distance(torch.from_numpy(ref["f_x"]), km.centroids, engine=engine)[0].cpu().numpy()).all()
    km2 = acc.KMeans(8, metric="cosine", max_iters=5, seed=3, engine=engine)
    km2.fit(iter([{"v": torch.from_numpy(ref["f_x"][:256])}, {"v": torch.from_numpy(ref["f_x"][256:])}]), column="v")
    nrm = np.linalg.norm(km2.centroids.cpu().numpy(), axis=1)       # means of unit vectors: inside the unit ball
    assert tuple(km2.centroids.shape) == (8, 40) and np.isfinite(nrm).all() and (nrm <= 1.0 + 1e-5).all() and (nrm > 0).all()
    with pytest.raises(ValueError, match="Only random initialization"):
        acc.KMeans(4, init="kmeans++")
    with pytest.raises(ValueError, match="not supported"):
        acc.KMeans(4, metric="hamming")


def test_accelerator_module_on_oracle_backed_engine(ref):
    """CPU: the wrapper logic (dtypes, NaN convention, cosine route, KMeans plumbing) with the device steps computed by
    the oracle stand-in used by the multi-rank build tests."""
    from lance_amd import accelerator
    from test_dist_gloo import OracleBuildEngine
    _check_accelerator_module(accelerator, OracleBuildEngine(), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("metric", ["l2", "dot"])
def test_hip_reproduces_frozen_e2e(eng, e2e, metric):
    from lance_amd.engine import DeviceIndex
    x, q = e2e["x"], e2e["q"]
    cent, loss, iters = eng.kmeans_train(x, 16, max_iters=20, tol=1e-4, balance_factor=1.0, init=e2e[f"{metric}_init"], seed=5)
    assert (_bits(cent.cpu().numpy()) == _bits(e2e[f"{metric}_centroids"])).all()
    assert loss == float(e2e[f"{metric}_ivf_loss"]) and iters == int(e2e[f"{metric}_ivf_iters"])
    part, codes, _ = eng.ivfpq_encode(x, e2e[f"{metric}_centroids"], e2e[f"{metric}_codebook"], metric)
    assert (part.cpu().numpy().view(np.uint32) == e2e[f"{metric}_part_ids"]).all()
    assert (codes.cpu().numpy() == e2e[f"{metric}_codes"]).all()
    g = DeviceIndex.from_storage(eng, metric, e2e[f"{metric}_centroids"], e2e[f"{metric}_codebook"], e2e[f"{metric}_offsets"],
                                 e2e[f"{metric}_codes_t"], e2e[f"{metric}_row_ids"], raw=x)
    ids, d = g.search(q, 10, 4, 0)
    assert (ids.cpu().numpy().view(np.uint64) == e2e[f"{metric}_ids_np4"]).all()
    assert (_bits(d.cpu().numpy()) == _bits(e2e[f"{metric}_dists_np4"])).all()
    ids, d = g.search(q, 10, 4, 5)
    assert (ids.cpu().numpy().view(np.uint64) == e2e[f"{metric}_ids_np4_rf5"]).all()
    assert (_bits(d.cpu().numpy()) == _bits(e2e[f"{metric}_dists_np4_rf5"])).all()
