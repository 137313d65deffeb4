"""Captured searches (search.hip: ivfpq_search_enqueue; on by default, LANCE_HIP_GRAPH=0 switches it off): the second call with
the same arguments is captured into a HIP graph and later ones replay it.  A replay must give what the plain path gives -- the oracle's answer, bit for bit -- also after another
call has grown the scratch arena (which drops every captured graph), through a second context, and with the per-query-table
filter on.  The switch is read once per process: the cases run in a child process."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
f32 = np.float32
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sift_like(n, d, seed, ncl=64):
    rng = np.random.default_rng(seed)
    centers = rng.uniform(0, 128, (ncl, d))
    x = centers[rng.integers(0, ncl, n)] + rng.normal(0, 24, (n, d))
    return np.clip(np.rint(x), 0, 218).astype(f32)


def _cases():
    sys.path.insert(0, ROOT)
    import torch
    import lance_amd
    import oracle
    from lance_amd.engine import Engine
    dev = torch.device("cuda", 0)
    checked = 0
    for (n, d, nlist, m) in ((60_000, 128, 64, 16), (30_000, 384, 32, 96)):
        x = _sift_like(n, d, 5)
        idx = lance_amd.create_index(x, "IVF_PQ", metric="l2", num_partitions=nlist, num_sub_vectors=m, max_iters=6)
        oidx = oracle.build_index(x, idx.centroids, idx.codebook)
        e2 = Engine()
        qs = {"big": torch.from_numpy(_sift_like(1500, d, 11)).to(dev), "small": torch.from_numpy(_sift_like(40, d, 12)).to(dev),
              "bigger": torch.from_numpy(_sift_like(4000, d, 13)).to(dev)}
        want = {}
        for name, (k, nprobes, rf) in (("big", (10, 8, 5)), ("small", (7, nlist, 0)), ("bigger", (10, 4, 0))):
            oi, od = oidx.search(qs[name].cpu().numpy(), k, nprobes, refine=rf, raw=x if rf else None)
            want[name] = (k, nprobes, rf, oi, od)

        def check(name, engine=None, out=None, tag=""):
            nonlocal checked
            k, nprobes, rf, oi, od = want[name]
            gi, gd = idx.search_device(qs[name], k, nprobes, rf, out=out, engine=engine)
            gi = gi.cpu().numpy().view(np.uint64); gd = gd.cpu().numpy()
            bad = np.argwhere((gi != oi).any(axis=1)).reshape(-1)
            assert bad.size == 0, (n, m, name, tag, f"{bad.size} of {gi.shape[0]} queries differ, first {bad[:5]}", gi[bad[0]], oi[bad[0]])
            assert (gd.view(np.uint32) == od.view(np.uint32)).all(), (n, m, name, tag)
            checked += 1

        outs = {name: (torch.empty((qs[name].shape[0], want[name][0]), dtype=torch.int64, device=dev),
                       torch.empty((qs[name].shape[0], want[name][0]), dtype=torch.float32, device=dev)) for name in qs}
        for rep in range(4):                       # plain, capture, replay, replay
            check("big", out=outs["big"], tag=f"a{rep}"); check("small", out=outs["small"], tag=f"a{rep}")
        check("bigger", out=outs["bigger"], tag="grow")       # grows the arena: every captured graph is dropped
        for rep in range(4):
            check("big", out=outs["big"], tag=f"b{rep}"); check("small", out=outs["small"], tag=f"b{rep}"); check("bigger", out=outs["bigger"], tag=f"b{rep}")
        for rep in range(4):                       # a second context has its own cache
            check("big", engine=e2, out=outs["big"], tag=f"c{rep}")
        # asynchronous enqueues on the second context, results read after one synchronisation
        k, nprobes, rf, oi, od = want["big"]
        for rep in range(3):
            idx.search_device(qs["big"], k, nprobes, rf, out=outs["big"], sync=False, engine=e2)
        e2.synchronize()
        assert (outs["big"][0].cpu().numpy().view(np.uint64) == oi).all()
        e2.close()
    print(f"graph cases ok: {checked}")


@pytest.mark.parametrize("qpt", ["plain", "0", "1"])
def test_captured_search_graphs_replay_the_plain_answers(qpt):
    env = dict(os.environ, LANCE_HIP_GRAPH="0" if qpt == "plain" else "1")      # "plain": the same cases without capture (control)
    if qpt == "1":
        env["LANCE_HIP_QPT"] = "1"
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import tests.test_zz_gpu_graph as t; t._cases()" % ROOT],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "graph cases ok" in r.stdout
