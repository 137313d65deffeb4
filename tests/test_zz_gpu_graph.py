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


def _changed_contents(idx, oidx, x, d, dev, graphs_on):
    """The serving pattern (bench.py: `host_step`): ONE device query buffer and ONE pair of output buffers, overwritten in place with a
    different batch before every call -- the captured graph is keyed on the pointers, so from the third call on every batch is answered
    by a replay that must read the NEW contents.  Outputs are filled with a sentinel before each call; every answer is compared with
    the oracle's for the batch that is in the buffer.  The same for a prefiltered search (the allow bitmap is rebuilt outside the graph,
    its scratch pointer is inside) and a distance-range search."""
    import torch
    import lance_amd
    eng = lance_amd.default_engine()
    n = x.shape[0]
    nq, k, nprobes, rf = 1500, 10, 8, 5
    batches = [_sift_like(nq, d, 100 + i) for i in range(3)]
    hq = [torch.from_numpy(b).pin_memory() for b in batches]
    qbuf = torch.empty((nq, d), dtype=torch.float32, device=dev)
    out = (torch.empty((nq, k), dtype=torch.int64, device=dev), torch.empty((nq, k), dtype=torch.float32, device=dev))
    h_ids = torch.empty((nq, k), dtype=torch.int64).pin_memory(); h_d = torch.empty((nq, k), dtype=torch.float32).pin_memory()
    want = [oidx.search(b, k, nprobes, refine=rf, raw=x) for b in batches]
    replays = lambda: eng.timing_query("count:graph_replay")[1]
    r0 = replays()
    done = 0
    for rep in range(7):                     # plain, capture, then replays -- the batch in the buffer changes every time
        b = rep % 3
        qbuf.copy_(hq[b], non_blocking=True)
        out[0].fill_(-7); out[1].fill_(float("nan"))
        torch.cuda.current_stream().synchronize()
        idx.search_device(qbuf, k, nprobes, rf, out=out, sync=False)
        eng.synchronize()
        h_ids.copy_(out[0], non_blocking=True); h_d.copy_(out[1], non_blocking=True)
        torch.cuda.current_stream().synchronize()
        oi, od = want[b]
        bad = np.argwhere((h_ids.numpy().view(np.uint64) != oi).any(axis=1)).reshape(-1)
        assert bad.size == 0, ("changed contents", rep, b, f"{bad.size} of {nq} queries differ, first {bad[:5]}")
        assert (h_d.numpy().view(np.uint32) == od.view(np.uint32)).all(), ("changed contents", rep, b)
        done += 1
    if graphs_on:
        assert replays() - r0 >= 5, f"expected the calls from the third on to be graph replays, counted {replays() - r0}"
    else:
        assert replays() == r0
    # prefilter: two different masks through the same buffers (the bitmap is rebuilt per call, the graph only holds its address)
    rng = np.random.default_rng(3)
    masks = [rng.random(n) < 0.5, rng.random(n) < 0.2]
    r0 = replays()
    for rep in range(6):
        b, mk = rep % 3, masks[rep % 2]
        qbuf.copy_(hq[b]); out[0].fill_(-7); out[1].fill_(float("nan"))
        torch.cuda.synchronize()
        gi, gd = idx._ix.search_filtered(qbuf, k, nprobes, mk, refine_factor=0, out=out)
        oi, od = oidx.search(batches[b], k, nprobes, prefilter=mk)
        assert (gi.cpu().numpy().view(np.uint64) == oi).all(), ("filtered", rep)
        assert (gd.cpu().numpy().view(np.uint32) == od.view(np.uint32)).all(), ("filtered", rep)
        done += 1
    if graphs_on:
        assert replays() - r0 >= 4, f"filtered searches: {replays() - r0} replays"
    # distance range through the same buffers
    lo, hi = 0.0, float(np.median(want[0][1][:, -1]))
    r0 = replays()
    for rep in range(5):
        b = rep % 3
        qbuf.copy_(hq[b]); out[0].fill_(-7); out[1].fill_(float("nan"))
        torch.cuda.synchronize()
        gi, gd = idx._ix.search_range(qbuf, k, nprobes, lower=lo, upper=hi, out=out)
        oi, od = oidx.search(batches[b], k, nprobes, lower=lo, upper=hi)
        assert (gi.cpu().numpy().view(np.uint64) == oi).all(), ("range", rep)
        assert (gd.cpu().numpy().view(np.uint32) == od.view(np.uint32)).all(), ("range", rep)
        done += 1
    if graphs_on:
        assert replays() - r0 >= 2, f"range searches: {replays() - r0} replays"
    # A caller that never repeats a call (fresh output buffers, kept alive so that the allocator cannot hand an address out twice) must
    # neither create graph entries nor push the steady-state graph out of the cache (the cache used to be dropped wholesale).
    captures = lambda: eng.timing_query("count:graph_capture")[1]
    qbuf.copy_(hq[0]); torch.cuda.synchronize()
    for _ in range(3):      # the steady-state key (the filtered / range loops above may have grown a scratch slot, which drops every graph:
        idx.search_device(qbuf, k, nprobes, rf, out=out)      # plain, capture, replay bring it back)
    c0, r0 = captures(), replays()
    keep_alive = []
    qsmall = qbuf[:40]
    for i in range(150):
        keep_alive.append(idx.search_device(qsmall, 5, 4, 0))
    assert captures() == c0, "calls that never repeat were captured"
    out[0].fill_(-7); torch.cuda.synchronize()
    idx.search_device(qbuf, k, nprobes, rf, out=out)
    if graphs_on:
        assert replays() == r0 + 1, "the steady-state graph was evicted by calls that never repeat"
    assert (out[0].cpu().numpy().view(np.uint64) == want[0][0]).all()
    gi, _ = keep_alive[-1]
    oi, _ = oidx.search(batches[0][:40], 5, 4)
    assert (gi.cpu().numpy().view(np.uint64) == oi).all()
    return done + 2


def _cases():
    sys.path.insert(0, ROOT)
    import torch
    import lance_amd
    import oracle
    from lance_amd.engine import Engine
    dev = torch.device("cuda", 0)
    checked = 0
    graphs_on = os.environ.get("LANCE_HIP_GRAPH", "1") != "0"
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
            if out is not None:      # a replay that did nothing must not pass on the previous repetition's answer
                out[0].fill_(-7); out[1].fill_(float("nan"))
                torch.cuda.synchronize()
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
        checked += _changed_contents(idx, oidx, x, d, dev, graphs_on)
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
