"""C1 batched: 1M x 128 f32 flat L2 k=10, 10,000-query batch (and 1 / 1000 queries), MFMA filter vs the exact VALU filter."""
import json, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(tag):
    import numpy as np, torch, lance_amd
    from lance_amd.testing import sift_like
    eng = lance_amd.default_engine()
    x = sift_like(1_000_000, 128, seed=1234, device="cuda")
    out = {"path": tag}
    for nq in (1, 1000, 10000):
        q = sift_like(nq, 128, seed=4321, device="cuda")
        ids, d = eng.flat_topk(x, q, 10); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); ids, d = eng.flat_topk(x, q, 10); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        out[f"nq{nq}"] = {"ms": min(ts) * 1e3, "qps": nq / min(ts)}
        np.save(f"/tmp/flat_{tag}_{nq}.npy", ids.cpu().numpy()); np.save(f"/tmp/flat_{tag}_{nq}_d.npy", d.cpu().numpy())
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        import numpy as np
        env = dict(os.environ)
        subprocess.check_call([sys.executable, __file__, "mfma"], env=env)
        env["LANCE_HIP_NO_MFMA_FLAT"] = "1"
        subprocess.check_call([sys.executable, __file__, "exact"], env=env)
        for nq in (1, 1000, 10000):
            a, b = np.load(f"/tmp/flat_mfma_{nq}.npy"), np.load(f"/tmp/flat_exact_{nq}.npy")
            da, db = np.load(f"/tmp/flat_mfma_{nq}_d.npy"), np.load(f"/tmp/flat_exact_{nq}_d.npy")
            print(f"nq={nq}: ids equal {bool((a == b).all())}, dists bit-equal {bool((da.view(np.uint32) == db.view(np.uint32)).all())}")
