"""Times lance_hip_assign (1M x 128 vs 256 centroids, and a C4-like 4096-centroid case) on the MFMA candidate path and on the
exact VALU path (LANCE_HIP_NO_MFMA=1 in a child process), and checks that both return the same ids / distances."""
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(tag):
    import numpy as np
    import torch
    import lance_amd
    from lance_amd.testing import sift_like
    eng = lance_amd.default_engine()
    out = {"path": tag}
    for name, n, k in (("1Mx128_k256", 1_000_000, 256), ("1Mx128_k4096", 1_000_000, 4096), ("65536x128_k256", 65536, 256)):
        x = sift_like(n, 128, seed=1234, device="cuda")
        c = sift_like(k, 128, seed=99, device="cuda") + 0.25
        ids, d = eng.assign(x, c)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            ids, d = eng.assign(x, c)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        out[name] = {"ms": min(ts) * 1e3, "ids_sum": int(ids.long().sum().item()), "d_sum": float(d.double().sum().item()),
                     "ids_head": ids[:8].tolist()}
        np.save(f"/tmp/assign_{tag}_{name}.npy", ids.cpu().numpy())
        np.save(f"/tmp/assign_{tag}_{name}_d.npy", d.cpu().numpy())
    # PQ encode: 1M residual-like rows, M = 16 sub-quantisers of 8 dimensions (and M = 32 x 4, M = 8 x 16)
    for m in (16, 32, 8):
        x = sift_like(1_000_000, 128, seed=77, device="cuda") - 60.0
        cb = (torch.randn((m, 256, 128 // m), device="cuda", generator=torch.Generator(device="cuda").manual_seed(5 + m)) * 40.0).contiguous()
        codes = eng.pq_encode(x, cb)
        torch.cuda.synchronize()
        ts = []
        for _ in range(4):
            t0 = time.perf_counter()
            codes = eng.pq_encode(x, cb)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        name = f"pq_encode_1Mx128_m{m}"
        out[name] = {"ms": min(ts) * 1e3, "sum": int(codes.long().sum().item())}
        np.save(f"/tmp/assign_{tag}_{name}.npy", codes.cpu().numpy())
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        import numpy as np
        env = dict(os.environ)
        subprocess.check_call([sys.executable, __file__, "mfma"], env=env)
        env["LANCE_HIP_NO_MFMA"] = "1"
        subprocess.check_call([sys.executable, __file__, "exact"], env=env)
        for name in ("1Mx128_k256", "1Mx128_k4096", "65536x128_k256"):
            a, b = np.load(f"/tmp/assign_mfma_{name}.npy"), np.load(f"/tmp/assign_exact_{name}.npy")
            da, db = np.load(f"/tmp/assign_mfma_{name}_d.npy"), np.load(f"/tmp/assign_exact_{name}_d.npy")
            print(name, "ids equal:", bool((a == b).all()), "mismatches:", int((a != b).sum()), "dists bit-equal:",
                  bool((da.view(np.uint32) == db.view(np.uint32)).all()))
        for m in (16, 32, 8):
            name = f"pq_encode_1Mx128_m{m}"
            a, b = np.load(f"/tmp/assign_mfma_{name}.npy"), np.load(f"/tmp/assign_exact_{name}.npy")
            print(name, "codes equal:", bool((a == b).all()), "mismatches:", int((a != b).sum()))
