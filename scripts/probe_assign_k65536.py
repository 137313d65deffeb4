"""C5-shaped coarse quantiser: exact assign of int8 rows (d = 128) against nlist = 65,536 centroids on one GPU -- the MFMA
surrogate + exact re-check path against the exact VALU kernels on a slice (ids must be equal).  This is the measurement behind
DESIGN.md's "no 2-level coarse quantiser needed" (reference: SimpleIndex, lance-index/src/vector/utils.rs:47-108)."""
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(tag, n):
    import torch
    import lance_amd
    eng = lance_amd.default_engine()
    g = torch.Generator(device="cuda").manual_seed(7)
    k, d = 65536, 128
    cent = (torch.randn((k, d), device="cuda", generator=g) * 40.0).contiguous()
    x = torch.clamp(cent[torch.randint(0, k, (n,), device="cuda", generator=g)] + torch.randn((n, d), device="cuda", generator=g) * 12.0,
                    -128, 127).to(torch.int8).contiguous()
    ids, dd = eng.assign(x, cent)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        ids, dd = eng.assign(x, cent)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print(json.dumps({"path": tag, "rows": n, "nlist": k, "d": d, "dtype": "int8", "ms": min(ts) * 1e3,
                      "ms_per_million_rows": min(ts) * 1e3 * 1e6 / n, "ids_sum": int(ids.long().sum().item()),
                      "dist_bits_sum": int(dd.view(torch.int32).long().sum().item())}))


if __name__ == "__main__":
    if len(sys.argv) > 2:
        run(sys.argv[1], int(sys.argv[2]))
    else:
        env = dict(os.environ)
        subprocess.check_call([sys.executable, __file__, "mfma", "1000000"], env=env)
        subprocess.check_call([sys.executable, __file__, "mfma", "100000"], env=env)
        env["LANCE_HIP_NO_MFMA"] = "1"
        subprocess.check_call([sys.executable, __file__, "exact", "100000"], env=env)
