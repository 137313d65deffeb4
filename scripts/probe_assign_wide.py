"""C3-shaped coarse quantiser: exact assign of 1M x 1536 f32 rows against 1024 centroids -- the K-tiled MFMA surrogate + exact
re-check against the exact VALU kernel (LANCE_HIP_NO_MFMA_WIDE=1); ids and distance bits must agree."""
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(tag):
    import torch
    import lance_amd
    eng = lance_amd.default_engine()
    g = torch.Generator(device="cuda").manual_seed(5)
    n, k, d = 1_000_000, 1024, 1536
    cent = torch.nn.functional.normalize(torch.randn((k, d), device="cuda", generator=g), dim=1).contiguous()
    x = torch.nn.functional.normalize(cent[torch.randint(0, k, (n,), device="cuda", generator=g)] + torch.randn((n, d), device="cuda", generator=g) * 0.05,
                                      dim=1).contiguous()
    out = {"path": tag, "rows": n, "nlist": k, "d": d}
    for metric in ("l2", "dot"):
        ids, dd = eng.assign(x, cent, metric)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            ids, dd = eng.assign(x, cent, metric)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        out[metric] = {"ms": min(ts) * 1e3, "ids_sum": int(ids.long().sum().item()), "dist_bits_sum": int(dd.view(torch.int32).long().sum().item())}
        if tag == "mfma":
            eng.timing(True)
            eng.assign(x, cent, metric)
            eng.synchronize()
            eng.timing(False)
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        env = dict(os.environ)
        subprocess.check_call([sys.executable, __file__, "mfma"], env=env)
        env["LANCE_HIP_NO_MFMA_WIDE"] = "1"
        subprocess.check_call([sys.executable, __file__, "exact"], env=env)
