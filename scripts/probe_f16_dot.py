"""Float16 column under the dot metric: exact assign in the 32-lane order of dot_scalar::<f16, f32, 32> -- the MFMA surrogate +
32-lane exact re-check (rows read as f16) against the two-pass exact kernel of wide.hip (LANCE_HIP_NO_MFMA=1); ids and
distance bits must agree.  Also the f16 L2 assign for scale."""
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
    g = torch.Generator(device="cuda").manual_seed(3)
    out = {"path": tag}
    for n, k in ((1_000_000, 256), (1_000_000, 4096)):
        cent = (torch.randn((k, 128), device="cuda", generator=g) * 2.0).to(torch.float16).contiguous()
        x = (cent[torch.randint(0, k, (n,), device="cuda", generator=g)].float() + torch.randn((n, 128), device="cuda", generator=g) * 0.7)
        x = x.to(torch.float16).contiguous()
        for metric in ("dot", "l2"):
            ids, dd = eng.assign(x, cent, metric)
            torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                ids, dd = eng.assign(x, cent, metric)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            out[f"f16_{metric}_1Mx128_k{k}"] = {"ms": min(ts) * 1e3, "ids_sum": int(ids.long().sum().item()),
                                                "dist_bits_sum": int(dd.view(torch.int32).long().sum().item())}
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        env = dict(os.environ)
        subprocess.check_call([sys.executable, __file__, "mfma"], env=env)
        env["LANCE_HIP_NO_MFMA"] = "1"
        subprocess.check_call([sys.executable, __file__, "exact"], env=env)
