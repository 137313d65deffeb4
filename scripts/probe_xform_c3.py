"""C3-shaped transform (1M x 1536 f32, cosine, IVF1024, PQ96): wall time of lance_hip_ivfpq_encode; run under rocprofv3 --kernel-trace --stats
for the kernel breakdown.  usage: python scripts/probe_xform_c3.py [n]"""
import json, os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import lance_amd
eng = lance_amd.default_engine()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d, nlist, m = 1536, 1024, 96
g = torch.Generator(device="cuda").manual_seed(7)
cl = torch.randn((2048, d), device="cuda", generator=g)
x = torch.empty((n, d), device="cuda")
for a in range(0, n, 100_000):
    b = min(n, a + 100_000)
    x[a:b] = cl[torch.randint(0, 2048, (b - a,), device="cuda", generator=g)] + 0.35 * torch.randn((b - a, d), device="cuda", generator=g)
xn = eng.normalize(x[:200_000])
cent = xn[torch.randperm(200_000, device="cuda", generator=g)[:nlist]].contiguous()
part, _ = eng.assign(xn, cent, "l2")
res = xn - cent[part.long()]
cb = torch.stack([res[torch.randperm(200_000, device="cuda", generator=g)[:256]][:, i * 16:(i + 1) * 16] for i in range(m)]).contiguous()
del xn, res
for _ in range(2):
    eng.ivfpq_encode(x, cent, cb, "cosine", want_loss=False)
ts = []
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    part, codes, _ = eng.ivfpq_encode(x, cent, cb, "cosine", want_loss=False)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("C3_XFORM", json.dumps({"n": n, "ms": [round(t * 1e3, 2) for t in ts], "codes_sum": int(codes.long().sum().item()), "part_sum": int(part.long().sum().item())}), flush=True)
