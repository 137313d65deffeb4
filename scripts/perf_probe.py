"""Quick per-stage timing probe (not the contract bench): SIFT-like 1M x 128."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import lance_amd
from lance_amd.engine import Engine

def sift_like_gpu(n, d, seed, ncl=256):
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    centers = torch.rand((ncl, d), generator=g, device="cuda") * 128
    a = torch.randint(0, ncl, (n,), generator=g, device="cuda")
    x = centers[a] + torch.randn((n, d), generator=g, device="cuda") * 24
    return torch.clamp(torch.round(x), 0, 218).float().contiguous()

def t(fn, reps=3):
    torch.cuda.synchronize(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

eng = lance_amd.default_engine()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
x = sift_like_gpu(n, 128, 1234)
q = sift_like_gpu(10000, 128, 4321)
cent = x[:256].clone()
dt = t(lambda: eng.assign(x, cent))
print(f"assign {n}x128 vs 256: {dt*1e3:.2f} ms  ({n*256*400/dt/1e12:.1f} Tops/s of 157 peak)")
dt = t(lambda: eng.kmeans_train(x[:65536], 256, max_iters=10, balance_factor=1.0, tol=0.0), reps=1)
print(f"kmeans 10 iters 65536x128 k=256: {dt*1e3:.2f} ms")
t0 = time.perf_counter()
idx = lance_amd.create_index(x, "IVF_PQ", num_partitions=256, num_sub_vectors=16)
print("create_index stages:", {k: round(v * 1e3, 2) for k, v in idx.stats.seconds.items()}, "total ms", round(idx.stats.total * 1e3, 1),
      "ivf_iters", idx.stats.ivf_iters, "pq_iters", idx.stats.pq_iters)
for nprobes, rf in ((1, 0), (10, 0), (10, 10), (50, 0), (256, 0)):
    eng.timing(True)
    dt = t(lambda: idx.search_device(q, 10, nprobes, rf), reps=2)
    parts = {k: eng.timing_query(k) for k in ("dist_matrix", "select_probes", "ivfpq_scan", "ivfpq_merge", "ivfpq_exact", "refine")}
    eng.timing(False)
    print(f"search nq=10000 k=10 nprobes={nprobes} refine={rf}: {dt*1e3:.2f} ms -> {10000/dt:.0f} QPS ",
          {k: round(v[0] / max(v[1], 1), 3) for k, v in parts.items()})
qs = q[:1000]
dt = t(lambda: eng.flat_topk(x, qs, 10), reps=1)
print(f"flat 1000 queries: {dt*1e3:.1f} ms -> {1000/dt:.0f} QPS")
gt, _ = eng.flat_topk(x, qs, 10)
for nprobes, rf in ((1, 0), (10, 0), (10, 10), (50, 10), (256, 0)):
    ids, _ = idx.search_device(qs, 10, nprobes, rf)
    rec = (ids.unsqueeze(2) == gt.unsqueeze(1)).any(dim=2).float().mean().item()
    print(f"recall@10 nprobes={nprobes} refine={rf}: {rec:.4f}")
