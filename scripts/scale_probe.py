"""Scale check (C4-shaped, reduced): n x 128 (f32 or f16), nlist 4096 (hierarchical k-means), M 16.
python scripts/scale_probe.py [n] [f16]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import lance_amd
from lance_amd.testing import sift_like
eng = lance_amd.default_engine()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
f16 = len(sys.argv) > 2 and sys.argv[2] == "f16"
chunks = []
for i in range(0, n, 4_000_000):
    c = sift_like(min(4_000_000, n - i), 128, seed=1000 + i, device="cuda", n_clusters=4096, latent=24)
    chunks.append((c / 256.0).half() if f16 else c)     # f16: keep the f16 M-step sums (kmeans.rs:403-406) far from 65504
x = torch.cat(chunks); del chunks
q = sift_like(2000, 128, seed=5, device="cuda", n_clusters=4096, latent=24)
if f16: q = (q / 256.0).half()
print("data", tuple(x.shape), x.dtype, f"{x.numel()*x.element_size()/1e9:.1f} GB", flush=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
idx = lance_amd.create_index(x, "IVF_PQ", metric="l2", num_partitions=4096, num_sub_vectors=16)
torch.cuda.synchronize(); bt = time.perf_counter() - t0
print(f"build {bt:.2f} s", {k: round(v * 1e3, 1) for k, v in idx.stats.seconds.items()}, flush=True)
offs, codes, rid = idx.export_storage()
sizes = np.diff(offs.astype(np.int64))
print("partitions: min/mean/max", sizes.min(), sizes.mean(), sizes.max(), " rows covered", int(sizes.sum()), " rid sum ok", int(rid.astype(np.uint64).sum()) == n * (n - 1) // 2, flush=True)
t0 = time.perf_counter(); gt, _ = eng.flat_topk(x, q[:200], 10); torch.cuda.synchronize(); print(f"flat 200 queries {1e3*(time.perf_counter()-t0):.1f} ms", flush=True)
for nprobes, rf in ((10, 0), (10, 10), (50, 10)):
    idx.search_device(q, 10, nprobes, rf); torch.cuda.synchronize()
    t0 = time.perf_counter(); ids, dd = idx.search_device(q, 10, nprobes, rf); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    rec = (ids[:200].unsqueeze(2) == gt.unsqueeze(1)).any(dim=2).float().mean().item()
    srt = bool((dd[:, 1:] >= dd[:, :-1]).all().item())
    print(f"nprobes={nprobes} refine={rf}: {dt*1e3:.2f} ms / 2000 queries = {2000/dt:.0f} QPS  recall@10={rec:.4f} sorted={srt} replays={eng.search_stats()}", flush=True)
