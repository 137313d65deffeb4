"""C3-shaped build (1M x 1536 f32 cosine, IVF_PQ nlist 1024 -> hierarchical k-means, M 96): stage times, best of 2.  GPU only."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lance_amd
from lance_amd.testing import sift_like

dev = torch.device("cuda", 0)
n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 1536
x = torch.nn.functional.normalize(sift_like(n, d, seed=77, device=dev, n_clusters=1024, latent=48, model_seed=77) - 64.0, dim=1).contiguous()
best = None
for _ in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    idx = lance_amd.create_index(x, "IVF_PQ", metric="cosine", num_partitions=1024, num_sub_vectors=96)
    torch.cuda.synchronize(); w = time.perf_counter() - t
    st = {k: round(v * 1e3, 2) for k, v in idx.stats.seconds.items()}
    st["wall_ms"] = round(w * 1e3, 2)
    if best is None or st["wall_ms"] < best["wall_ms"]:
        best = st
    idx.close() if hasattr(idx, "close") else None
print(json.dumps({"graph": os.environ.get("LANCE_HIP_KMEANS_GRAPH", "0"), **best}))
