import sys, itertools
import numpy as np, torch
sys.path.insert(0, ".")
import lance_amd
from lance_amd.testing import sift_like
eng = lance_amd.default_engine()
for latent, within, noise in [(16, 1.0, 0.12), (16, 1.4, 0.15), (20, 1.2, 0.12)]:
    x = sift_like(1_000_000, 128, 1234, device="cuda", latent=latent, within=within, noise=noise)
    q = sift_like(1000, 128, 4321, device="cuda", latent=latent, within=within, noise=noise)
    idx = lance_amd.create_index(x, "IVF_PQ", num_partitions=256, num_sub_vectors=16)
    gt, _ = eng.flat_topk(x, q, 10)
    out = []
    for nprobes, rf in ((1, 0), (10, 0), (10, 10), (25, 10), (50, 10), (256, 0)):
        ids, _ = idx.search_device(q, 10, nprobes, rf)
        rec = (ids.unsqueeze(2) == gt.unsqueeze(1)).any(dim=2).float().mean().item()
        out.append(f"np{nprobes}/rf{rf}:{rec:.3f}")
    print(f"latent={latent} within={within} noise={noise} build={idx.stats.total*1e3:.0f}ms iters={idx.stats.ivf_iters}", " ".join(out), flush=True)
