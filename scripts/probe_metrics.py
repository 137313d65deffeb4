"""C2-shaped IVF_PQ search (1M x 128 f32, 256 lists, PQ16, 10,000-query batches, nprobes 10, refine 10) under l2 / cosine / dot on ONE engine
context: queries/s and which scan served the batch.  The dot metric has no quantised flow: this is what that costs.  GPU only."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lance_amd
from lance_amd.testing import sift_like

n, d, nq = 1_000_000, 128, 10_000
x = sift_like(n, d, seed=1)
q = torch.from_numpy(sift_like(nq, d, seed=2)).cuda()
eng = lance_amd.default_engine()
for metric in ("l2", "cosine", "dot"):
    idx = lance_amd.create_index(x, "IVF_PQ", metric=metric, num_partitions=256, num_sub_vectors=16)
    for _ in range(3):
        out = idx.search_device(q, 10, 10, 10)
    torch.cuda.synchronize()
    ms0 = eng.timing_query("count:ivfpq_mscan")[1]
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        out = idx.search_device(q, 10, 10, 10)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{metric}: {dt * 1e3:.3f} ms per {nq}-query batch = {nq / dt / 1e6:.2f} M q/s (matrix-core scan served {eng.timing_query('count:ivfpq_mscan')[1] - ms0} of {reps} batches)", flush=True)
    del idx
