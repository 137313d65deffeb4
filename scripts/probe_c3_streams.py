"""C3-shaped search (1M x 1536 f32 cosine, IVF_PQ nlist 1024, M 96): 1000-query batches enqueued alternately on 1 / 2 / 3 engine
contexts (own HIP stream + scratch each, one read-only index) -- the arrangement bench.py uses for the C2 headline, so that the
launch gaps and the latency-bound tail of one batch overlap the scan of the next.  GPU only.
Usage: python scripts/probe_c3_streams.py [n_rows] [queries per batch]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lance_amd
from lance_amd.engine import Engine
from lance_amd.testing import sift_like

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
NQ = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
d, nlist, m = 1536, 1024, 96
dev = torch.device("cuda", 0)
eng = lance_amd.default_engine()
x = torch.nn.functional.normalize(sift_like(n, d, seed=77, device=dev, n_clusters=1024, latent=48, model_seed=77) - 64.0, dim=1).contiguous()
qs = [torch.nn.functional.normalize(sift_like(NQ, d, seed=78 + i, device=dev, n_clusters=1024, latent=48, model_seed=77) - 64.0, dim=1).contiguous()
      for i in range(4)]
idx = lance_amd.create_index(x, "IVF_PQ", metric="cosine", num_partitions=nlist, num_sub_vectors=m)
torch.cuda.synchronize()
engines = [eng] + [Engine(device=eng.device) for _ in range(2)]
outs = [(torch.empty((NQ, 10), dtype=torch.int64, device=dev), torch.empty((NQ, 10), dtype=torch.float32, device=dev)) for _ in range(3)]
ref = {}
out = {"n": n, "queries_per_batch": NQ, "batches_timed": 24}
for nprobes, rf in ((10, 0), (10, 10), (50, 10)):
    ref_ids = [idx.search_device(q, 10, nprobes, rf)[0].clone() for q in qs]
    row = {}
    for ns in (1, 2, 3):
        def step(i):
            idx.search_device(qs[i % 4], 10, nprobes, rf, out=outs[i % ns], sync=False, engine=engines[i % ns])
        for i in range(6):
            step(i)
        for e in engines:
            e.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(24):
            step(i)
        for e in engines:
            e.synchronize()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 24
        # the last batch each context produced equals the single-context answer
        same = all(bool((outs[(23 - j) % ns][0] == ref_ids[(23 - j) % 4]).all()) for j in range(ns))
        row[f"contexts_{ns}"] = {"ms_per_batch": round(dt * 1e3, 4), "qps": round(NQ / dt), "ids_equal_single_context": same}
    out[f"nprobes{nprobes}_refine{rf}"] = row
print(json.dumps(out))
