"""C3-shaped search (1M x 1536 f32 cosine, IVF_PQ nlist 1024, M 96) with the per-kernel HIP-event timers on: where a 1000-query
batch spends its time at (nprobes, refine) = (10, 0) / (10, 10) / (50, 10).  LANCE_HIP_Q_STATS=1 adds the filter's survivor counts.
GPU only.  Usage: python scripts/probe_c3_search.py [n_rows] [queries_per_batch]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lance_amd
from lance_amd.testing import sift_like

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
d, nlist, m = 1536, 1024, 96
dev = torch.device("cuda", 0)
eng = lance_amd.default_engine()
x = torch.nn.functional.normalize(sift_like(n, d, seed=77, device=dev, n_clusters=1024, latent=48, model_seed=77) - 64.0, dim=1).contiguous()
q = torch.nn.functional.normalize(sift_like(nq, d, seed=78, device=dev, n_clusters=1024, latent=48, model_seed=77) - 64.0, dim=1).contiguous()
idx = lance_amd.create_index(x, "IVF_PQ", metric="cosine", num_partitions=nlist, num_sub_vectors=m)
torch.cuda.synchronize()
names = ("dist_matrix", "select_probes", "pm_group", "ivfpq_scan", "ivfpq_scan_c0", "q_residual", "ivfpq_scan_c1", "q_pt_tables", "q_pt_table_only", "ivfpq_scan_cb",
         "ivfpq_merge", "ivfpq_exact", "refine")
out = {"n": n, "queries_per_batch": nq, "build_stages_ms": {k: round(v * 1e3, 2) for k, v in idx.stats.seconds.items()}}
CFGS = ((10, 0), (10, 10), (50, 10))
wall = {}
outb = (torch.empty((q.shape[0], 10), dtype=torch.int64, device=dev), torch.empty((q.shape[0], 10), dtype=torch.float32, device=dev))
for nprobes, rf in CFGS:
    for _ in range(3):      # (LANCE_HIP_GRAPH=1: plain, capture, replay)
        idx.search_device(q, 10, nprobes, rf, out=outb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        idx.search_device(q, 10, nprobes, rf, out=outb)
    torch.cuda.synchronize()
    wall[(nprobes, rf)] = (time.perf_counter() - t0) / 5 * 1e3
# one context, batches enqueued back to back (asynchronous entry point), one synchronisation at the end
awall = {}
for nprobes, rf in CFGS:
    for _ in range(3):
        idx.search_device(q, 10, nprobes, rf, out=outb, sync=False)
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        idx.search_device(q, 10, nprobes, rf, out=outb, sync=False)
    eng.synchronize()
    awall[(nprobes, rf)] = (time.perf_counter() - t0) / 20 * 1e3
eng.timing(True)
for nprobes, rf in CFGS:
    base = {k: eng.timing_query(k) for k in names}
    for _ in range(5):
        idx.search_device(q, 10, nprobes, rf)
    eng.synchronize()
    kt = {}
    for k in names:
        ms, cnt = eng.timing_query(k)
        if cnt > base[k][1]:
            kt[k] = round((ms - base[k][0]) / 5, 4)
    out[f"nprobes{nprobes}_refine{rf}"] = {"wall_ms_per_batch": round(wall[(nprobes, rf)], 4), "async_ms_per_batch": round(awall[(nprobes, rf)], 4),
                                          "qps_async_one_context": round(q.shape[0] / awall[(nprobes, rf)] * 1e3), "kernel_ms_per_batch": kt,
                                          "sum_ms": round(sum(kt.values()), 4), "exact_replays": eng.search_stats()}
eng.timing(False)
print(json.dumps(out))
