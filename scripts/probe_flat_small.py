"""C1 single-query flat scan (1M x 128 f32): flat_small.hip's one-pass kernel against the batch path (LANCE_HIP_NO_FLAT_SMALL=1 in a
second run), host-timed over repeated synchronous calls and by the "flat_scan" HIP-event timer.  GPU only."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lance_amd
from lance_amd.testing import sift_like

eng = lance_amd.default_engine()
x = sift_like(1_000_000, 128, 1234, device="cuda")
q = sift_like(16, 128, 4321, device="cuda")
out = {"small_path": os.environ.get("LANCE_HIP_NO_FLAT_SMALL") is None}
for nq in (1, 2, 4, 8):
    qs = q[:nq].contiguous()
    for _ in range(3):
        eng.flat_topk(x, qs, 10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        eng.flat_topk(x, qs, 10)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 20
    eng.timing(True)
    for _ in range(10):
        eng.flat_topk(x, qs, 10)
    ms, cnt = eng.timing_query("flat_scan")
    eng.timing(False)
    out[f"nq{nq}"] = {"wall_ms": round(wall * 1e3, 4), "flat_scan_event_ms_per_call": round(ms / 10, 4), "timer_intervals": cnt,
                      "GBps_wall": round(x.numel() * 4 / wall / 1e9, 1), "GBps_kernels": round(x.numel() * 4 / (ms / 10 * 1e-3) / 1e9, 1) if ms else None}
print(json.dumps(out))
