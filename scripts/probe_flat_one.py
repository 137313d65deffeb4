"""One query against 1M x 128 f32 rows (C1, SURVEY 8d: N*d*4 bytes per query): wall time per call, for the kernel trace."""
import sys, time
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lance_amd
from lance_amd.testing import sift_like
eng = lance_amd.default_engine()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
x = sift_like(n, 128, 1234, device="cuda")
q = sift_like(64, 128, 4321, device="cuda")
for nq in (1, 2, 4):
    for k in (10, 100):
        qs = q[:nq].contiguous()
        eng.flat_topk(x, qs, k); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50): eng.flat_topk(x, qs, k)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 50
        print(f"nq={nq} k={k}: {dt*1e3:.4f} ms per call  {x.numel()*4/dt/1e9:.0f} GB/s", flush=True)
