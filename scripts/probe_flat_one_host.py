"""Where the host's share of a single-query flat call goes: output allocation, the wait for torch's stream, the C call."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lance_amd
from lance_amd.testing import sift_like
from lance_amd.engine import _ptr, METRICS, check
eng = lance_amd.default_engine()
x = sift_like(1_000_000, 128, 1234, device="cuda")
q = sift_like(8, 128, 4321, device="cuda")[:1].contiguous()
k = 10
for _ in range(5): eng.flat_topk(x, q, k)
torch.cuda.synchronize()
R = 300
t = [0.0] * 5
for _ in range(R):
    a = time.perf_counter()
    ids = torch.empty((1, k), dtype=torch.int64, device=x.device); dists = torch.empty((1, k), dtype=torch.float32, device=x.device)
    b = time.perf_counter()
    torch.cuda.synchronize()
    c = time.perf_counter()
    st = torch.cuda.current_stream(x.device); done = st.query()
    d = time.perf_counter()
    check(eng.lib.lance_hip_flat_topk(eng.h, 0, METRICS["l2"], _ptr(x), _ptr(None), x.shape[0], 128, _ptr(q), 1, k, _ptr(ids), _ptr(dists)))
    e = time.perf_counter()
    t[0] += b - a; t[1] += c - b; t[2] += d - c; t[3] += e - d
print("us per call: torch.empty x2 %.2f | torch.cuda.synchronize %.2f | current_stream + query %.2f | C call %.2f" % tuple(v / R * 1e6 for v in t[:4]))
t0 = time.perf_counter()
for _ in range(R): eng.flat_topk(x, q, k)
torch.cuda.synchronize()
print("eng.flat_topk: %.2f us per call" % ((time.perf_counter() - t0) / R * 1e6))
