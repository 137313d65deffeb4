import sys, time
import torch
sys.path.insert(0, ".")
import lance_amd
from lance_amd.testing import sift_like
eng = lance_amd.default_engine()
x = sift_like(1_000_000, 128, 1234, device="cuda")
q = sift_like(10_000, 128, 4321, device="cuda")
for nq in (1, 10, 100, 1000, 10000):
    qs = q[:nq]
    eng.flat_topk(x, qs, 10); torch.cuda.synchronize()
    reps = 5 if nq <= 1000 else 2
    t0 = time.perf_counter()
    for _ in range(reps): eng.flat_topk(x, qs, 10)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"nq={nq}: {dt*1e3:.3f} ms  {nq/dt:.0f} QPS  {x.numel()*4/dt/1e9:.0f} GB/s/batch  {1e6*nq*128*3/dt/1e12:.1f} Tops/s", flush=True)
xs = torch.sort(x[:, 0], descending=False)[1]
# adversarial order: rows sorted by distance to q[0], farthest first
dd = ((x - q[0]) ** 2).sum(1)
xo = x[torch.argsort(dd, descending=True)].contiguous()
t0 = time.perf_counter(); i1, d1 = eng.flat_topk(xo, q[:4], 10); torch.cuda.synchronize(); print("adversarial ms", (time.perf_counter()-t0)*1e3)
