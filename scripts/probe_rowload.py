"""How fast does the lanes-own-rows shape stream rows?  assign() of 1M x 128 against k = 1..64 centroids."""
import sys, time
import torch
sys.path.insert(0, ".")
import lance_amd
from lance_amd.testing import sift_like
eng = lance_amd.default_engine()
x = sift_like(1_000_000, 128, 1234, device="cuda")
for k in (1, 4, 16, 64, 256):
    c = x[:k].clone()
    eng.assign(x, c); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): eng.assign(x, c)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"k={k}: {dt*1e3:.3f} ms  rows {x.numel()*4/dt/1e9:.0f} GB/s  {1e6*k*128*3/dt/1e12:.1f} Tops/s", flush=True)
