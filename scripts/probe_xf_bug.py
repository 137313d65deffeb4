"""Is the one wrong code of the fused transform (row 351773, sub-quantiser 11 at the C2 probe shape) deterministic?"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import lance_amd
from lance_amd.testing import sift_like
eng = lance_amd.default_engine()
x = sift_like(1_000_000, 128, 1234, device="cuda")
g = torch.Generator(device="cuda").manual_seed(1)
cent = x[torch.randperm(1_000_000, device="cuda", generator=g)[:256]].contiguous()
part, _ = eng.assign(x[:200000], cent, "l2")
res = x[:200000] - cent[part.long()]
cb = torch.stack([res[torch.randperm(200000, device="cuda", generator=g)[:256]][:, i * 8:(i + 1) * 8] for i in range(16)]).contiguous()
runs = []
for i in range(4):
    p, c, _ = eng.ivfpq_encode(x, cent, cb, "l2")
    runs.append(c.cpu().numpy())
    print("run", i, "code[351773, 11] =", runs[-1][351773, 11], "sum", int(runs[-1].astype(np.int64).sum()))
for i in range(1, 4):
    r, m = np.nonzero(runs[i] != runs[0])
    print("run", i, "vs run 0:", len(r), "items differ", list(zip(r[:5], m[:5])))
for lo, hi in ((351744 - 1024, 351744 + 1024), (351744, 351744 + 2048), (350000, 354096)):
    p, c, _ = eng.ivfpq_encode(x[lo:hi].contiguous(), cent, cb, "l2")
    print("window", lo, hi, "code =", int(c[351773 - lo, 11]))
