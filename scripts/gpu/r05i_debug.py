"""gpurun r05i: what the zero rows / zero query do under cosine in the flat scan -- oracle (x86 arithmetic) against the exact GPU kernel and the
matrix-core filter path.  Prints the first queries' ids / distance bits side by side."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import oracle
import lance_amd

f32 = np.float32
eng = lance_amd.default_engine()
rng = np.random.default_rng(9)
n, d = 20_000, 256
centers = rng.normal(0, 1.0, (24, d))
x = (centers[rng.integers(0, 24, n)] + rng.normal(0, 0.35, (n, d))).astype(f32)
q = (centers[rng.integers(0, 24, 160)] + rng.normal(0, 0.35, (160, d))).astype(f32)
which = sys.argv[1]
if which == "rows":
    x[17] = 0.0; x[9000] = 0.0
elif which == "inf":
    x[31] = 3e19
elif which == "tiny":
    x[32] *= f32(1e-20)
elif which == "query":
    q[5] = 0.0
gi, gd = eng.flat_topk(x, q, 10, "cosine")
oi, od = oracle.flat_knn(x, q, 10, "cosine")
gi = gi.cpu().numpy().view(np.uint64); gd = gd.cpu().numpy()
bad = np.nonzero((gi != oi).any(axis=1) | (gd.view(np.uint32) != od.view(np.uint32)).any(axis=1))[0]
print(which, "mode", "exact" if os.environ.get("LANCE_HIP_NO_MFMA_FLAT_WIDE") else "mfma", "differing queries:", bad.size, bad[:8])
for qi in list(bad[:2]):
    print(" q", qi, "gpu ids", gi[qi][:6], "bits", [hex(v) for v in gd[qi].view(np.uint32)[:6]])
    print(" q", qi, "ora ids", oi[qi][:6], "bits", [hex(v) for v in od[qi].view(np.uint32)[:6]])
