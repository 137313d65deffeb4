import sys
import numpy as np
sys.path.insert(0, ".")
import oracle
from lance_amd.engine import Engine, DeviceFlatIndex
eng = Engine()
f32 = np.float32
def sift_like(n, d, seed, ncl=32):
    rng = np.random.default_rng(seed)
    centers = rng.uniform(0, 128, (ncl, d))
    x = centers[rng.integers(0, ncl, n)] + rng.normal(0, 24, (n, d))
    return np.clip(np.rint(x), 0, 218).astype(f32)
d, metric = 40, "l2"
n, nlist = 20000, 24
x = sift_like(n, d, 90 + d); x[50:60] = x[7]
q = sift_like(150, d, 91 + d)
cent, _, _, _ = oracle.kmeans_train(x[:4096], nlist, max_iters=6, seed=2)
part, _ = eng.assign(x, cent, metric)
g = DeviceFlatIndex.create(eng, metric, cent, x, part)
gi, gd = g.search(q, 50, nlist)
gi = gi.cpu().numpy().view(np.uint64); gd = gd.cpu().numpy()
oi, od = oracle.ivfflat_search(x, cent, q, 50, nlist, metric)
bad = np.argwhere(gi != oi)
print("mismatches", len(bad), bad[:10])
for qi, j in bad[:3]:
    print(qi, j, gi[qi, max(0,j-2):j+3], gd[qi, max(0,j-2):j+3], oi[qi, max(0,j-2):j+3], od[qi, max(0,j-2):j+3])
fi, fd = oracle.flat_knn(x, q, 50, metric)
print("oracle ivfflat == oracle flat:", (oi == fi).all(), " gpu == oracle flat:", (gi == fi).all())
