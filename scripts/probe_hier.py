"""Wall time of the hierarchical IVF trainer (HIER_K = 4096 lists, HIER_D = 128: a 1M x 128 f32 sample -- the C4 shape; HIER_K=1024 HIER_D=1536: C3) for the LANCE_HIP_HIER_CONTEXTS in the
environment, and that the centroids equal the library's own sequential loop bit for bit.  GPU only."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lance_amd import vector as lv
from lance_amd.vector import default_engine

k = int(os.environ.get("HIER_K", "4096")); d = int(os.environ.get("HIER_D", "128"))
rng = np.random.default_rng(7)
cent = rng.standard_normal((512, d)).astype(np.float32) * 3
x = (cent[rng.integers(0, 512, k * 256)] + rng.standard_normal((k * 256, d)).astype(np.float32)).astype(np.float32)
xd = torch.from_numpy(x).cuda()
eng = default_engine()
p = lv.IvfPqParams(num_partitions=k, num_sub_vectors=16)
out = []
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    c, _, _ = lv.train_ivf_centroids(xd, p, engine=eng)
    torch.cuda.synchronize(); out.append(time.perf_counter() - t0)
nctx = len(lv._hier_engine_pool(eng) or [0])
os.environ["LANCE_HIP_HIER_CONTEXTS"] = "1"
lv._HIER_POOLS.clear()
torch.cuda.synchronize(); t0 = time.perf_counter()
c1, _, _ = lv.train_ivf_centroids(xd, p, engine=eng)
torch.cuda.synchronize(); t1 = time.perf_counter() - t0
print("contexts", nctx, "k", k, "sec", [round(t, 3) for t in out], "sequential", round(t1, 3),
      "bit-identical", bool(torch.equal(c.view(torch.int32), c1.view(torch.int32))))
