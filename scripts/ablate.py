import os, sys, time
import torch
sys.path.insert(0, ".")
import lance_amd
from lance_amd.testing import sift_like
eng = lance_amd.default_engine()
x = sift_like(1_000_000, 128, 1234, device="cuda")
q = sift_like(10000, 128, 4321, device="cuda")
idx = lance_amd.create_index(x, "IVF_PQ", num_partitions=256, num_sub_vectors=16)
idx = lance_amd.create_index(x, "IVF_PQ", num_partitions=256, num_sub_vectors=16)
print("build ms", {k: round(v*1e3,2) for k,v in idx.stats.seconds.items()}, "iters", idx.stats.ivf_iters, flush=True)
def run(nprobes, rf):
    for _ in range(2): idx.search_device(q, 10, nprobes, rf, sync=False)
    eng.synchronize(); eng.timing(True)
    for _ in range(5): idx.search_device(q, 10, nprobes, rf, sync=False)
    eng.synchronize(); eng.timing(False)
    return {k: round(eng.timing_query(k)[0]/5, 3) for k in ("dist_matrix","select_probes","ivfpq_scan","ivfpq_merge","refine")}
for ab in ("0", "1", "2", "3", "4", "5"):
    os.environ["LANCE_HIP_ABLATE"] = ab
    print("ablate", ab, "np10:", run(10, 10), "np50:", run(50, 0), flush=True)
os.environ["LANCE_HIP_ABLATE"] = "0"
c = x[:256].clone()
for n in (65536, 1_000_000):
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(5): eng.assign(x[:n], c)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/5
    print(f"assign n={n}: {dt*1e3:.3f} ms {n*256*400/dt/1e12:.1f} Tops")
