import os, sys, time
import torch
sys.path.insert(0, ".")
import lance_amd
from lance_amd.testing import sift_like
eng = lance_amd.default_engine()
x = sift_like(1_000_000, 128, 1234, device="cuda")
q = sift_like(10000, 128, 4321, device="cuda")
idx = lance_amd.create_index(x, "IVF_PQ", num_partitions=256, num_sub_vectors=16)
def run(nprobes, rf, reps=10):
    for _ in range(3): idx.search_device(q, 10, nprobes, rf, sync=False)
    eng.synchronize(); eng.timing(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): idx.search_device(q, 10, nprobes, rf, sync=False)
    eng.synchronize(); dt = (time.perf_counter() - t0) / reps
    eng.timing(False)
    print('  replays', eng.search_stats(), end=' ')
    return round(dt * 1e3, 3), {k: round(eng.timing_query(k)[0]/reps, 3) for k in ("dist_matrix","select_probes","pm_group","ivfpq_scan","ivfpq_merge","ivfpq_exact","refine")}
print("persist" if not os.environ.get("LANCE_HIP_NO_PERSIST") else "plain", "np10/rf10", run(10, 10), "np50", run(50, 0, 4), "np1", run(1, 0), flush=True)
# flagged-query census over different batches
for s in range(6):
    qq = sift_like(10000, 128, 5000 + s, device="cuda")
    idx.search_device(qq, 10, 10, 10)
    print("batch", s, "replays", eng.search_stats(), flush=True)
