import os, sys
import torch
sys.path.insert(0, ".")
import lance_amd
from lance_amd.testing import sift_like
eng = lance_amd.default_engine()
x = sift_like(1_000_000, 128, 1234, device="cuda")
q = sift_like(10000, 128, 4321, device="cuda")
idx = lance_amd.create_index(x, "IVF_PQ", num_partitions=256, num_sub_vectors=16)
def run(nprobes, rf):
    for _ in range(2): idx.search_device(q, 10, nprobes, rf, sync=False)
    eng.synchronize(); eng.timing(True)
    for _ in range(8): idx.search_device(q, 10, nprobes, rf, sync=False)
    eng.synchronize(); eng.timing(False)
    return {k: round(eng.timing_query(k)[0]/8, 3) for k in ("ivfpq_scan_c0","ivfpq_scan_c1","ivfpq_merge","refine")}
print("bound pass:", run(10, 10), run(50, 10), flush=True)
os.environ["LANCE_HIP_PM_NOBOUND"] = "1"
print("two-class  :", run(10, 10), run(50, 10), flush=True)
