"""Per-phase shader clocks of the partition-major main pass (LANCE_HIP_PM_PROF=1 makes the library print them)."""
import os, sys
os.environ["LANCE_HIP_PM_PROF"] = "1"
import torch
sys.path.insert(0, ".")
import lance_amd
from lance_amd.testing import sift_like
eng = lance_amd.default_engine()
x = sift_like(1_000_000, 128, 1234, device="cuda")
q = sift_like(10000, 128, 4321, device="cuda")
idx = lance_amd.create_index(x, "IVF_PQ", num_partitions=256, num_sub_vectors=16)
for _ in range(3):
    idx.search_device(q, 10, 10, 10)
