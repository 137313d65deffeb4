"""How many (query, probe) pairs could skip the ADC scan exactly?  LB(q,p) = sum_m min_c LUT[m][c] is a lower bound of
every row's ADC distance in partition p; if LB > T_final(q) (the k*refine-th best ADC distance) no row of p can enter."""
import sys
import torch
sys.path.insert(0, ".")
import lance_amd
from lance_amd.testing import sift_like
eng = lance_amd.default_engine()
x = sift_like(1_000_000, 128, 1234, device="cuda")
q = sift_like(10000, 128, 4321, device="cuda")
idx = lance_amd.create_index(x, "IVF_PQ", num_partitions=256, num_sub_vectors=16)
cent = idx._ix.centroids; cb = idx._ix.codebook            # [256,128], [16,256,8]
for nprobes, keff in ((10, 100), (10, 10), (50, 100)):
    probes, _ = eng.find_partitions(q, cent, nprobes)
    ids, dd = idx.search_device(q, keff, nprobes, 0)
    T = dd[:, keff - 1]                                      # final keff-th ADC distance
    r = q[:, None, :] - cent[probes.long()]                  # [nq, nprobes, d]
    rs = r.reshape(q.shape[0], nprobes, 16, 1, 8)
    lut = ((rs - cb[None, None]) ** 2).sum(-1)               # [nq, nprobes, 16, 256]
    lb = lut.min(-1).values.sum(-1)                          # [nq, nprobes]
    skip = lb > T[:, None]
    print(f"nprobes={nprobes} keff={keff}: skippable pairs {skip.float().mean().item():.3f}; by probe rank:",
          [round(v, 3) for v in skip.float().mean(0).tolist()][:12], flush=True)
    # half-LUT bound after 8 sub-quantisers for rows: fraction of rows whose partial sum over m<8 of their codes exceeds T (sampled)
