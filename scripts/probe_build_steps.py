"""Where the build's wall time goes, step by step (one MI355X, BASELINE config 2): every host-level step of train_ivf_centroids /
train_pq_codebook / the transform timed with a device synchronisation on both sides.  Prints JSON lines (best of 3)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lance_amd
from lance_amd import vector as lv
from lance_amd.testing import sift_like

eng = lance_amd.default_engine()
dev = torch.device("cuda")
x = sift_like(1_000_000, 128, seed=1234, device=dev)
params = lv.IvfPqParams(256, 16, 8, "l2", 50, 256, 42)


def timed(fn, reps=3):
    best, out = 1e9, None
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); out = fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    return out, best * 1e3


steps = {}
idx, steps["ivf: numpy sample indices"] = timed(lambda: lv._sample_rows(x.shape[0], 256 * 256, np.random.default_rng(42)))
sample, steps["ivf: gather sample"] = timed(lambda: x[torch.from_numpy(idx).to(dev)])
sample, steps["ivf: isfinite filter"] = timed(lambda: sample[torch.isfinite(sample).all(dim=1)])
(cent, loss, iters), steps["ivf: kmeans_train"] = timed(lambda: eng.kmeans_train(sample, 256, max_iters=50, balance_factor=1.0, seed=42, metric="l2"))
steps["ivf: iterations"] = int(iters)
idx2, steps["pq: numpy sample indices"] = timed(lambda: lv.pq_sample_indices(x.shape[0], params))
s2, steps["pq: gather sample"] = timed(lambda: x[torch.from_numpy(idx2).to(dev)])
s2, steps["pq: isfinite filter"] = timed(lambda: s2[torch.isfinite(s2).all(dim=1)])
(part, _), steps["pq: assign sample"] = timed(lambda: eng.assign(s2, cent, "l2"))
res, steps["pq: residual"] = timed(lambda: eng.residual(s2, cent, part))
(cb, pq_iters), steps["pq: pq_train"] = timed(lambda: eng.pq_train(res, 16, 8, 50, 256, 44))
steps["pq: iterations"] = [int(v) for v in np.asarray(pq_iters).reshape(-1)] if pq_iters is not None else None
(_, _, _), steps["transform: ivfpq_encode"] = timed(lambda: eng.ivfpq_encode(x, cent, cb, "l2"))
import ctypes as C
out = np.empty(256, np.uint64)
t = time.perf_counter()
for s in range(16):
    eng.lib.lance_hip_kmeans_init_indices(C.c_uint64(65536), C.c_uint32(256), C.c_uint64(100 + s), out.ctypes.data_as(C.c_void_p))
steps["host: 16 reservoir initialisations, one thread"] = (time.perf_counter() - t) * 1e3
steps["host: hardware threads"] = os.cpu_count()
eng.timing(True)
for name in ("pq_mfma_estep", "kmeans_mstep"):
    eng.timing_query(name)
eng.pq_train(res, 16, 8, 50, 256, 44)
steps["pq_train kernel ms by stage"] = {name: round(eng.timing_query(name)[0], 3) for name in ("pq_mfma_estep", "kmeans_mstep")}
eng.timing(False)
print(json.dumps({k_: (round(v, 3) if isinstance(v, float) else v) for k_, v in steps.items()}))
