"""Generates the committed fixtures of tests/golden/ (run in the build container, where /root/reference exists).

  ref_torch_assign.npz  -- outputs of the REFERENCE's own Python accelerator kernels
                           (python/python/lance/torch/distance.py: l2_distance :214-243, pairwise_l2 :134-175,
                           dot_distance :246-266), imported from /root/reference and run on torch-CPU.  The inputs are
                           small integers stored as f32, so every f32 operation is exact and the result does not depend
                           on the summation order: the reference's matmul formulation and the Rust path's lane-ordered
                           sums (what the oracle and the HIP kernels restate) must agree bit for bit.  Rows whose minimum
                           is tied are dropped (torch.min does not promise the first index; the Rust argmin does).
                           A second, non-integer case is kept for a tolerance check (1e-4 relative).
  e2e_small.npz         -- a seeded end-to-end IVF_PQ case (train -> encode -> search) produced by the CPU oracle
                           (oracle/lance_oracle.c).  It freezes the oracle: the CPU suite re-runs the oracle against
                           it (catches drift from compiler/host changes), the GPU suite checks the HIP path against it.

Only the native module of pylance is missing in this image, so `lance` itself cannot be imported; the three pure-Python
files needed (lance/dependencies.py, lance/log.py, lance/torch/{__init__,distance}.py) are loaded through a stub parent
package.  Nothing here is read at test time on the GPU box -- only the .npz files are.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_PY = "/root/reference/python/python"


def load_reference_distance():
    pkg = types.ModuleType("lance")
    pkg.__path__ = [os.path.join(REF_PY, "lance")]          # sub-modules resolve into the reference tree
    sys.modules["lance"] = pkg
    import importlib
    return importlib.import_module("lance.torch.distance")


def make_ref_torch_assign():
    import torch
    dist = load_reference_distance()
    rng = np.random.default_rng(20240521)
    out = {}
    # exact (integer-valued) case
    x = rng.integers(0, 16, (600, 32)).astype(np.float32)
    c = rng.integers(0, 16, (64, 32)).astype(np.float32)
    tx, tc = torch.from_numpy(x), torch.from_numpy(c)
    full = dist.pairwise_l2(tx, tc).numpy()                 # exact integers
    ids, _ = dist.l2_distance(tx, tc)                       # cdist -> min -> pow(2): ids exact, dists sqrt-rounded
    ids = ids.numpy()
    srt = np.sort(full, axis=1)
    keep = srt[:, 0] < srt[:, 1]                            # untied minimum only
    assert (full[np.arange(len(x)), ids][keep] == srt[:, 0][keep]).all()
    out["int_x"], out["int_c"] = x[keep], c
    out["int_l2_ids"] = ids[keep].astype(np.uint32)
    out["int_l2_min"] = srt[:, 0][keep].astype(np.float32)
    out["int_l2_matrix"] = full[keep].astype(np.float32)
    ddot = dist.dot_distance(tx, tc)
    ddot = (ddot[0] if isinstance(ddot, tuple) else ddot)
    # dot_distance returns (part ids, dists) of the argmin of 1 - x.y
    dfull = 1.0 - x.astype(np.float64) @ c.astype(np.float64).T       # exact in f64 and in f32 (|values| < 2^24)
    ds = np.sort(dfull, axis=1)
    keepd = ds[:, 0] < ds[:, 1]
    did, dval = dist.dot_distance(tx, tc)
    assert (did.numpy()[keepd] == dfull.argmin(1)[keepd]).all()
    out["int_dot_x"] = x[keepd]
    out["int_dot_ids"] = did.numpy()[keepd].astype(np.uint32)
    out["int_dot_min"] = dval.numpy()[keepd].astype(np.float32)
    # tolerance case
    xf = rng.standard_normal((512, 40)).astype(np.float32)
    cf = rng.standard_normal((48, 40)).astype(np.float32)
    idf, df = dist.l2_distance(torch.from_numpy(xf), torch.from_numpy(cf))
    out["f_x"], out["f_c"] = xf, cf
    out["f_l2_ids"] = idf.numpy().astype(np.uint32)
    out["f_l2_min"] = df.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "ref_torch_assign.npz"), **out)
    print("ref_torch_assign.npz:", {k: v.shape for k, v in out.items()})


def make_e2e_small():
    sys.path.insert(0, ROOT)
    import oracle as orc
    rng = np.random.default_rng(7)
    n, d, nlist, m, k = 6000, 32, 16, 8, 10
    centers = rng.standard_normal((24, d)) * 2.0
    x = (centers[rng.integers(0, 24, n)] + rng.standard_normal((n, d))).astype(np.float32)
    q = (centers[rng.integers(0, 24, 48)] + rng.standard_normal((48, d))).astype(np.float32)
    out = {"x": x, "q": q}
    for metric in ("l2", "dot"):
        init = x[orc.kmeans_init_indices(n, nlist, 99)]
        cent, loss, iters = orc.kmeans_train(x, nlist, metric="l2", max_iters=20, tol=1e-4, balance_factor=1.0 / n, init=init, seed=5)[:3]
        part, _ = orc.assign(x, cent, metric="l2")
        res = orc.residual(x, cent, part) if metric == "l2" else x
        cb, _ = orc.pq_train(res, m, max_iters=12, seed=11)[:2]
        idx = orc.build_index(x, cent, cb, metric=metric)
        ids0, d0 = idx.search(q, k, 4, refine=0)
        ids1, d1 = idx.search(q, k, 4, refine=5, raw=x)
        out.update({f"{metric}_init": init, f"{metric}_part_ids": idx.part_ids, f"{metric}_codes": idx.codes_rowmajor,
                    f"{metric}_centroids": cent, f"{metric}_codebook": cb, f"{metric}_offsets": idx.part_offsets,
                    f"{metric}_codes_t": idx.codes_t, f"{metric}_row_ids": idx.row_ids,
                    f"{metric}_ids_np4": ids0, f"{metric}_dists_np4": d0, f"{metric}_ids_np4_rf5": ids1, f"{metric}_dists_np4_rf5": d1,
                    f"{metric}_ivf_loss": np.float64(loss), f"{metric}_ivf_iters": np.int64(iters)})
    np.savez_compressed(os.path.join(HERE, "e2e_small.npz"), **out)
    print("e2e_small.npz:", {k_: getattr(v, "shape", ()) for k_, v in out.items()})


if __name__ == "__main__":
    make_ref_torch_assign()
    make_e2e_small()
