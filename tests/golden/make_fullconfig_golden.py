"""Writes tests/golden/fullconfig.npz: the CPU oracle's results for BASELINE configs 3 and 5 at their real index parameters
(see tests/fullconfig_spec.py).  Run in the build container (minutes of CPU time):

    python tests/golden/make_fullconfig_golden.py [c3] [c5]
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import oracle  # noqa: E402
from fullconfig_spec import C3, C4, C5, C5T, c3_data, c4_data, c5_data, digest, f32  # noqa: E402

OUT = os.path.join(HERE, "fullconfig.npz")


def tick(t0, what):
    print(f"  {what}: {time.time() - t0:.1f} s", flush=True)
    return time.time()


def make_c3(out):
    c = C3
    x, q = c3_data()
    t = time.time()
    xs = oracle.normalize(x)
    oc = oracle.kmeans_train_hierarchical(xs, c["nlist"], max_iters=c["ivf_iters"], balance_factor_scaled=f32(1.0) / f32(c["n"]), seed=c["seed"])
    t = tick(t, "c3 hierarchical IVF training")
    assert oc.shape[0] == c["nlist"]
    part, _ = oracle.assign(xs, oc)
    res = oracle.residual(xs, oc, part)
    ocb, its = oracle.pq_train(res[:65536], c["m"], max_iters=c["pq_iters"], seed=c["seed"] + 1)
    t = tick(t, "c3 assign + residual + PQ training")
    oidx = oracle.build_index(x, oc, ocb, "cosine")
    t = tick(t, "c3 build_index")
    out["c3_centroids"] = digest(oc); out["c3_codebook"] = digest(ocb); out["c3_pq_iters"] = its.astype(np.uint32)
    out["c3_part_ids"] = digest(oidx.part_ids); out["c3_codes"] = digest(oidx.codes_rowmajor)
    out["c3_part_offsets"] = oidx.part_offsets.astype(np.uint32)
    for (k, nprobes, rf) in c["searches"]:
        oi, od = oidx.search(q, k, nprobes, refine=rf, raw=x if rf else None)
        out[f"c3_ids_{k}_{nprobes}_{rf}"] = oi; out[f"c3_dists_{k}_{nprobes}_{rf}"] = od
    oi, od = oracle.flat_knn(x, q[:50], 10, "cosine")
    out["c3_flat_ids"] = oi; out["c3_flat_dists"] = od
    tick(t, "c3 searches + flat")


def make_c5(out):
    c = C5
    xi, qi = c5_data()
    x, q = xi.astype(f32), qi.astype(f32)
    t = time.time()
    init_rows = oracle.kmeans_init_indices(c["n"], c["nlist"], c["seed"])
    # the trainers (reference and engine alike) stop at 4096 centroids per call; the C5 coarse quantiser here is the reference's
    # k-means INITIALISATION -- 65,536 distinct random rows (kmeans_random_init, kmeans.rs:149-170) -- which exercises
    # assign / encode / find_partitions / search at nlist = 65,536 all the same (duplicate rows give exactly tied centroids)
    oc = np.ascontiguousarray(x[init_rows.astype(np.int64)])
    part, _ = oracle.assign(x, oc)
    res = oracle.residual(x, oc, np.where(part == oracle.NONE, 0, part))
    ocb, pits = oracle.pq_train(res[:65536], c["m"], max_iters=c["pq_iters"], seed=c["seed"] + 1)
    t = tick(t, "c5 assign + residual + PQ training")
    oidx = oracle.build_index(x, oc, ocb, "l2")
    t = tick(t, "c5 build_index")
    out["c5_init_rows"] = init_rows.astype(np.uint64)
    out["c5_centroids"] = digest(oc); out["c5_codebook"] = digest(ocb)
    out["c5_pq_iters"] = pits.astype(np.uint32)
    out["c5_part_ids"] = digest(oidx.part_ids); out["c5_codes"] = digest(oidx.codes_rowmajor)
    out["c5_part_offsets_digest"] = digest(oidx.part_offsets.astype(np.uint32))
    pi, pd = oracle.find_partitions(q[:200], oc, 64)
    out["c5_probe_ids"] = pi; out["c5_probe_dists"] = pd
    for (k, nprobes, rf) in c["searches"]:
        oi, od = oidx.search(q, k, nprobes, refine=rf, raw=x if rf else None)
        out[f"c5_ids_{k}_{nprobes}_{rf}"] = oi; out[f"c5_dists_{k}_{nprobes}_{rf}"] = od
    tick(t, "c5 searches")


def make_c4(out):
    c = C4
    x, q = c4_data()                      # float16
    t = time.time()
    oc = oracle.kmeans_train_hierarchical(x, c["nlist"], max_iters=c["ivf_iters"], balance_factor_scaled=f32(1.0) / f32(c["n"]), seed=c["seed"])
    t = tick(t, "c4 hierarchical IVF training (f16 M-step)")
    assert oc.shape[0] == c["nlist"] and oc.dtype == np.float16
    part, _ = oracle.assign(x, oc)
    res = oracle.residual(x, oc, part)
    ocb, its = oracle.pq_train(res[:65536], c["m"], max_iters=c["pq_iters"], seed=c["seed"] + 1)
    t = tick(t, "c4 assign + residual + PQ training")
    oidx = oracle.build_index(x, oc, ocb, "l2")
    t = tick(t, "c4 build_index")
    out["c4_centroids"] = digest(oc); out["c4_codebook"] = digest(ocb); out["c4_pq_iters"] = its.astype(np.uint32)
    out["c4_part_ids"] = digest(oidx.part_ids); out["c4_codes"] = digest(oidx.codes_rowmajor)
    out["c4_part_offsets"] = oidx.part_offsets.astype(np.uint32)
    for (k, nprobes, rf) in c["searches"]:
        oi, od = oidx.search(q, k, nprobes, refine=rf, raw=x if rf else None)
        out[f"c4_ids_{k}_{nprobes}_{rf}"] = oi; out[f"c4_dists_{k}_{nprobes}_{rf}"] = od
    tick(t, "c4 searches")


def make_c5t(out):
    c, ct = C5, C5T
    xi, qi = c5_data()
    x, q = xi.astype(f32), qi.astype(f32)
    t = time.time()
    oc = oracle.kmeans_train_hierarchical(x, ct["nlist"], max_iters=ct["ivf_iters"], balance_factor_scaled=f32(1.0) / f32(c["n"]), seed=ct["seed"])
    t = tick(t, f"c5t hierarchical IVF training to {oc.shape[0]} centroids")
    out["c5t_ncent"] = np.array([oc.shape[0]], np.uint32)
    out["c5t_centroids"] = digest(oc)
    part, _ = oracle.assign(x, oc)
    res = oracle.residual(x, oc, np.where(part == oracle.NONE, 0, part))
    ocb, pits = oracle.pq_train(res[:65536], c["m"], max_iters=c["pq_iters"], seed=ct["seed"] + 1)
    oidx = oracle.build_index(x, oc, ocb, "l2")
    t = tick(t, "c5t assign + PQ + build_index")
    out["c5t_codebook"] = digest(ocb); out["c5t_part_ids"] = digest(oidx.part_ids); out["c5t_codes"] = digest(oidx.codes_rowmajor)
    for (k, nprobes, rf) in ct["searches"]:
        oi, od = oidx.search(q, k, nprobes, refine=rf, raw=x if rf else None)
        out[f"c5t_ids_{k}_{nprobes}_{rf}"] = oi; out[f"c5t_dists_{k}_{nprobes}_{rf}"] = od
    tick(t, "c5t searches")


if __name__ == "__main__":
    which = [a for a in sys.argv[1:]] or ["c3", "c5"]
    out = dict(np.load(OUT)) if os.path.exists(OUT) else {}
    if "c3" in which:
        make_c3(out)
    if "c5" in which:
        make_c5(out)
    if "c4" in which:
        make_c4(out)
    if "c5t" in which:
        make_c5t(out)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
