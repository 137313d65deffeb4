"""C2-shaped dot search (1M x 128 f32, 256 lists, PQ16, 10,000-query batches, nprobes 10, refine 10) on ONE engine context: the list-size
skew the dot metric produces, per-stage times of the quantised flow (matrix-core bound pass + scan) against the exact pair scan
(LANCE_HIP_NO_DOT_FLOW=1 in a child), for (a) SIFT-like rows as they are (all components >= 0: the lists follow the rows' norms) and (b)
the same rows centred (x - 64: dot products of both signs, even lists).  GPU only."""
import os, subprocess, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lance_amd
from lance_amd.testing import sift_like

STAGES = ["find_partitions", "pm_group", "ivfpq_scan_c0", "q_residual", "ivfpq_scan_c1", "ivfpq_scan_cb", "ivfpq_merge", "refine"]

def run(tag, x, qn, metric, quiet=False):
    q = torch.from_numpy(qn).cuda()
    eng = lance_amd.default_engine()
    idx = lance_amd.create_index(x, "IVF_PQ", metric=metric, num_partitions=256, num_sub_vectors=16)
    sizes = np.bincount(idx.part_ids.cpu().numpy().astype(np.int64).ravel(), minlength=256) if getattr(idx, "part_ids", None) is not None else None
    for _ in range(3):
        idx.search_device(q, 10, 10, 10)
    torch.cuda.synchronize()
    ms0 = eng.timing_query("count:ivfpq_mscan")[1]
    t0 = time.perf_counter(); reps = 20
    for _ in range(reps):
        idx.search_device(q, 10, 10, 10)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    took = eng.timing_query("count:ivfpq_mscan")[1] - ms0
    line = f"{tag} {metric}: {dt * 1e3:.3f} ms per batch = {len(qn) / dt / 1e6:.2f} M q/s, matrix-core scan served {took} of {reps}"
    if sizes is not None:
        line += f"; list sizes min {sizes.min()} median {int(np.median(sizes))} max {sizes.max()}"
    if quiet:
        del idx
        return {"ms_per_batch": round(dt * 1e3, 4), "qps_one_context": round(len(qn) / dt), "matrix_core_scan_batches": int(took), "batches": reps,
                "list_sizes_min_median_max": None if sizes is None else [int(sizes.min()), int(np.median(sizes)), int(sizes.max())]}
    print(line, flush=True)
    # per-stage times (plain path, events around every stage)
    eng.timing(True)
    b = {s: eng.timing_query(s) for s in STAGES}
    for _ in range(5):
        idx.search_device(q, 10, 10, 10)
    eng.synchronize()
    a = {s: eng.timing_query(s) for s in STAGES}
    eng.timing(False)
    print("   stages (ms per batch): " + ", ".join(f"{s} {(a[s][0] - b[s][0]) / 5:.3f}" for s in STAGES if a[s][1] > b[s][1]), flush=True)
    del idx

if __name__ == "__main__":
    n, d, nq = 1_000_000, 128, 10_000
    x = sift_like(n, d, seed=1); qn = sift_like(nq, d, seed=2)
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which == "json":      # bench.py's child: unit-normalised rows, the metrics named on the command line -> one JSON line
        import json
        xu = (x / np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-9)).astype(np.float32)
        qu = (qn / np.maximum(np.linalg.norm(qn, axis=1, keepdims=True), 1e-9)).astype(np.float32)
        print(json.dumps({mt: run("unit", xu, qu, mt, quiet=True) for mt in sys.argv[2:]}), flush=True)
    elif which == "all":
        for env in ({}, {"LANCE_HIP_NO_DOT_FLOW": "1"}):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env), capture_output=True, text=True)
            print(("quantised flow" if not env else "exact pair scan (LANCE_HIP_NO_DOT_FLOW=1)") + ":\n" + "".join(l + "\n" for l in r.stdout.splitlines() if "amdgpu.ids" not in l) + r.stderr[-1500:], flush=True)
    else:
        run("as-is  ", x, qn, "l2")
        run("as-is  ", x, qn, "dot")
        run("centred", x - 64.0, qn - 64.0, "dot")
        xu = (x / np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-9)).astype(np.float32)
        qu = (qn / np.maximum(np.linalg.norm(qn, axis=1, keepdims=True), 1e-9)).astype(np.float32)
        run("unit   ", xu, qu, "dot")
        run("unit   ", xu, qu, "cosine")
        run("unit   ", xu, qu, "l2")
