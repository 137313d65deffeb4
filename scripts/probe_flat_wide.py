"""Timing of the batched flat scan over long rows (BASELINE config 3's ground truth: 1M x 1536 f32, cosine, 1000 queries, k = 10) with the
K-tiled matrix-core filter (flat_mfma_wide.hip) and, in a child process, with the exact kernel it replaces (LANCE_HIP_NO_MFMA_FLAT_WIDE=1).
Prints one JSON line per configuration; ids of the two runs are compared through a checksum."""
import json
import os
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(n, d, nq, metric, k=10, reps=3):
    import lance_amd
    eng = lance_amd.default_engine()
    g = torch.Generator(device="cuda").manual_seed(77)
    cent = torch.randn((1024, d), generator=g, device="cuda")
    x = cent[torch.randint(0, 1024, (n,), generator=g, device="cuda")] + 0.35 * torch.randn((n, d), generator=g, device="cuda")
    q = cent[torch.randint(0, 1024, (nq,), generator=g, device="cuda")] + 0.35 * torch.randn((nq, d), generator=g, device="cuda")
    if metric == "cosine":
        x = x / x.norm(dim=1, keepdim=True); q = q / q.norm(dim=1, keepdim=True)
    x = x.contiguous(); q = q.contiguous()
    eng.flat_topk(x[:50000], q, k, metric)      # warm-up (kernel load, scratch)
    eng.timing(True)
    eng.timing_query("flat_mfma_wide"); eng.timing_query("flat_scan")
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ids, _ = eng.flat_topk(x, q, k, metric)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    wide_ms, wide_n = eng.timing_query("flat_mfma_wide")
    scan_ms, scan_n = eng.timing_query("flat_scan")
    eng.timing(False)
    flop = 2.0 * n * nq * d
    out = {"n": n, "d": d, "nq": nq, "metric": metric, "k": k, "call_ms_best": min(ts) * 1e3, "call_ms_all": [round(t * 1e3, 2) for t in ts],
           "filter_launches_per_call": wide_n / reps, "filter_ms_per_call": wide_ms / reps, "all_epochs_ms_per_call": scan_ms / reps,
           "product_tflops_of_the_filter": flop / (wide_ms / reps * 1e-3) / 1e12 if wide_n else None,
           "ids_checksum": int((ids.to(torch.int64) * torch.arange(1, k + 1, device="cuda")).sum().item()),
           "exact_only": os.environ.get("LANCE_HIP_NO_MFMA_FLAT_WIDE") == "1"}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        run(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], reps=int(sys.argv[6]))
        sys.exit(0)
    for n, d, nq, metric, reps_exact in ((1_000_000, 1536, 1000, "cosine", 1), (1_000_000, 1536, 1000, "l2", 0), (1_000_000, 960, 1000, "l2", 0), (200_000, 256, 2000, "dot", 1)):
        run(n, d, nq, metric)
        if reps_exact:
            env = dict(os.environ, LANCE_HIP_NO_MFMA_FLAT_WIDE="1")
            subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(n), str(d), str(nq), metric, str(reps_exact)], env=env, timeout=500)
