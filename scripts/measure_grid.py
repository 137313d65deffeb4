"""SURVEY 8(d) measurement grid on one MI355X (not the contract bench -- bench.py is).

Writes one JSON document with: measured device-copy bandwidth (the "peak_measured" denominator), the C1 flat
scan, the C2 recall/QPS grid nprobes x refine, IVF_FLAT on the same data,
and (with --c3) the C3-shaped cosine build + search.  Usage: python scripts/measure_grid.py [--c3] > out.json
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import lance_amd
from lance_amd.testing import sift_like


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


def recall_of(ids, gt):
    return (ids.unsqueeze(2) == gt.unsqueeze(1)).any(dim=2).float().mean().item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--c3", action="store_true")
    ap.add_argument("--c3-n", type=int, default=1_000_000)
    ap.add_argument("--skip-c2", action="store_true")
    args = ap.parse_args()
    eng = lance_amd.default_engine()
    dev = torch.device("cuda", 0)
    out = {"device": torch.cuda.get_device_name(0)}

    # ---- peak_measured: device-to-device copy of 4 GiB (read + write counted) ------------------------
    a = torch.empty(1 << 30, dtype=torch.float32, device=dev)
    b = torch.empty_like(a)
    dt = timed(lambda: b.copy_(a), reps=5)
    out["peak_measured"] = {"copy_GBps": 2 * a.numel() * 4 / dt / 1e9, "bytes": a.numel() * 4}
    del a, b

    if not args.skip_c2:
        d, nlist, m = 128, 256, 16
        x = sift_like(1_000_000, d, seed=1234, device=dev)
        q = sift_like(10_000, d, seed=4321, device=dev)

        # ---- C1: flat scan ------------------------------------------------------------------------
        flat = {}
        for nq in (1, 100, 1000, 10_000):
            qs = q[:nq]
            dt = timed(lambda: eng.flat_topk(x, qs, 10), reps=2 if nq >= 1000 else 5)
            flat[str(nq)] = {"ms": dt * 1e3, "qps": nq / dt, "algorithmic_GBps_per_batch": x.numel() * 4 / dt / 1e9,
                             "algorithmic_TFLOPs": 3 * x.shape[0] * d * nq / dt / 1e12}
        # roofline objects in bench.py's schema (flat scan = N*d*s bytes per batch, 2*N*d flops per query)
        t1, t10k = flat["1"]["ms"] * 1e-3, flat["10000"]["ms"] * 1e-3
        out["c1_roofline_single_query"] = {
            "kernel": "flat_small_scan_kernel<L2,f32,1> + flat_small_merge_kernel (one query: every row read once; flat_small.hip)", "bound": "hbm",
            "achieved": x.numel() * 4 / t1 / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": x.numel() * 4 / t1 / 1e9 / 8000.0, "traffic": None,
            "note": "wall time of one synchronous call through the Python binding (two launches + a stream wait); the scan kernel alone streams the rows at ~5.3 TB/s (profiles/r06zzo_flat_one_kernel_stats.csv); algorithmic bytes = N*d*4"}
        out["c1_roofline_batch_10k"] = {
            "kernel": "flat_filter_mfma_kernel<KS=8,L2,f32> (+ flat_mfma_eval_kernel, flat_select_kernel)", "bound": "mfma",
            "achieved": 2.0 * x.shape[0] * d * 10_000 / t10k / 1e12, "peak": 2500.0, "unit": "TFLOP/s (algorithmic 2*N*d per query; dense bf16 peak)",
            "frac": 2.0 * x.shape[0] * d * 10_000 / t10k / 1e12 / 2500.0,
            "executed_bf16_TFLOPs": 3 * 2.0 * x.shape[0] * d * 10_000 / t10k / 1e12,
            "frac_executed": 3 * 2.0 * x.shape[0] * d * 10_000 / t10k / 1e12 / 2500.0, "traffic": None,
            "note": "three bf16 products per pair (hi*hi, hi*lo, lo*hi) stand in for one f32 product; wall time of the whole call"}
        out["c1_flat"] = {"n": 1_000_000, "d": d, "k": 10, "by_batch_size": flat,
                          "note": "bytes = N*d*4 once per batch (SURVEY 8d); flops counted as 3 per element (sub, mul, add: no FMA by contract)"}
        gt, _ = eng.flat_topk(x, q[:1000], 10)

        # ---- C2: build + recall/QPS grid ---------------------------------------------------------------
        idx = lance_amd.create_index(x, "IVF_PQ", metric="l2", num_partitions=nlist, num_sub_vectors=m)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        idx = lance_amd.create_index(x, "IVF_PQ", metric="l2", num_partitions=nlist, num_sub_vectors=m)
        torch.cuda.synchronize()
        out["c2_build"] = {"sec": time.perf_counter() - t0, "stages_ms": {k: round(v * 1e3, 3) for k, v in idx.stats.seconds.items()},
                           "ivf_iters": idx.stats.ivf_iters, "pq_iters": idx.stats.pq_iters}
        grid = []
        for nprobes in (1, 10, 25, 50, nlist):
            for rf in (0, 10):
                nq = 10_000 if nprobes <= 50 else 2_000
                qq = q[:nq]
                dt = timed(lambda: idx.search_device(qq, 10, nprobes, rf), reps=3)
                ids, _ = idx.search_device(q[:1000], 10, nprobes, rf)
                grid.append({"nprobes": nprobes, "refine_factor": rf, "recall_at_10": recall_of(ids, gt), "nq": nq,
                             "ms_per_batch": dt * 1e3, "qps": nq / dt, "exact_replays": eng.search_stats()})
        out["c2_grid"] = grid
        # small batches (the reference publishes single-query latencies: benchmarks/sift/lance_sift1m_stats.csv) -- wall time of one
        # call through the Python binding, results on the device
        lat = []
        for nq in (1, 16, 256, 2048):
            qq = q[:nq].contiguous()
            dt = timed(lambda: idx.search_device(qq, 10, 10, 10), reps=20 if nq <= 256 else 5)
            lat.append({"nq": nq, "nprobes": 10, "refine_factor": 10, "ms_per_call": dt * 1e3, "qps": nq / dt})
        out["c2_small_batches"] = lat

        # (the bit-exact id check at nprobes = nlist against the CPU oracle lives in tests/test_gpu_parity.py::
        #  test_full_size_properties_sift1m -- only tests, smoke() and bench.py's cpu_baseline may touch oracle/)
        raw = None
        # ---- IVF_FLAT (N4) on the same data: exact distances inside the probed partitions -------------------
        fx = lance_amd.create_index(x, "IVF_FLAT", metric="l2", num_partitions=nlist)
        fl = []
        for nprobes in (1, 10, 50):
            qq = q[:2000]
            dt = timed(lambda: fx.search_device(qq, 10, nprobes), reps=2)
            ids, _ = fx.search_device(q[:1000], 10, nprobes)
            fl.append({"nprobes": nprobes, "recall_at_10": recall_of(ids, gt), "nq": 2000, "ms_per_batch": dt * 1e3, "qps": 2000 / dt})
        out["c2_ivf_flat"] = {"build_stages_ms": {k: round(v * 1e3, 3) for k, v in fx.stats.seconds.items()}, "grid": fl}
        del x, q, idx, raw, fx

    if args.c3:
        d, nlist, m = 1536, 1024, 96
        n = args.c3_n
        x = sift_like(n, d, seed=77, device=dev, n_clusters=1024, latent=48, model_seed=77)
        x = x - 64.0                                   # centre: embeddings are not all-positive
        q = (sift_like(1000, d, seed=78, device=dev, n_clusters=1024, latent=48, model_seed=77) - 64.0).contiguous()
        x = torch.nn.functional.normalize(x, dim=1).contiguous()
        q = torch.nn.functional.normalize(q, dim=1).contiguous()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        idx = lance_amd.create_index(x, "IVF_PQ", metric="cosine", num_partitions=nlist, num_sub_vectors=m)
        torch.cuda.synchronize()
        c3 = {"n": n, "d": d, "nlist": nlist, "m": m, "build_sec_cold": time.perf_counter() - t0,
              "stages_ms_cold": {k: round(v * 1e3, 3) for k, v in idx.stats.seconds.items()},
              "ivf_iters": idx.stats.ivf_iters, "pq_iters": idx.stats.pq_iters}
        del idx          # the first build pays for the scratch arenas (3 x 7.7 GB of hipMalloc at this shape); time a second one
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        idx = lance_amd.create_index(x, "IVF_PQ", metric="cosine", num_partitions=nlist, num_sub_vectors=m)
        torch.cuda.synchronize()
        c3["build_sec"] = time.perf_counter() - t0
        c3["stages_ms"] = {k: round(v * 1e3, 3) for k, v in idx.stats.seconds.items()}
        dt = timed(lambda: eng.flat_topk(x, q, 10, metric="cosine"), reps=1)
        c3["flat_1000q_ms"] = dt * 1e3
        gt, _ = eng.flat_topk(x, q, 10, metric="cosine")
        grid = []
        for nprobes, rf in ((1, 0), (10, 0), (10, 10), (25, 10), (50, 10)):
            dt = timed(lambda: idx.search_device(q, 10, nprobes, rf), reps=2)
            ids, _ = idx.search_device(q, 10, nprobes, rf)
            grid.append({"nprobes": nprobes, "refine_factor": rf, "recall_at_10": recall_of(ids, gt), "nq": 1000,
                         "ms_per_batch": dt * 1e3, "qps": 1000 / dt, "exact_replays": eng.search_stats()})
        c3["grid"] = grid
        # roofline of the dominant search kernel at (nprobes 10, refine 10): the tiled filter scan (search_qt.hip).  Its time is the
        # 4-query table build -- SURVEY 8d: 2 * 256 * d flops per (query, probe) -- on the packed-f32 VALU (157 TFLOP/s vector peak)
        eng.timing(True)
        b0 = eng.timing_query("ivfpq_scan_c1")
        for _ in range(5):
            idx.search_device(q, 10, 10, 10)
        eng.synchronize()
        b1 = eng.timing_query("ivfpq_scan_c1")
        eng.timing(False)
        if b1[1] > b0[1]:
            t_scan = (b1[0] - b0[0]) / (b1[1] - b0[1]) * 1e-3
            flops = 1000 * 10 * 2.0 * 256 * d
            c3["roofline"] = {"kernel": "ivfpq_qscan_tiled_kernel<SD=16,MU=6,NT=3> (4-query u16 table build tiled over m + row scan)",
                              "bound": "valu", "achieved": flops / t_scan / 1e12, "peak": 157.3, "unit": "TFLOP/s (f32 vector)",
                              "frac": flops / t_scan / 1e12 / 157.3, "traffic": None, "avg_launch_ms": t_scan * 1e3,
                              "algorithmic_flops_per_launch": flops,
                              "note": "LUT build 2*256*d flops per (query, probe), SURVEY 8(d); executed as sub + FMA (2 packed VALU per 2 "
                                      "MACs), so the executed rate is twice the algorithmic one"}
        out["c3"] = c3
    print(json.dumps(out, indent=1, default=lambda o: o.tolist() if hasattr(o, 'tolist') else str(o)))


if __name__ == "__main__":
    main()
