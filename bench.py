#!/usr/bin/env python
"""bench.py -- QPS @ recall@10 of the IVF_PQ hot path on MI355X (+ index-build seconds).

Workload (BASELINE.json configs[1]): SIFT-1M-like synthetic base set (1,000,000 x 128 f32,
integer-valued, see lance_amd/testing/datagen.py), IVF_PQ(nlist=256, M=16, nbits=8) built on
the GPU, then batched k-NN: one STEP = one batch of 10,000 queries through
find_partitions -> residual LUT -> ADC scan -> per-partition top-k -> (dist,rowid) merge ->
refine, with k=10, nprobes=10, refine_factor=10.  Inputs are resident in HBM before the timed
region.  value = whole-job queries/s (N ranks x 10,000 queries / max-over-ranks time).

Multi-GPU (one process per GPU, torch.distributed/RCCL): the index build shards the k-means
E-step over the ranks with one all-reduce per Lloyd iteration and the encode pass by rows
(lance_amd/dist.py); search is embarrassingly parallel -- every rank holds a replica of the
16 MB code table and serves its own query batch ("weak" scaling, no data-path collective).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable copy rate)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--nq", type=int, default=10_000)
    ap.add_argument("--nprobes", type=int, default=10)
    ap.add_argument("--refine", type=int, default=10)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if rank != 0:
        # only rank 0 reports; keep the other ranks' stdout (RCCL prints a version banner through C stdio) out of the
        # launcher's combined output so that the JSON line stays the only thing on it
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no HIP device visible); there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force_dist = os.environ.get("LANCE_BENCH_FORCE_DIST") == "1"   # exercise the sharded build with world_size 1
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import lance_amd
    from lance_amd.testing import sift_like
    from lance_amd import vector as lv

    eng = lance_amd.default_engine()
    d, nlist, m = 128, 256, 16
    x = sift_like(args.n, d, seed=1234, device=dev)
    # 4 different query batches per rank, cycled over the steps
    qbatches = [sift_like(args.nq, d, seed=4321 + 100 * rank + i, device=dev) for i in range(4)]

    # ---- index build (timed separately: "index-build sec") ----------------------------
    def build_once():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        if world > 1 or force_dist:
            from lance_amd import dist as ld
            ix = ld.create_index_sharded(x, metric="l2", num_partitions=nlist, num_sub_vectors=m)
        else:
            ix = lance_amd.create_index(x, "IVF_PQ", metric="l2", num_partitions=nlist, num_sub_vectors=m)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        return ix, time.perf_counter() - t0

    idx, _ = build_once()            # warm-up build (kernel load, scratch allocation)
    build_secs = []
    for _ in range(2):
        idx, bs = build_once()
        build_secs.append(bs)
    build_sec = min(build_secs)

    # ---- recall@10 on a 1000-query sample against exact flat top-10 (GPU flat kernel,
    # itself parity-tested against the oracle) ----------------------------------------------
    qs = qbatches[0][:1000]
    gt, _ = eng.flat_topk(x, qs, args.k)
    ids_s, _ = idx.search_device(qs, args.k, args.nprobes, args.refine)
    recall = (ids_s.unsqueeze(2) == gt.unsqueeze(1)).any(dim=2).float().mean().item()

    # ---- algorithmic bytes of the ADC scan: sum over (query, probe) pairs of n_p * M code bytes.
    # The scan runs as two launches of ivfpq_scan_pm_kernel: a bound pass over each query's nearest partition
    # (seeds the per-query threshold, keeps nothing) and the main pass over ALL nprobes partitions (the dominant
    # launch; with LANCE_HIP_PM_NOBOUND=1 the earlier flow: class 0 / class 1 = the other nprobes-1 partitions).
    offs = torch.from_numpy(idx.export_storage()[0].astype(np.int64)).to(dev)
    sizes = offs[1:] - offs[:-1]
    scan_bytes, scan_bytes_c1 = [], []
    for qb in qbatches:
        probes, _ = eng.find_partitions(qb, idx._ix.centroids, args.nprobes)
        per = sizes[probes.long()]
        scan_bytes.append(int(per.sum().item()) * m)
        scan_bytes_c1.append(int(per[:, 1:].sum().item()) * m)

    # ---- timed region ---------------------------------------------------------------------
    out_ids = torch.empty((args.nq, args.k), dtype=torch.int64, device=dev)
    out_d = torch.empty((args.nq, args.k), dtype=torch.float32, device=dev)

    def step(i):
        idx.search_device(qbatches[i % 4], args.k, args.nprobes, args.refine, out=(out_ids, out_d), sync=False)

    for i in range(args.warmup):
        step(i)
    eng.synchronize()
    eng.timing(True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    eng.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    eng.timing(False)
    exact_replays = eng.search_stats()
    kt = {kname: eng.timing_query(kname) for kname in ("dist_matrix", "select_probes", "pm_group", "ivfpq_scan", "ivfpq_scan_c0",
                                                       "ivfpq_scan_c1", "ivfpq_merge", "ivfpq_exact", "refine")}
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- informational (never `value`): the same steps when the boundary hands over HOST buffers -- every step
    # copies its query batch from pinned host memory and its results back (PCIe inclusive) ------------------
    pcie_qps = None
    if world == 1:
        hq = [qb.cpu().pin_memory() for qb in qbatches]
        h_ids = torch.empty((args.nq, args.k), dtype=torch.int64).pin_memory()
        h_d = torch.empty((args.nq, args.k), dtype=torch.float32).pin_memory()
        dq = torch.empty_like(qbatches[0])

        def host_step(i):
            dq.copy_(hq[i % 4], non_blocking=True)
            torch.cuda.current_stream().synchronize()
            idx.search_device(dq, args.k, args.nprobes, args.refine, out=(out_ids, out_d), sync=False)
            eng.synchronize()
            h_ids.copy_(out_ids, non_blocking=True); h_d.copy_(out_d, non_blocking=True)
            torch.cuda.current_stream().synchronize()

        for i in range(2):
            host_step(i)
        t1 = time.perf_counter()
        for i in range(args.steps):
            host_step(i)
        pcie_qps = args.nq * args.steps / (time.perf_counter() - t1)

    ms_per_step = elapsed / args.steps * 1e3
    qps = world * args.nq * args.steps / elapsed
    if kt["ivfpq_scan_c1"][1] > 0:      # partition-major path: dominant launch = class 1
        scan_ms, scan_launches = kt["ivfpq_scan_c1"]
        if os.environ.get("LANCE_HIP_PM_NOBOUND"):
            bytes_list, kernel_name = scan_bytes_c1, "ivfpq_scan_pm_kernel<SD=8,L2,MU=1,RPL=2> (class-1 launch: the nprobes-1 farther partitions)"
        else:
            bytes_list, kernel_name = scan_bytes, "ivfpq_scan_pm_kernel<SD=8,L2,MU=1,RPL=2> (main pass: all nprobes partitions of every query)"
    else:
        scan_ms, scan_launches = kt["ivfpq_scan"]
        bytes_list, kernel_name = scan_bytes, "ivfpq_scan_kernel<SD=8,L2,MU=1>"
    avg_scan_ms = scan_ms / max(scan_launches, 1)
    avg_bytes = float(np.mean([bytes_list[i % 4] for i in range(args.steps)]))
    achieved = avg_bytes / (avg_scan_ms * 1e-3) / 1e9 if avg_scan_ms > 0 else 0.0
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "r01_scan_pmc.json")   # rocprofv3 --pmc summary of this same command (committed)
    if os.path.exists(pmc_path):
        try:
            traffic = json.load(open(pmc_path)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    result = {
        "metric": "QPS @ recall@10 (SIFT-1M IVF_PQ nlist=256 M=16) + index-build sec",
        "value": qps,
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "SIFT-1M-like 1Mx128 f32, IVF_PQ nlist=256 M=16 nbits=8, build+search on MI355X",
                   "n": args.n, "d": d, "nlist": nlist, "m": m, "queries_per_step_per_gpu": args.nq, "k": args.k,
                   "nprobes": args.nprobes, "refine_factor": args.refine,
                   "parallelism": f"replica x{world}" if world > 1 else "single"},
        "recall_at_10": recall,
        "exact_replays_last_step": exact_replays,
        "host_buffers_qps_pcie_inclusive": pcie_qps,
        "build_sec": build_sec,
        "build_stages_ms": {k_: round(v * 1e3, 3) for k_, v in (idx.stats.seconds.items() if idx.stats else [])},
        "kernel_ms_per_step": {k_: round(v[0] / max(v[1], 1), 4) for k_, v in kt.items()},
        "roofline": {"kernel": kernel_name, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "algorithmic_bytes_per_launch": avg_bytes, "avg_launch_ms": avg_scan_ms},
    }

    if not args.no_cpu_baseline and world == 1:
        import oracle as orc
        native = orc.use_native_build()      # time the CPU side with code generated for THIS host
        o_offs, o_codes_t, o_rid = idx.export_storage()
        oidx = orc.IvfPqIndex("l2", idx.centroids, idx.codebook, o_offs, o_codes_t, o_rid)
        xq = qbatches[0].cpu().numpy()
        raw = x.cpu().numpy()
        cores = orc.num_threads()
        oidx.search(xq[:256], args.k, args.nprobes, refine=args.refine, raw=raw)  # warm
        best = float("inf")
        reps = 0
        t_all = time.perf_counter()
        while reps < 3 or (time.perf_counter() - t_all < 10 and reps < 20):
            t1 = time.perf_counter()
            oi, _ = oidx.search(xq, args.k, args.nprobes, refine=args.refine, raw=raw)
            best = min(best, time.perf_counter() - t1)
            reps += 1
        gi, _ = idx.search_device(qbatches[0], args.k, args.nprobes, args.refine)
        same = bool((oi == gi.cpu().numpy().view(np.uint64)).all())
        # CPU index build on a bounded sample: 3 Lloyd iterations of the IVF k-means on the 65,536-row sample
        samp = raw[: nlist * 256]
        t1 = time.perf_counter()
        orc.kmeans_train(samp, nlist, max_iters=3, tol=0.0, balance_factor=1.0 / samp.shape[0], seed=1)
        cpu_iter = (time.perf_counter() - t1) / 3
        result["cpu_baseline"] = {"value": args.nq / best, "unit": "queries/s", "cores": cores, "kind": "port",
                                  "sample": f"the same {args.nq}-query batch, same index/nprobes/refine, best of {reps} runs "
                                            f"of oracle/lance_oracle.c (OpenMP over queries)",
                                  "ids_equal_gpu": same,
                                  "oracle_march": "native" if native else "x86-64-v3",
                                  "ivf_kmeans_sec_per_iter_65536x128_k256": cpu_iter}
    elif world == 1:
        result["cpu_baseline"] = None
    if world > 1 or force_dist:
        dist.destroy_process_group()
    # RCCL's banner sits in the C stdio buffer until exit: push it out first so that the JSON line is the LAST line
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
