#!/usr/bin/env python
"""bench.py -- QPS @ recall@10 of the IVF_PQ hot path on MI355X (+ index-build seconds).

Workload (BASELINE.json configs[1]): SIFT-1M-like synthetic base set (1,000,000 x 128 f32,
integer-valued, see lance_amd/testing/datagen.py), IVF_PQ(nlist=256, M=16, nbits=8) built on
the GPU, then batched k-NN: one STEP = one batch of 10,000 queries through
find_partitions -> residual LUT -> ADC scan -> per-partition top-k -> (dist,rowid) merge ->
refine, with k=10, nprobes=10, refine_factor=10.  Inputs are resident in HBM before the timed
region.  value = whole-job queries/s (N ranks x 10,000 queries / max-over-ranks time).

Multi-GPU (one process per GPU, torch.distributed/RCCL): every rank generates and keeps only its block of the rows.  The
build runs the IVF k-means with one all-reduce of the fused [k*d sums | k counts] buffer per Lloyd iteration, trains the PQ
sub-quantisers model-parallel and encodes its own rows (lance_amd/dist.py: create_index_rowsharded); `build_sec` is that
build plus the all-gather of the 20 bytes per row that gives every rank a replica.  `value` = the replicas serving their own
query batches ("weak" scaling, no data-path collective); `multi_gpu.list_sharded_qps_strong_scaling` = the same batch
answered jointly by IVF lists sharded over the ranks (all_to_all by list owner, one all-gather + device merge per batch).
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
LDS_B64_CONFLICT_FREE_PER_CLK_CU = 32.0   # MI355X_MICROARCH.md, LDS table: ds_read_b64 = 2 cycles per wave-instruction


def collect_pmc_traffic(kernel_substr, child_args, timeout_s=300):
    """HBM bytes per launch of the dominant kernel, measured as MI355X_MICROARCH.md's HBM / rocprofv3 section prescribes: two
    separate `rocprofv3 --pmc` passes (FETCH_SIZE, then WRITE_SIZE; --kernel-trace only) of THIS script as a child process (same
    workload, a few steps, one stream, no CPU leg), counters in units of 1024 B, FETCH_SIZE x 2 on gfx950 (wide coalesced reads are
    reported at half their bytes).  Only the full-size launches count (the most frequent grid size of the kernel: the recall
    check's 1000-query launch is left out).  Returns (dict or None, note)."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    from collections import Counter
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    vals = {}
    tmp = tempfile.mkdtemp(prefix="lance_bench_pmc_")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__)] + child_args
            env = dict(os.environ, LANCE_BENCH_PMC_CHILD="1", TMPDIR=tmp)
            proc = subprocess.Popen(cmd, cwd=tmp, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                proc.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(proc.pid, signal.SIGKILL)
                proc.wait()
                return None, f"rocprofv3 --pmc {counter} pass did not finish in {timeout_s} s"
            if proc.returncode != 0:
                return None, f"rocprofv3 --pmc {counter} pass exited with {proc.returncode}"
            rows = []
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r.get("Counter_Name") == counter and kernel_substr in r.get("Kernel_Name", ""):
                        rows.append((r.get("Grid_Size", ""), float(r["Counter_Value"])))
            if not rows:
                return None, f"no {counter} rows for a kernel matching {kernel_substr!r}"
            grid = Counter(g for g, _ in rows).most_common(1)[0][0]
            sel = [v for g, v in rows if g == grid]
            vals[counter] = (sum(sel) / len(sel) * 1024.0, len(sel))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    fetch_raw, nf = vals["FETCH_SIZE"]
    write_raw, nw = vals["WRITE_SIZE"]
    return ({"hbm_bytes_per_launch": 2.0 * fetch_raw + write_raw, "fetch_bytes_raw": fetch_raw, "fetch_bytes_x2": 2.0 * fetch_raw,
             "write_bytes": write_raw, "launches_averaged": [nf, nw]},
            "collected in this run: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate, --kernel-trace only) of "
            "`bench.py " + " ".join(child_args) + "`; bytes = 1024 x counter, FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md HBM section)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)      # a step is 0.5 ms: 200 of them time 0.1 s (20 steps = 10 ms moved the line by 5 % run to run)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=["c2", "c4", "c5"], default="c2",
                    help="c2 (default, the contract workload): SIFT-1M-like f32, IVF_PQ(256, 16).  c4: BASELINE config 4's shape -- "
                         "f16 rows, IVF_PQ(4096, 16), 12.5M rows PER RANK (100M on 8 GPUs): the configuration where the E-step "
                         "outweighs the per-iteration all-reduce of the sharded k-means")
    ap.add_argument("--n", type=int, default=None, help="total rows (default: 1,000,000 for c2; 12,500,000 x ranks for c4 / c5).  c5: BASELINE config 5's "
                    "shape -- int8 rows x 128, IVF_PQ(65536, 32), hierarchical k-means")
    ap.add_argument("--nq", type=int, default=10_000)
    ap.add_argument("--nprobes", type=int, default=10)
    ap.add_argument("--refine", type=int, default=10)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 --pmc child passes that measure roofline.traffic")
    ap.add_argument("--streams", type=int, default=3, help="engine contexts (HIP streams) the query batches alternate on")
    ap.add_argument("--no-grid", action="store_true", help="skip the small-batch latencies and the SURVEY 8(d) recall grid (a few seconds)")
    ap.add_argument("--nlist", type=int, default=None, help="override the configuration's number of IVF lists (tests: a count the ranks do not divide)")
    ap.add_argument("--no-extras", action="store_true", help="skip the child run on the f32 refine source and the PCIe-inclusive build (round 6 additions to the line)")
    # ranks started by this script's own launcher take their arguments from the environment: torchrun's argument parser refuses
    # `--n` after the script name (an abbreviation of several of ITS options: gpurun r05a)
    args = ap.parse_args(json.loads(os.environ["LANCE_BENCH_ARGV"])) if "LANCE_BENCH_ARGV" in os.environ else ap.parse_args()

    # `python bench.py --gpus N` on its own starts the N ranks itself (one process per GPU under torch.distributed.run, rendezvous
    # on 127.0.0.1); under an external launcher (WORLD_SIZE already set: the driver's own command form) the flag is checked
    # against the world size instead of being ignored.
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)]
        os.environ["LANCE_BENCH_ARGV"] = json.dumps(sys.argv[1:])
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ.get('WORLD_SIZE')} ranks")

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
    # First-contact insurance for the N-rank path on a box with ONE GPU (tests/test_zz_gpu_two_ranks.py): LANCE_BENCH_ONE_GPU=1 puts every
    # rank on device 0 and LANCE_BENCH_BACKEND=gloo carries the collectives through host memory (RCCL refuses two ranks on one device) --
    # every kernel, buffer layout and exchange of the multi-GPU build / search runs for real; only the transport differs.  Never a
    # measurement: the line says so in `multi_gpu.transport`.
    one_gpu = os.environ.get("LANCE_BENCH_ONE_GPU") == "1"
    backend = os.environ.get("LANCE_BENCH_BACKEND", "nccl")
    if one_gpu:
        local_rank = 0
    if not torch.cuda.is_available() or (not one_gpu and torch.cuda.device_count() < world):
        raise SystemExit(f"bench.py --gpus {world} needs {world} MI355X GPUs (rank {rank} sees {torch.cuda.device_count() if torch.cuda.is_available() else 0} "
                         f"HIP devices); there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force_dist = os.environ.get("LANCE_BENCH_FORCE_DIST") == "1"   # exercise the sharded build with world_size 1
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import lance_amd
    from lance_amd.testing import sift_like
    from lance_amd import vector as lv

    eng = lance_amd.default_engine()
    d, nlist, m = {"c2": (128, 256, 16), "c4": (128, 4096, 16), "c5": (128, 65536, 32)}[args.config]
    if args.nlist:
        nlist = args.nlist
    half = args.config == "c4"                      # Float16 column (C4)
    int8c = args.config == "c5"                     # Int8 column (C5: BigANN-shaped)
    if args.n is None:
        args.n = 1_000_000 if args.config == "c2" else 12_500_000 * world
    multi = world > 1 or force_dist

    def gen_rows(count, seed):
        # c4: 100M rows do not fit through one f32 staging tensor comfortably -- generate in 4M-row pieces, keep f16
        if int8c:      # descriptors as signed bytes, 16,384 clusters, generated in 4M-row pieces
            out = torch.empty((count, d), dtype=torch.int8, device=dev)
            for a in range(0, count, 4_000_000):
                b = min(count, a + 4_000_000)
                out[a:b] = (sift_like(b - a, d, seed=seed + 31 * (a // 4_000_000), device=dev, n_clusters=16384) - 100.0).clamp_(-128, 127).to(torch.int8)
            return out
        if not half:
            return sift_like(count, d, seed=seed, device=dev)
        out = torch.empty((count, d), dtype=torch.float16, device=dev)
        for a in range(0, count, 4_000_000):
            b = min(count, a + 4_000_000)
            # / 256: the reference accumulates the k-means M-step in the column's element type (kmeans.rs:380-406), so Float16 rows
            # must keep their cluster sums far from 65504 (scripts/scale_probe.py uses the same scaling)
            out[a:b] = (sift_like(b - a, d, seed=seed + 31 * (a // 4_000_000), device=dev, n_clusters=4096) / 256.0).to(torch.float16)
        return out
    # 4 different query batches per rank, cycled over the steps
    qbatches = [((sift_like(args.nq, d, seed=4321 + 100 * rank + i, device=dev, n_clusters=4096) / 256.0).to(torch.float16) if half else
                 (sift_like(args.nq, d, seed=4321 + 100 * rank + i, device=dev, n_clusters=16384) - 100.0).clamp_(-128, 127).to(torch.int8) if int8c else
                 sift_like(args.nq, d, seed=4321 + 100 * rank + i, device=dev)) for i in range(4)]
    mg = {}            # multi-GPU extras of the bench line

    if not multi:
        x = gen_rows(args.n, 1234)

        def build_once():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ix = lance_amd.create_index(x, "IVF_PQ", metric="l2", num_partitions=nlist, num_sub_vectors=m)
            torch.cuda.synchronize()
            return ix, time.perf_counter() - t0

        # a billion-row index (C5) fits the GPU once, not twice: one build, and the previous index is dropped before the next one is made
        big = args.n >= 200_000_000
        idx = None
        if not big:
            idx, _ = build_once()            # warm-up build (kernel load, scratch allocation)
        build_secs = []
        for _ in range(1 if big else 2):
            idx = None
            torch.cuda.empty_cache()
            idx, bs = build_once()
            build_secs.append(bs)
        build_sec = min(build_secs)
        # informational (never `build_sec`): the same build when the boundary hands over a HOST buffer -- the column crosses PCIe first
        build_sec_pcie = None
        if not args.no_extras and os.environ.get("LANCE_BENCH_CHILD") != "1":
            hx = x.cpu().pin_memory()
            xd = torch.empty_like(x)
            bt = []
            for _ in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                xd.copy_(hx, non_blocking=True)
                ixp = lance_amd.create_index(xd, "IVF_PQ", metric="l2", num_partitions=nlist, num_sub_vectors=m)
                torch.cuda.synchronize()
                bt.append(time.perf_counter() - t0)
                del ixp
            build_sec_pcie = min(bt)
            del hx, xd
    else:
        build_sec_pcie = None
        # One node, N ranks: every rank generates and keeps ONLY its block of the 1M rows (same mixture, own draws); the
        # vectors never leave their GPU during the build.  IVF k-means = Lloyd iterations with one RCCL all-reduce of the
        # fused [k*d sums | k counts] buffer per iteration ("sharded"; the replicated-training time is reported beside it),
        # PQ sub-quantisers model-parallel, transform local (lance_amd/dist.py: create_index_rowsharded).
        from lance_amd import dist as ld
        _, ranges = ld.block_ranges(args.n, world)
        lo, hi = ranges[rank]
        x_local = gen_rows(hi - lo, 1234 + 7919 * rank)

        def build_once(mode):
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            b = ld.create_index_rowsharded(x_local, metric="l2", num_partitions=nlist, num_sub_vectors=m, ivf_training=mode)
            ix, _ = ld.replica_index(b, None)          # all-gather of the 20 bytes per row (partition id, code): a replica per rank
            torch.cuda.synchronize()
            dist.barrier()
            return ix, b, time.perf_counter() - t0

        build_once("sharded")            # warm-up
        secs = {}
        for mode in ("replicated", "sharded"):
            best = float("inf")
            for _ in range(2):
                idx, bld, bs = build_once(mode)
                best = min(best, bs)
            secs[mode] = best
        build_sec = secs["sharded"]
        t0 = time.perf_counter()
        x, _ = ld.all_gather_var(x_local)             # refine on a replica needs every raw vector on every rank (512 MB over xGMI)
        idx._ix.set_raw(x)
        torch.cuda.synchronize()
        mg = {"rccl_ranks": world, "transport": "rccl" if backend == "nccl" else f"{backend} through host memory" + (", all ranks on one GPU" if one_gpu else ""),
              "rows_per_rank": hi - lo, "build_sec_ivf_sharded_allreduce": secs["sharded"],
              "build_sec_ivf_replicated": secs["replicated"], "raw_vectors_allgather_sec": time.perf_counter() - t0,
              "build_stages_ms_sharded": {k_: round(v * 1e3, 3) for k_, v in bld.stats.seconds.items()},
              # what "sharded" IVF training means at this nlist, and what it can be checked against
              "ivf_training_sharded_is": ("hierarchical k-means (nlist > 256, kmeans.rs:1027) with its splits spread over the ranks: bit-identical to the "
                                          "single-GPU trainer and to the CPU oracle (tests/test_dist_gloo.py)" if nlist > 256 else
                                          "row-sharded Lloyd loop, one all-reduce per iteration: the sums arrive in rank order, not row order -- equal to the "
                                          "single-GPU trainer to f32 round-off only (no oracle equality; one rank: identical)"),
              "ivf_hierarchical_rounds": getattr(bld.stats, "ivf_hierarchical", None)}
        # strong-scaling line: the IVF lists sharded over the ranks (list p -> rank p % N, rows moved by all_to_all), every
        # rank answers the SAME query batches with its lists, one all-gather + device (dist, rowid) merge per batch
        shard, l2g = ld.list_shard_index(bld, x_local)
        common = [((sift_like(args.nq, d, seed=9000 + i, device=dev, n_clusters=4096) / 256.0).to(torch.float16) if half else
                   sift_like(args.nq, d, seed=9000 + i, device=dev)) for i in range(2)]

        def lstep(i):
            return ld.search_list_sharded(lambda qq, kk, npb, rf: shard.search(qq, kk, npb, rf), l2g, common[i % 2], args.k, args.nprobes,
                                          args.refine, engine=eng, local_candidates=lambda qq, ke, npb: shard.search_candidates(qq, ke, npb))

        li, _ = lstep(0)
        ri, _ = idx.search_device(common[0], args.k, args.nprobes, args.refine)
        mg["list_sharded_equals_replica"] = bool(torch.equal(li, ri))
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            lstep(i)
        torch.cuda.synchronize()
        dist.barrier()
        mg["list_sharded_qps_strong_scaling"] = args.nq * args.steps / (time.perf_counter() - t0)
        shard.close()

    # ---- recall@10 on a 1000-query sample against exact flat top-10 (GPU flat kernel,
    # itself parity-tested against the oracle) ----------------------------------------------
    qs = qbatches[0][:1000]
    gt, _ = eng.flat_topk(x, qs, args.k)
    ids_s, _ = idx.search_device(qs, args.k, args.nprobes, args.refine)
    recall = (ids_s.unsqueeze(2) == gt.unsqueeze(1)).any(dim=2).float().mean().item()

    # ---- small batches and the SURVEY 8(d) grid (benchmarks/sift/metrics.py:78-94,154-155: the reference publishes single-query latencies and
    # recall@k over nprobes x refine_factor).  Latency = wall time of one synchronous call (host launch + device + the flags read-back),
    # median of 50 after 5 warm calls, queries resident on the device; grid = recall@10 of the 1000-query sample + ms per 1000-query batch.
    latency, recall_grid = None, None
    if not args.no_grid and os.environ.get("LANCE_BENCH_PMC_CHILD") != "1" and rank == 0:
        latency = {}
        for nq_l in (1, 16, 256):
            ql = qbatches[1][:nq_l].contiguous()
            o_l = (torch.empty((nq_l, args.k), dtype=torch.int64, device=dev), torch.empty((nq_l, args.k), dtype=torch.float32, device=dev))
            for _ in range(5):
                idx.search_device(ql, args.k, args.nprobes, args.refine, out=o_l)
            ts = []
            for _ in range(50):
                t1 = time.perf_counter()
                idx.search_device(ql, args.k, args.nprobes, args.refine, out=o_l)
                ts.append(time.perf_counter() - t1)
            ts.sort()
            latency[f"nq{nq_l}"] = {"median_ms": round(ts[len(ts) // 2] * 1e3, 4), "p95_ms": round(ts[int(len(ts) * 0.95)] * 1e3, 4)}
        latency["what"] = (f"one synchronous lance_hip_ivfpq_search call, k={args.k} nprobes={args.nprobes} refine={args.refine}, device-resident queries, "
                           "wall clock incl. host launch and result-flag read-back; the reference's published figures for this path are "
                           "single-query latencies (BASELINE.md: 1.24-8.67 ms, IVF512, unnamed hardware)")
        recall_grid = []
        o_g = (torch.empty((1000, args.k), dtype=torch.int64, device=dev), torch.empty((1000, args.k), dtype=torch.float32, device=dev))
        for npb in (1, 10, 25, 50, min(nlist, 256)):      # (c4 / c5: 256 probes stand in for the exhaustive column)
            for rf in (0, 10):
                for _ in range(3):
                    idx.search_device(qs, args.k, npb, rf, out=o_g)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(5):
                    idx.search_device(qs, args.k, npb, rf, out=o_g)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t1) / 5 * 1e3
                rc = (o_g[0].unsqueeze(2) == gt.unsqueeze(1)).any(dim=2).float().mean().item()
                recall_grid.append({"nprobes": npb, "refine_factor": rf, "recall_at_10": round(rc, 4), "ms_per_1000_queries": round(ms, 4)})

    # ---- algorithmic bytes of the ADC scan: sum over (query, probe) pairs of n_p * M code bytes.
    # The scan runs as two launches of ivfpq_scan_pm_kernel: a bound pass over each query's nearest partition
    # (seeds the per-query threshold, keeps nothing) and the main pass over ALL nprobes partitions (the dominant
    # launch; with LANCE_HIP_PM_NOBOUND=1 the earlier flow: class 0 / class 1 = the other nprobes-1 partitions).
    offs = torch.from_numpy(idx._ix.part_offsets().astype(np.int64)).to(dev)
    sizes = offs[1:] - offs[:-1]
    scan_bytes, scan_bytes_c1 = [], []
    for qb in qbatches:
        probes, _ = eng.find_partitions(qb, idx._ix.centroids, args.nprobes)
        per = sizes[probes.long()]
        scan_bytes.append(int(per.sum().item()) * m)
        scan_bytes_c1.append(int(per[:, 1:].sum().item()) * m)

    # ---- timed region ---------------------------------------------------------------------
    # Batches are independent: they are enqueued alternately on `--streams` engine contexts (own HIP stream + scratch arena
    # each, one shared read-only index), so the latency-bound tail of one batch (survivor re-evaluation, merge, refine)
    # overlaps the scan of the next.  A step is still one 10,000-query batch; the region ends when every batch has finished.
    from lance_amd.engine import Engine
    nstreams = max(1, args.streams)
    engines = [eng] + [Engine(device=eng.device) for _ in range(nstreams - 1)]
    outs = [(torch.empty((args.nq, args.k), dtype=torch.int64, device=dev), torch.empty((args.nq, args.k), dtype=torch.float32, device=dev))
            for _ in range(nstreams)]
    out_ids, out_d = outs[0]

    def step(i):
        s_ = i % nstreams
        idx.search_device(qbatches[i % 4], args.k, args.nprobes, args.refine, out=outs[s_], sync=False, engine=engines[s_])

    def sync_all():
        for e in engines:
            e.synchronize()

    # Untimed priming: a search is captured into a HIP graph on its second call with the same arguments (context, query batch,
    # output buffers) and replayed from the third -- every (context, batch) pair of the cycle below is taken through that once, so
    # that the warm-up and the timed region measure the steady state a serving loop runs in (LANCE_HIP_GRAPH=0: no capture, no priming)
    graph_priming = 0
    if os.environ.get("LANCE_HIP_GRAPH", "1") != "0" and os.environ.get("LANCE_BENCH_PMC_CHILD") != "1":
        graph_priming = 2 * 4 * nstreams
        for i in range(graph_priming):
            step(i)
        sync_all()
    for i in range(max(args.warmup, nstreams)):
        step(i)
    sync_all()
    # per-kernel HIP-event timing (roofline) is taken in a separate, single-stream pass below so that the events do not
    # serialise the streams of the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    sync_all()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if os.environ.get("LANCE_BENCH_PMC_CHILD") == "1":      # a counter pass of collect_pmc_traffic: the launches are all it is for
        return
    # kernel-level timing pass (same batches, one stream, events on that stream)
    eng.timing(True)
    for i in range(args.steps):
        idx.search_device(qbatches[i % 4], args.k, args.nprobes, args.refine, out=outs[0], sync=False)
    eng.synchronize()
    eng.timing(False)
    exact_replays = eng.search_stats()
    kt = {kname: eng.timing_query(kname) for kname in ("dist_matrix", "select_probes", "pm_group", "ivfpq_scan", "ivfpq_scan_c0",
                                                       "q_residual", "ivfpq_scan_c1", "ivfpq_merge", "ivfpq_exact", "refine")}
    mscan = eng.timing_query("ivfpq_mscan")[1] > 0      # the main pass ran as the matrix-core filter scan (search_ms.hip)
    if world > 1:
        from lance_amd import dist as ld
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        ld._all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
        if "list_sharded_qps_strong_scaling" in mg:
            tl = torch.tensor([mg["list_sharded_qps_strong_scaling"]], dtype=torch.float64, device=dev)
            ld._all_reduce(tl, op=dist.ReduceOp.MIN)
            mg["list_sharded_qps_strong_scaling"] = tl.item()

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
    quantised = kt["ivfpq_scan_c1"][1] > 0 and not os.environ.get("LANCE_HIP_NO_QSCAN") and not os.environ.get("LANCE_HIP_PM_NOBOUND")
    if kt["ivfpq_scan_c1"][1] > 0:      # partition-major path: dominant launch = the main pass
        scan_ms, scan_launches = kt["ivfpq_scan_c1"]
        if quantised:
            bytes_list = scan_bytes
            kernel_name = "ivfpq_qscan_kernel<SD=8,MU=1> (main pass: 4-query u16 filter scan of all nprobes partitions)"
            queries_per_gather = 4
        elif os.environ.get("LANCE_HIP_PM_NOBOUND"):
            bytes_list, kernel_name = scan_bytes_c1, "ivfpq_scan_pm_kernel<SD=8,L2,MU=1,RPL=2> (class-1 launch: the nprobes-1 farther partitions)"
            queries_per_gather = 2
        else:
            bytes_list, kernel_name = scan_bytes, "ivfpq_scan_pm_kernel<SD=8,L2,MU=1,RPL=2> (main pass: all nprobes partitions of every query)"
            queries_per_gather = 2
    else:
        scan_ms, scan_launches = kt["ivfpq_scan"]
        bytes_list, kernel_name = scan_bytes, "ivfpq_scan_kernel<SD=8,L2,MU=1>"
        queries_per_gather = 1
    avg_scan_ms = scan_ms / max(scan_launches, 1)
    avg_bytes = float(np.mean([bytes_list[i % 4] for i in range(args.steps)]))
    # The scan's binding resource is the LDS gather (with the VALU issue slots next to it), NOT HBM: partition-major order
    # keeps the 16 MB code table in L2 (rocprofv3 FETCH/WRITE: 4 % of HBM peak, profiles/).  Roofline unit = lane-gathers:
    # one (row, sub-quantiser) table lookup serving `queries_per_gather` queries.  Algorithmic gathers per launch =
    # sum over (query, probe) pairs of n_p * M / queries_per_gather; peak = the random-gather rate of the same table shape
    # ([16][256] entries of 8 bytes) measured IN THIS RUN by lance_hip_ubench.
    lut_values = avg_bytes                     # n_p * M summed over pairs: one LUT value per (row, sub-quantiser, query)
    gathers = lut_values / queries_per_gather
    gather_rate = gathers / (avg_scan_ms * 1e-3) if avg_scan_ms > 0 else 0.0
    ceil_key = {1: "lds4", 2: "lds8", 4: "lds8"}[queries_per_gather]
    ceiling = eng.ubench(ceil_key)
    copy_bw = eng.ubench("copy")
    hbm_equiv = avg_bytes / (avg_scan_ms * 1e-3) / 1e9 if avg_scan_ms > 0 else 0.0

    # HBM traffic of the dominant kernel: measured IN THIS RUN by two rocprofv3 --pmc child passes (collect_pmc_traffic); the figure
    # recorded under profiles/ in an earlier round is kept beside it under its own key, never as `traffic`
    traffic, traffic_src, traffic_detail = None, None, None
    pmc_kernel = ("ivfpq_mscan_kernel<8, 8" if mscan else "ivfpq_qscan_kernel<8, 1") if (quantised and args.config == "c2") else None
    if pmc_kernel and not args.no_pmc and world == 1:
        child = ["--steps", "3", "--warmup", "1", "--streams", "1", "--no-cpu-baseline", "--no-pmc", "--config", args.config, "--n", str(args.n),
                 "--nq", str(args.nq), "--nprobes", str(args.nprobes), "--refine", str(args.refine), "--k", str(args.k)]
        traffic_detail, traffic_src = collect_pmc_traffic(pmc_kernel, child)
        if traffic_detail:
            traffic = traffic_detail["hbm_bytes_per_launch"]
    traffic_recorded = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_pmc_tcc.json")))
        for kn, kv in pmc["kernels"].items():
            if pmc_kernel and pmc_kernel in kn:
                traffic_recorded = {"hbm_bytes_per_launch": kv["hbm_bytes_per_launch"], "source": "profiles/r03_bench_pmc_tcc.json (round 3 tree; not this run)"}
    except Exception:
        pass
    # ---- the build half of the metric against ITS rooflines (world 1): (i) the transform pass (assign + residual + encode of every
    # row, one streaming read of the column) against HBM; (ii) one IVF E-step of the training sample (65,536 x nlist x d) against the
    # dense bf16 MFMA peak -- executed flop = 3 x algorithmic (two-term bf16 split: hi.hi + lo.hi + hi.lo)
    roofline_build = None
    if world == 1 and not multi:
        es = 2 if half else 1 if int8c else 4
        tr_bytes = float(args.n) * d * es + float(args.n) * (4 + m)
        tr_sec = idx.stats.seconds.get("transform") if idx.stats else None
        ns = min(args.n, nlist * 256)
        xs_ = x[:ns]
        cent_ = idx._ix.centroids
        for _ in range(3):
            eng.assign(xs_, cent_, "l2")
        torch.cuda.synchronize()
        reps_e = 20 if nlist <= 4096 else 2
        eng.timing(True)          # HIP events around the MFMA sweep kernel itself (a host clock over an 85 us call measured the launch gaps)
        eng.timing_query("ma_sweep"); eng.timing_query("ma_recheck")
        t1 = time.perf_counter()
        for _ in range(reps_e):
            eng.assign(xs_, cent_, "l2")
        torch.cuda.synchronize()
        e_wall = (time.perf_counter() - t1) / reps_e
        sw_ms, sw_n = eng.timing_query("ma_sweep")
        rc_ms, rc_n = eng.timing_query("ma_recheck")
        eng.timing(False)
        e_sec = (sw_ms / max(sw_n, 1)) * 1e-3 if sw_n else e_wall
        e_recheck = (rc_ms / max(rc_n, 1)) * 1e-3 if rc_n else None
        e_flop = 2.0 * ns * nlist * d
        MFMA_BF16_PEAK = 2500.0      # TFLOP/s dense bf16, MI355X_MICROARCH.md
        roofline_build = {
            "transform": {"bound": "hbm", "algorithmic_bytes": tr_bytes, "seconds": tr_sec,
                          "achieved": (tr_bytes / tr_sec / 1e9) if tr_sec else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": (tr_bytes / tr_sec / 1e9 / HBM_PEAK_GBS) if tr_sec else None,
                          "what": "lance_hip_ivfpq_encode over all rows as create_index calls it (host wall clock of the call): ONE kernel (xf_kernel, xform_fused.hip) "
                                  "reads every row once -- MFMA coarse assign, exact re-check, residual, MFMA PQ encode -- plus two clean-up kernels for the ~1 % undecided "
                                  "items; bytes = N*d*s read + N*(4+M) written.  The kernel is VALU-issue bound, not HBM bound: 3 VALU per (row, codeword) pair "
                                  "(profiles/r06_xform_fused_notes.txt)"},
            "estep_ivf": {"bound": "mfma", "rows": ns, "centroids": nlist, "d": d, "seconds": e_sec,
                          "achieved": e_flop / e_sec / 1e12, "achieved_executed": 3.0 * e_flop / e_sec / 1e12, "peak": MFMA_BF16_PEAK,
                          "unit": "TFLOP/s", "frac": e_flop / e_sec / 1e12 / MFMA_BF16_PEAK, "frac_executed": 3.0 * e_flop / e_sec / 1e12 / MFMA_BF16_PEAK,
                          "peak_measured": eng.ubench("mfma_bf16") / 1e12,
                          "peak_measured_source": "lance_hip_ubench(mfma_bf16): v_mfma_f32_32x32x16_bf16 register loop, measured in this run",
                          "exact_recheck_seconds": e_recheck, "call_wall_seconds": e_wall,
                          "what": "the E-step kernel of lance_hip_assign over the training sample against the trained centroids (f32 rows, d <= 128: xf_kernel<.., ASSIGN>, "
                                  "xform_fused.hip -- rows -> bf16x3 MFMA sweep, four smallest kept -> exact re-check of the candidates, ONE kernel; other "
                                  "shapes: ma_top3_kernel), mean of 20 launches by HIP events on the engine's stream; the recompute-list kernel and the "
                                  "whole call's wall time beside it; algorithmic flop = 2*n*k*d"},
            "build_stages_ms": {k_: round(v * 1e3, 3) for k_, v in (idx.stats.seconds.items() if idx.stats else [])},
        }
    guide_lds_peak = LDS_B64_CONFLICT_FREE_PER_CLK_CU * 256 * 2.4e9      # lane-gathers/s: ds_read_b64, 256 B/clk/CU, 256 CUs, 2.4 GHz
    c4 = args.config == "c4"
    result = {
        "metric": ("QPS @ recall@10 (SIFT-1M IVF_PQ nlist=256 M=16) + index-build sec" if args.config == "c2" else
                   "QPS @ recall@10 (synthetic f16 x128, IVF_PQ nlist=4096 M=16; BASELINE config 4 shape) + index-build sec" if c4 else
                   "QPS @ recall@10 (synthetic int8 x128, IVF_PQ nlist=65536 M=32, hierarchical k-means; BASELINE config 5 shape) + index-build sec"),
        "value": qps,
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32" if args.config == "c2" else "f16 rows, f32 arithmetic" if c4 else "int8 rows, f32 arithmetic",
        "data": "synthetic",
        "config": {"workload": ("SIFT-1M-like 1Mx128 f32, IVF_PQ nlist=256 M=16 nbits=8, build+search on MI355X" if args.config == "c2" else
                                f"C4 shape: {args.n} x 128 f16 rows, IVF_PQ nlist=4096 M=16 nbits=8, build+search on MI355X" if c4 else
                                f"C5 shape: {args.n} x 128 int8 rows, IVF_PQ nlist=65536 M=32 nbits=8 (hierarchical k-means), build+search on MI355X"),
                   "n": args.n, "d": d, "nlist": nlist, "m": m, "queries_per_step_per_gpu": args.nq, "k": args.k,
                   "nprobes": args.nprobes, "refine_factor": args.refine,
                   "parallelism": (f"build: rows sharded over {world} ranks, RCCL all-reduce per Lloyd iteration; search: replica x{world} "
                                   f"(weak) + list-sharded (strong)") if multi else "single"},
        "recall_at_10": recall,
        "exact_replays_last_step": exact_replays,
        "streams": nstreams,
        "graph_priming_calls": graph_priming,
        "host_buffers_qps_pcie_inclusive": pcie_qps,
        "build_sec": build_sec,
        "build_sec_pcie_inclusive": build_sec_pcie,
        "qps_f32_refine_source": None,
        # N > 1: the two numbers that say something about scaling (the replica `value` is linear by construction): the SAME
        # query batches answered jointly by list shards (strong scaling), and the row-sharded build with one all-reduce per
        # Lloyd iteration.  null at N = 1.  No multi-GPU hardware record exists for them before the driver's SCALE run.
        "strong_scaling_list_sharded_qps": mg.get("list_sharded_qps_strong_scaling") if mg else None,
        "build_sec_rows_sharded_allreduce": mg.get("build_sec_ivf_sharded_allreduce") if mg else None,
        "multi_gpu": mg or None,
        "build_stages_ms": {k_: round(v * 1e3, 3) for k_, v in (idx.stats.seconds.items() if idx.stats else [])},
        "kernel_ms_per_step": {k_: round(v[0] / max(v[1], 1), 4) for k_, v in kt.items()},
        # frac = achieved / peak with peak from MI355X_MICROARCH.md (LDS table: conflict-free ds_read_b64, 2 cycles per wave-instruction
        # = 32 lanes/clk/CU, x 256 CUs x 2.4 GHz); the same rate against the random-gather microbenchmark of this run is kept as
        # frac_of_measured_gather (that kernel shares the scan's bank-conflict pathology: it is a floor for "what this table shape
        # can deliver", not a roofline)
        "roofline_build": roofline_build,
        "latency": latency,
        "recall_grid": recall_grid,
        "roofline": None,
    }
    common = {"traffic": traffic, "traffic_unit": "HBM bytes per launch", "traffic_source": traffic_src, "traffic_detail": traffic_detail,
              "traffic_recorded": traffic_recorded, "traffic_vs_algorithmic_code_bytes": (traffic / avg_bytes) if (traffic and avg_bytes) else None,
              "avg_launch_ms": avg_scan_ms, "survey_8d_code_bytes_per_launch": avg_bytes, "survey_8d_byte_rate_GBps": hbm_equiv,
              "hbm_peak_GBps": HBM_PEAK_GBS, "device_copy_GBps_measured": copy_bw / 1e9}
    if mscan:
        # The main pass is the matrix-core filter scan: per partition a [queries x d] x [d x rows] f16 product (v_mfma_f32_32x32x16_f16)
        # whose accumulator starts at |c^|^2 - limit, one compare per (row, query).  Algorithmic flop per launch = 2 d x the number of
        # (row, query) cells = 2 d x sum over (query, probe) pairs of n_p (avg_bytes / M); peak = the dense f16 MFMA rate of the guide.
        cells = avg_bytes / m
        flop = 2.0 * d * cells
        MFMA_F16_PEAK = 2500.0      # TFLOP/s dense f16 / bf16, MI355X_MICROARCH.md
        mfma_meas = eng.ubench("mfma_f16") / 1e12      # the same instruction in a plain register loop on THIS box (lance_hip_ubench 10)
        ach = flop / (avg_scan_ms * 1e-3) / 1e12 if avg_scan_ms > 0 else 0.0
        la_rate = avg_bytes / (avg_scan_ms * 1e-3) if avg_scan_ms > 0 else 0.0
        result["roofline"] = dict(common, kernel="ivfpq_mscan_kernel<SD=8,KS=8> (main pass: matrix-core filter scan of all nprobes partitions, "
                                  "32 rows x 32 queries per v_mfma_f32_32x32x16_f16 chain)",
                                  bound="mfma", achieved=ach, peak=MFMA_F16_PEAK, unit="TFLOP/s", frac=ach / MFMA_F16_PEAK,
                                  peak_source="MI355X_MICROARCH.md: dense f16 MFMA ~2.5 PFLOP/s",
                                  peak_measured=mfma_meas, frac_of_peak_measured=(ach / mfma_meas) if mfma_meas else None,
                                  peak_measured_source="lance_hip_ubench(mfma_f16): v_mfma_f32_32x32x16_f16, four independent accumulator chains per wave, "
                                                       "two waves per SIMD, operands in registers, measured in this run",
                                  # SURVEY 8(d)'s unit against the same peak: the LUT formulation of the reference needs M lookup-adds per (row, query)
                                  # cell; counting each as ONE flop, the fraction of the MFMA peak's flop rate this kernel delivers them at
                                  frac_algorithmic=(la_rate / 1e12 / MFMA_F16_PEAK),
                                  frac_algorithmic_what="SURVEY 8(d) lookup-adds per second (M per (row, query) cell, one flop each) / dense f16 MFMA flop per second; "
                                                        "the executed-flop `frac` is 2 d / M = 16 x this",
                                  executed_flop_per_launch=flop, row_query_cells_per_launch=cells,
                                  cells_per_s=cells / (avg_scan_ms * 1e-3) if avg_scan_ms > 0 else 0.0,
                                  # SURVEY 8(d)'s own unit beside it: the LUT formulation needs M lookup-adds per (row, query) cell; the rate this
                                  # kernel delivers them at, against the guide's conflict-free LDS gather rate (what a LUT scan is bound by)
                                  survey_8d_lookup_adds_per_launch=avg_bytes,
                                  survey_8d_lookup_adds_per_s=avg_bytes / (avg_scan_ms * 1e-3) if avg_scan_ms > 0 else 0.0,
                                  lds_gather_peak_per_s=guide_lds_peak,
                                  lookup_add_rate_vs_lds_gather_peak=(avg_bytes / (avg_scan_ms * 1e-3) / guide_lds_peak) if avg_scan_ms > 0 else None,
                                  flop_note="executed flop: the filter's formulation (a dense [queries x d] x [d x rows] reconstruction product, 2 d flop "
                                            "per cell) -- 16 x the arithmetic of the LUT formulation (M lookup-adds per cell); frac is the fraction of the "
                                            "MFMA peak the kernel as written reaches",
                                  note="survivors of the filter (~250 per query) are re-evaluated in the reference's arithmetic by the merge kernel; "
                                       "HBM is not the bound (codes, codebook and residual blocks are L2-resident); SQ counters and phase stamps "
                                       "under profiles/r04*_mscan*")
    else:
        result["roofline"] = dict(common, kernel=kernel_name, bound="lds", achieved=gather_rate / 1e9, peak=guide_lds_peak / 1e9,
                                  unit="G lane-gathers/s", frac=gather_rate / guide_lds_peak,
                                  peak_source="MI355X_MICROARCH.md, LDS: ds_read_b64 conflict-free = 32 lanes/clk/CU x 256 CUs x 2.4 GHz",
                                  frac_of_measured_gather=gather_rate / ceiling if ceiling else None,
                                  measured_gather_G_per_s=ceiling / 1e9,
                                  measured_gather_source=f"lance_hip_ubench({ceil_key}): random ds_read_b64 gathers of a [16][256] table, measured in this run",
                                  queries_per_gather=queries_per_gather,
                                  lut_values_per_s=lut_values / (avg_scan_ms * 1e-3) if avg_scan_ms > 0 else 0.0,
                                  algorithmic_gathers_per_launch=gathers,
                                  note="HBM is not the bound (code table is L2-resident); PMC evidence (LDS busy, VALU issue, bank "
                                       "conflicts, FETCH/WRITE) is under profiles/")

    # ---- the other wide kernels of a step, each against the resource that bounds it (the small ones -- grouping, slice tables, pre-pass, coarse
    # quantiser -- are launch-latency chains that the other engine contexts' work hides: the step's throughput is set by these four)
    def per_launch(name):
        ms_, n_ = kt[name]
        return ms_ / max(n_, 1)
    keff_b = args.k * max(args.refine, 1)
    refine_u8 = eng.timing_query("count:refine_u8")[1] > 0      # the index kept a lossless u8 copy of the integer-valued f32 column (index.h raw_u8)
    refine_bytes = float(args.nq) * keff_b * d * (1 if (refine_u8 or int8c) else 2 if half else 4)
    result["refine_source"] = ("lossless u8 copy of the f32 raw column (every element is an integer in [0, 255], checked on the bits when the index was "
                               "built; same f32 values after widening, same arithmetic): 1 byte per element" if refine_u8 else
                               "the caller's raw column, " + ("2" if half else "4") + " bytes per element")
    near_bytes = float(np.mean([scan_bytes[i % 4] - scan_bytes_c1[i % 4] for i in range(args.steps)]))      # n_p * M over the (query, nearest partition) pairs
    rt = {}
    if per_launch("refine") > 0:
        r_ms = per_launch("refine")
        rt["refine_kernel"] = {"bound": "hbm", "algorithmic_bytes": refine_bytes, "avg_launch_ms": r_ms, "achieved": refine_bytes / (r_ms * 1e-3) / 1e9,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": refine_bytes / (r_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "what": f"{args.nq} x {keff_b} candidate rows of {d} elements read at random from the refine source (two lanes per row, whole row in flight)"}
    msbound = eng.timing_query("count:ivfpq_msbound")[1] > 0      # the bound pass ran on the matrix cores (search_ms.hip: ms_bound_kernel)
    if per_launch("ivfpq_scan_c0") > 0 and msbound:
        b_ms = per_launch("ivfpq_scan_c0")
        cells_b = near_bytes / m
        fl_b = 2.0 * d * cells_b
        rt["bound_pass"] = {"kernel": "ms_bound_kernel<SD=8,KS=8> (+ its two item-table kernels: the stage's HIP-event time)", "bound": "mfma",
                            "row_query_cells": cells_b, "executed_flop": fl_b, "avg_launch_ms": b_ms, "achieved": fl_b / (b_ms * 1e-3) / 1e12,
                            "peak": 2500.0, "unit": "TFLOP/s", "frac": fl_b / (b_ms * 1e-3) / 1e12 / 2500.0,
                            "what": "512-bin histogram of |c^|^2 - 2 r.c^ + |r|^2 over every query's NEAREST partition from the f16 product on "
                                    "v_mfma_f32_32x32x16_f16 (2 d flop per (row, query) cell, one FMA + one compare + one LDS atomic per cell behind it); "
                                    "a tenth of the main scan's cells, latency and LDS-atomic bound rather than MFMA bound -- the fraction says how far"}
    elif per_launch("ivfpq_scan_c0") > 0:
        b_ms = per_launch("ivfpq_scan_c0")
        g_ = near_bytes / 4.0
        rt["bound_pass"] = {"kernel": "ivfpq_qbound_kernel (integer tables)", "bound": "lds", "algorithmic_gathers": g_, "avg_launch_ms": b_ms, "achieved": g_ / (b_ms * 1e-3) / 1e9, "peak": guide_lds_peak / 1e9,
                            "unit": "G lane-gathers/s", "frac": g_ / (b_ms * 1e-3) / guide_lds_peak,
                            "what": "integer-table histogram over every query's nearest partition, four queries per 8-byte LDS gather (item tables + residual "
                                    "pre-pass + ivfpq_qbound_kernel)"}
    if per_launch("ivfpq_merge") > 0:
        m_ms = per_launch("ivfpq_merge")
        rt["merge"] = {"bound": "latency", "avg_launch_ms": m_ms, "queries": args.nq, "dependent_memory_round_trips_per_query": 8,
                       "workgroups_in_flight": 256 * 10, "us_per_workgroup_at_that_occupancy": m_ms * 1e3 * (256 * 10) / args.nq,
                       "what": "rescan of overflowed segments + ivfpq_qmerge1g_kernel: per query a chain of ~8 dependent L2 / HBM round trips (counts + probes -> "
                               "survivors + residual rows -> codes -> 4 x codebook -> row ids) around a few thousand lane-operations; bound by latency x "
                               "workgroups in flight (10 per CU by LDS), not by a throughput roofline"}
    result["roofline_tail"] = rt or None

    # the headline refine reads the lossless u8 copy that an integer-valued f32 column admits; the same steps with the refine kernel on the
    # caller's f32 rows (what a general f32 column gets), from a child run of this script with LANCE_HIP_NO_RAW_COMPACT=1
    if world == 1 and not multi and not args.no_extras and refine_u8 and os.environ.get("LANCE_BENCH_CHILD") != "1":
        import subprocess
        env = dict(os.environ, LANCE_HIP_NO_RAW_COMPACT="1", LANCE_BENCH_CHILD="1")
        cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(args.steps), "--warmup", str(args.warmup), "--streams", str(args.streams),
               "--no-pmc", "--no-cpu-baseline", "--no-grid", "--no-extras", "--config", args.config, "--n", str(args.n), "--nq", str(args.nq),
               "--nprobes", str(args.nprobes), "--refine", str(args.refine), "--k", str(args.k)]
        try:
            cp = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
            cj = json.loads(cp.stdout.strip().splitlines()[-1])
            result["qps_f32_refine_source"] = {"value": cj["value"], "ms_per_step": cj["ms_per_step"], "recall_at_10": cj["recall_at_10"],
                                               "refine_kernel_ms": cj["kernel_ms_per_step"].get("refine"),
                                               "what": "child run of this script with LANCE_HIP_NO_RAW_COMPACT=1: same index, same batches, the refine kernel "
                                                       "reads the caller's f32 rows (4 bytes per element)"}
        except Exception as e:      # the extra is informational: never fail the line over it
            result["qps_f32_refine_source"] = {"error": str(e)[:200]}

    # the other metrics on the same shape (one engine context, unit-normalised rows -- what a dot index is normally built on): L2 / cosine / dot
    # through the matrix-core flow, and dot on the exact pair scan it replaces there (LANCE_HIP_NO_DOT_FLOW=1); children of scripts/probe_dot_flow.py
    if world == 1 and not multi and not args.no_extras and args.config == "c2" and os.environ.get("LANCE_BENCH_CHILD") != "1":
        import subprocess
        probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scripts", "probe_dot_flow.py")
        om = {"what": "1M x 128 unit-normalised SIFT-like rows, IVF_PQ(256,16), 10,000-query batches, k 10, nprobes 10, refine 10, ONE engine context, "
                      "wall clock over 20 batches (scripts/probe_dot_flow.py json)"}
        try:
            cp = subprocess.run([sys.executable, probe, "json", "l2", "cosine", "dot"], env=dict(os.environ, LANCE_BENCH_CHILD="1"), capture_output=True, text=True, timeout=300)
            om.update(json.loads([l for l in cp.stdout.strip().splitlines() if l.startswith("{")][-1]))
            cp = subprocess.run([sys.executable, probe, "json", "dot"], env=dict(os.environ, LANCE_BENCH_CHILD="1", LANCE_HIP_NO_DOT_FLOW="1"), capture_output=True, text=True, timeout=300)
            om["dot_exact_pair_scan"] = json.loads([l for l in cp.stdout.strip().splitlines() if l.startswith("{")][-1])["dot"]
        except Exception as e:      # informational: never fail the line over it
            om["error"] = str(e)[:200]
        result["other_metrics"] = om

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
        tot_cpu = 0.0
        while reps < 3 or (time.perf_counter() - t_all < 10 and reps < 20):
            t1 = time.perf_counter()
            oi, _ = oidx.search(xq, args.k, args.nprobes, refine=args.refine, raw=raw)
            dt_ = time.perf_counter() - t1
            best = min(best, dt_)
            tot_cpu += dt_
            reps += 1
        gi, _ = idx.search_device(qbatches[0], args.k, args.nprobes, args.refine)
        same = bool((oi == gi.cpu().numpy().view(np.uint64)).all())
        # CPU index build, full (the other half of the metric): sample -> IVF k-means (<= 50 iterations, tol 1e-4) -> residuals
        # -> 16 PQ k-means -> assign + encode all rows + per-partition layout, mirroring the reference's stage log lines
        # (ivf.rs:1239-1272, builder.rs:415-466)
        cpu_build = {}
        samp = raw[: nlist * 256]
        t1 = time.perf_counter()
        ocent, _, oiters, _ = orc.kmeans_train(samp, nlist, max_iters=50, balance_factor=np.float32(1.0) / np.float32(samp.shape[0]), seed=1)
        cpu_build["train_ivf"] = time.perf_counter() - t1
        cpu_iter = cpu_build["train_ivf"] / max(int(oiters), 1)
        t1 = time.perf_counter()
        opart, _ = orc.assign(samp, ocent)
        ores = orc.residual(samp, ocent, opart)
        ocb, _ = orc.pq_train(ores, m, max_iters=50, seed=2)
        cpu_build["train_pq"] = time.perf_counter() - t1
        t1 = time.perf_counter()
        orc.build_index(raw, ocent, ocb)
        cpu_build["transform+partitions"] = time.perf_counter() - t1
        cpu_build_sec = sum(cpu_build.values())
        result["cpu_baseline"] = {"value": args.nq * reps / tot_cpu, "value_best_run": args.nq / best, "unit": "queries/s", "cores": cores, "kind": "port",
                                  "sample": f"the same {args.nq}-query batch, same index/nprobes/refine, MEAN over {reps} runs "
                                            f"of oracle/lance_oracle.c (OpenMP over queries) -- the same statistic as the GPU figure; the best run beside it",
                                  "ids_equal_gpu": same,
                                  "oracle_march": "native" if native else "x86-64-v3",
                                  "build_sec": cpu_build_sec,
                                  "build_stages_sec": {k_: round(v, 3) for k_, v in cpu_build.items()},
                                  "build_sample": "the full build: 65,536-row training sample, 1,000,000 rows encoded",
                                  "ivf_kmeans_iterations": int(oiters),
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
