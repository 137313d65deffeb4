"""Multi-GPU index build: one process per GPU, torch.distributed (RCCL over xGMI).

The reference has no distributed path; this is the MI355X-native extension:

  * IVF k-means: the training rows are sharded by contiguous row ranges; centroids are replicated.  Each Lloyd iteration runs the local E-step and the
    local per-centroid partial sums on the device (lance_hip_kmeans_estep_partial), then ONE
    all-reduce(sum) of the fused f32 buffer [k*d sums | k counts] (129 KiB for SIFT IVF256),
    one all-reduce(sum) of the f64 per-cluster losses and one all-reduce(max) of the radii,
    after which every rank finalises identical centroids and evaluates the same convergence
    test (kmeans.rs:665-712).  Empty-cluster splits use a seed shared by all ranks.
  * PQ codebook: the M sub-quantisers are independent k-means problems (pq/builder.rs:109-138), so they
    are spread over the ranks (M=16 >= 8) and the 8 KiB codebook slices are all-gathered.
  * transform (assign + residual + PQ encode): embarrassingly parallel by rows; the
    (part id, code) columns are all-gathered so that every rank holds a full replica of the
    index for search (16 MB of codes for SIFT-1M).
  * search: replicas, the query batch is split by the caller -- no data-path collective.

Sharded sums are added in a different order than the single-GPU row-order chain, so the
multi-GPU centroids agree with the single-GPU ones to f32 round-off, not bit for bit; the
inference kernels (assign / encode / search) stay bit-exact given the same model.

The loop is written against a small engine interface (estep_partial / finalize) so the
world_size-2 gloo tests can drive it on CPU with a stand-in engine.
"""
import numpy as np
import torch
import torch.distributed as dist

from ._rng import Rng, kmeans_init_indices

FLT_MAX = float(np.finfo(np.float32).max)


# ---- collectives ------------------------------------------------------------------------------------------------------------
# Production: backend "nccl" (= RCCL over xGMI), device tensors, everything ordered on the current stream.  The same host logic must be
# exercisable where RCCL cannot run -- two ranks sharing ONE GPU (RCCL refuses a duplicate device), or a CPU box: under the gloo backend
# a device tensor takes the round trip through host memory here (synchronous with the current stream), so every code path above the
# transport -- sharding, fused buffers, update kernels, list exchange, device merge -- runs on the real kernels at world size > 1
# (tests/test_zz_gpu_two_ranks.py; `LANCE_BENCH_BACKEND=gloo LANCE_BENCH_ONE_GPU=1 python bench.py --gpus 2`).
def _host_transport(t, group=None):
    return isinstance(t, torch.Tensor) and t.is_cuda and dist.get_backend(group) == "gloo"


def _all_reduce(t, op, group=None):
    if _host_transport(t, group):
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
        return
    dist.all_reduce(t, op=op, group=group)


def _broadcast(t, src, group=None):
    if _host_transport(t, group):
        h = t.cpu()
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
        return
    dist.broadcast(t, src=src, group=group)


def _all_gather(outs, t, group=None):
    if _host_transport(t, group):
        hs = [o.cpu() for o in outs]
        dist.all_gather(hs, t.cpu(), group=group)
        for o, h in zip(outs, hs):
            o.copy_(h)
        return
    dist.all_gather(outs, t, group=group)


def _all_gather_into_tensor(out, t, group=None):
    if _host_transport(t, group):
        h = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(h, t.cpu(), group=group)
        out.copy_(h)
        return
    dist.all_gather_into_tensor(out, t, group=group)


def _all_to_all_single(out, t, output_split_sizes=None, input_split_sizes=None, group=None):
    if _host_transport(t, group):
        h = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(h, t.cpu(), output_split_sizes=output_split_sizes, input_split_sizes=input_split_sizes, group=group)
        out.copy_(h)
        return
    dist.all_to_all_single(out, t, output_split_sizes=output_split_sizes, input_split_sizes=input_split_sizes, group=group)


def _split_clusters(n, cnts, centroids, rng):
    """split_clusters (kmeans.rs:174-207) on host numpy arrays (f32), shared-seed RNG."""
    f32 = np.float32
    k, dim = centroids.shape
    eps = f32(1.0 / 1024.0)
    for i in range(k):
        if cnts[i] == 0:
            j = 0
            while True:
                p = (f32(cnts[j]) - f32(1.0)) / f32(n - k)
                if f32(rng.next_f32()) < p:
                    break
                j = (j + 1) % k
            cnts[i] = cnts[j] // 2
            cnts[j] -= cnts[i]
            even = np.arange(dim) % 2 == 0
            ci = np.where(even, centroids[j] * (f32(1) + eps), centroids[j] * (f32(1) - eps)).astype(f32)
            cj = np.where(even, centroids[j] * (f32(1) - eps), centroids[j] * (f32(1) + eps)).astype(f32)
            centroids[i] = ci
            centroids[j] = cj


def train_kmeans_sharded(engine, x_local, k, n_total, max_iters=50, tol=1e-4, balance_factor=0.0, init=None, seed=0,
                         metric="l2", group=None):
    """Distributed KMeans::train_kmeans.  x_local: this rank's rows; n_total: rows over all ranks.
    init: [k,d] initial centroids (identical on all ranks) or None -> rank 0 draws k of ITS rows
    (kmeans_random_init shape) and broadcasts them.  -> (centroids, loss, iters)"""
    f32 = np.float32
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    d = x_local.shape[1]
    dev = x_local.device if isinstance(x_local, torch.Tensor) else torch.device("cpu")
    if init is None:
        cent = torch.empty((k, d), dtype=torch.float32, device=dev)
        if rank == 0:
            idx = kmeans_init_indices(int(x_local.shape[0]), k, seed)
            rows = x_local[torch.from_numpy(idx.astype(np.int64)).to(dev)] if isinstance(x_local, torch.Tensor) \
                else torch.from_numpy(np.asarray(x_local)[idx.astype(np.int64)])
            cent.copy_(rows)
        if world > 1:
            _broadcast(cent, src=0, group=group)
    else:
        cent = torch.as_tensor(init, dtype=torch.float32).to(dev).clone()
    bf_param = f32(balance_factor) / f32(n_total)          # train_kmeans :1344
    if hasattr(engine, "kmeans_shard_begin") and isinstance(x_local, torch.Tensor) and x_local.is_cuda:
        return _train_kmeans_sharded_device(engine, x_local, cent, k, n_total, max_iters, tol, float(bf_param), seed, metric, group, world)
    split_rng = Rng(seed ^ 0x5BD1E995)
    sizes = np.zeros(k, np.int64)
    adjusted = f32(FLT_MAX)
    loss = float(np.finfo(np.float64).max)
    last_loss = loss
    iters = 0
    for it in range(1, max_iters + 1):
        iters = it
        bf = min(adjusted, bf_param)
        bias = None
        if bf_param != 0:
            bias = torch.from_numpy((f32(bf) * sizes.astype(f32)).astype(f32)).to(dev)
        buf, losses, radius = engine.kmeans_estep_partial(x_local, cent, metric, bias)
        if world > 1:
            _all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
            _all_reduce(losses, op=dist.ReduceOp.SUM, group=group)
            _all_reduce(radius, op=dist.ReduceOp.MAX, group=group)
        cent = engine.kmeans_finalize(buf, k, d)
        counts = buf[k * d:].cpu().numpy().astype(np.int64)
        lh = losses.cpu().numpy()
        rh = radius.cpu().numpy()
        sizes = counts.copy()
        max_id = int(np.argmax(sizes))                       # first maximal cluster
        adjusted = (f32(rh[max_id]) - f32(lh[max_id]) / f32(sizes[max_id])) / f32(n_total)
        size_loss = f32(int((sizes.astype(object) ** 2).sum()))
        balance_loss = f32(bf) * (size_loss - f32(int(n_total) ** 2) / f32(k))
        last_loss = float(lh.sum()) + float(balance_loss)
        if (sizes == 0).any():
            ch = cent.cpu().numpy()
            _split_clusters(int(n_total), sizes, ch, split_rng)
            cent = torch.from_numpy(ch).to(dev)
        if abs(loss - last_loss) < tol * last_loss:
            break
        loss = last_loss
    return cent, last_loss, iters


def _train_kmeans_sharded_device(engine, x_local, cent, k, n_total, max_iters, tol, bf_scaled, seed, metric, group, world):
    """The same loop with every step enqueued on ONE stream (the engine twin bound to torch's current stream, on which
    torch.distributed orders its RCCL kernels): local E-step + partials -> all-reduce SUM of the fused f32 buffer [k*d sums | k
    counts] and of the f64 losses, all-reduce MAX of the radii -> update kernel (centroids, loss, balance factor,
    convergence, shared-seed split, next bias).  The host only looks at the state every 8 iterations to stop enqueueing."""
    from .engine import Engine
    twin = getattr(engine, "_torch_stream_twin", None)
    if twin is None:
        # a dedicated (non-default) torch stream: kernels enqueued through the twin context and the collectives torch.distributed
        # enqueues while this stream is current are ordered on it, without the legacy default stream's implicit joins
        side = torch.cuda.Stream(device=engine.device)
        with torch.cuda.stream(side):
            twin = Engine(device=engine.device, use_torch_stream=True)
        twin._side_stream = side
        engine._torch_stream_twin = twin
    side = twin._side_stream
    # conversions run on torch's current stream: enqueue them BEFORE the side stream takes its dependency on that stream, so
    # the E-step never reads a widened / compacted copy that is still being written (f16 or strided shards)
    x_local = x_local.to(torch.float32).contiguous()
    cent = cent.contiguous()
    side.wait_stream(torch.cuda.current_stream())
    x_local.record_stream(side)
    cent.record_stream(side)
    loss, iters = 0.0, 0
    with torch.cuda.stream(side):
        st = twin.kmeans_shard_begin(k, x_local.shape[1], bf_scaled, seed)
        for it in range(1, max_iters + 1):
            twin.kmeans_shard_estep(st, x_local, cent, metric)
            if world > 1:
                _all_reduce(st["buf"], op=dist.ReduceOp.SUM, group=group)
                _all_reduce(st["losses"], op=dist.ReduceOp.SUM, group=group)
                _all_reduce(st["radius"], op=dist.ReduceOp.MAX, group=group)
            twin.kmeans_shard_update(st, cent, n_total, tol, it)
            if it % 8 == 0 or it == max_iters:
                loss, iters, active = twin.kmeans_shard_end(st)
                if not active:
                    break
        if iters == 0:
            loss, iters, _ = twin.kmeans_shard_end(st)
    torch.cuda.current_stream().wait_stream(side)
    return cent, loss, iters


# ---- hierarchical k-means (k > 256: BASELINE configs 4 / 5) with its splits spread over the ranks ------------------------------------
class _HCluster:
    __slots__ = ("id", "idx", "centroid", "finalized")

    def __init__(self, cid, idx, centroid):
        self.id, self.idx, self.centroid, self.finalized = cid, idx, centroid, False


def _hc_le(a, b):
    """Ord of the reference's Cluster (kmeans.rs:769-790): unfinalized before finalized, then by size -- the `<=` a max-heap sift uses"""
    ka, kb = (0 if a.finalized else 1), (0 if b.finalized else 1)
    if ka != kb:
        return ka < kb
    return len(a.idx) <= len(b.idx)


class _HHeap:
    """std::collections::BinaryHeap restated (push = sift_up; pop = swap_remove(0) + sift_down_to_bottom + sift_up): the ORDER in which equal-sized
    clusters are popped depends on it, and the run counter that seeds every split follows the pop order (lance_amd/csrc/kmeans.hip HHeap
    is the same restatement)."""

    def __init__(self, d=None):
        self.d = [] if d is None else d

    def _sift_up(self, start, pos):
        e = self.d[pos]
        while pos > start:
            parent = (pos - 1) // 2
            if _hc_le(e, self.d[parent]):
                break
            self.d[pos] = self.d[parent]
            pos = parent
        self.d[pos] = e

    def push(self, c):
        self.d.append(c)
        self._sift_up(0, len(self.d) - 1)

    def pop(self):
        item = self.d.pop()
        if self.d:
            item, self.d[0] = self.d[0], item
            end = len(self.d)
            pos, child = 0, 1
            e = self.d[0]
            while end >= 2 and child <= end - 2:
                if _hc_le(self.d[child], self.d[child + 1]):
                    child += 1
                self.d[pos] = self.d[child]
                pos = child
                child = 2 * pos + 1
            if child == end - 1:
                self.d[pos] = self.d[child]
                pos = child
            self.d[pos] = e
            self._sift_up(0, pos)
        return item

    def shadow(self):
        return _HHeap(list(self.d))          # same structure, same objects: pops on the copy replay the original's next pops


def _hier_cluster_k(size, remaining_k, hk):
    if size <= hk:
        return min(2, remaining_k, size)
    return max(min(size // hk, remaining_k, hk), 2)


def train_kmeans_hierarchical_sharded(engine, sample, k, max_iters=50, tol=1e-4, balance_factor=1.0, hierarchical_k=16, seed=0, metric="l2",
                                      group=None, window=None, stats=None, engines=None):
    """train_hierarchical_kmeans (rust/lance-index/src/vector/kmeans.rs:746-1003) with the splits of the largest clusters computed on
    DIFFERENT ranks at the same time.  `sample` is the whole training sample, identical on every rank (at C5: 16.7M x 128 f32 = 8.6 GB per
    GPU; the caller all-gathers it once).  The reference pops the largest cluster, runs a k-means over its rows (seeded by the number of
    runs so far), pushes the children, and repeats until k clusters exist: a sequential recurrence -- but the next pops are almost always
    the next-largest clusters already in the heap, whatever the split in progress produces.  So every round SPECULATES the next
    `window` pops on a copy of the heap, rank r runs the speculated splits j = r (mod world) (engine.kmeans_split = one C-ABI call each),
    the results are exchanged, and the reference's loop is then replayed on the real heap: a speculated split is applied only if the
    cluster the real heap pops next is the one it was computed for, with the same k and therefore the same seed; the first mismatch
    (a child outgrew a waiting cluster, or the remaining budget changed k) throws the rest of the round away.  The result is the
    single-GPU trainer's BIT FOR BIT (same sub-problems, same seeds, same order) -- unlike the row-sharded Lloyd loop it has an oracle to
    be equal to.  -> centroids [<= k, d] float32 tensor on the sample's device.
    stats (optional dict): rounds, splits applied, splits thrown away.
    engines (optional list): further engine contexts on THIS GPU -- the rank's share of a round then runs on host threads, one context
    (own stream + scratch) each: thousands of the splits are a few thousand rows against sixteen centroids and do not fill a GPU one
    at a time (C5, one GPU: 4,369 splits, 10 s one after the other)."""
    world = dist.get_world_size(group) if (dist.is_initialized() and group is not False) else 1
    rank = dist.get_rank(group) if (dist.is_initialized() and group is not False) else 0
    pool = list(engines) if engines else [engine]
    n, d = sample.shape
    hk = int(hierarchical_k)
    f16 = (isinstance(sample, torch.Tensor) and sample.dtype == torch.float16) or (isinstance(sample, np.ndarray) and sample.dtype == np.float16)
    out_dev = sample.device if isinstance(sample, torch.Tensor) else None
    if isinstance(sample, torch.Tensor) and sample.is_cuda and sample.dtype != torch.float32:
        sample = sample.float().contiguous()      # f16 / int8 samples: widened (exactly) ONCE -- thousands of splits read the sample
    if isinstance(sample, torch.Tensor) and sample.is_cuda:
        if not sample.is_contiguous():
            sample = sample.contiguous()
        torch.cuda.synchronize()                  # the engine contexts run on their own streams: the sample must be complete before the first split
    kw = {"f16_arith": True} if (f16 and isinstance(sample, torch.Tensor) and sample.is_cuda) else {}
    bfs = float(np.float32(balance_factor) / np.float32(n))        # train_kmeans :1344 divides once, by the whole sample
    window = int(window) if window else max(1, 2 * world * len(pool))
    run = 0
    initial_k = min(hk, k, n)
    # first level: one k-means over the whole sample -- computed on every rank (a single problem: nothing to spread)
    c0, mem0 = engine.kmeans_split(sample, None, initial_k, max_iters=max_iters, tol=tol, balance_factor_scaled=bfs, seed=seed + run, metric=metric, **kw)
    run += 1
    heap = _HHeap()
    next_id = 0
    allrows = np.arange(n, dtype=np.uint32)
    for i in range(initial_k):
        idx = allrows[mem0 == i]
        if idx.size == 0:
            continue
        heap.push(_HCluster(next_id, idx, c0[i].copy()))
        next_id += 1

    def next_job(h):
        """the head of the reference's loop on heap h: -> (cluster, cluster_k) with the cluster popped, or None when the loop ends"""
        if len(h.d) >= k or not h.d:
            return None
        big = h.pop()
        if big.finalized or len(big.idx) <= 1:
            h.push(big)
            return None
        return big, _hier_cluster_k(len(big.idx), k - len(h.d), hk)

    n_rounds = n_applied = n_wasted = 0
    pending = None
    tpool = None
    while True:
        first = pending if pending is not None else next_job(heap)
        pending = None
        if first is None:
            break
        # speculate the pops after `first` on a copy: children are assumed non-empty (each split adds cluster_k - 1 clusters) and small
        jobs = [first]
        sh = heap.shadow()
        virt = len(heap.d) + first[1]
        while len(jobs) < window and virt < k and sh.d:
            c = sh.pop()
            if c.finalized or len(c.idx) <= 1:
                break
            ck = _hier_cluster_k(len(c.idx), k - (virt - 1), hk)     # the reference computes remaining_k after the pop
            jobs.append((c, ck))
            virt += ck - 1
        mine = {}
        my_jobs = list(range(rank, len(jobs), world))

        def run_job(t):
            slot, j = t
            c, ck = jobs[j]
            return j, pool[slot % len(pool)].kmeans_split(sample, c.idx, ck, max_iters=max_iters, tol=tol, balance_factor_scaled=bfs,
                                                          seed=seed + run + j, metric=metric, **kw)
        if len(pool) > 1 and len(my_jobs) > 1:
            if tpool is None:
                from concurrent.futures import ThreadPoolExecutor
                tpool = ThreadPoolExecutor(max_workers=len(pool))
            for j, res in tpool.map(run_job, list(enumerate(my_jobs))):
                mine[j] = res
        else:
            for t in enumerate(my_jobs):
                j, res = run_job(t)
                mine[j] = res
        if world > 1:
            gathered = [None] * world
            dist.all_gather_object(gathered, mine, group=group)
            results = {}
            for g in gathered:
                results.update(g)
        else:
            results = mine
        n_rounds += 1
        # replay the reference's loop on the real heap
        for j, (c, ck) in enumerate(jobs):
            if j > 0:
                nj = next_job(heap)
                if nj is None:
                    n_wasted += len(jobs) - j
                    pending = None
                    jobs = jobs[:j]
                    first = None
                    break
                if nj[0] is not c or nj[1] != ck:
                    n_wasted += len(jobs) - j
                    pending = nj
                    break
            cent, mem = results[j]
            run += 1
            n_applied += 1
            valid = mem[mem != 0xFFFFFFFF]
            if valid.size == 0 or (valid == valid[0]).all():      # every row in one child: the cluster cannot be split further
                c.finalized = True
                heap.push(c)
                continue
            for i in range(ck):
                idx = c.idx[mem == i]
                if idx.size == 0:
                    continue
                heap.push(_HCluster(next_id, idx, cent[i].copy()))
                next_id += 1
        else:
            continue
        if first is None and pending is None:
            break
    if tpool is not None:
        tpool.shutdown(wait=True)
    out = sorted(heap.d, key=lambda c: c.id)
    if stats is not None:
        stats.update(rounds=n_rounds, splits_applied=n_applied, splits_thrown_away=n_wasted, window=window, world=world)
    cent = np.stack([c.centroid for c in out]).astype(np.float32) if out else np.zeros((0, d), np.float32)
    t = torch.from_numpy(cent)
    return t.to(out_dev) if out_dev is not None else t


def block_ranges(total, world):
    """contiguous blocks of ceil(total / world) items per rank (the last ranks may be short or empty)"""
    per = (total + world - 1) // world
    return per, [(min(r * per, total), min((r + 1) * per, total)) for r in range(world)]


def all_gather_blocks(local, total, group=None):
    """Every rank holds the rows [lo_r, hi_r) of a `total`-row array (block_ranges); returns the whole array on every
    rank.  One all_gather_into_tensor of equally sized, zero-padded blocks; the padding is dropped afterwards."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per, ranges = block_ranges(total, world)
    lo, hi = ranges[rank]
    assert local.shape[0] == hi - lo, (local.shape, lo, hi)
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: hi - lo] = local
    out = torch.empty((per * world,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    _all_gather_into_tensor(out, pad.contiguous(), group=group)
    if per * world == total:
        return out
    keep = torch.cat([torch.arange(r * per, r * per + (b - a), device=local.device) for r, (a, b) in enumerate(ranges)])
    return out[keep].contiguous()


def create_index_sharded(x, metric="l2", num_partitions=256, num_sub_vectors=16, num_bits=8, max_iters=50, sample_rate=256,
                         seed=42, engine=None, group=None, keep_raw=True, ivf_training="auto", index_factory=None):
    """create_index over N ranks.  `x` is the full matrix on every rank in this version (the bench generates it
    from a shared seed).  Work split: the PQ sub-quantisers are trained model-parallel, the transform (assign +
    residual + encode) is sharded by rows, and the IVF k-means is either
      * "replicated": every rank runs the same deterministic single-GPU training (no collective; the E-step of a
        65,536-row sample takes 0.17 ms, less than one all-reduce round trip), or
      * "sharded": rows split over the ranks, one all-reduce per Lloyd iteration (train_kmeans_sharded) -- for
        training sets large enough that the E-step dominates (C4 and up).
      * "hierarchical" (nlist > 256, where the reference trains hierarchically): the splits of the largest clusters are computed on
        different ranks and applied in the reference's order (train_kmeans_hierarchical_sharded) -- bit-identical to one GPU.
    "auto" picks by the E-step size; nlist > 256 goes hierarchical on more than one rank.  With replicated or hierarchical training the
    whole index is bit-identical to the single-GPU build (same sample, same seeds, independent sub-quantisers, per-row encode).
    `index_factory` (default DeviceIndex.create) exists so that the collective logic can be driven on CPU by the
    world_size > 1 gloo tests with a stand-in engine."""
    import time

    from . import vector as lv
    from .engine import DeviceIndex, to_device
    eng = engine or lv.default_engine()
    on_gpu = torch.cuda.is_available()
    make_index = index_factory or DeviceIndex.create
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    params = lv.IvfPqParams(num_partitions, num_sub_vectors, num_bits, lv._normalize_metric_type(metric), max_iters, sample_rate, seed)
    x = to_device(x) if on_gpu else torch.as_tensor(x)
    n, d = x.shape
    stats = lv.BuildStats()
    kmetric = "l2" if params.metric == "cosine" else params.metric

    def timed(name, fn):
        if on_gpu:
            torch.cuda.synchronize()
        t = time.perf_counter()
        out = fn()
        if on_gpu:
            torch.cuda.synchronize()
        stats.seconds[name] = time.perf_counter() - t
        return out

    def shard(t):
        per = (t.shape[0] + world - 1) // world
        return t[rank * per: min(t.shape[0], (rank + 1) * per)]

    def prep(sample):
        if params.metric == "cosine":
            sample = eng.normalize(sample)
        return sample[torch.isfinite(sample).all(dim=1)]

    # IVF: same global sample on every rank (shared seed), each rank keeps its slice
    rng = np.random.default_rng(seed)
    idx = lv._sample_rows(n, num_partitions * sample_rate, rng)
    sample = prep(x if idx is None else x[torch.from_numpy(idx).to(x.device)])
    n_total = sample.shape[0]
    if ivf_training == "auto":
        # nlist > 256: the reference trains hierarchically (kmeans.rs:1027) -- its splits are spread over the ranks, result bit-identical to
        # the single-GPU trainer (round 6; round 5 kept such builds replicated: every rank trained the same 65,536 clusters)
        ivf_training = ("hierarchical" if num_partitions > 256 and world > 1 else
                        "replicated" if n_total * num_partitions * d <= (1 << 36) or num_partitions > 256 else "sharded")
    stats.ivf_training = ivf_training
    if ivf_training == "hierarchical":
        hst = {}
        cent = timed("train_ivf", lambda: train_kmeans_hierarchical_sharded(eng, sample, num_partitions, max_iters=max_iters, balance_factor=1.0,
                                                                            seed=seed, metric=kmetric, group=group, stats=hst))
        stats.ivf_loss, stats.ivf_iters, stats.ivf_hierarchical = 0.0, 0, hst      # "Loss is not meaningful for hierarchical clustering" (:1001)
    elif ivf_training == "replicated":
        cent, stats.ivf_loss, stats.ivf_iters = timed("train_ivf", lambda: eng.kmeans_train(
            sample, num_partitions, max_iters=max_iters, balance_factor=1.0, seed=seed, metric=kmetric))
    else:
        cent, stats.ivf_loss, stats.ivf_iters = timed("train_ivf", lambda: train_kmeans_sharded(
            eng, shard(sample), num_partitions, n_total, max_iters, 1e-4, 1.0, None, seed, kmetric, group))
    # PQ: residual sample, sub-quantisers trained one after the other, each sharded by rows
    rng2 = np.random.default_rng(seed + 1)
    idx2 = lv._sample_rows(n, sample_rate * (1 << num_bits), rng2)
    psample = prep(x if idx2 is None else x[torch.from_numpy(idx2).to(x.device)])
    if kmetric == "l2":
        part, _ = eng.assign(psample, cent, "l2")
        psample = eng.residual(psample, cent, part)
    kc = 1 << num_bits
    sd = d // num_sub_vectors

    def train_pq():
        # model-parallel over the M independent sub-quantisers (pq/builder.rs:109-138 trains them one after the
        # other): rank r trains a contiguous block with the batched single-GPU trainer on the (small, replicated)
        # residual sample, seeds seed+2+m as in the single-GPU build, then the codebook slices are all-gathered.
        _, ranges = block_ranges(num_sub_vectors, world)
        m0, m1 = ranges[rank]
        mine = torch.zeros((m1 - m0, kc, sd), dtype=torch.float32, device=x.device)
        its = torch.zeros(m1 - m0, dtype=torch.int32, device=x.device)
        if m1 > m0:
            cols = psample[:, m0 * sd: m1 * sd].contiguous()
            c, it = eng.pq_train(cols, m1 - m0, num_bits, max_iters, sample_rate, seed + 2 + m0)
            mine[:] = c
            its[:] = torch.from_numpy(it.astype(np.int32)).to(x.device)
        cb_all = all_gather_blocks(mine, num_sub_vectors, group)
        its_all = all_gather_blocks(its, num_sub_vectors, group)
        return cb_all, its_all.cpu().numpy().astype(np.uint32)

    cb, stats.pq_iters = timed("train_pq", train_pq)

    # transform: each rank encodes its row shard, then all-gather the shuffle-buffer columns
    def transform():
        _, ranges = block_ranges(n, world)
        lo, hi = ranges[rank]
        if hi > lo:
            part_l, codes_l, _ = eng.ivfpq_encode(x[lo:hi], cent, cb, params.metric, want_loss=False)
        else:
            part_l = torch.empty(0, dtype=torch.int32, device=x.device)
            codes_l = torch.empty((0, num_sub_vectors if num_bits == 8 else num_sub_vectors // 2), dtype=torch.uint8, device=x.device)
        return all_gather_blocks(part_l, n, group), all_gather_blocks(codes_l, n, group)

    part, codes = timed("transform", transform)
    ix = timed("build_partitions", lambda: make_index(eng, params.metric, cent, cb, part, codes, None,
                                                      raw=x if keep_raw else None))
    return lv.IvfPqIndex(ix, params, stats, part, codes)


def all_gather_var(local, group=None):
    """all-gather of row blocks of DIFFERENT lengths (rank order): one all-gather of the lengths, one of the zero-padded
    blocks."""
    world = dist.get_world_size(group)
    nloc = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(nloc) for _ in range(world)]
    _all_gather(counts, nloc, group=group)
    counts = [int(c.item()) for c in counts]
    per = max(counts) if counts else 0
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((per * world,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    _all_gather_into_tensor(out, pad.contiguous(), group=group)
    return torch.cat([out[r * per: r * per + c] for r, c in enumerate(counts)]).contiguous(), counts


def exchange_by_owner(owner, tensors, group=None):
    """Rows go to the rank that owns them (`owner` int64 [n], -1 = dropped): one all_to_all_single per tensor after one for
    the counts.  Rows arrive grouped by source rank, in the source's order.  -> list of received tensors."""
    world = dist.get_world_size(group)
    keep = owner >= 0
    order = torch.argsort(owner[keep], stable=True)
    sel = torch.nonzero(keep).reshape(-1)[order]
    send_counts = torch.bincount(owner[keep], minlength=world).to(torch.int64)
    recv_counts = torch.empty_like(send_counts)
    _all_to_all_single(recv_counts, send_counts, group=group)
    sc, rc = [int(v) for v in send_counts.tolist()], [int(v) for v in recv_counts.tolist()]
    out = []
    for t in tensors:
        src = t[sel].contiguous()
        dst = torch.empty((sum(rc),) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        _all_to_all_single(dst, src, output_split_sizes=rc, input_split_sizes=sc, group=group)
        out.append(dst)
    return out


class RowShardedBuild:
    """What create_index_rowsharded leaves on every rank."""

    def __init__(self, centroids, codebook, part_local, codes_local, row0, n_total, stats, params):
        self.centroids, self.codebook = centroids, codebook
        self.part_local, self.codes_local = part_local, codes_local    # this rank's rows only
        self.row0, self.n_total = row0, n_total                          # global row id of local row i = row0 + i
        self.stats, self.params = stats, params


def create_index_rowsharded(x_local, metric="l2", num_partitions=256, num_sub_vectors=16, num_bits=8, max_iters=50, sample_rate=256,
                            seed=42, engine=None, group=None, ivf_training="sharded"):
    """IVF_PQ build where every rank holds ONLY its contiguous block of the rows (rank r: rows block_ranges(n_total)[r]; the
    vectors never leave their GPU).  Per stage:
      * IVF k-means -- each rank samples its share of the num_partitions * sample_rate training rows from its own block;
        "sharded": Lloyd iterations with one all-reduce of [k*d sums | k counts] (+ the per-cluster loss / radius reductions)
        per iteration over RCCL (train_kmeans_sharded); "replicated": the 33 MB sample is all-gathered once and every rank
        runs the same deterministic single-GPU training (no collective inside the loop);
      * PQ -- each rank computes the residuals of its share of the training rows, the residual sample is all-gathered and the
        M independent sub-quantisers are trained model-parallel, then the codebook slices are all-gathered;
      * transform -- every rank encodes its own rows.  Nothing else is exchanged here: `replica_index` / `list_shard_index`
        move the 20 bytes per row of (partition id, PQ code) -- and the raw vectors only if refine needs them on another GPU.
    -> RowShardedBuild"""
    import time

    from . import vector as lv
    from .engine import to_device
    eng = engine or lv.default_engine()
    on_gpu = torch.cuda.is_available()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    params = lv.IvfPqParams(num_partitions, num_sub_vectors, num_bits, lv._normalize_metric_type(metric), max_iters, sample_rate, seed)
    x_local = to_device(x_local) if on_gpu else torch.as_tensor(x_local)
    n_local, d = x_local.shape
    dev = x_local.device
    nl = torch.tensor([n_local], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(nl) for _ in range(world)]
    _all_gather(counts, nl, group=group)
    counts = [int(c.item()) for c in counts]
    n_total, row0 = sum(counts), sum(counts[:rank])
    stats = lv.BuildStats()
    stats.ivf_training = ivf_training
    kmetric = "l2" if params.metric == "cosine" else params.metric

    def timed(name, fn):
        if on_gpu:
            torch.cuda.synchronize()
        t = time.perf_counter()
        out = fn()
        if on_gpu:
            torch.cuda.synchronize()
        stats.seconds[name] = time.perf_counter() - t
        return out

    def local_sample(target_total, salt):
        # this rank's proportional share of a `target_total`-row training sample, drawn from its own rows
        share = n_local if n_total <= target_total else int(round(target_total * n_local / n_total))
        rng = np.random.default_rng(seed + salt + 7919 * rank)
        idx = lv._sample_rows(n_local, share, rng)
        s = x_local if idx is None else x_local[torch.from_numpy(idx).to(dev)]
        if params.metric == "cosine":
            s = eng.normalize(s)
        return s[torch.isfinite(s).all(dim=1)]

    def train_ivf():
        samp = local_sample(num_partitions * sample_rate, 0)
        if ivf_training == "hierarchical" or (ivf_training == "sharded" and num_partitions > 256):
            # nlist > 256 is a hierarchical training in the reference (kmeans.rs:1027): the sample (16.7M x 128 at C5) is all-gathered once,
            # the splits are computed on different ranks side by side and applied in the reference's order (bit-identical to one GPU)
            full, _ = all_gather_var(samp, group)
            hst = {}
            c = train_kmeans_hierarchical_sharded(eng, full, num_partitions, max_iters=max_iters, balance_factor=1.0, seed=seed, metric=kmetric,
                                                  group=group, stats=hst)
            stats.ivf_hierarchical = hst
            return c, 0.0, 0
        if ivf_training == "replicated":
            full, _ = all_gather_var(samp, group)
            return eng.kmeans_train(full, num_partitions, max_iters=max_iters, balance_factor=1.0, seed=seed, metric=kmetric)
        nt = torch.tensor([samp.shape[0]], dtype=torch.int64, device=dev)
        _all_reduce(nt, op=dist.ReduceOp.SUM, group=group)
        return train_kmeans_sharded(eng, samp, num_partitions, int(nt.item()), max_iters, 1e-4, 1.0, None, seed, kmetric, group)

    cent, stats.ivf_loss, stats.ivf_iters = timed("train_ivf", train_ivf)
    kc = 1 << num_bits
    sd = d // num_sub_vectors

    def train_pq():
        ps = local_sample(sample_rate * kc, 1)
        if kmetric == "l2":
            part, _ = eng.assign(ps, cent, "l2")
            ps = eng.residual(ps, cent, part)
        psample, _ = all_gather_var(ps.to(torch.float32), group)
        _, ranges = block_ranges(num_sub_vectors, world)
        m0, m1 = ranges[rank]
        mine = torch.zeros((m1 - m0, kc, sd), dtype=torch.float32, device=dev)
        its = torch.zeros(m1 - m0, dtype=torch.int32, device=dev)
        if m1 > m0:
            cols = psample[:, m0 * sd: m1 * sd].contiguous()
            c, it = eng.pq_train(cols, m1 - m0, num_bits, max_iters, sample_rate, seed + 2 + m0)
            mine[:] = c
            its[:] = torch.from_numpy(np.asarray(it).astype(np.int32)).to(dev)
        return all_gather_blocks(mine, num_sub_vectors, group), all_gather_blocks(its, num_sub_vectors, group).cpu().numpy().astype(np.uint32)

    cb, stats.pq_iters = timed("train_pq", train_pq)

    def transform():
        if n_local == 0:
            return (torch.empty(0, dtype=torch.int32, device=dev),
                    torch.empty((0, num_sub_vectors if num_bits == 8 else num_sub_vectors // 2), dtype=torch.uint8, device=dev))
        p, c, _ = eng.ivfpq_encode(x_local, cent, cb, params.metric, want_loss=False)
        return p, c

    part_l, codes_l = timed("transform", transform)
    return RowShardedBuild(cent, cb, part_l, codes_l, row0, n_total, stats, params)


def replica_index(build, x_local=None, engine=None, group=None, index_factory=None):
    """Full index replica on every rank from a row-sharded build: all-gather of the (partition id, code) columns -- 20 bytes per
    row -- and, when refine is wanted (x_local given), of the raw vectors.  Row ids are global row numbers."""
    from . import vector as lv
    from .engine import DeviceIndex
    eng = engine or lv.default_engine()
    make_index = index_factory or DeviceIndex.create
    part, _ = all_gather_var(build.part_local, group)
    codes, _ = all_gather_var(build.codes_local, group)
    raw = None
    if x_local is not None:
        raw, _ = all_gather_var(torch.as_tensor(x_local), group)
    ix = make_index(eng, build.params.metric, build.centroids, build.codebook, part, codes, None, raw=raw)
    return lv.IvfPqIndex(ix, build.params, build.stats, part, codes), raw


def list_shard_index(build, x_local=None, engine=None, group=None, index_factory=None):
    """IVF lists sharded over the ranks (list p -> rank p % world) from a row-sharded build: every row's (partition id, code,
    global row id [, raw vector]) goes to the rank that owns its list with one all_to_all per column.  The device index
    stores local row numbers (ascending in global id within a source rank, sources in rank order -- so (dist, rowid)
    comparisons on the device order rows as the global ids would) and `l2g` maps them back.  -> (DeviceIndex, l2g)"""
    from . import vector as lv
    from .engine import DeviceIndex
    eng = engine or lv.default_engine()
    make_index = index_factory or DeviceIndex.create
    world = dist.get_world_size(group)
    part = build.part_local.to(torch.int64)
    owner = torch.where((part >= 0) & (part < (1 << 31)), part % world, torch.full_like(part, -1))
    gid = build.row0 + torch.arange(part.shape[0], dtype=torch.int64, device=part.device)
    cols = [build.part_local, build.codes_local, gid]
    if x_local is not None:
        cols.append(torch.as_tensor(x_local))
    got = exchange_by_owner(owner, cols, group)
    pl, cl, l2g = got[0], got[1], got[2]
    # sources arrive in rank order and every source block is ascending in global id, and rank blocks are contiguous in the
    # global numbering: l2g is ascending, local order == global order
    ix = make_index(eng, build.params.metric, build.centroids, build.codebook, pl, cl, None, raw=got[3] if x_local is not None else None)
    return ix, l2g


# ---------------------------------------------------------------------------------------------------------
# Search over IVF lists SHARDED across the ranks (the C5 shape: 32 GB of codes + 8 GB of row ids do
# not belong on one GPU next to the raw vectors).  List p lives on rank p % world.  Centroids and codebook are
# replicated, so every rank picks the same probes for a query and simply finds the lists it does not own empty.
# Each rank answers with its local top-keff (k * refine_factor) by PQ distance, ONE all-gather moves
# nq * keff * 16 bytes per rank, and every rank merges by (distance, row id) -- the order of the reference's
# SortExec over the per-partition heaps (scanner.rs:3440-3468), which is a total order, so the merged result is
# identical to the single-GPU one.  With refine the merged PQ top-keff is re-ranked by the exact distances the
# owning ranks computed for their own candidates (scanner.rs:2884-2904), again identical.

def _order_key(d):
    """f32::total_cmp as an int64 sort key (lance-index graph.rs:66-82)."""
    u = d.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    return torch.where((u & 0x80000000) != 0, (~u) & 0xFFFFFFFF, u | 0x80000000)


def merge_topk(ids, dists, k):
    """ids [nq, C] int64 (-1 = none), dists [nq, C] f32 -> top-k per row by (dist, row id); missing = (-1, +inf)."""
    none = ids < 0
    big = torch.iinfo(torch.int64).max
    idk = torch.where(none, torch.full_like(ids, big), ids)
    dk = torch.where(none, torch.full_like(ids, 1 << 40), _order_key(dists))
    o1 = torch.sort(idk, dim=1, stable=True).indices
    dk1 = torch.gather(dk, 1, o1)
    o2 = torch.sort(dk1, dim=1, stable=True).indices
    order = torch.gather(o1, 1, o2)[:, :k]
    out_i = torch.gather(ids, 1, order)
    out_d = torch.where(out_i < 0, torch.full_like(dists[:, :1], float("inf")).expand(-1, order.shape[1]), torch.gather(dists, 1, order))
    if out_i.shape[1] < k:
        pad = k - out_i.shape[1]
        out_i = torch.cat([out_i, torch.full((out_i.shape[0], pad), -1, dtype=out_i.dtype, device=out_i.device)], 1)
        out_d = torch.cat([out_d, torch.full((out_d.shape[0], pad), float("inf"), dtype=out_d.dtype, device=out_d.device)], 1)
    return out_i, out_d


def local_list_rows(part_ids, world, rank):
    """indices (ascending) of the rows whose IVF list is owned by `rank` (list p -> rank p % world)."""
    p = torch.as_tensor(part_ids).to(torch.int64)
    return torch.nonzero((p >= 0) & (p < (1 << 31)) & (p % world == rank)).reshape(-1)


def create_list_shard(engine, metric, centroids, codebook, part_ids, codes, raw=None, group=None):
    """Device index over THIS rank's lists only.  Local row ids are the ranks of the global ones (0..n_local-1,
    ascending), so every (dist, rowid) comparison made on the device orders rows exactly as the global ids would;
    `l2g` maps them back.  raw (optional, [n][d] by global row id) is sliced to the local rows for refine."""
    from .engine import DeviceIndex
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    part = torch.as_tensor(part_ids)
    rows = local_list_rows(part, world, rank)       # part ids are int32 on the device, NONE = -1
    pl = part[rows.to(part.device)]
    cl = torch.as_tensor(codes)[rows.to(torch.as_tensor(codes).device)]
    raw_l = None if raw is None else torch.as_tensor(raw)[rows.to(torch.as_tensor(raw).device)].contiguous()
    ix = DeviceIndex.create(engine, metric, centroids, codebook, pl, cl, None, raw=raw_l)
    return ix, rows


def shard_index_contents(c, world, rank):
    """The part of an index file's contents (lance_amd.index_file.IndexFileContents) that rank `rank` of `world` owns
    under the list -> rank (p % world) placement: same centroids / codebook, foreign lists emptied.
    -> (part_offsets u32 [nlist+1], codes u8 (stored layout of the owned lists, concatenated), row_ids u64)"""
    nlist = len(c.part_offsets) - 1
    cb = c.code_bytes
    lens = np.diff(c.part_offsets.astype(np.int64))
    own = (np.arange(nlist) % world) == rank
    offs = np.zeros(nlist + 1, np.uint32)
    np.cumsum(np.where(own, lens, 0), out=offs[1:])
    codes = [c.codes[int(c.part_offsets[p]) * cb:int(c.part_offsets[p + 1]) * cb] for p in range(nlist) if own[p]]
    rids = [c.row_ids[int(c.part_offsets[p]):int(c.part_offsets[p + 1])] for p in range(nlist) if own[p]]
    return (offs, np.concatenate(codes) if codes else np.empty(0, np.uint8),
            np.concatenate(rids) if rids else np.empty(0, np.uint64))


def load_list_shard(engine, index_dir, raw=None, dtype=None, group=None):
    """Opens an IVF_PQ index directory and puts THIS rank's lists into HBM (lance_hip_index_load_lists): every rank maps
    the files, parses the (small) metadata and copies only its own code blocks -- at C5 scale 1/8 of 40 GB per GPU.  Row ids are the stored ones, so `search_list_sharded`
    needs no local->global map (pass the returned empty tensor as l2g).  raw: the column's vectors for refine, indexed
    by row id (only meaningful when the ids are row offsets)."""
    from .engine import DeviceIndex
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    ix = DeviceIndex.load(engine, index_dir, dtype=dtype, raw=raw, lists=(world, rank))   # lance_hip_index_load_lists
    return ix, torch.empty(0, dtype=torch.int64)


def search_list_sharded(local_search, l2g, q, k, nprobes, refine_factor=0, group=None, engine=None, local_candidates=None):
    """local_search(q, kk, nprobes, refine_factor) -> (local ids int64 [-1 = none], dists) over this rank's lists.
    Every rank passes the SAME query batch and gets the same (ids [nq,k] int64 global, dists [nq,k]).
    engine: merge the gathered candidates with the device kernel (lance_hip_merge_topk) instead of torch sorts.
    local_candidates(q, keff, nprobes) -> (ids, PQ dists, exact dists) from ONE scan (DeviceIndex.search_candidates =
    lance_hip_ivfpq_search_candidates): with a refine factor the local half is then a single C-ABI call -- no second scan with
    refine_factor = 1, no sorts to align the two candidate lists (the strong-scaling path paid 2 x scan + torch glue)."""
    world = dist.get_world_size(group)
    keff = k * refine_factor if refine_factor else k
    ex = None
    if refine_factor and local_candidates is not None:
        li, ld, ex = local_candidates(q, keff, nprobes)
    else:
        li, ld = local_search(q, keff, nprobes, 0)
    li = torch.as_tensor(li).to(torch.int64); ld = torch.as_tensor(ld).to(torch.float32)
    l2g = torch.as_tensor(l2g).to(li.device)
    gi = torch.where(li < 0, li, l2g[li.clamp(min=0)]) if l2g.numel() else li
    payload = [gi.contiguous(), ld.contiguous()]
    if ex is not None:
        payload.append(torch.as_tensor(ex).to(torch.float32).contiguous())
    elif refine_factor:
        # exact distances of the same keff local candidates (k = keff, refine_factor = 1 re-ranks without dropping any)
        ri, rd = local_search(q, keff, nprobes, 1)
        ri = torch.as_tensor(ri).to(torch.int64); rd = torch.as_tensor(rd).to(torch.float32)
        # align the exact distances with the PQ-ordered candidate list (same id set, different order)
        so = torch.sort(torch.where(ri < 0, torch.full_like(ri, torch.iinfo(torch.int64).max), ri), dim=1)
        qo = torch.sort(torch.where(li < 0, torch.full_like(li, torch.iinfo(torch.int64).max), li), dim=1)
        ex_sorted = torch.gather(rd, 1, so.indices)
        ex = torch.empty_like(ld)
        ex.scatter_(1, qo.indices, ex_sorted)
        payload.append(ex.contiguous())
    gathered = []
    for t in payload:
        buf = [torch.empty_like(t) for _ in range(world)]
        _all_gather(buf, t, group=group)
        gathered.append(torch.cat(buf, dim=1))
    if engine is not None and gathered[0].shape[1] <= 4096:
        return engine.merge_topk(gathered[0], gathered[1], k, exact=gathered[2] if refine_factor else None, keff=keff)
    if not refine_factor:
        return merge_topk(gathered[0], gathered[1], k)
    # global top-keff by (PQ distance, row id), then order those by (exact distance, row id) and fetch k
    ci, cd = gathered[0], gathered[1]
    none = ci < 0
    big = torch.iinfo(torch.int64).max
    o1 = torch.sort(torch.where(none, torch.full_like(ci, big), ci), dim=1, stable=True).indices
    dk1 = torch.gather(torch.where(none, torch.full_like(ci, 1 << 40), _order_key(cd)), 1, o1)
    order = torch.gather(o1, 1, torch.sort(dk1, dim=1, stable=True).indices)[:, :keff]
    return merge_topk(torch.gather(ci, 1, order), torch.gather(gathered[2], 1, order), k)
