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
            dist.broadcast(cent, src=0, group=group)
    else:
        cent = torch.as_tensor(init, dtype=torch.float32).to(dev).clone()
    split_rng = Rng(seed ^ 0x5BD1E995)
    bf_param = f32(balance_factor) / f32(n_total)          # train_kmeans :1344
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
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
            dist.all_reduce(losses, op=dist.ReduceOp.SUM, group=group)
            dist.all_reduce(radius, op=dist.ReduceOp.MAX, group=group)
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


def create_index_sharded(x, metric="l2", num_partitions=256, num_sub_vectors=16, num_bits=8, max_iters=50, sample_rate=256,
                         seed=42, engine=None, group=None, keep_raw=True):
    """create_index over N ranks.  `x` is the full matrix on every rank in this version (the
    bench generates it from a shared seed); each rank TRAINS on and ENCODES only its row shard."""
    import time

    from . import vector as lv
    from .engine import DeviceIndex, to_device
    eng = engine or lv.default_engine()
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    params = lv.IvfPqParams(num_partitions, num_sub_vectors, num_bits, lv._normalize_metric_type(metric), max_iters, sample_rate, seed)
    x = to_device(x)
    n, d = x.shape
    stats = lv.BuildStats()
    kmetric = "l2" if params.metric == "cosine" else params.metric

    def timed(name, fn):
        torch.cuda.synchronize()
        t = time.perf_counter()
        out = fn()
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
        per = (num_sub_vectors + world - 1) // world
        m0, m1 = min(rank * per, num_sub_vectors), min((rank + 1) * per, num_sub_vectors)
        cb_all = torch.zeros((per * world, kc, sd), dtype=torch.float32, device=x.device)
        its_all = torch.zeros(per * world, dtype=torch.int32, device=x.device)
        mine = torch.zeros((per, kc, sd), dtype=torch.float32, device=x.device)
        its = torch.zeros(per, dtype=torch.int32, device=x.device)
        if m1 > m0:
            cols = psample[:, m0 * sd: m1 * sd].contiguous()
            c, it = eng.pq_train(cols, m1 - m0, num_bits, max_iters, sample_rate, seed + 2 + m0)
            mine[: m1 - m0] = c
            its[: m1 - m0] = torch.from_numpy(it.astype(np.int32)).to(x.device)
        dist.all_gather_into_tensor(cb_all, mine, group=group)
        dist.all_gather_into_tensor(its_all, its, group=group)
        # blocks are laid out rank-major with `per` slots each; compact to the first M
        keep = torch.cat([torch.arange(r * per, r * per + max(0, min((r + 1) * per, num_sub_vectors) - min(r * per, num_sub_vectors)),
                                       device=x.device) for r in range(world)])
        return cb_all[keep].contiguous(), its_all[keep].cpu().numpy().astype(np.uint32)

    cb, stats.pq_iters = timed("train_pq", train_pq)

    # transform: each rank encodes its row shard, then all-gather the shuffle-buffer columns
    def transform():
        per = (n + world - 1) // world
        lo, hi = rank * per, min(n, (rank + 1) * per)
        part_l, codes_l, _ = eng.ivfpq_encode(x[lo:hi], cent, cb, params.metric)
        part = torch.empty(per * world, dtype=torch.int32, device=x.device)
        codes = torch.empty((per * world, num_sub_vectors), dtype=torch.uint8, device=x.device)
        pl = torch.full((per,), -1, dtype=torch.int32, device=x.device); pl[: hi - lo] = part_l
        cl = torch.zeros((per, num_sub_vectors), dtype=torch.uint8, device=x.device); cl[: hi - lo] = codes_l
        dist.all_gather_into_tensor(part, pl, group=group)
        dist.all_gather_into_tensor(codes, cl, group=group)
        return part[:n].contiguous(), codes[:n].contiguous()

    part, codes = timed("transform", transform)
    ix = timed("build_partitions", lambda: DeviceIndex.create(eng, params.metric, cent, cb, part, codes, None,
                                                              raw=x if keep_raw else None))
    return lv.IvfPqIndex(ix, params, stats, part, codes)
