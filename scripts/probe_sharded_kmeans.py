"""Times the device-resident sharded Lloyd loop (world_size 1, RCCL initialised) step by step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29811")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import lance_amd
from lance_amd import dist as ld
from lance_amd.testing import sift_like
eng = lance_amd.default_engine()
x = sift_like(65536, 128, seed=1, device="cuda")
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    c, loss, it = ld.train_kmeans_sharded(eng, x, 256, 65536, 50, 1e-4, 1.0, None, 42, "l2", None)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    c2, loss2, it2 = eng.kmeans_train(x, 256, max_iters=50, balance_factor=1.0, seed=42)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"sharded(device loop) {1e3*(t1-t0):.2f} ms, {it} iterations, loss {loss:.6g} | single {1e3*(t2-t1):.2f} ms, {it2} iterations, loss {loss2:.6g}", flush=True)
twin = eng._torch_stream_twin
st = twin.kmeans_shard_begin(256, 128, 1.0 / 65536, 42)
cent = x[:256].clone()
torch.cuda.synchronize()
for name, fn in (("estep", lambda: twin.kmeans_shard_estep(st, x, cent, "l2")), ("update", lambda: twin.kmeans_shard_update(st, cent, 65536, 1e-4, 1)),
                 ("end", lambda: twin.kmeans_shard_end(st))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        fn()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize(); t1 = time.perf_counter() - t0
    print(f"{name}: enqueue {1e6*t_enq/20:.1f} us/call, with drain {1e6*t1/20:.1f} us/call", flush=True)
dist.destroy_process_group()
