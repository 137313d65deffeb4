"""find_partitions of a query batch against thousands of centroids (C5: 65,536 x 128; C4: 4096 x 128): per-stage times (dist_matrix =
the matrix-core sweep, select_probes = threshold + candidates + exact re-check) with the per-group keys on / off (child process with
LANCE_HIP_COARSE_GROUPS=0), ids compared.  GPU only."""
import os, subprocess, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(nlist, nq, nprobes, kind):
    from lance_amd.engine import Engine
    eng = Engine()
    rng = np.random.default_rng(5)
    d = 128
    if kind == "int8":
        cent = rng.integers(-100, 100, (nlist, d)).astype(np.float32)
        q = rng.integers(-100, 100, (nq, d)).astype(np.float32)
    else:
        cent = rng.standard_normal((nlist, d)).astype(np.float32)
        q = (cent[rng.integers(0, nlist, nq)] + 0.7 * rng.standard_normal((nq, d))).astype(np.float32)
    cd, qd = torch.from_numpy(cent).cuda(), torch.from_numpy(q).cuda()
    for _ in range(2):
        ids, dist = eng.find_partitions(qd, cd, nprobes, "l2")
    eng.timing(True)
    eng.timing_query("dist_matrix"); eng.timing_query("select_probes")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        ids, dist = eng.find_partitions(qd, cd, nprobes, "l2")
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 5
    a, an = eng.timing_query("dist_matrix"); b, bn = eng.timing_query("select_probes")
    eng.timing(False)
    print(f"nlist {nlist} nq {nq} nprobes {nprobes} {kind}: groups={os.environ.get('LANCE_HIP_COARSE_GROUPS', 'default')} sweep {a / max(an, 1):.3f} ms select {b / max(bn, 1):.3f} ms wall {wall * 1e3:.3f} ms "
          f"ids checksum {int(ids.cpu().numpy().astype(np.int64).sum())} dist checksum {float(dist.double().sum()):.6e}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
    else:
        for nlist, nq, nprobes, kind in ((65536, 10000, 10, "int8"), (65536, 10000, 32, "f32"), (4096, 10000, 10, "f32"), (1024, 10000, 10, "f32")):
            for sw in (None, "0"):
                env = dict(os.environ)
                if sw is not None:
                    env["LANCE_HIP_COARSE_GROUPS"] = sw
                subprocess.run([sys.executable, os.path.abspath(__file__), str(nlist), str(nq), str(nprobes), kind], env=env, timeout=600)
