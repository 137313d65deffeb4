"""find_partitions at query time on the matrix cores (mfma_assign.hip: SUR kernels + coarse_select_kernel; round 4).

`IvfModel::find_partitions` -> `kmeans_find_partitions` (kmeans.rs:1134-1158): distances to every centroid, the nprobes smallest
ascending by (distance, index).  The device path computes a bf16x3 MFMA surrogate matrix, takes every centroid within the
surrogate's error margin of the nprobes-th smallest, recomputes those exactly in the reference's order and sorts them: partition
ids AND distances must equal the oracle's bit for bit.

The path switches itself on by problem size (nq * nlist * d >= 2^27); LANCE_HIP_MFMA_COARSE=1 (read once per process) forces it for
every shape it takes, so the small cases below run in a child process with the switch on (`_cases`), the large ones in-process."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
f32 = np.float32
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _eq(eng, oracle, q, cent, nprobes, metric, tag):
    ids, d = eng.find_partitions(q, cent, nprobes, metric)
    oi, od = oracle.find_partitions(q, cent, nprobes, metric)
    ids = ids.cpu().numpy().view(np.uint32); d = d.cpu().numpy()
    assert np.array_equal(ids, oi), (tag, np.argwhere(ids != oi)[:5])
    assert np.array_equal(d.view(np.uint32), od.view(np.uint32)), tag


def _cases():
    """Runs inside the child process (LANCE_HIP_MFMA_COARSE=1)."""
    sys.path.insert(0, ROOT)
    import oracle
    from lance_amd.engine import Engine
    eng = Engine()
    rng = np.random.default_rng(2024)
    n_cases = 0
    # (from 1024 lists on -- or from LANCE_HIP_COARSE_GROUPS lists -- rows of <= 128 elements take the per-group keys instead of the matrix)
    for d, nlist, nq in ((16, 32, 300), (64, 100, 257), (128, 256, 1000), (96, 300, 129), (128, 5000, 200), (48, 70, 64),
                         (1536, 1024, 130), (200, 64, 260), (132, 333, 70), (4096, 65, 5), (32, 1024, 300), (16, 65536, 40), (64, 2049, 150)):
        for metric in ("l2", "dot", "cosine"):
            cent = (rng.standard_normal((nlist, d)) * 3).astype(f32)
            q = (cent[rng.integers(0, nlist, nq)] + rng.standard_normal((nq, d)).astype(f32)).astype(f32)
            if metric == "cosine":     # the index path normalises rows and queries, then L2
                cent = oracle.normalize(cent); q = oracle.normalize(q)
            for nprobes in (1, 10, 50, 64):
                if nprobes > nlist:
                    continue
                _eq(eng, oracle, q, cent, nprobes, "l2" if metric == "cosine" else metric, (d, nlist, nq, metric, nprobes))
                n_cases += 1
    # ties: duplicate centroids (first index wins), a block of identical centroids larger than the candidate list (exact path),
    # NaN / inf queries, a NaN centroid, integer-valued SIFT-like rows (many equal distances)
    d, nlist = 64, 512
    cent = np.rint(rng.uniform(0, 40, (nlist, d))).astype(f32)
    cent[7] = cent[300]
    cent[100:260] = cent[100]                      # 160 identical centroids > 128 candidates
    q = np.rint(rng.uniform(0, 40, (400, d))).astype(f32)
    q[3] = cent[100]                               # distance 0 to the whole block
    q[5, 2] = np.nan
    q[6] = np.inf
    q[8] = 1e30                                    # squares overflow: inf distances
    # dot: inf x 0 is a NaN whose SIGN is the platform's (x86: the negative "real indefinite", which total_cmp sorts first; gfx950:
    # the positive default NaN, sorted last), and 1 - NaN keeps the NaN's sign on x86 while the GPU's subtract-as-negated-add may
    # flip it -- the reference itself answers differently on x86 and ARM there, so the NaN / infinite queries stay L2 cases
    qd = q.copy(); qd[6] = q[7]; qd[5] = q[4]
    for nprobes in (1, 10, 50, 64):
        _eq(eng, oracle, q, cent, nprobes, "l2", ("ties", nprobes)); n_cases += 1
        _eq(eng, oracle, qd, cent, nprobes, "dot", ("ties-dot", nprobes)); n_cases += 1
    # a non-finite centroid: every surrogate row carries a NaN (inf - inf in the bf16 split) -> every query takes the exact path.
    # (+inf, not NaN: x - NaN keeps the NaN's sign on x86 while the GPU's subtract may flip it, and total_cmp sorts -NaN first)
    cent2 = cent.copy(); cent2[11, 0] = np.inf
    _eq(eng, oracle, q, cent2, 10, "l2", "inf-centroid"); n_cases += 1
    # the same kinds of ties over 2048 lists (per-group keys): two and five of a kind inside one group of 16 and across groups, a block
    # of identical centroids larger than the candidate list, a NaN centroid
    nlist = 2048
    cent = np.rint(rng.uniform(0, 40, (nlist, d))).astype(f32)
    cent[7] = cent[300]; cent[65] = cent[64]; cent[129] = cent[133]; cent[1000:1005] = cent[1000]
    cent[1200:1400] = cent[1200]
    q = np.rint(rng.uniform(0, 40, (300, d))).astype(f32)
    q[3] = cent[1200]; q[4] = cent[64]; q[9] = cent[1000]; q[10] = cent[129]
    q[5, 2] = np.nan; q[6] = np.inf; q[8] = 1e30
    qd = q.copy(); qd[6] = q[7]; qd[5] = q[4]
    for nprobes in (1, 10, 50, 64):
        _eq(eng, oracle, q, cent, nprobes, "l2", ("ties-2048", nprobes)); n_cases += 1
        _eq(eng, oracle, qd, cent, nprobes, "dot", ("ties-2048-dot", nprobes)); n_cases += 1
    cent2 = cent.copy(); cent2[11, 0] = np.inf
    _eq(eng, oracle, q[:40], cent2, 10, "l2", "inf-centroid-2048"); n_cases += 1
    groups = eng.timing_query("count:coarse_groups")[1]
    eng.close()
    print(f"coarse mfma cases ok: {n_cases} (per-group keys served {groups} calls)")


def test_find_partitions_per_group_keys_from_256_lists():
    """every case again with the per-group keys taken from 256 lists on (the default is 1024): the tie / NaN / overflow cases of the
    512-list set then run through it too"""
    env = dict(os.environ, LANCE_HIP_MFMA_COARSE="1", LANCE_HIP_COARSE_GROUPS="256")
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import tests.test_zz_gpu_coarse_mfma as t; t._cases()" % ROOT],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    import re
    m = re.search(r"per-group keys served (\d+) calls", r.stdout)
    assert m and int(m.group(1)) >= 60, r.stdout[-500:]


def test_find_partitions_on_matrix_cores_forced_for_small_shapes():
    env = dict(os.environ, LANCE_HIP_MFMA_COARSE="1")
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import tests.test_zz_gpu_coarse_mfma as t; t._cases()" % ROOT],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "coarse mfma cases ok" in r.stdout


def test_find_partitions_large_batches_take_the_matrix_core_path(oracle):
    """C2 / C3 / C5-shaped coarse quantisers at sizes where the path is on by default."""
    from lance_amd.engine import Engine
    eng = Engine()
    rng = np.random.default_rng(77)
    for d, nlist, nq, nprobes in ((128, 256, 5000, 10), (1536, 4096, 2100, 10), (128, 65536, 64, 32), (128, 4096, 512, 50)):
        cent = np.rint(rng.uniform(0, 128, (nlist, d))).astype(f32)
        q = np.clip(cent[rng.integers(0, nlist, nq)] + np.rint(rng.normal(0, 20, (nq, d))), 0, 218).astype(f32)
        assert nq * nlist * d >= 1 << 27 and (d <= 128 or nq * nlist >= 1 << 23)
        _eq(eng, oracle, q, cent, nprobes, "l2", (d, nlist, nq, nprobes))
    eng.close()
