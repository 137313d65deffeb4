"""Shared by tests/golden/make_fullconfig_golden.py (CPU, oracle: writes tests/golden/fullconfig.npz) and
tests/test_zz_gpu_fullconfig.py (GPU: reproduces every recorded digest / result).  The oracle is too slow to run these sizes
inside the GPU suite (minutes of CPU work per case), so its outputs are recorded once, here in the build container, as SHA-256
digests of the large arrays (trained centroids, codebook, partition ids, PQ codes) plus the small search results verbatim.

Cases = BASELINE.json configs 3 and 5 at their real index parameters on a row count the oracle can build in minutes:
  c3: 100,000 x 1536 f32 unit-norm, cosine, IVF_PQ nlist 1024 (hierarchical k-means), M 96 (sub-dimension 16)
  c5: 300,000 x 128 int8, L2, nlist 65,536 (centroids = the reference's random-row k-means initialisation), M 32 (sub-dim 4)
Round 4:
  c4: 1,000,000 x 128 FLOAT16 rows, L2, IVF_PQ nlist 4096 trained by the HIERARCHICAL k-means in its Float16Type instantiation
      (f16 M-step in every inner k-means, kmeans.rs:1030-1033), M 16 -- BASELINE config 4's index parameters on 1/100 of its rows
  c5t: the c5 rows with the coarse quantiser TRAINED to nlist 65,536 by the hierarchical k-means (the reference's own route for
      k > 256; clusters run down to a handful of rows, so the cluster_size <= hierarchical_k and the two-way split arms are taken)
"""
import hashlib

import numpy as np

f32 = np.float32

C3 = dict(n=100_000, d=1536, nlist=1024, m=96, nq=200, metric="cosine", ivf_iters=10, pq_iters=10, seed=31,
          searches=((10, 1, 0), (10, 10, 10), (10, 40, 0), (100, 1024, 0), (10, 1024, 10)))
C5 = dict(n=300_000, d=128, nlist=65536, m=32, nq=1000, metric="l2", pq_iters=10, seed=51,
          searches=((10, 1, 0), (10, 32, 10), (10, 256, 0), (100, 64, 0)))


C4 = dict(n=1_000_000, d=128, nlist=4096, m=16, nq=200, metric="l2", ivf_iters=50, pq_iters=50, seed=41,
          searches=((10, 1, 0), (10, 10, 10), (10, 50, 10), (10, 4096, 0)))
C5T = dict(nlist=65536, ivf_iters=20, seed=57, searches=((10, 32, 10), (10, 256, 0)))


def digest(a):
    a = np.ascontiguousarray(a)
    return np.frombuffer(hashlib.sha256(a.view(np.uint8).reshape(-1).tobytes()).digest(), np.uint8).copy()


def c3_data():
    rng = np.random.default_rng(3003)
    c = C3
    centers = rng.standard_normal((256, c["d"])).astype(f32)
    x = centers[rng.integers(0, 256, c["n"])] + rng.standard_normal((c["n"], c["d"]), dtype=f32) * f32(0.5)
    x = (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(f32)            # ada-002 embeddings are unit norm
    q = centers[rng.integers(0, 256, c["nq"])] + rng.standard_normal((c["nq"], c["d"]), dtype=f32) * f32(0.5)
    return x, q.astype(f32)


def c5_data():
    rng = np.random.default_rng(5005)
    c = C5
    centers = rng.integers(0, 256, (4096, c["d"]))
    x = np.clip(centers[rng.integers(0, 4096, c["n"])] + rng.integers(-12, 13, (c["n"], c["d"])), 0, 255) - 128     # BigANN bytes, stored as i8
    q = np.clip(centers[rng.integers(0, 4096, c["nq"])] + rng.integers(-12, 13, (c["nq"], c["d"])), 0, 255) - 128
    return x.astype(np.int8), q.astype(np.int8)


def c4_data():
    """SIFT-like integer rows scaled by 1/256 (exactly representable in binary16; cluster sums stay far below 65504, so the
    reference's f16 M-step does not overflow -- bench.py --config c4 uses the same scaling)"""
    rng = np.random.default_rng(4004)
    c = C4
    centers = rng.uniform(0, 128, (4096, c["d"]))
    def draw(n):
        out = np.empty((n, c["d"]), np.float16)
        for a in range(0, n, 100_000):
            b = min(n, a + 100_000)
            v = centers[rng.integers(0, 4096, b - a)] + rng.normal(0, 24, (b - a, c["d"]))
            out[a:b] = (np.clip(np.rint(v), 0, 218) / 256.0).astype(np.float16)
        return out
    return draw(c["n"]), draw(c["nq"])
