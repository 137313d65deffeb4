"""Python twin of lance_amd/csrc/rng.h (xoshiro256++ / splitmix64 and the k-means
reservoir), used by the multi-GPU host loop so all ranks draw identical streams."""
import numpy as np

M64 = (1 << 64) - 1


def _rotl(x, k):
    return ((x << k) | (x >> (64 - k))) & M64


class Rng:
    def __init__(self, seed):
        self.s = []
        x = seed & M64
        for _ in range(4):
            x = (x + 0x9E3779B97F4A7C15) & M64
            z = x
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
            self.s.append(z ^ (z >> 31))

    def next(self):
        s = self.s
        result = (_rotl((s[0] + s[3]) & M64, 23) + s[0]) & M64
        t = (s[1] << 17) & M64
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]
        s[2] ^= t
        s[3] = _rotl(s[3], 45)
        return result

    def next_f32(self):
        return np.float32(self.next() >> 40) * np.float32(1.0 / 16777216.0)

    def upto(self, n):
        rng_ = n + 1
        mask = rng_ - 1
        for sh in (1, 2, 4, 8, 16, 32):
            mask |= mask >> sh
        while True:
            v = self.next() & mask
            if v < rng_:
                return v


def kmeans_init_indices(n, k, seed):
    """the k initial rows; the native helper when the library is built (identical stream), else the Python twin"""
    try:
        import ctypes as C
        from . import _lib
        out = np.empty(k, np.uint64)
        _lib.check(_lib.load().lance_hip_kmeans_init_indices(int(n), int(k), int(seed) & M64, out.ctypes.data_as(C.c_void_p)))
        return out
    except (ImportError, OSError, AttributeError):
        pass
    r = Rng(seed)
    out = np.arange(k, dtype=np.uint64)
    for i in range(k, n):
        j = r.upto(i)
        if j < k:
            out[j] = i
    return out
