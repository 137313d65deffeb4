"""Seeded SIFT-like synthetic vectors.

Real SIFT descriptors are integer-valued in [0, 218], 128-d, but of low intrinsic
dimension -- that is what makes PQ-16 reach ~0.6 recall@10 un-refined and ~0.97 with
refine (benchmarks/sift/lance_sift1m_stats.csv).  An isotropic Gaussian mixture in 128-d
(the first recipe tried) has no usable neighbour structure: distances concentrate and even
an exhaustive PQ scan only gets recall@10 ~ 0.12.  This generator therefore draws a
Gaussian mixture in a LATENT space of `latent` dims, maps it to 128-d with a fixed random
linear map, adds a little isotropic noise, then shifts/scales into the SIFT value range and
rounds to integers stored as f32.
"""
import numpy as np


def sift_like(n, d=128, seed=1234, n_clusters=256, latent=16, device=None, spread=1.0, within=1.0, noise=0.12,
              model_seed=1234):
    """-> float32 [n, d] (torch tensor on `device` if given, else numpy).
    `model_seed` fixes the mixture (centres, linear map); `seed` draws the points, so base and
    query sets share one distribution but are disjoint draws."""
    mrng = np.random.default_rng(model_seed)
    centers = mrng.standard_normal((n_clusters, latent)) * spread
    W = mrng.standard_normal((latent, d)) / np.sqrt(latent)
    if device is not None:
        import torch
        g = torch.Generator(device=device)
        g.manual_seed(int(seed))
        c = torch.from_numpy(centers).float().to(device)
        Wt = torch.from_numpy(W).float().to(device)
        a = torch.randint(0, n_clusters, (n,), generator=g, device=device)
        z = c[a] + torch.randn((n, latent), generator=g, device=device) * within
        x = z @ Wt + torch.randn((n, d), generator=g, device=device) * noise
        return torch.clamp(torch.round(64.0 + 28.0 * x), 0, 218).float().contiguous()
    rng = np.random.default_rng(seed)
    a = rng.integers(0, n_clusters, n)
    z = centers[a] + rng.standard_normal((n, latent)) * within
    x = z @ W + rng.standard_normal((n, d)) * noise
    return np.clip(np.rint(64.0 + 28.0 * x), 0, 218).astype(np.float32)
