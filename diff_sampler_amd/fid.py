"""FID moment accumulation and the only data collective of the hot path.

Reference: diff-solvers-main/fid.py:23-87.  ``calculate_inception_stats`` shards the image list over ranks exactly like
``sample.py`` does (fid.py:54-56), accumulates ``mu += sum(f)`` and ``sigma += f^T f`` in fp64 over InceptionV3 pool
features (fid.py:69-71), then SUM-all-reduces both (fid.py:74-75: 16 KiB + 32 MiB fp64) and finalises
(fid.py:76-78).  The InceptionV3 forward itself is a SURVEY section 8(f) "next" row and is supplied by the caller here
(any callable ``images -> [B, 2048]``); everything after it is implemented.

The fp64 second-moment update is a plain library GEMM (rocBLAS through ``torch.matmul``); the all-reduce goes through
``torch.distributed`` -- backend ``nccl`` is RCCL over xGMI on ROCm, ``gloo`` in the CPU tests.
"""
from __future__ import annotations

import numpy as np
import torch


def shard_items(num_items: int, max_batch_size: int, rank: int, world: int):
    """fid.py:54-56 (same rule as sample.py:167-169)."""
    num_batches = ((num_items - 1) // (max_batch_size * world) + 1) * world
    all_batches = torch.arange(num_items).tensor_split(num_batches)
    return all_batches[rank::world]


class MomentAccumulator:
    def __init__(self, feature_dim=2048, device='cuda'):
        self.mu = torch.zeros([feature_dim], dtype=torch.float64, device=device)
        self.sigma = torch.zeros([feature_dim, feature_dim], dtype=torch.float64, device=device)
        self.count = 0

    def update(self, features: torch.Tensor):
        f = features.to(torch.float64)
        self.mu += f.sum(0)
        self.sigma += f.T @ f
        self.count += f.shape[0]

    def all_reduce(self):
        """SUM over ranks (fid.py:74-75).  Returns the wall time of the two collectives in seconds (informational)."""
        import time
        import torch.distributed as dist
        self.last_timing = dict(mu_s=0.0, sigma_s=0.0)
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return 0.0
        sync = torch.cuda.synchronize if self.mu.is_cuda else (lambda: None)
        sync()
        t0 = time.perf_counter()
        dist.all_reduce(self.mu)
        sync()
        t1 = time.perf_counter()
        dist.all_reduce(self.sigma)
        sync()
        t2 = time.perf_counter()
        self.last_timing = dict(mu_s=t1 - t0, sigma_s=t2 - t1)
        return t2 - t0

    def finalize(self, num_total: int):
        """fid.py:76-78 -> (mu, sigma) as numpy fp64."""
        mu = self.mu / num_total
        sigma = self.sigma - mu.ger(mu) * num_total
        sigma = sigma / (num_total - 1)
        return mu.cpu().numpy(), sigma.cpu().numpy()


def calculate_inception_stats(feature_fn, images, max_batch_size=64, device='cuda', feature_dim=2048):
    """Sharded moment computation.  ``images``: indexable collection; ``feature_fn(batch) -> [b, feature_dim]``."""
    import torch.distributed as dist
    rank, world = (dist.get_rank(), dist.get_world_size()) if (dist.is_available() and dist.is_initialized()) else (0, 1)
    acc = MomentAccumulator(feature_dim, device)
    for idx in shard_items(len(images), max_batch_size, rank, world):
        if world > 1:
            dist.barrier()                                   # fid.py:64
        if len(idx) == 0:
            continue
        acc.update(feature_fn(images[idx]))
    acc.all_reduce()
    return acc.finalize(len(images))


def calculate_fid_from_inception_stats(mu, sigma, mu_ref, sigma_ref):
    """fid.py:83-87."""
    import scipy.linalg
    m = np.square(mu - mu_ref).sum()
    s, _ = scipy.linalg.sqrtm(np.dot(sigma, sigma_ref), disp=False)
    return float(np.real(m + np.trace(sigma + sigma_ref - s * 2)))
