"""FID moment accumulation and the only data collective of the hot path.

Reference: diff-solvers-main/fid.py:23-87.  ``calculate_inception_stats`` shards the image list over ranks exactly like
``sample.py`` does (fid.py:54-56), accumulates ``mu += sum(f)`` and ``sigma += f^T f`` in fp64 over InceptionV3 pool
features (fid.py:69-71), then SUM-all-reduces both (fid.py:74-75: 16 KiB + 32 MiB fp64) and finalises
(fid.py:76-78).  The InceptionV3 weights live in a pickle the reference downloads (fid.py:34); there is no network here, so the
detector is INJECTED: any callable ``uint8 images [b, 3, H, W] -> [b, 2048]`` -- ``load_detector`` accepts the reference's
pickle (called with ``return_features=True`` like fid.py:35,68), a TorchScript file or a ``module:function`` factory and runs
it as stock PyTorch-ROCm.  Everything around it is implemented: image-folder listing with the reference's subset rule, rank
sharding, the fp64 moment accumulation, the two SUM all-reduces over RCCL, finalisation, the Frechet distance, and the
``calc`` / ``ref`` command line (fid.py:92-166).

The moment update of a batch on the GPU is ONE launch of ``ds_fid_moments`` (csrc/fid.hip: fp64 MFMA ``f^T f`` + column sums,
accumulated in place; no fp64 copy of the features, no rocBLAS); there is no fallback when libdsamd.so is missing.  Accumulators
on the CPU (the gloo tests of the sharding / all-reduce logic, ``bench.py --stub``) use the two torch expressions of fid.py:69-71.
The all-reduce goes through ``torch.distributed`` -- backend ``nccl`` is RCCL over xGMI on ROCm, ``gloo`` in the CPU tests.
The detector itself is stock PyTorch: its architecture is not in the reference repository (fid.py:34 downloads a pickle).
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
        """fid.py:69-71 for one batch of features [b, feature_dim]."""
        if self.mu.is_cuda:
            import ctypes as C
            from . import _lib
            f = features.to(self.mu.device)
            if f.dtype not in (torch.float32, torch.float64):
                f = f.to(torch.float32)                      # fp16 / bf16 detector outputs widen exactly
            f = f.contiguous()
            assert f.dim() == 2 and f.shape[1] == self.mu.numel(), (tuple(f.shape), self.mu.numel())
            _lib.check(_lib.load().ds_fid_moments(C.c_void_p(f.data_ptr()), int(f.dtype == torch.float64), max(f.stride(0), f.shape[1]), f.shape[0],
                                                  f.shape[1], C.c_void_p(self.mu.data_ptr()), C.c_void_p(self.sigma.data_ptr()),
                                                  _lib.stream_ptr()), 'ds_fid_moments')
        else:                                                # host-side accumulators (gloo tests, bench.py --stub): no kernels
            f = features.to(torch.float64)
            self.mu += f.sum(0)
            self.sigma += f.T @ f
        self.count += features.shape[0]

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


def calculate_inception_stats(feature_fn, images, max_batch_size=64, device='cuda', feature_dim=None):
    """Sharded moment computation.  ``images``: indexable collection; ``feature_fn(batch) -> [b, feature_dim]``
    (feature_dim None: taken from the first batch; 2048 for the Inception pool features of fid.py:33)."""
    import torch.distributed as dist
    rank, world = (dist.get_rank(), dist.get_world_size()) if (dist.is_available() and dist.is_initialized()) else (0, 1)
    acc = MomentAccumulator(feature_dim, device) if feature_dim else None
    for idx in shard_items(len(images), max_batch_size, rank, world):
        if world > 1:
            dist.barrier()                                   # fid.py:64
        if len(idx) == 0:
            continue
        f = feature_fn(images[idx])
        if acc is None:
            acc = MomentAccumulator(f.shape[1], device)
        acc.update(f)
    if acc is None:                                          # a rank without images still takes part in the reduction
        acc = MomentAccumulator(feature_dim or 2048, device)
    acc.all_reduce()
    return acc.finalize(len(images))


class ImageFolder:
    """Recursive image listing with the reference's subset rule (dataset.py:44-48 via fid.py:41: ``max_size=num_expected,
    random_seed=seed`` -> shuffle all indices with ``RandomState(seed % 2**31)``, keep the first ``max_size``, sort)."""

    EXT = ('.png', '.jpg', '.jpeg', '.bmp')

    def __init__(self, path, max_size=None, random_seed=0):
        import os
        names = sorted(os.path.relpath(os.path.join(r, f), path) for r, _, fs in os.walk(path) for f in fs if f.lower().endswith(self.EXT))
        self.path, self.names = path, names
        idx = np.arange(len(names), dtype=np.int64)
        if max_size is not None and idx.size > max_size:
            np.random.RandomState(random_seed % (1 << 31)).shuffle(idx)
            idx = np.sort(idx[:max_size])
        self.idx = idx

    def __len__(self):
        return int(self.idx.size)

    def __getitem__(self, items):
        """items: int tensor / list of dataset indices -> uint8 tensor [b, C, H, W] (single-channel images stay 1-channel)."""
        import os
        import PIL.Image
        out = []
        for i in (items.tolist() if torch.is_tensor(items) else list(items)):
            a = np.asarray(PIL.Image.open(os.path.join(self.path, self.names[int(self.idx[i])])))
            a = a[:, :, None] if a.ndim == 2 else a
            out.append(torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))))
        return torch.stack(out)


def load_detector(spec, device='cuda'):
    """-> callable(uint8 images [b, 3, H, W] on ``device``) -> features [b, 2048].
      *.pkl                 the reference's detector pickle (fid.py:34-36), called with return_features=True;
      *.pt / *.ts           TorchScript module;
      package.module:name   factory ``name(device) -> callable`` (tests, other feature extractors)."""
    import importlib
    import pickle
    if ':' in spec and not spec.endswith(('.pkl', '.pt', '.ts')):
        mod, fn = spec.split(':', 1)
        return getattr(importlib.import_module(mod), fn)(device)
    if spec.endswith('.pkl'):
        with open(spec, 'rb') as f:
            net = pickle.load(f).to(device)
        return lambda images: net(images, return_features=True)
    net = torch.jit.load(spec, map_location=device).eval()
    return lambda images: net(images)


def calculate_inception_stats_for_path(image_path, detector, num_expected=None, seed=0, max_batch_size=64, device='cuda', log=print):
    """fid.py:23-79 with the detector injected: list, check the count, shard, accumulate, reduce, finalise."""
    import torch.distributed as dist
    ds = ImageFolder(image_path, max_size=num_expected, random_seed=seed)
    if num_expected is not None and len(ds) < num_expected:
        raise ValueError(f'Found {len(ds)} images, but expected at least {num_expected}')
    if len(ds) < 2:
        raise ValueError(f'Found {len(ds)} images, but need at least 2 to compute statistics')
    log(f'Calculating statistics for {len(ds)} images...')

    def feature_fn(images):
        images = images.to(device)
        if images.shape[1] == 1:
            images = images.repeat([1, 3, 1, 1])                 # fid.py:66-67
        with torch.no_grad():
            return detector(images)

    return calculate_inception_stats(feature_fn, ds, max_batch_size=max_batch_size, device=device)


def calculate_fid_from_inception_stats(mu, sigma, mu_ref, sigma_ref):
    """fid.py:83-87."""
    import scipy.linalg
    m = np.square(mu - mu_ref).sum()
    s, _ = scipy.linalg.sqrtm(np.dot(sigma, sigma_ref), disp=False)
    return float(np.real(m + np.trace(sigma + sigma_ref - s * 2)))


# ------------------------------------------------------------------------------------------------------------------
# Command line (fid.py:92-166): `fid.py calc --images DIR --ref NPZ --detector SPEC`, `fid.py ref --data DIR --dest NPZ --detector SPEC`

def _init_dist():
    import os
    import torch.distributed as dist
    if 'RANK' in os.environ and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend='nccl' if torch.cuda.is_available() else 'gloo', init_method='env://')
    rank = dist.get_rank() if dist.is_initialized() else 0
    return dist, rank


try:
    import click
except ImportError:                        # pragma: no cover
    click = None

if click is not None:
    @click.group()
    def main():
        """Calculate Frechet Inception Distance (FID) -- the reference's fid.py surface with the detector injected."""

    @main.command()
    @click.option('--images', 'image_path', help='Path to the images', metavar='PATH', type=str, required=True)
    @click.option('--ref', 'ref_path', help='Dataset reference statistics', metavar='NPZ', type=str, required=True)
    @click.option('--num', 'num_expected', help='Number of images to use', metavar='INT', type=click.IntRange(min=2), show_default=True)
    @click.option('--seed', help='Random seed for selecting the images', metavar='INT', type=int, default=0, show_default=True)
    @click.option('--batch', help='Maximum batch size', metavar='INT', type=click.IntRange(min=1), default=250, show_default=True)
    @click.option('--detector', help='Feature extractor: reference .pkl, TorchScript .pt/.ts, or module:factory', type=str, required=True)
    @click.option('--device', type=str, default=None)
    def calc(image_path, ref_path, num_expected, seed, batch, detector, device):
        """Calculate FID for a given set of images."""
        import os
        dist, rank = _init_dist()
        device = device or ('cuda:%d' % int(os.environ.get('LOCAL_RANK', 0)) if torch.cuda.is_available() else 'cpu')
        log = print if rank == 0 else (lambda *a, **k: None)
        log(f'Loading dataset reference statistics from "{ref_path}"...')
        ref = dict(np.load(ref_path)) if rank == 0 else None
        mu, sigma = calculate_inception_stats_for_path(image_path, load_detector(detector, device), num_expected=num_expected, seed=seed,
                                                       max_batch_size=batch, device=device, log=log)
        log('Calculating FID...')
        if rank == 0:
            print(f'{calculate_fid_from_inception_stats(mu, sigma, ref["mu"], ref["sigma"]):g}')
        if dist.is_initialized():
            dist.barrier()

    @main.command()
    @click.option('--data', 'dataset_path', help='Path to the dataset', metavar='PATH', type=str, required=True)
    @click.option('--dest', 'dest_path', help='Destination .npz file', metavar='NPZ', type=str, required=True)
    @click.option('--batch', help='Maximum batch size', metavar='INT', type=click.IntRange(min=1), default=500, show_default=True)
    @click.option('--detector', help='Feature extractor: reference .pkl, TorchScript .pt/.ts, or module:factory', type=str, required=True)
    @click.option('--device', type=str, default=None)
    def ref(dataset_path, dest_path, batch, detector, device):
        """Calculate dataset reference statistics needed by 'calc'."""
        import os
        dist, rank = _init_dist()
        device = device or ('cuda:%d' % int(os.environ.get('LOCAL_RANK', 0)) if torch.cuda.is_available() else 'cpu')
        log = print if rank == 0 else (lambda *a, **k: None)
        mu, sigma = calculate_inception_stats_for_path(dataset_path, load_detector(detector, device), max_batch_size=batch, device=device, log=log)
        log(f'Saving dataset reference statistics to "{dest_path}"...')
        if rank == 0:
            if os.path.dirname(dest_path):
                os.makedirs(os.path.dirname(dest_path), exist_ok=True)
            np.savez(dest_path, mu=mu, sigma=sigma)
        if dist.is_initialized():
            dist.barrier()
        log('Done.')

    if __name__ == '__main__':
        main()
