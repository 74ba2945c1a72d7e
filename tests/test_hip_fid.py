"""FID moment path on the GPU (reference fid.py:54-87): fp64 first / second moments of [N, 2048] pool features
accumulated on the device by `fid.MomentAccumulator` (the second-moment update is the library fp64 GEMM), finalised as
fid.py:76-78, compared with numpy fp64; Frechet distance of two such statistics against the closed form.  Bit-level equality is
not the bar for an fp64 GEMM with a different summation order: 1e-10 relative (fp64 has 2^-53)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sampler_amd import fid as F  # noqa: E402

pytestmark = pytest.mark.gpu


def _features(n, d, seed, shift=0.0):
    g = np.random.RandomState(seed)
    a = g.randn(d, d) / np.sqrt(d)
    return (g.randn(n, d) @ a + shift + 0.3 * g.randn(d)).astype(np.float32)


@pytest.mark.parametrize('n,d,batch', [(1000, 2048, 250), (333, 2048, 64), (50, 64, 7)])
def test_moments_on_device_match_numpy_fp64(n, d, batch):
    f = _features(n, d, seed=n)
    acc = F.MomentAccumulator(d, 'cuda')
    for idx in F.shard_items(n, batch, 0, 1):               # the reference's batch rule (fid.py:54-56), one rank
        acc.update(torch.from_numpy(f[idx.numpy()]).cuda())
    assert acc.count == n
    assert acc.all_reduce() == 0.0                          # no process group: nothing to reduce
    mu, sigma = acc.finalize(n)
    f64 = f.astype(np.float64)
    mu_ref = f64.mean(0)
    sigma_ref = (f64.T @ f64 - np.outer(mu_ref, mu_ref) * n) / (n - 1)                      # fid.py:76-78, literally
    assert mu.dtype == np.float64 and sigma.dtype == np.float64
    assert np.abs(mu - mu_ref).max() <= 1e-10 * np.abs(mu_ref).max()
    assert np.abs(sigma - sigma_ref).max() <= 1e-10 * np.abs(sigma_ref).max()
    assert np.allclose(sigma_ref, np.cov(f64, rowvar=False), rtol=0, atol=1e-9 * np.abs(sigma_ref).max())


def test_calculate_inception_stats_with_a_device_feature_fn_and_frechet_distance():
    """`calculate_inception_stats` end to end with the detector replaced by a fixed random projection evaluated on the GPU; FID of
    two Gaussians with equal covariance is the squared distance of the means (fid.py:83-87)."""
    d, n = 96, 400
    g = torch.Generator().manual_seed(5)
    proj = torch.randn(3 * 8 * 8, d, generator=g).cuda() / 14.0
    images = torch.randint(0, 256, (n, 3, 8, 8), generator=g, dtype=torch.uint8)

    def feature_fn(batch):
        return batch.cuda().to(torch.float32).reshape(batch.shape[0], -1) @ proj

    mu, sigma = F.calculate_inception_stats(feature_fn, images, max_batch_size=64, device='cuda')
    f64 = (images.to(torch.float32).reshape(n, -1) @ proj.cpu()).double().numpy()
    assert np.abs(mu - f64.mean(0)).max() <= 1e-5 * np.abs(f64.mean(0)).max()            # fp32 features, fp32 projection on the device
    assert np.abs(sigma - np.cov(f64, rowvar=False)).max() <= 1e-4 * np.abs(sigma).max()
    assert abs(F.calculate_fid_from_inception_stats(mu, sigma, mu, sigma)) < 1e-6 * np.trace(sigma)
    shift = np.full(d, 0.25)
    want = float((shift ** 2).sum())
    got = F.calculate_fid_from_inception_stats(mu + shift, sigma, mu, sigma)
    assert abs(got - want) < 1e-6 * max(1.0, np.trace(sigma))
