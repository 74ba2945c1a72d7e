"""FID moment path on the GPU (reference fid.py:54-87): fp64 first / second moments of [N, 2048] pool features
accumulated on the device by `fid.MomentAccumulator` through ds_fid_moments (csrc/fid.hip: v_mfma_f64_16x16x4_f64, in-place
accumulation), compared with numpy fp64 on the raw sums (1e-12 relative: an fp64 contraction with a different summation order; fp64
has 2^-53) and after the finalisation of fid.py:76-78 (1e-10: the covariance subtracts two nearly equal numbers); Frechet distance of
two such statistics against the closed form."""
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


@pytest.mark.parametrize('rows,dim,dtype', [(64, 2048, torch.float32), (250, 2048, torch.float32), (7, 64, torch.float32), (33, 96, torch.float64),
                                            (1, 2048, torch.float32), (5, 50, torch.float32), (130, 200, torch.float64)])
def test_ds_fid_moments_raw_sums_match_numpy_fp64(rows, dim, dtype):
    """The kernel itself through the C ABI: mu += f.sum(0), sigma += f^T f IN PLACE on non-zero accumulators, feature rows with a
    leading dimension larger than dim, ragged dims (not multiples of 16 / 64) and row counts (not multiples of 4); 1e-12 of the
    result's scale; the update must be exactly symmetric and must not touch memory beyond [dim] / [dim][dim]."""
    import ctypes as C
    from diff_sampler_amd import _lib
    lib = _lib.load()
    g = np.random.RandomState(rows * 1000 + dim)
    ld = dim + 8
    f = (g.randn(rows, ld) * 2.0 + 0.5).astype(np.float32 if dtype == torch.float32 else np.float64)
    mu0, s0 = g.randn(dim), g.randn(dim, dim)
    s0 = s0 + s0.T
    fd = torch.from_numpy(f).cuda()
    mu = torch.from_numpy(np.concatenate([mu0, [777.0]])).cuda()                 # one guard element behind each accumulator
    sg = torch.from_numpy(np.concatenate([s0.reshape(-1), [777.0]])).cuda()
    rc = lib.ds_fid_moments(C.c_void_p(fd.data_ptr()), int(dtype == torch.float64), ld, rows, dim, C.c_void_p(mu.data_ptr()),
                            C.c_void_p(sg.data_ptr()), _lib.stream_ptr())
    torch.cuda.synchronize()
    assert rc == 0, lib.ds_error_string(rc)
    f64 = f[:, :dim].astype(np.float64)
    mu_ref, s_ref = mu0 + f64.sum(0), s0 + f64.T @ f64
    got_mu, got_s = mu.cpu().numpy(), sg.cpu().numpy()
    assert got_mu[-1] == 777.0 and got_s[-1] == 777.0
    got_s = got_s[:-1].reshape(dim, dim)
    assert np.abs(got_mu[:-1] - mu_ref).max() <= 1e-12 * np.abs(mu_ref).max()
    assert np.abs(got_s - s_ref).max() <= 1e-12 * np.abs(s_ref).max()
    assert np.array_equal(got_s, got_s.T)


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
