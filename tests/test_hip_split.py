"""GPU: the fp32-EMULATED mode (split fp16 hi/lo operands, three MFMA products per multiplication in the 3x3 convolutions) against
the SAME real-reference goldens and the SAME fp32 tolerances as the exact fp32 path (2e-4 per evaluation, 5e-4 per trajectory).  The
mode is an experiment the round-1 verdict asked for; it is reported next to the fp32 headline only because it passes these."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu
G = os.path.join(ROOT, 'tests', 'golden')


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-6))


@pytest.mark.parametrize('name', ['cifar10', 'ffhq', 'imagenet64'])
def test_split_denoiser_matches_reference_goldens_within_fp32_tolerance(name):
    import ctypes as C
    from diff_sampler_amd import _lib
    from diff_sampler_amd.engine import EDMDenoiser
    dev = torch.device('cuda')
    z = np.load(os.path.join(G, f'net_{name}.npz'))
    net = EDMDenoiser.from_config(name, seed=int(z['seed']), split_fp16=True)
    lab = torch.from_numpy(z['labels']).to(dev) if z['labels'].size else None
    out = net(torch.from_numpy(z['x']).to(dev), torch.from_numpy(z['sigma']).to(dev), class_labels=lab)
    torch.cuda.synchronize()
    err = _rel(out.cpu(), torch.from_numpy(z['out_vec']))
    assert err < 2e-4, err
    lib = _lib.load()
    plan = next(iter(net.engine._plans.values()))
    modes = [op.keep[0].wgt_f16 for op in plan.ops if op.fn is lib.ds_conv2d_nhwc and op.keep[0].taps == 9]
    assert modes.count(2) >= 0.5 * len(modes), modes          # the mode is on (at B = 1..2 the 8x8 layers cannot form tiles of four images and stay exact fp32)


def test_split_headline_sampler_matches_reference_golden():
    """The benchmarked sampler (DPM-Solver++(2M), logSNR, NFE=10, CIFAR-10 net, B=64) in the emulated mode vs the real reference."""
    from diff_sampler_amd import solvers
    from diff_sampler_amd.engine import EDMDenoiser
    dev = torch.device('cuda')
    z = np.load(os.path.join(G, 'sampler_cifar10_dpmpp2m_nfe10_b64.npz'))
    net = EDMDenoiser.from_config('cifar10', seed=int(z['seed']), split_fp16=True)
    latents = torch.randn(64, 3, 32, 32, generator=torch.Generator().manual_seed(int(z['latent_seed']))).to(dev)
    out = solvers.dpm_pp_sampler(net, latents, num_steps=11, sigma_min=0.002, sigma_max=80., schedule_type='logsnr', schedule_rho=7,
                                 max_order=2, predict_x0=True, lower_order_final=True)
    torch.cuda.synchronize()
    assert _rel(out.cpu(), torch.from_numpy(z['out'])) < 5e-4


def test_split_headline_sampler_at_the_benchmark_batch_b256_matches_reference_golden():
    """`bench.py --dtype fp16x3` times the headline sampler at 256 images (the driver's `other_configs` line): four rounds of 256-pixel tiles
    per 32x32 layer -- tilings the 64-image call above does not reach.  The 64 latents of the real reference's golden run are scattered over
    the 256-image batch (every residue mod 4, all four quarters; the other 192 slots carry different latents) and must reproduce the golden
    within the FP32 trajectory tolerance; every 3x3 layer of the body must be on the emulating kernel (ds_conv_kernel_id 2563)."""
    import ctypes as C
    from diff_sampler_amd import _lib, solvers
    from diff_sampler_amd.engine import EDMDenoiser
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from _parity import record
    dev = torch.device('cuda')
    z = np.load(os.path.join(G, 'sampler_cifar10_dpmpp2m_nfe10_b64.npz'))
    net = EDMDenoiser.from_config('cifar10', seed=int(z['seed']), split_fp16=True)
    gold = torch.randn(64, 3, 32, 32, generator=torch.Generator().manual_seed(int(z['latent_seed'])))
    latents = torch.randn(256, 3, 32, 32, generator=torch.Generator().manual_seed(99))
    slots = torch.arange(64) * 4 + torch.arange(64) % 4
    latents[slots] = gold
    out = solvers.dpm_pp_sampler(net, latents.to(dev), num_steps=11, sigma_min=0.002, sigma_max=80., schedule_type='logsnr', schedule_rho=7,
                                 max_order=2, predict_x0=True, lower_order_final=True)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    err = _rel(out.cpu()[slots], torch.from_numpy(z['out']))
    lib = _lib.load()
    ids = [lib.ds_conv_kernel_id(C.byref(op.keep[0])) for op in net.engine.plan(256, 1).ops
           if op.fn is lib.ds_conv2d_nhwc and op.keep[0].taps == 9]
    record('cifar10_dpmpp2m_nfe10_b256_fp16x3', final=err, bound=5e-4, conv_kernel_ids={str(k): ids.count(k) for k in sorted(set(ids))})
    assert err < 5e-4, err
    assert ids.count(2563) >= 60 and set(ids) <= {2563, 2570}, ids          # split kernel everywhere but the 3-channel head (VALU kernel)
