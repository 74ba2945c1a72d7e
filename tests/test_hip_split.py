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
