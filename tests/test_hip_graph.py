"""GPU: a hipGraph-captured sampler call replays bit-identically to the eager call, for new latents and labels."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('case', ['ipndm_uncond', 'dpmpp_cond'])
def test_graphed_sampler_equals_eager(case):
    from diff_sampler_amd import solvers
    from diff_sampler_amd.engine import EDMDenoiser
    from diff_sampler_amd.graph import GraphedSampler
    dev = torch.device('cuda')
    if case == 'ipndm_uncond':
        net = EDMDenoiser.from_config('tiny_song', seed=9)
        fn, kw, lab_shape = solvers.ipndm_sampler, dict(num_steps=7, max_order=4), None
    else:
        net = EDMDenoiser.from_config('tiny_song_cond', seed=9)
        fn, kw, lab_shape = solvers.dpm_pp_sampler, dict(num_steps=6, max_order=2, schedule_type='logsnr'), (4, 10)
    g = GraphedSampler(fn, net, (4, 3, 16, 16), class_labels_shape=lab_shape, **kw)
    gen = torch.Generator().manual_seed(0)
    for trial in range(3):
        lat = torch.randn(4, 3, 16, 16, generator=gen).to(dev)
        lab = torch.eye(10)[torch.randint(10, (4,), generator=gen)].to(dev) if lab_shape else None
        eager = fn(net, lat, class_labels=lab, **kw)
        graphed = g(lat, lab)
        torch.cuda.synchronize()
        assert torch.equal(eager, graphed), (case, trial)


def test_graphed_cfg_sampler_equals_eager():
    """The config-5 solver on the latent-diffusion denoiser (CFG-doubled evaluations) captured into one hipGraph."""
    from diff_sampler_amd import solvers
    from diff_sampler_amd.graph import GraphedSampler
    from diff_sampler_amd.ldm_engine import CFGDenoiser
    dev = torch.device('cuda')
    net = CFGDenoiser.from_config('tiny_ldm_1res', seed=9, guidance_rate=7.5)
    kw = dict(num_steps=4, sigma_min=net.sigma_min, sigma_max=net.sigma_max, schedule_type='discrete', schedule_rho=1, max_order=2,
              predict_x0=False, lower_order_final=True)
    g = GraphedSampler(solvers.dpm_pp_sampler, net, (2, 4, 16, 16), condition_shape=(2, 77, 64), uncond_shape=(2, 77, 64), **kw)
    gen = torch.Generator().manual_seed(0)
    for trial in range(2):
        lat = torch.randn(2, 4, 16, 16, generator=gen).to(dev)
        c, uc = torch.randn(2, 77, 64, generator=gen).to(dev), torch.randn(2, 77, 64, generator=gen).to(dev)
        eager = solvers.dpm_pp_sampler(net, lat, condition=c, unconditional_condition=uc, **kw)
        graphed = g(lat, condition=c, unconditional_condition=uc)
        torch.cuda.synchronize()
        assert torch.equal(eager, graphed), trial
