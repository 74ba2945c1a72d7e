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
