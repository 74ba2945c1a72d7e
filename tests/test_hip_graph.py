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
    # the eager comparison runs on a SECOND denoiser (same weights, its own plan buffers): an eager call on `net` would refresh the very
    # context / K / V buffers a replay that lost its context nodes reads, and hide that
    ref_net = CFGDenoiser.from_config('tiny_ldm_1res', seed=9, guidance_rate=7.5)
    gen = torch.Generator().manual_seed(0)
    prev = None
    for trial in range(3):
        lat = torch.randn(2, 4, 16, 16, generator=gen).to(dev) if trial < 2 else lat       # trial 2: only the conditions change
        c, uc = torch.randn(2, 77, 64, generator=gen).to(dev), torch.randn(2, 77, 64, generator=gen).to(dev)
        graphed = g(lat, condition=c, unconditional_condition=uc)                            # replay FIRST, before any eager call
        eager = solvers.dpm_pp_sampler(ref_net, lat, condition=c, unconditional_condition=uc, **kw)
        torch.cuda.synchronize()
        assert torch.equal(eager, graphed), trial
        if trial == 2:
            assert not torch.equal(prev, graphed), 'a replay must follow the condition tensors'
        prev = graphed
    # an eager call on the CAPTURED denoiser after a replay, with a context it had cached before the replay overwrote the K / V buffers
    c0, uc0 = torch.randn(2, 77, 64, generator=gen).to(dev), torch.randn(2, 77, 64, generator=gen).to(dev)
    a = solvers.dpm_pp_sampler(net, lat, condition=c0, unconditional_condition=uc0, **kw)    # caches (c0, uc0)
    g(lat, condition=c, unconditional_condition=uc)                                         # replay: buffers now hold (c, uc)
    b = solvers.dpm_pp_sampler(net, lat, condition=c0, unconditional_condition=uc0, **kw)
    torch.cuda.synchronize()
    assert torch.equal(a, b)


def test_context_cache_follows_in_place_updates_and_pins_nothing():
    """CFGDenoiser's per-context K / V cache: same tensor object + same version => projections skipped; an in-place update or another
    tensor => recomputed; the cache holds only weak references."""
    import gc
    import weakref
    from diff_sampler_amd.ldm_engine import CFGDenoiser
    dev = torch.device('cuda')
    net = CFGDenoiser.from_config('tiny_ldm_1res', seed=9, guidance_rate=7.5)
    ref = CFGDenoiser.from_config('tiny_ldm_1res', seed=9, guidance_rate=7.5)
    ref.cache_context = False
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(2, 4, 16, 16, generator=gen).to(dev)
    c, uc = torch.randn(2, 77, 64, generator=gen).to(dev), torch.randn(2, 77, 64, generator=gen).to(dev)
    for trial in range(3):
        if trial == 1:
            c.mul_(0.5)                                              # in-place: `_version` moves, the projections must rerun
        assert torch.equal(net(x, 1.3, condition=c, unconditional_condition=uc), ref(x, 1.3, condition=c, unconditional_condition=uc)), trial
    w = weakref.ref(c)
    del c
    gc.collect()
    assert w() is None, 'the context cache must not keep the caller\'s condition tensors alive'
