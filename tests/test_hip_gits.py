"""GPU: GITS schedule search (cost matrix from trajectory moments, DP, AFS slot search) against the real reference's
outputs on the same seeded warm-up latents (tests/golden/gits.npz, made by oracle/gen_golden.py --part gits)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

from oracle import cases  # noqa: E402

G = os.path.join(ROOT, 'tests', 'golden')


def test_cal_deviation_matches_reference():
    from diff_sampler_amd import gits_utils
    z = np.load(os.path.join(G, 'gits.npz'))
    dev = gits_utils.cal_deviation(torch.from_numpy(z['dev_traj']).cuda(), 3, 16, bs=3)
    assert np.allclose(dev.cpu().numpy(), z['dev_out'], rtol=2e-5, atol=1e-5)


@pytest.mark.parametrize('tag', [c[0] for c in cases.GITS_CASES])
def test_get_dp_list_matches_reference(tag):
    from diff_sampler_amd import gits_utils
    from diff_sampler_amd.engine import EDMDenoiser
    z = np.load(os.path.join(G, 'gits.npz'))
    net = EDMDenoiser.from_config('tiny_song', seed=int(z['seed']))
    gk = dict(cases.GITS_CASES)[tag]
    kwargs = dict(cases.GITS_COMMON); kwargs.update(gk)
    rounds = kwargs['num_warmup'] // (kwargs['max_batch_size'] + 1) + 1
    lat = cases.gits_warmup_latents(1000 + len(tag), rounds, kwargs['max_batch_size'], (3, 16, 16))
    dp_list = gits_utils.get_dp_list(net, torch.device('cuda'), warmup_latents=lat, **kwargs)
    assert list(dp_list) == list(z[f'{tag}_dp_list']), (tag, dp_list, z[f'{tag}_dp_list'])
    # the searched schedule is consumed exactly like the reference does (gits-main/solver_utils.py:52-53)
    from diff_sampler_amd import solver_utils
    ts = solver_utils.get_schedule(kwargs['num_steps_tea'], 0.002, 80., device='cuda', schedule_type=kwargs['schedule_type'],
                                   schedule_rho=kwargs['schedule_rho'], dp_list=dp_list)
    assert ts.shape[0] == len(dp_list) and float(ts[0]) > float(ts[-1])


def test_get_dp_list_on_the_full_size_cifar10_net_matches_reference():
    """The schedule search itself (gits-main/gits_utils.py:42-255) on the FULL-size CIFAR-10 net against the real reference
    (oracle/gen_golden.py --part fullgits): 21-step iPNDM-4 teacher on 8 warm-up latents, 'dev' cost, 6-step student."""
    from diff_sampler_amd import gits_utils
    from diff_sampler_amd.engine import EDMDenoiser
    z = np.load(os.path.join(G, 'gits_cifar10.npz'))
    net = EDMDenoiser.from_config('cifar10', seed=int(z['seed']))
    tag, gk = cases.GITS_FULL_CASE
    kwargs = dict(cases.GITS_COMMON); kwargs.update(gk)
    rounds = kwargs['num_warmup'] // (kwargs['max_batch_size'] + 1) + 1
    lat = cases.gits_warmup_latents(int(z['warmup_seed']), rounds, kwargs['max_batch_size'], (3, 32, 32))
    dp_list = gits_utils.get_dp_list(net, torch.device('cuda'), warmup_latents=lat, **kwargs)
    assert list(dp_list) == list(z['dp_list']), (dp_list, z['dp_list'])


def test_get_dp_list_on_a_latent_diffusion_denoiser_matches_reference():
    """model_source == 'ldm' (gits-main/gits_utils.py:86-108): text conditions instead of labels, classifier-free guidance inside the
    denoiser, 'discrete' schedule from the net's own sigma(t).  Golden = the REAL reference's get_dp_list on the tiny LDM U-Net with a
    shimmed text encoder (oracle/gen_golden.py --part gitsldm); here the same latents and conditions go through the HIP CFGDenoiser."""
    from diff_sampler_amd import gits_utils
    from diff_sampler_amd.ldm_engine import CFGDenoiser
    import diff_sampler_amd.ldm_arch as la
    z = np.load(os.path.join(G, 'gits_ldm.npz'))
    net = CFGDenoiser.from_config('tiny_ldm', seed=int(z['seed']), guidance_rate=7.5)
    assert abs(net.sigma_min - float(z['sigma_min'])) < 1e-6 and abs(net.sigma_max - float(z['sigma_max'])) < 1e-4
    tag, gk = cases.GITS_LDM_CASE
    kwargs = dict(cases.GITS_COMMON); kwargs.update(gk)
    kwargs['sigma_min'], kwargs['sigma_max'] = net.sigma_min, net.sigma_max
    kw = la.NAMED_LDM_CONFIGS['tiny_ldm']
    rounds = kwargs['num_warmup'] // (kwargs['max_batch_size'] + 1) + 1
    lat = cases.gits_warmup_latents(int(z['warmup_seed']), rounds, kwargs['max_batch_size'], (kw['in_channels'], kw['img_resolution'], kw['img_resolution']))
    conds = cases.gits_ldm_conditions(int(z['cond_seed']), rounds, kwargs['max_batch_size'], kw['context_dim'])
    dp_list = gits_utils.get_dp_list(net, torch.device('cuda'), warmup_latents=lat, warmup_conditions=conds, **kwargs)
    assert list(dp_list) == list(z['dp_list']), (dp_list, z['dp_list'])
    # without the test hooks the search draws its own conditions (seeded N(0,1) CLIP-shaped states for an engine net) and still returns a path
    dp2 = gits_utils.get_dp_list(net, torch.device('cuda'), **kwargs)
    assert dp2[0] == 0 and dp2[-1] == kwargs['num_steps_tea'] - 1 and len(dp2) == kwargs['num_steps']
