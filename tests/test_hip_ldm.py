"""GPU parity of the latent-diffusion path (BASELINE config 5): HIP ``CFGDenoiser`` (through the C ABI) against the golden
vectors the real reference produced (tests/golden/ldm_*.npz) -- tiny configs with every layer type, and one full-size
Stable-Diffusion-v1.5 evaluation (859.5M parameters, 64x64x4 latents, 77x768 context, classifier-free guidance 7.5).

Tolerance (fp32 path, stated): 2e-4 of the output scale per evaluation (observed ~1e-5), 1e-3 for 5-step trajectories."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu
G = os.path.join(ROOT, 'tests', 'golden')
TOL = 2e-4


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-6))


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda')


@pytest.mark.parametrize('name', ['tiny_ldm_1res', 'tiny_ldm', 'sd15'])
def test_cfg_denoiser_matches_golden(name, dev):
    from diff_sampler_amd.ldm_engine import CFGDenoiser
    z = np.load(os.path.join(G, f'ldm_{name}.npz'))
    net = CFGDenoiser.from_config(name, seed=int(z['seed']), guidance_rate=7.5)
    assert abs(net.sigma_min - float(z['sigma_min'])) < 1e-6 and abs(net.sigma_max - float(z['sigma_max'])) < 1e-4
    assert np.allclose(net.sigma_inv(torch.from_numpy(z['probe_sigma'])).numpy(), z['probe_sigma_inv'], rtol=2e-6, atol=1e-7)
    assert np.allclose(net.sigma(torch.from_numpy(z['probe_t'])).numpy(), z['probe_sigma_of_t'], rtol=2e-6, atol=1e-7)
    x, cond, uncond = (torch.from_numpy(z[k]).to(dev) for k in ('x', 'cond', 'uncond'))
    out = net(x, torch.from_numpy(z['sigma']).to(dev), condition=cond, unconditional_condition=uncond)
    torch.cuda.synchronize()
    assert _rel(out.cpu(), torch.from_numpy(z['out_vec'])) < TOL
    if name != 'sd15':
        out = net(x, torch.tensor(1.9), condition=cond, unconditional_condition=uncond)
        assert _rel(out.cpu(), torch.from_numpy(z['out_scalar'])) < TOL
        out = net(x, 1.9, condition=cond, unconditional_condition=None)
        assert _rel(out.cpu(), torch.from_numpy(z['out_nocfg'])) < TOL


@pytest.mark.parametrize('name', ['tiny_ldm_1res', 'tiny_ldm'])
def test_ldm_layer_taps_match_oracle(name, dev):
    """Every layer output of the plan against the oracle's taps (localises a failure to a layer)."""
    import diff_sampler_amd.ldm_arch as la
    from diff_sampler_amd.ldm_engine import CFGDenoiser
    from oracle.ldm_net import unet_forward
    z = np.load(os.path.join(G, f'ldm_{name}.npz'))
    kw = dict(la.NAMED_LDM_CONFIGS[name])
    spec = la.ldm_unet_spec(**kw)
    params = la.init_ldm_params(spec, seed=int(z['seed']))
    net = CFGDenoiser(spec, params, guidance_rate=1.0)
    x, cond = torch.from_numpy(z['x']), torch.from_numpy(z['cond'])
    sigma = 1.9
    f, plan, _ = net.raw(x.to(dev), sigma, cond.to(dev), None)
    torch.cuda.synchronize()
    taps = {}
    c_in = 1 / (sigma ** 2 + 1) ** 0.5
    cn = (net.M * net.sigma_inv(torch.tensor(sigma)) - 1.).expand(x.shape[0])
    with torch.no_grad():
        ref = unet_forward(params, kw, c_in * x, cn, cond, taps=taps)
    B = x.shape[0]
    for key, t in taps.items():
        got = plan.bufs[key].cpu().reshape(B, t.shape[2], t.shape[3], t.shape[1]).permute(0, 3, 1, 2)
        assert _rel(got, t) < TOL, key
    got = f.cpu().reshape(B, ref.shape[2], ref.shape[3], 4).permute(0, 3, 1, 2)[:, :ref.shape[1]]
    assert _rel(got, ref) < TOL


@pytest.mark.parametrize('name', ['tiny_ldm_1res', 'tiny_ldm'])
def test_ldm_samplers_match_golden(name, dev):
    """Config-5 solver (DPM-Solver++(2M), noise prediction, discrete rho=1 schedule, CFG-doubled evaluations) and friends on
    the HIP path against the real reference's trajectories."""
    from diff_sampler_amd import solvers, solver_utils
    from diff_sampler_amd.ldm_engine import CFGDenoiser
    z = np.load(os.path.join(G, f'ldm_{name}.npz'))
    net = CFGDenoiser.from_config(name, seed=int(z['seed']), guidance_rate=7.5)
    lat, cond, uncond = (torch.from_numpy(z[k]).to(dev) for k in ('latents', 'cond', 'uncond'))
    sched = solver_utils.get_schedule(6, net.sigma_min, net.sigma_max, device=dev, schedule_type='discrete', schedule_rho=1, net=net)
    assert np.allclose(sched.cpu().numpy(), z['sched_discrete_6'], rtol=5e-6)
    cases = [('dpmpp2m_eps', solvers.dpm_pp_sampler, dict(max_order=2, predict_x0=False, lower_order_final=True)),
             ('dpmpp2m_x0', solvers.dpm_pp_sampler, dict(max_order=2, predict_x0=True, lower_order_final=True)),
             ('euler', solvers.euler_sampler, {}), ('ipndm3', solvers.ipndm_sampler, dict(max_order=3))]
    for tag, fn, kws in cases:
        tr = fn(net, lat, condition=cond, unconditional_condition=uncond, num_steps=6, sigma_min=net.sigma_min, sigma_max=net.sigma_max,
                schedule_type='discrete', schedule_rho=1, return_inters=True, **kws)
        torch.cuda.synchronize()
        ref = torch.from_numpy(z[f'traj_{tag}'])
        assert tr.shape == ref.shape
        assert _rel(tr.cpu(), ref) < 1e-3, tag


def test_cross_attention_kv_projections_are_cached_per_context(dev):
    """to_k(context) / to_v(context) (ldm/modules/attention.py:168-176) do not depend on x or sigma: they run once per context TENSOR, not
    once per network evaluation.  Same results; a new tensor object, or an in-place change of the cached one, recomputes them."""
    from diff_sampler_amd.ldm_engine import CFGDenoiser
    z = np.load(os.path.join(G, 'ldm_tiny_ldm.npz'))
    net = CFGDenoiser.from_config('tiny_ldm', seed=int(z['seed']), guidance_rate=7.5)
    x, cond, uncond = (torch.from_numpy(z[k]).to(dev) for k in ('x', 'cond', 'uncond'))
    f, plan, _ = net.raw(x, 1.9, cond, uncond)
    assert len(plan.ctx.ops) > 0 and all(op.name.endswith('.attn2.kv') for op in plan.ctx.ops)
    assert not any(op.name.endswith('.attn2.kv') for op in plan.ops)
    a = net(x, 1.9, condition=cond, unconditional_condition=uncond).clone()
    runs = []
    real = plan.ctx.run
    plan.ctx.run = lambda st: (runs.append(1), real(st))[1]
    b = net(x, 1.9, condition=cond, unconditional_condition=uncond).clone()          # same tensors: projections reused
    assert runs == [] and torch.equal(a, b)
    net.cache_context = False
    c = net(x, 1.9, condition=cond, unconditional_condition=uncond).clone()
    assert runs == [1] and torch.equal(a, c)
    net.cache_context = True
    cond2 = cond.clone()                                                                # a different tensor object: recomputed
    d = net(x, 1.9, condition=cond2, unconditional_condition=uncond).clone()
    assert runs == [1, 1] and torch.equal(a, d)
    cond2.mul_(0.5)                                                                     # in-place change of the cached tensor: recomputed
    e = net(x, 1.9, condition=cond2, unconditional_condition=uncond).clone()
    assert runs == [1, 1, 1] and not torch.equal(a, e)
    ref = CFGDenoiser.from_config('tiny_ldm', seed=int(z['seed']), guidance_rate=7.5)
    ref.cache_context = False
    assert torch.equal(e, ref(x, 1.9, condition=cond2, unconditional_condition=uncond))
