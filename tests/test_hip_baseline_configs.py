"""BASELINE.json configs 3 and 4 at FULL network size on the HIP path against the CPU oracle (same seeded weights, same
latents), complementing the golden-vector tests (config 1/2: tests/test_hip_samplers.py; config 5: tests/test_hip_ldm.py).

  config 3: EDM ImageNet-64 class-conditional DhariwalUNet (295.9M params), iPNDM (max_order 4) on an 11-point schedule of
            the GITS form (a fixed literal through `t_steps`, as `sample.py --t_steps`), NFE = 10
  config 4: AMED-Solver on the FFHQ-64 SongUNet (61.8M params) with a seeded AMED_predictor, num_steps = 4, afs = True,
            time_uniform rho = 1  =>  5 NFE  (amed-solver-main/launch.sh:21-24)
Tolerances as stated in DESIGN.md section 2: 5e-4 (config 3), 1e-3 (AMED) -- of EACH step's own scale, the final image included
(tests/_parity.py)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

import diff_sampler_amd.arch as arch  # noqa: E402
from oracle import cases, solvers_ref  # noqa: E402
from oracle.edm_net import OracleNet  # noqa: E402
from _parity import per_step_rel, record, step_scales  # noqa: E402


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-6))


def test_config3_imagenet64_ipndm_gits_schedule_nfe10():
    from diff_sampler_amd import solvers
    from diff_sampler_amd.engine import EDMDenoiser
    dev = torch.device('cuda')
    kw = dict(arch.NAMED_CONFIGS['imagenet64'])
    spec = arch.edm_precond_spec(**kw)
    params = arch.init_params(spec, seed=31)
    g = torch.Generator().manual_seed(32)
    B = 2
    latents = torch.randn(B, 3, 64, 64, generator=g)
    labels = torch.eye(1000)[torch.randint(1000, (B,), generator=g)]
    t_steps = torch.tensor([80.0, 31.78, 14.51, 7.42, 3.88, 2.05, 1.06, 0.5666, 0.2531, 0.0631, 0.002])   # 11 points => NFE 10
    with torch.no_grad():
        ref = solvers_ref.sample('ipndm', OracleNet(params, kw), latents, t_steps, class_labels=labels, max_order=4, want_inters=True)
    net = EDMDenoiser(spec, params)
    out = solvers.ipndm_sampler(net, latents.to(dev), class_labels=labels.to(dev), max_order=4, t_steps=t_steps.to(dev),
                                num_steps=11, return_inters=True)
    torch.cuda.synchronize()
    assert out.shape == ref.shape == (11, B, 3, 64, 64)
    errs = per_step_rel(out.cpu(), ref)
    record('config3_imagenet64_vs_oracle_b2_fp32', per_step=errs, final=errs[-1], step_scales=step_scales(ref), bound=5e-4)
    assert max(errs) < 5e-4 and errs[-1] < 5e-4, errs


def test_config4_ffhq64_amed_solver_nfe5():
    from diff_sampler_amd import solvers_amed
    from diff_sampler_amd.engine import EDMDenoiser
    dev = torch.device('cuda')
    kw = dict(arch.NAMED_CONFIGS['ffhq'])
    spec = arch.edm_precond_spec(**kw)
    params = arch.init_params(spec, seed=41)
    g = torch.Generator().manual_seed(42)
    B = 2
    latents = torch.randn(B, 3, 64, 64, generator=g)
    pk = dict(scale_dir=0.01, scale_time=0)
    pp = cases.amed_predictor_params(43, pk['scale_dir'], pk['scale_time'])
    ts = solvers_ref.schedule(4, 0.002, 80., kind='time_uniform', rho=1)
    with torch.no_grad():
        ref = solvers_ref.sample('amed', OracleNet(params, kw), latents, ts, afs=True, num_steps=4,
                                 predictor=lambda b, tc, tn: solvers_ref.amed_predict(pp, pk, b, tc, tn), want_inters=True)
    net = EDMDenoiser(spec, params)
    pred = solvers_amed.AMEDPredictor(pp, device=dev, num_steps=4, sampler_stu='amed', schedule_type='time_uniform', schedule_rho=1,
                                      afs=True, **pk)
    out = solvers_amed.amed_sampler(net, latents.to(dev), num_steps=4, sigma_min=0.002, sigma_max=80., schedule_type='time_uniform',
                                    schedule_rho=1, afs=True, return_inters=True, AMED_predictor=pred)
    torch.cuda.synchronize()
    assert tuple(out.shape) == tuple(ref.shape)
    errs = per_step_rel(out.cpu(), ref)
    record('config4_ffhq64_amed_vs_oracle_b2_fp32', per_step=errs, final=errs[-1], step_scales=step_scales(ref), bound=1e-3)
    assert max(errs) < 1e-3 and errs[-1] < 1e-3, errs
