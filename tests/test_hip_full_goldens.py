"""GPU parity at FULL size against golden vectors the REAL reference produced (oracle/gen_golden.py --part full):

  * the benchmarked configuration itself -- CIFAR-10 net, DPM-Solver++(2M) logSNR NFE=10 (the sampler bench.py times) at B=64,
    where the 32x32 layers dispatch to the 256-pixel-tile halo kernel (conv3x3_halo_kernel<4>, the dominant kernel of the
    headline); the dispatch is asserted through ds_conv_kernel_id; and the same 64 golden samples embedded in a call at the BENCHMARK
    batch of 256 images;
  * one full-size evaluation each of the FFHQ-64 SongUNet and the ImageNet-64 DhariwalUNet (BASELINE configs 4 / 3);
  * BASELINE configs 3 and 4 through the reference's own sampler calls at full size (ImageNet-64 iPNDM-4 on the GITS-form schedule,
    FFHQ-64 AMED-Solver): whole trajectories;
  * one full-size Stable-Diffusion-v1.5 config-5 trajectory (DPM-Solver++(2M) eps-prediction, discrete rho=1, CFG 7.5).

Tolerances (fp32 path): 2e-4 per evaluation, 5e-4 per EDM trajectory, 1e-3 for the 5-step SD trajectory.
Trajectories are bounded PER STEP, each step against its own golden scale (tests/_parity.py: a trajectory runs from scale ~300 at sigma_max
to an image of scale ~3, so one normalisation over the whole trajectory would leave the final image unconstrained), and the final image
explicitly.  What every test observed goes to gpurun_out/r4_parity.json (kept copy: profiles/r4_parity.json)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from _parity import per_step_rel, record, step_scales  # noqa: E402  (tests/ is on sys.path under pytest rootdir-less collection)

pytestmark = pytest.mark.gpu
G = os.path.join(ROOT, 'tests', 'golden')


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-6))


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda')


def test_headline_sampler_b64_matches_reference_and_uses_the_256_tile_kernel(dev):
    from diff_sampler_amd import _lib, solvers
    from diff_sampler_amd.engine import EDMDenoiser
    z = np.load(os.path.join(G, 'sampler_cifar10_dpmpp2m_nfe10_b64.npz'))
    net = EDMDenoiser.from_config('cifar10', seed=int(z['seed']))
    latents = torch.randn(64, 3, 32, 32, generator=torch.Generator().manual_seed(int(z['latent_seed']))).to(dev)
    out = solvers.dpm_pp_sampler(net, latents, num_steps=11, sigma_min=0.002, sigma_max=80., schedule_type='logsnr', schedule_rho=7,
                                 max_order=2, predict_x0=True, lower_order_final=True)
    torch.cuda.synchronize()
    assert _rel(out.cpu(), torch.from_numpy(z['out'])) < 5e-4
    # which kernel ran: every 3x3 layer at 32x32 with a multiple-of-128 channel count must be on the 256-pixel tile -- the 8-wave
    # kernel <4>: 256 = its 128-column tiles, 2565 = its 256-column tiles (the benchmarked kernel at 256 output channels)
    lib = _lib.load()
    plan = net.engine.plan(64, 1)
    ids = []
    for op in plan.ops:
        if op.fn is lib.ds_conv2d_nhwc:
            a = op.keep[0]
            if a.taps == 9 and a.h == 32 and a.cout % 128 == 0:
                ids.append(lib.ds_conv_kernel_id(C.byref(a)))
    assert len(ids) >= 20 and all(i in (256, 2565) for i in ids) and sum(i == 2565 for i in ids) >= 20, ids


def test_headline_sampler_at_the_benchmark_batch_b256_matches_reference(dev):
    """The sampler call bench.py times, at ITS batch (256 images: four rounds of 256 x 256 tiles per 32x32 layer, one round per 16x16
    layer -- tilings a 64-image call does not reach).  Samples are independent (per-image GroupNorm, per-sample thresholding), so the 64
    latents of the real reference's golden run are scattered over the 256-image batch (every fourth slot, i.e. all four rounds and
    every tile position of an image) and their outputs compared with that golden; the other 192 slots carry different latents and
    must not leak into them."""
    from diff_sampler_amd import solvers
    from diff_sampler_amd.engine import EDMDenoiser
    z = np.load(os.path.join(G, 'sampler_cifar10_dpmpp2m_nfe10_b64.npz'))
    net = EDMDenoiser.from_config('cifar10', seed=int(z['seed']))
    gold = torch.randn(64, 3, 32, 32, generator=torch.Generator().manual_seed(int(z['latent_seed'])))
    latents = torch.randn(256, 3, 32, 32, generator=torch.Generator().manual_seed(99))
    slots = torch.arange(64) * 4 + torch.arange(64) % 4            # 0, 5, 10, 15, 16, 21, ...: every residue mod 4, all four quarters
    latents[slots] = gold
    out = solvers.dpm_pp_sampler(net, latents.to(dev), num_steps=11, sigma_min=0.002, sigma_max=80., schedule_type='logsnr', schedule_rho=7,
                                 max_order=2, predict_x0=True, lower_order_final=True)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    assert _rel(out.cpu()[slots], torch.from_numpy(z['out'])) < 5e-4


@pytest.mark.parametrize('name', ['ffhq', 'imagenet64'])
def test_full_size_64px_nets_match_reference(name, dev):
    from diff_sampler_amd.engine import EDMDenoiser
    z = np.load(os.path.join(G, f'net_{name}.npz'))
    net = EDMDenoiser.from_config(name, seed=int(z['seed']))
    lab = torch.from_numpy(z['labels']).to(dev) if z['labels'].size else None
    out = net(torch.from_numpy(z['x']).to(dev), torch.from_numpy(z['sigma']).to(dev), class_labels=lab)
    torch.cuda.synchronize()
    assert _rel(out.cpu(), torch.from_numpy(z['out_vec'])) < 2e-4


def test_every_solver_family_on_the_full_size_cifar10_net_matches_reference(dev):
    """The real reference's solvers on the full-size CIFAR-10 net at NFE = 10, B = 4 (oracle/gen_golden.py --part fullsolv): every solver
    family north_star names beyond the headline's DPM-Solver++(2M) -- Heun, DPM-Solver-2, iPNDM on a polynomial and on the GITS-form
    schedule, iPNDM_v with AFS, DEIS tAB3 on time_uniform, DPM-Solver++(3M) / (2M, eps form), UniPC bh2 -- final images."""
    from diff_sampler_amd import solvers, solver_utils
    from diff_sampler_amd.engine import EDMDenoiser
    from oracle import cases
    z = np.load(os.path.join(G, 'sampler_cifar10_solvers_nfe10_b4.npz'))
    net = EDMDenoiser.from_config('cifar10', seed=int(z['seed']))
    latents = torch.randn(4, 3, 32, 32, generator=torch.Generator().manual_seed(int(z['latent_seed']))).to(dev)
    for tag, fn, kind, rho, n, extra in cases.FULL_SOLVER_CASES:
        extra = dict(extra)
        ts = torch.from_numpy(z[f'{tag}_t']).to(dev)
        if fn == 'deis_sampler':
            extra['coeff_list'] = solver_utils.get_deis_coeff_list(ts, extra['max_order'], deis_mode=extra.pop('deis_mode'))
        out = getattr(solvers, fn)(net, latents, num_steps=n, t_steps=ts, **extra)
        torch.cuda.synchronize()
        assert _rel(out.cpu(), torch.from_numpy(z[f'{tag}_out'])) < 5e-4, (tag, _rel(out.cpu(), torch.from_numpy(z[f'{tag}_out'])))


def test_config3_imagenet64_trajectory_matches_reference(dev):
    """BASELINE config 3 at full size against the REAL reference's own sampler call (oracle/gen_golden.py --part full3): ImageNet-64
    DhariwalUNet with a one-hot label, ipndm_sampler max_order 4 on the GITS-form schedule literal, NFE = 10, every intermediate."""
    from diff_sampler_amd import solvers
    from diff_sampler_amd.engine import EDMDenoiser
    z = np.load(os.path.join(G, 'sampler_imagenet64_ipndm_gits_nfe10_b1.npz'))
    net = EDMDenoiser.from_config('imagenet64', seed=int(z['seed']))
    out = solvers.ipndm_sampler(net, torch.from_numpy(z['latents']).to(dev), class_labels=torch.from_numpy(z['labels']).to(dev), max_order=4,
                                t_steps=torch.from_numpy(z['t_steps']).to(dev), num_steps=11, return_inters=True)
    torch.cuda.synchronize()
    assert tuple(out.shape) == tuple(z['traj'].shape)
    gold = torch.from_numpy(z['traj'])
    errs = per_step_rel(out.cpu(), gold)
    record('config3_imagenet64_ipndm4_gits_b1_fp32', per_step=errs, final=errs[-1], step_scales=step_scales(gold), bound=5e-4)
    assert max(errs) < 5e-4 and errs[-1] < 5e-4, errs


def test_config4_ffhq64_amed_trajectory_matches_reference(dev):
    """BASELINE config 4 at full size against the REAL reference's amed_sampler (oracle/gen_golden.py --part full4): FFHQ-64 SongUNet +
    AMED_predictor(num_steps = 4, afs, time_uniform rho = 1, scale_dir 0.01), 5 NFE; tolerance 1e-3 (per-sample powf / expm1f on the device)."""
    from diff_sampler_amd import solvers_amed
    from diff_sampler_amd.engine import EDMDenoiser
    from oracle import cases
    z = np.load(os.path.join(G, 'sampler_ffhq_amed_nfe5_b2.npz'))
    net = EDMDenoiser.from_config('ffhq', seed=int(z['seed']))
    pk = dict(scale_dir=float(z['scale_dir']), scale_time=float(z['scale_time']))
    pp = cases.amed_predictor_params(int(z['predictor_seed']), pk['scale_dir'], pk['scale_time'])
    pred = solvers_amed.AMEDPredictor(pp, device=dev, num_steps=4, sampler_stu='amed', schedule_type='time_uniform', schedule_rho=1,
                                      afs=True, **pk)
    out = solvers_amed.amed_sampler(net, torch.from_numpy(z['latents']).to(dev), num_steps=4, sigma_min=0.002, sigma_max=80.,
                                    schedule_type='time_uniform', schedule_rho=1, afs=True, return_inters=True, AMED_predictor=pred)
    torch.cuda.synchronize()
    assert tuple(out.shape) == tuple(z['traj'].shape)
    gold = torch.from_numpy(z['traj'])
    errs = per_step_rel(out.cpu(), gold)
    record('config4_ffhq64_amed_nfe5_b2_fp32', per_step=errs, final=errs[-1], step_scales=step_scales(gold), bound=1e-3)
    assert max(errs) < 1e-3 and errs[-1] < 1e-3, errs


def test_sd15_config5_trajectory_matches_reference(dev):
    from diff_sampler_amd import solvers
    from diff_sampler_amd.ldm_engine import CFGDenoiser
    z = np.load(os.path.join(G, 'ldm_sd15_traj.npz'))
    net = CFGDenoiser.from_config('sd15', seed=int(z['seed']), guidance_rate=7.5)
    lat, cond, uncond = (torch.from_numpy(z[k]).to(dev) for k in ('latents', 'cond', 'uncond'))
    tr = solvers.dpm_pp_sampler(net, lat, condition=cond, unconditional_condition=uncond, num_steps=6, sigma_min=net.sigma_min,
                                sigma_max=net.sigma_max, schedule_type='discrete', schedule_rho=1, return_inters=True,
                                max_order=2, predict_x0=False, lower_order_final=True)
    torch.cuda.synchronize()
    ref = torch.from_numpy(z['traj'])
    assert tr.shape == ref.shape
    errs = per_step_rel(tr.cpu(), ref)
    record('config5_sd15_dpmpp2m_b_fp32', per_step=errs, final=errs[-1], step_scales=step_scales(ref), bound=1e-3)
    assert max(errs) < 1e-3 and errs[-1] < 1e-3, errs


# ---- the OTHER benchmarked configurations at THEIR bench batches (bench.py --config imagenet64 --batch 64 / --config ffhq --batch 128),
# ---- fp32 and the reference's fp16 mode: golden samples of the real reference scattered over the batch, kernel routing asserted --------
def _conv_kernel_ids(net, B, emb_rows):
    """{ds_conv_kernel_id: number of 3x3 launches} of the plan an evaluation at batch B runs."""
    from diff_sampler_amd import _lib
    lib = _lib.load()
    ids = {}
    for op in net.engine.plan(B, emb_rows).ops:
        if op.fn is lib.ds_conv2d_nhwc and op.keep[0].taps == 9:
            k = lib.ds_conv_kernel_id(C.byref(op.keep[0]))
            ids[k] = ids.get(k, 0) + 1
    return ids


@pytest.mark.parametrize('mode', ['fp32', 'fp16'])
def test_config3_imagenet64_at_the_benchmark_batch_b64(mode, dev):
    """`bench.py --config imagenet64 --batch 64 [--dtype fp16]`: the real reference's config-3 trajectory (B = 1, label, iPNDM-4 on the
    GITS-form schedule, NFE = 10) occupies slots 0 / 21 / 42 / 63 of a 64-image call whose other slots carry different latents and labels;
    every copy must reproduce the golden (samples are independent), on the tilings only this batch dispatches.  fp32: 5e-4 of the trajectory
    scale.  use_fp16 (the checkpoint's own mode, networks_edm.py:486): 1e-2 over the 10-step trajectory against the FP32 reference, and
    5e-3 for one evaluation at this batch (golden net_imagenet64.npz scattered the same way)."""
    from diff_sampler_amd import solvers
    from diff_sampler_amd.engine import EDMDenoiser
    z = np.load(os.path.join(G, 'sampler_imagenet64_ipndm_gits_nfe10_b1.npz'))
    f16 = mode == 'fp16'
    net = EDMDenoiser.from_config('imagenet64', seed=int(z['seed']), use_fp16=f16)
    B, slots = 64, [0, 21, 42, 63]
    g = torch.Generator().manual_seed(991)
    latents = torch.randn(B, 3, 64, 64, generator=g)
    labels = torch.eye(1000)[torch.randint(1000, (B,), generator=g)]
    latents[slots] = torch.from_numpy(z['latents'])
    labels[slots] = torch.from_numpy(z['labels'])
    out = solvers.ipndm_sampler(net, latents.to(dev), class_labels=labels.to(dev), max_order=4, t_steps=torch.from_numpy(z['t_steps']).to(dev),
                                num_steps=11, return_inters=True)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    gold = torch.from_numpy(z['traj'])                       # [11, 1, 3, 64, 64]
    tol = 1e-2 if f16 else 5e-4                              # of EACH step's own scale; the final image explicitly
    worst = [0.0] * gold.shape[0]
    for s in slots:
        errs = per_step_rel(out[:, s:s + 1].cpu(), gold)
        worst = [max(a, b) for a, b in zip(worst, errs)]
    record(f'config3_imagenet64_ipndm4_gits_b64_{mode}', per_step=worst, final=worst[-1], step_scales=step_scales(gold), bound=tol)
    assert max(worst) < tol and worst[-1] < tol, (mode, worst)
    ids = _conv_kernel_ids(net, B, B)
    if f16:
        assert ids.get(2562, 0) + ids.get(2566, 0) >= 60, ids       # fp16-operand 3x3 kernels (2562: 256 x 128 tiles, 2566: 256 x 256 tiles)
    else:
        assert ids.get(2568, 0) + ids.get(2565, 0) + ids.get(256, 0) + ids.get(128, 0) + ids.get(1284, 0) >= 60, ids     # the LDS-halo fp32 family
        assert ids.get(2568, 0) >= 30, ids                   # the 192 / 384-channel layers at 64x64 / 32x32: 256 x 192 tiles
    # one evaluation at this batch against the real reference's net_imagenet64.npz
    zn = np.load(os.path.join(G, 'net_imagenet64.npz'))
    netn = net if int(zn['seed']) == int(z['seed']) else EDMDenoiser.from_config('imagenet64', seed=int(zn['seed']), use_fp16=f16)
    sig = torch.full((B,), 0.7)
    x = torch.randn(B, 3, 64, 64, generator=g) * 0.7
    x[slots], sig[slots], labels[slots] = torch.from_numpy(zn['x']), torch.from_numpy(zn['sigma']), torch.from_numpy(zn['labels'])
    o = netn(x.to(dev), sig.to(dev), class_labels=labels.to(dev)).cpu()
    for s in slots:
        e = _rel(o[s:s + 1], torch.from_numpy(zn['out_vec']))
        record(f'net_imagenet64_b64_{mode}', **{f'slot{s}': e}, bound=(5e-3 if f16 else 2e-4))
        assert e < (5e-3 if f16 else 2e-4), (mode, s, e)


@pytest.mark.parametrize('mode', ['fp32', 'fp16'])
def test_ffhq64_headline_solver_at_the_benchmark_batch_b128(mode, dev):
    """`bench.py --config ffhq --batch 128 [--dtype fp16]`: the real reference's DPM-Solver++(2M) logSNR NFE = 10 call on the full-size
    FFHQ-64 SongUNet (B = 2, oracle/gen_golden.py --part fullffhq) scattered over a 128-image call (each golden latent in two slots).
    fp32: 5e-4; use_fp16: 1e-2 over the 10-evaluation trajectory against the FP32 reference."""
    from diff_sampler_amd import solvers
    from diff_sampler_amd.engine import EDMDenoiser
    z = np.load(os.path.join(G, 'sampler_ffhq_dpmpp2m_nfe10_b2.npz'))
    f16 = mode == 'fp16'
    net = EDMDenoiser.from_config('ffhq', seed=int(z['seed']), use_fp16=f16)
    B, slots = 128, [0, 77, 50, 127]                        # golden latent 0 in slots 0 and 50, latent 1 in slots 77 and 127
    latents = torch.randn(B, 3, 64, 64, generator=torch.Generator().manual_seed(992))
    gl = torch.from_numpy(z['latents'])
    latents[slots] = torch.cat([gl, gl])
    out = solvers.dpm_pp_sampler(net, latents.to(dev), num_steps=11, sigma_min=0.002, sigma_max=80., schedule_type='logsnr', schedule_rho=7,
                                 max_order=2, predict_x0=True, lower_order_final=True).cpu()
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    gold = torch.from_numpy(z['out'])
    tol = 1e-2 if f16 else 5e-4                              # final images against their own scale
    e = _rel(out[slots], torch.cat([gold, gold]))
    record(f'ffhq64_dpmpp2m_nfe10_b128_{mode}', final=e, final_scale=float(gold.abs().max()), bound=tol)
    assert e < tol, (mode, e)
    ids = _conv_kernel_ids(net, B, 1)
    if f16:
        assert ids.get(2562, 0) + ids.get(2566, 0) >= 60, ids
    else:
        assert ids.get(2565, 0) >= 20 and ids.get(256, 0) >= 10, ids       # 256-channel layers on 256 x 256 tiles, 128-channel layers on 256 x 128
