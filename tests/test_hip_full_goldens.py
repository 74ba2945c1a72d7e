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
explicitly.  What every test observed goes to gpurun_out/r6_parity.json (kept copy: profiles/r6_parity.json; round 5: profiles/r5_parity.json; round 4: profiles/r4_parity.json)."""
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


def test_headline_sampler_at_the_largest_swept_batch_b4096_matches_reference(dev):
    """SURVEY 8d config 2 sweeps B over {64, 256, 1024, 4096}; `bench.py`'s `throughput_by_batch` times all four.  4 096 images per call is
    where 32-bit offsets would first overflow (a 32x32 x 256-channel fp32 tensor of that batch is 4.3 GB) and where a 32x32 layer runs 64
    rounds of 256-tile waves: the 64 golden samples of the real reference's run, one per 64-image stride (slot 64 i + i: every residue mod
    64), must come out as in the reference; all other slots carry different latents.  ~180 GB of plan workspaces: skipped (loudly) only if
    the device does not have them free."""
    from diff_sampler_amd import solvers
    from diff_sampler_amd.engine import EDMDenoiser
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info()
    if free < 215e9:
        pytest.skip(f'needs ~200 GB of free device memory for the 4 096-image plan, {free / 1e9:.0f} GB free of {total / 1e9:.0f}')
    z = np.load(os.path.join(G, 'sampler_cifar10_dpmpp2m_nfe10_b64.npz'))
    net = EDMDenoiser.from_config('cifar10', seed=int(z['seed']))
    gold = torch.randn(64, 3, 32, 32, generator=torch.Generator().manual_seed(int(z['latent_seed'])))
    latents = torch.randn(4096, 3, 32, 32, generator=torch.Generator().manual_seed(77))
    slots = torch.arange(64) * 64 + torch.arange(64)
    latents[slots] = gold
    out = solvers.dpm_pp_sampler(net, latents.to(dev), num_steps=11, sigma_min=0.002, sigma_max=80., schedule_type='logsnr', schedule_rho=7,
                                 max_order=2, predict_x0=True, lower_order_final=True)
    torch.cuda.synchronize()
    got = out[slots.to(dev)].cpu()
    fin = bool(torch.isfinite(out).all())
    del out, net
    torch.cuda.empty_cache()
    assert fin
    e = _rel(got, torch.from_numpy(z['out']))
    record('headline_b4096', final_image_rel=e)
    assert e < 5e-4, e


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


@pytest.mark.parametrize('mode', ['fp32', 'fp16'])
def test_config5_sd15_at_the_benchmark_batch_b16(mode, dev):
    """`bench.py --config sd15 --batch 16 [--dtype fp16]` (the SD-1.5 line of the driver's bench): the REAL reference's config-5 trajectory
    for two latents (oracle/gen_golden.py --part full5b: dpm_pp_sampler eps form, discrete rho = 1, num_steps = 6, CFG 7.5 -- sample.py:293-301,
    networks_edm.py:688-762) occupies slots 0 / 9 (latent 0) and 6 / 15 (latent 1) of a 16-latent call whose other slots carry other latents
    and conditions: 32 U-Net images per evaluation, the batch at which the 8x8 stage, the split-K layers, the NB = 4 tiles and the gather
    `Downsample` GEMM run -- none of which a B = 1 call reaches.  fp32: 1e-3 per step, each step against its own scale.  use_fp16 (the
    reference's autocast mode): every step within 1.5 x the fp16-stream ORACLE's own distance from the fp32 golden at that step (the noise
    floor of the mode, stored by oracle/gen_f16_golden.py --traj; absolute ceiling 2e-2) and within 2 x that noise of the oracle's fp16
    trajectory (two independent roundings of the same tensors); the first evaluation's raw U-Net outputs within 2.5e-3 of the fp16 oracle's (same tensors rounded); the plan must run
    EVERY body convolution on the fp16 kernels (fp32_convs == 0), split K somewhere, use a 256-column tile and the gather GEMM."""
    from diff_sampler_amd import _lib, solvers
    from diff_sampler_amd.ldm_engine import CFGDenoiser
    z = np.load(os.path.join(G, 'ldm_sd15_traj_b2.npz'))
    f16 = mode == 'fp16'
    net = CFGDenoiser.from_config('sd15', seed=int(z['seed']), guidance_rate=7.5, use_fp16=f16)
    B, slots = 16, [0, 6, 9, 15]                            # golden latent 0 in slots 0 and 9, latent 1 in slots 6 and 15
    g = torch.Generator().manual_seed(993)
    lat = torch.randn(B, 4, 64, 64, generator=g)
    cond = torch.randn(B, 77, 768, generator=g)
    uncond = torch.randn(B, 77, 768, generator=g)
    gl, gc, gu = (torch.from_numpy(z[k]) for k in ('latents', 'cond', 'uncond'))
    lat[slots], cond[slots], uncond[slots] = torch.cat([gl, gl]), torch.cat([gc, gc]), torch.cat([gu, gu])
    lat, cond, uncond = lat.to(dev), cond.to(dev), uncond.to(dev)
    tr = solvers.dpm_pp_sampler(net, lat, condition=cond, unconditional_condition=uncond, num_steps=6, sigma_min=net.sigma_min,
                                sigma_max=net.sigma_max, schedule_type='discrete', schedule_rho=1, return_inters=True,
                                max_order=2, predict_x0=False, lower_order_final=True)
    torch.cuda.synchronize()
    assert torch.isfinite(tr).all()
    gold = torch.from_numpy(z['traj'])                       # [6, 2, 4, 64, 64]
    gold4 = torch.cat([gold, gold], dim=1)
    errs = per_step_rel(tr[:, slots].cpu(), gold4)
    lib = _lib.load()
    plan = net.engine.plan(2 * B, 1, 77)
    convs = [op.keep[0] for op in plan.ops if op.fn is lib.ds_conv2d_nhwc]
    c33 = [a for a in convs if a.taps == 9]
    ids = [lib.ds_conv_kernel_id(C.byref(a)) for a in c33]
    if not f16:
        record('config5_sd15_dpmpp2m_b16_fp32', per_step=errs, final=errs[-1], step_scales=step_scales(gold), bound=1e-3)
        assert max(errs) < 1e-3 and errs[-1] < 1e-3, errs
        return
    z16 = np.load(os.path.join(G, 'ldm_sd15_traj_b2_f16ops.npz'))
    from _f16_names import ldm_prefixes, ldm_stored_prefixes
    assert sorted(ldm_prefixes(plan)) == [str(v) for v in z16['f16_layers']]                  # the oracle rounded exactly these layers' operands
    assert plan.stream16 and sorted(ldm_stored_prefixes(plan)) == [str(v) for v in z16['f16_stored']]
    # routing at THIS batch: every 3x3 convolution of the body on the fp16-activation kernels (the 4-channel head alone stays fp32, VALU kernel),
    # the three Downsample convolutions on the gather GEMM, 256-column tiles in use
    fp32_body = [i for a, i in zip(c33, ids) if not a.wgt_f16 and i != 2570]
    assert fp32_body == [] and ids.count(2570) == 1, (fp32_body, ids)
    assert ids.count(2571) == 3 and ids.count(2566) >= 45, ids
    noise = [float(v) for v in z16['per_step_rel_vs_fp32_golden']]
    gold16 = torch.from_numpy(z16['traj_f16ops'])
    errs16 = per_step_rel(tr[:, slots].cpu(), torch.cat([gold16, gold16], dim=1))
    # the first evaluation alone: raw U-Net outputs (unconditional / conditional halves of the doubled batch) against the fp16 oracle's
    t0 = solvers.get_schedule(6, net.sigma_min, net.sigma_max, device=dev, schedule_type='discrete', schedule_rho=1, net=net)[0]
    f_rows = net.raw(lat * t0, float(t0), cond, uncond)[0]
    torch.cuda.synchronize()
    eps = f_rows.reshape(2 * B, 64, 64, 4).permute(0, 3, 1, 2).cpu()
    eps_gold = torch.from_numpy(z16['eps0_f16ops'])          # [uncond 0, uncond 1, cond 0, cond 1]
    rows = [0, 6, B + 0, B + 6], [9, 15, B + 9, B + 15]
    e_eps = max(_rel(eps[r], eps_gold) for r in rows)
    record('config5_sd15_dpmpp2m_b16_fp16', per_step_vs_fp32_golden=errs, per_step_vs_fp16_oracle=errs16, oracle_noise_per_step=noise,
           first_evaluation_unet_outputs_vs_fp16_oracle=e_eps, step_scales=step_scales(gold), fp32_body_convs=len(fp32_body),
           conv_kernel_ids={str(k): ids.count(k) for k in sorted(set(ids))})
    assert errs[0] < 1e-6                                    # step 0 is latents * sigma_max: no network yet
    for i in range(1, len(errs)):
        assert errs[i] < min(max(1.5 * noise[i], 2e-3), 2e-2), (i, errs[i], noise[i])
        assert errs16[i] < max(2.0 * noise[i], 2e-3), (i, errs16[i], noise[i])      # two independent fp16 noises (expected ~1.4 x)
    assert e_eps < 2.5e-3, e_eps
    # what only this batch exercises (host mirrors of the launcher's rules, tests/_f16_names.py): split-K on the 8x8 stage, 256-column tiles
    from _f16_names import f16dma_splits, f16dma_tile_widths
    s1 = [a for a in c33 if a.in_f16 and (a.stride or 1) == 1]
    assert sum(1 for a in s1 if a.h == 8 and f16dma_splits(a) > 1) >= 10, [(a.h, a.c0, a.cout, f16dma_splits(a)) for a in s1]
    assert sum(1 for a in s1 if 4 in f16dma_tile_widths(a)) >= 10


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
        assert ids.get(2562, 0) + ids.get(2566, 0) + ids.get(2572, 0) >= 60, ids       # fp16-activation 3x3 kernels (2566; 2572: with the fused input normalisation, the 64x64 one-column-tile layers)
        assert ids.get(2572, 0) >= 8, ids                    # engine.fuse_norm16 = 'auto': the 64x64 layers with 192 output channels normalise their own halo
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
        assert ids.get(2562, 0) + ids.get(2566, 0) + ids.get(2572, 0) >= 60, ids
    else:
        assert ids.get(2565, 0) >= 20 and ids.get(256, 0) >= 10, ids       # 256-channel layers on 256 x 256 tiles, 128-channel layers on 256 x 128
