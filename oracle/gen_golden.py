"""ORACLE tooling: generate tests/golden/*.npz by running the REAL reference (``/root/reference``) on CPU.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python oracle/gen_golden.py            # all parts, each in its own interpreter
    python oracle/gen_golden.py --part net # one part

Every fixture stores the inputs and the reference's outputs.  Network weights are NOT stored: they are
regenerated on any machine from ``arch.init_params(spec, seed)`` (CPU generator, deterministic), and this
script loads exactly those tensors into the reference ``EDMPrecond`` before running it.

Parts
  net       reference EDMPrecond.forward on tiny SongUNet / class-cond SongUNet / DhariwalUNet and the full
            CIFAR-10 SongUNet                      (diff-solvers-main/models/networks_edm.py)
  sched     get_schedule (4 kinds incl. GITS dp_list), DEIS coefficient lists, dynamic thresholding
            (diff-solvers-main/solver_utils.py, gits-main/solver_utils.py)
  samplers  every sampler of diff-solvers-main/solvers.py on a tiny net, with trajectories; config-1
            (Euler, NFE=10, B=8, CIFAR-10 net) final images
  amed      amed-solver-main/solvers_amed.py (AMED-Solver + plugins) with a random-init AMED_predictor
  gits      gits-main/gits_utils.py get_dp_list (cost matrix + dynamic programme) on a tiny net
  ldm       reference CFGPrecond + ldm UNetModel (tiny configs and full SD-1.5 size): denoiser outputs with
            classifier-free guidance, sigma / sigma_inv / discrete schedule probes, sampler trajectories
  full      full-size pins of the BENCHMARKED configurations (round 2): CIFAR-10 net, DPM-Solver++(2M) logSNR NFE=10
            at B=64 (the bench.py headline sampler; final images); one full-size evaluation each of the FFHQ-64
            SongUNet and the ImageNet-64 DhariwalUNet (B=1, labels); one full-size SD-1.5 config-5 trajectory
            (DPM-Solver++(2M) eps-prediction, discrete rho=1, num_steps=6, CFG 7.5, B=1)
  fullsolv  every solver family north_star names on the FULL-size CIFAR-10 net at NFE = 10, B = 4 (Heun, DPM-Solver-2, iPNDM on a
            polynomial and on the GITS-form schedule, iPNDM_v with AFS, DEIS tAB3 on time_uniform, DPM-Solver++(3M) and (2M, eps form),
            UniPC bh2): final images
  fullgits  gits-main get_dp_list on the FULL-size CIFAR-10 net (21-step iPNDM-4 teacher, 6-step student, 'dev' metric, 8 warm-up latents)
  full3     BASELINE config 3 at full size through the reference sampler: ImageNet-64 DhariwalUNet (295.9M params, one-hot labels),
            ipndm_sampler max_order=4 on the 11-point GITS-form schedule literal (t_steps), NFE=10, B=1: the whole trajectory
  full5b    BASELINE config 5 at full size for TWO latents (round 5): the reference's dpm_pp_sampler (eps form, discrete rho=1, num_steps=6,
            CFG 7.5) on latents / conditions the GPU tests scatter over the BENCHMARK batch of 16 latents (bench.py --config sd15 --batch 16)
  ldmw16    the full-size SD-1.5 evaluation of `ldm` with every U-Net weight rounded through fp16 (what a public fp16 .ckpt holds): the
            expectation of the checkpoint loader test (round 6)
  full4     BASELINE config 4 at full size through the reference sampler: FFHQ-64 SongUNet (61.8M params) + AMED_predictor
            (num_steps=4, afs=True, time_uniform rho=1, scale_dir=0.01, scale_time=0; amed-solver-main/launch.sh:21-24), 5 NFE, B=2
"""
import argparse
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
OUT = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, ROOT)
from oracle import cases  # noqa: E402
from oracle.cases import (FULL_SOLVER_CASES, FULL_SOLVER_TSTEPS, GITS_TSTEPS, SAMPLER_CASES, AMED_CASES, GITS_CASES, GITS_COMMON, GITS_FULL_CASE, make_inputs as _inputs,  # noqa: E402
                          amed_predictor_params, gits_warmup_latents)

def _ref_net(name, seed):
    import diff_sampler_amd.arch as arch
    from models.networks_edm import EDMPrecond
    kw = dict(arch.NAMED_CONFIGS[name])
    spec = arch.edm_precond_spec(**kw)
    params = arch.init_params(spec, seed=seed)
    net = EDMPrecond(**kw).eval()
    missing, unexpected = net.load_state_dict(params, strict=False)
    assert not unexpected and all('resample_filter' in m for m in missing)
    return net, kw


def part_net():
    sys.path.insert(0, os.path.join(REF, 'diff-solvers-main'))
    for name, B, seed in [('tiny_song', 3, 11), ('tiny_song_cond', 3, 12), ('tiny_adm', 3, 13), ('tiny_song_amed', 2, 14),
                          ('cifar10', 2, 15)]:
        net, kw = _ref_net(name, seed)
        x, lab = _inputs(kw, B, seed + 100)
        sig = torch.tensor([80.0, 1.7, 0.05][:B])
        x = x * sig.reshape(-1, 1, 1, 1)
        with torch.no_grad():
            out_vec = net(x, sig, class_labels=lab)                 # per-sample sigma
            out_sc = net(x, torch.tensor(0.6), class_labels=lab)    # 0-dim sigma (the samplers' usual call)
        np.savez(os.path.join(OUT, f'net_{name}.npz'), config=name, seed=seed, x=x.numpy(), sigma=sig.numpy(),
                 labels=(lab.numpy() if lab is not None else np.zeros(0, np.float32)),
                 out_vec=out_vec.numpy(), out_scalar=out_sc.numpy())
        print('net', name, float(out_vec.abs().max()))


def part_sched():
    sys.path.insert(0, os.path.join(REF, 'gits-main'))
    import solver_utils as su
    d = {}
    for kind, rho in [('polynomial', 7), ('logsnr', 7), ('time_uniform', 2), ('time_uniform', 1)]:
        for n in [2, 6, 7, 11, 36]:
            d[f'{kind}_rho{rho}_n{n}'] = su.get_schedule(n, 0.002, 80., schedule_type=kind, schedule_rho=rho).numpy()
    dp = [0, 12, 25, 37, 48, 55, 60]
    d['gits_poly7_n61_dp'] = su.get_schedule(61, 0.002, 80., schedule_type='polynomial', schedule_rho=7, dp_list=dp).numpy()
    d['gits_dp_list'] = np.array(dp)
    np.savez(os.path.join(OUT, 'schedule.npz'), **d)

    d = {}
    for tag, ts in [('tu2_n7', su.get_schedule(7, 0.002, 80., schedule_type='time_uniform', schedule_rho=2)),
                    ('poly7_n11', su.get_schedule(11, 0.002, 80., schedule_type='polynomial', schedule_rho=7)),
                    ('gits', torch.tensor(GITS_TSTEPS))]:
        d[f'{tag}_t'] = ts.numpy()
        for mo in [2, 3, 4]:
            C = su.get_deis_coeff_list(ts, mo, deis_mode='tab')
            for i, row in enumerate(C):
                d[f'{tag}_tab{mo}_{i}'] = np.array([float(c) for c in row], dtype=np.float32)
        C = su.get_deis_coeff_list(ts, 4, deis_mode='rhoab')
        for i, row in enumerate(C):
            d[f'{tag}_rhoab4_{i}'] = np.array([float(c) for c in row], dtype=np.float32)
    np.savez(os.path.join(OUT, 'deis.npz'), **d)

    g = torch.Generator().manual_seed(5)
    d = {}
    for tag, shape, amp in [('c32', (4, 3, 32, 32), 1.5), ('c64', (2, 3, 64, 64), 0.7), ('sd', (2, 4, 64, 64), 3.0), ('small', (3, 3, 8, 8), 1.0)]:
        x = torch.randn(shape, generator=g) * amp
        d[f'{tag}_x'] = x.numpy()
        d[f'{tag}_y'] = su.dynamic_thresholding_fn(x).numpy()
    np.savez(os.path.join(OUT, 'threshold.npz'), **d)
    print('sched ok')


def part_samplers():
    sys.path.insert(0, os.path.join(REF, 'diff-solvers-main'))
    import solvers
    import solver_utils as su
    for netname, seed in [('tiny_song', 21), ('tiny_song_cond', 22)]:
        net, kw = _ref_net(netname, seed)
        latents, lab = _inputs(kw, 2, seed + 100)
        d = dict(latents=latents.numpy(), labels=(lab.numpy() if lab is not None else np.zeros(0, np.float32)), seed=seed, config=netname)
        for tag, fn, kind, rho, n, extra in SAMPLER_CASES:
            if netname == 'tiny_song_cond' and tag not in ('euler', 'ipndm4', 'dpmpp2m', 'heun', 'unipc3_bh2'):
                continue
            extra = dict(extra)
            if kind is None:
                ts = torch.tensor(GITS_TSTEPS)
            else:
                ts = su.get_schedule(n, 0.002, 80., schedule_type=kind, schedule_rho=rho)
            if fn == 'deis_sampler':
                extra['coeff_list'] = su.get_deis_coeff_list(ts, extra['max_order'], deis_mode=extra.pop('deis_mode'))
            want_eps = fn != 'unipc_sampler'
            res = getattr(solvers, fn)(net, latents, class_labels=lab, num_steps=n, t_steps=ts, return_inters=True,
                                        return_eps=want_eps, **extra)
            if want_eps:
                inters, eps = res
                d[f'{tag}_eps'] = eps.numpy()
            else:
                inters = res
            d[f'{tag}_t'] = ts.numpy()
            d[f'{tag}_inters'] = inters.numpy()
            print(netname, tag, tuple(inters.shape), float(inters[-1].abs().max()))
        np.savez_compressed(os.path.join(OUT, f'sampler_{netname}.npz'), **d)

    # BASELINE config 1: EDM CIFAR-10, Euler NFE=10, batch 8.
    net, kw = _ref_net('cifar10', 31)
    latents = torch.randn(8, 3, 32, 32, generator=torch.Generator().manual_seed(0))
    ts = su.get_schedule(11, 0.002, 80., schedule_type='polynomial', schedule_rho=7)
    out = solvers.euler_sampler(net, latents, num_steps=11, t_steps=ts)
    out2 = solvers.dpm_pp_sampler(net, latents[:2], num_steps=6, max_order=2, schedule_type='logsnr')
    np.savez_compressed(os.path.join(OUT, 'sampler_cifar10_config1.npz'), seed=31, latents=latents.numpy(), t=ts.numpy(),
                        euler_nfe10=out.numpy(), dpmpp2m_nfe5_b2=out2.numpy())
    print('config1', float(out.abs().max()))


def part_amed():
    sys.path.insert(0, os.path.join(REF, 'amed-solver-main'))
    import solvers_amed
    from training.networks import AMED_predictor
    from models.networks_edm import EDMPrecond
    import diff_sampler_amd.arch as arch
    for netname, seed in [('tiny_song_amed', 41), ('tiny_song_amed_cond', 42)]:
        kw = dict(arch.NAMED_CONFIGS[netname])
        spec = arch.edm_precond_spec(**kw)
        net = EDMPrecond(**kw).eval()
        net.load_state_dict(arch.init_params(spec, seed=seed), strict=False)
        latents, lab = _inputs(kw, 2, seed + 100)
        d = dict(latents=latents.numpy(), labels=(lab.numpy() if lab is not None else np.zeros(0, np.float32)), seed=seed, config=netname)
        for tag, stu, n, kind, rho, afs, pk, sk in AMED_CASES:
            if netname.endswith('cond') and tag not in ('amed', 'amed_ipndm'):
                continue
            pred = AMED_predictor(num_steps=n, sampler_stu=stu, sampler_tea='heun', M=1, schedule_type=kind, schedule_rho=rho,
                                  afs=afs, **pk, **{k: v for k, v in sk.items() if k in ('max_order', 'predict_x0', 'lower_order_final')}).eval()
            pp = amed_predictor_params(1000 + len(tag), pk['scale_dir'], pk['scale_time'])
            pred.load_state_dict(pp, strict=True)
            fn = {'amed': solvers_amed.amed_sampler, 'euler': solvers_amed.euler_sampler, 'ipndm': solvers_amed.ipndm_sampler,
                  'dpm': solvers_amed.dpm_2_sampler, 'dpmpp': solvers_amed.dpm_pp_sampler}[stu]
            with torch.no_grad():
                inters = fn(net, latents, class_labels=lab, num_steps=n, sigma_min=0.002, sigma_max=80., schedule_type=kind,
                            schedule_rho=rho, afs=afs, return_inters=True, AMED_predictor=pred, **sk)
            d[f'{tag}_inters'] = inters.numpy()
            d[f'{tag}_pseed'] = 1000 + len(tag)
            print(netname, tag, tuple(inters.shape), float(inters[-1].abs().max()))
        np.savez_compressed(os.path.join(OUT, f'sampler_{netname}.npz'), **d)


def part_gits():
    """gits-main/gits_utils.py: cal_deviation, dp and the full get_dp_list on a tiny net (single-rank gloo group)."""
    sys.path.insert(0, os.path.join(REF, 'gits-main'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29571', RANK='0', WORLD_SIZE='1')
    torch.distributed.init_process_group('gloo', rank=0, world_size=1)
    import gits_utils
    net, kw = _ref_net('tiny_song', 51)
    d = dict(seed=51, config='tiny_song')
    g = torch.Generator().manual_seed(9)
    traj = torch.randn(7, 3, 3, 16, 16, generator=g).cumsum(0)
    d['dev_traj'] = traj.numpy()
    d['dev_out'] = gits_utils.cal_deviation(traj, 3, 16, bs=3).numpy()
    cm = torch.rand(12, 12, generator=g).numpy().astype(np.float64) + np.triu(np.ones((12, 12)), 1) * 0.1
    d['dp_cost'] = cm
    for ns, coeff in [(4, 1.0), (6, 1.15), (8, 0.85)]:
        d[f'dp_{ns}_{coeff}'] = np.array(gits_utils.dp(cm, ns, 12, coeff))
    for tag, gk in GITS_CASES:
        kwargs = dict(GITS_COMMON); kwargs.update(gk)
        kwargs['solver_kwargs_seed'] = 1000 + len(tag)
        torch.manual_seed(kwargs['solver_kwargs_seed'])
        dp_list = gits_utils.get_dp_list(net, torch.device('cpu'), **kwargs)
        d[f'{tag}_dp_list'] = np.array(dp_list)
        print('gits', tag, dp_list)
    np.savez_compressed(os.path.join(OUT, 'gits.npz'), **d)
    torch.distributed.destroy_process_group()


def _ref_cfg_net(name, seed, guidance_rate=7.5):
    """The reference CFGPrecond (networks_edm.py:630-762) around the reference UNetModel (openaimodel.py:413-742), with
    weights from ldm_arch.init_ldm_params.  LatentDiffusion itself needs pytorch_lightning/CLIP/VAE and is replaced by a
    shim exposing exactly what CFGPrecond uses: ``alphas_cumprod`` and ``apply_model`` (= DiffusionWrapper 'crossattn',
    ddpm.py:1409-1411)."""
    import types
    if 'omegaconf' not in sys.modules:
        try:
            import omegaconf  # noqa: F401
        except ImportError:
            m, lc = types.ModuleType('omegaconf'), types.ModuleType('omegaconf.listconfig')
            lc.ListConfig = type('ListConfig', (list,), {})
            m.listconfig = lc
            sys.modules['omegaconf'], sys.modules['omegaconf.listconfig'] = m, lc
    import diff_sampler_amd.ldm_arch as la
    from models.ldm.modules.diffusionmodules.openaimodel import UNetModel
    from models.networks_edm import CFGPrecond
    kw = dict(la.NAMED_LDM_CONFIGS[name])
    spec = la.ldm_unet_spec(**kw)
    unet = UNetModel(image_size=32, in_channels=kw['in_channels'], out_channels=kw['out_channels'], model_channels=kw['model_channels'],
                     attention_resolutions=list(kw['attention_resolutions']), num_res_blocks=kw['num_res_blocks'],
                     channel_mult=list(kw['channel_mult']), num_heads=kw['num_heads'], use_spatial_transformer=True,
                     transformer_depth=1, context_dim=kw['context_dim'], use_checkpoint=False, legacy=False).eval()
    unet.load_state_dict(la.init_ldm_params(spec, seed=seed), strict=True)

    class Shim(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.unet = unet
            self.register_buffer('alphas_cumprod', la.alphas_cumprod(spec))

        def apply_model(self, x, t, cond):
            return self.unet(x, t, context=cond)

    net = CFGPrecond(Shim(), img_resolution=kw['img_resolution'], img_channels=kw['in_channels'], guidance_rate=guidance_rate,
                     guidance_type='classifier-free', label_dim=True).eval()
    return net, unet, kw, spec


def part_gitsldm():
    """gits-main/gits_utils.py:get_dp_list with model_source == 'ldm' on the tiny latent-diffusion net (real reference code path:
    autocast + ema_scope + get_learned_conditioning), text encoder and EMA scope shimmed."""
    import contextlib
    sys.path.insert(0, os.path.join(REF, 'gits-main'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29573', RANK='0', WORLD_SIZE='1')
    torch.distributed.init_process_group('gloo', rank=0, world_size=1)
    import gits_utils
    torch.set_grad_enabled(False)
    net, unet, kw, spec = _ref_cfg_net('tiny_ldm', 52)
    tag, gk = cases.GITS_LDM_CASE
    kwargs = dict(cases.GITS_COMMON); kwargs.update(gk)
    kwargs['sigma_min'], kwargs['sigma_max'] = net.sigma_min, net.sigma_max
    rounds = kwargs['num_warmup'] // (kwargs['max_batch_size'] + 1) + 1
    conds = cases.gits_ldm_conditions(77, rounds, kwargs['max_batch_size'], kw['context_dim'])
    calls = []

    def get_learned_conditioning(prompts):
        r, which = divmod(len(calls), 2)
        calls.append(list(prompts))
        return conds[r][2] if which == 0 else conds[r][1]       # unconditional first, then the prompts (gits_utils.py:97-101)
    net.model.get_learned_conditioning = get_learned_conditioning
    net.model.ema_scope = contextlib.nullcontext
    torch.manual_seed(4321)
    dp_list = gits_utils.get_dp_list(net, torch.device('cpu'), **kwargs)
    assert len(calls) == 2 * rounds and calls[0] == [''] * kwargs['max_batch_size'] and calls[1] == ['a photo'] * kwargs['max_batch_size']
    np.savez_compressed(os.path.join(OUT, 'gits_ldm.npz'), seed=52, warmup_seed=4321, cond_seed=77, dp_list=np.array(dp_list),
                        sigma_min=net.sigma_min, sigma_max=net.sigma_max)
    print('gitsldm', tag, dp_list, flush=True)
    torch.distributed.destroy_process_group()


def part_ldm():
    sys.path.insert(0, os.path.join(REF, 'diff-solvers-main'))
    import solvers
    import solver_utils
    for name, B, seed in [('tiny_ldm', 2, 21), ('tiny_ldm_1res', 3, 22), ('sd15', 1, 23)]:
        net, unet, kw, spec = _ref_cfg_net(name, seed)
        g = torch.Generator().manual_seed(seed + 100)
        R, Cc, ctx = kw['img_resolution'], kw['in_channels'], kw['context_dim']
        sig = torch.tensor([11.0, 0.7, 2.3][:B])
        x = torch.randn(B, Cc, R, R, generator=g) * sig.reshape(-1, 1, 1, 1)
        cond = torch.randn(B, 77, ctx, generator=g)
        uncond = torch.randn(B, 77, ctx, generator=g)
        d = dict(config=name, seed=seed, x=x.numpy(), sigma=sig.numpy(), cond=cond.numpy(), uncond=uncond.numpy(),
                 sigma_min=net.sigma_min, sigma_max=net.sigma_max)
        with torch.no_grad():
            d['out_vec'] = net(x, sig, condition=cond, unconditional_condition=uncond).numpy()
            if name != 'sd15':
                d['out_scalar'] = net(x, torch.tensor(1.9), condition=cond, unconditional_condition=uncond).numpy()
                d['out_nocfg'] = net(x, torch.tensor(1.9), condition=cond, unconditional_condition=None).numpy()
                tt = torch.tensor([3.0, 998.0, 421.7][:B])
                d['unet_t'] = tt.numpy()
                d['unet_out'] = unet(x, tt, context=cond).numpy()
            probe = torch.tensor([0.03, 0.1, 0.5, 1.0, 2.5, 7.0, 14.6], dtype=torch.float32)
            d['probe_sigma'] = probe.numpy()
            d['probe_sigma_inv'] = net.sigma_inv(probe).numpy()
            tpr = torch.tensor([0.001, 0.0105, 0.2, 0.5, 0.77, 1.0], dtype=torch.float32)
            d['probe_t'] = tpr.numpy()
            d['probe_sigma_of_t'] = net.sigma(tpr).numpy()
            d['sched_discrete_6'] = solver_utils.get_schedule(6, net.sigma_min, net.sigma_max, device='cpu', schedule_type='discrete',
                                                              schedule_rho=1, net=net).numpy()
            if name != 'sd15':
                lat = torch.randn(B, Cc, R, R, generator=g)
                d['latents'] = lat.numpy()
                for tag, fn, kws in [
                    ('dpmpp2m_eps', solvers.dpm_pp_sampler, dict(max_order=2, predict_x0=False, lower_order_final=True)),
                    ('dpmpp2m_x0', solvers.dpm_pp_sampler, dict(max_order=2, predict_x0=True, lower_order_final=True)),
                    ('euler', solvers.euler_sampler, {}),
                    ('ipndm3', solvers.ipndm_sampler, dict(max_order=3)),
                ]:
                    tr = fn(net, lat, condition=cond, unconditional_condition=uncond, num_steps=6, sigma_min=net.sigma_min,
                            sigma_max=net.sigma_max, schedule_type='discrete', schedule_rho=1, return_inters=True, **kws)
                    d[f'traj_{tag}'] = tr.numpy()
        np.savez_compressed(os.path.join(OUT, f'ldm_{name}.npz'), **d)
        print('ldm', name, float(np.abs(d['out_vec']).max()), net.sigma_min, net.sigma_max)


def part_full():
    sys.path.insert(0, os.path.join(REF, 'diff-solvers-main'))
    import solvers
    torch.set_grad_enabled(False)
    # (1) the headline sampler of bench.py at a batch whose 32x32 layers dispatch to the 256-pixel-tile kernel
    net, kw = _ref_net('cifar10', 31)
    latents = torch.randn(64, 3, 32, 32, generator=torch.Generator().manual_seed(1))
    out = solvers.dpm_pp_sampler(net, latents, num_steps=11, sigma_min=0.002, sigma_max=80., schedule_type='logsnr', schedule_rho=7,
                                 max_order=2, predict_x0=True, lower_order_final=True)
    np.savez_compressed(os.path.join(OUT, 'sampler_cifar10_dpmpp2m_nfe10_b64.npz'), seed=31, latent_seed=1, out=out.numpy())
    print('full cifar10 dpmpp2m nfe10 b64', float(out.abs().max()), flush=True)
    del net
    # (2) one full-size evaluation of the other two EDM nets of BASELINE.json (configs 3 and 4)
    for name, seed in [('ffhq', 61), ('imagenet64', 62)]:
        net, kw = _ref_net(name, seed)
        x, lab = _inputs(kw, 1, seed + 100)
        sig = torch.tensor([1.3])
        x = x * sig.reshape(-1, 1, 1, 1)
        out = net(x, sig, class_labels=lab)
        np.savez_compressed(os.path.join(OUT, f'net_{name}.npz'), config=name, seed=seed, x=x.numpy(), sigma=sig.numpy(),
                            labels=(lab.numpy() if lab is not None else np.zeros(0, np.float32)), out_vec=out.numpy())
        print('full net', name, float(out.abs().max()), flush=True)
        del net
    # (3) BASELINE config 5 at full size: one latent through the whole sampler call
    net, unet, kw, spec = _ref_cfg_net('sd15', 23)
    g = torch.Generator().manual_seed(223)
    lat = torch.randn(1, 4, 64, 64, generator=g)
    cond = torch.randn(1, 77, 768, generator=g)
    uncond = torch.randn(1, 77, 768, generator=g)
    tr = solvers.dpm_pp_sampler(net, lat, condition=cond, unconditional_condition=uncond, num_steps=6, sigma_min=net.sigma_min,
                                sigma_max=net.sigma_max, schedule_type='discrete', schedule_rho=1, return_inters=True,
                                max_order=2, predict_x0=False, lower_order_final=True)
    np.savez_compressed(os.path.join(OUT, 'ldm_sd15_traj.npz'), seed=23, input_seed=223, latents=lat.numpy(), cond=cond.numpy(),
                        uncond=uncond.numpy(), traj=tr.numpy())
    print('full sd15 config-5 trajectory', tuple(tr.shape), float(tr[-1].abs().max()), flush=True)


def part_full5b():
    """BASELINE config 5 through the REAL reference at B = 2 (4 U-Net images per evaluation under guidance): the whole trajectory.  The GPU
    test scatters the two latents (with their conditions) over a 16-latent call -- the batch `bench.py --config sd15` times."""
    sys.path.insert(0, os.path.join(REF, 'diff-solvers-main'))
    import solvers
    torch.set_grad_enabled(False)
    net, unet, kw, spec = _ref_cfg_net('sd15', 23)
    g = torch.Generator().manual_seed(523)
    lat = torch.randn(2, 4, 64, 64, generator=g)
    cond = torch.randn(2, 77, 768, generator=g)
    uncond = torch.randn(2, 77, 768, generator=g)
    tr = solvers.dpm_pp_sampler(net, lat, condition=cond, unconditional_condition=uncond, num_steps=6, sigma_min=net.sigma_min,
                                sigma_max=net.sigma_max, schedule_type='discrete', schedule_rho=1, return_inters=True,
                                max_order=2, predict_x0=False, lower_order_final=True)
    np.savez_compressed(os.path.join(OUT, 'ldm_sd15_traj_b2.npz'), seed=23, input_seed=523, latents=lat.numpy(), cond=cond.numpy(),
                        uncond=uncond.numpy(), traj=tr.numpy())
    print('full5b sd15 config-5 trajectory, two latents', tuple(tr.shape), float(tr[-1].abs().max()), flush=True)


CONFIG3_TSTEPS = [80.0, 31.78, 14.51, 7.42, 3.88, 2.05, 1.06, 0.5666, 0.2531, 0.0631, 0.002]     # 11 points => NFE 10 (GITS form)


def part_full3():
    sys.path.insert(0, os.path.join(REF, 'diff-solvers-main'))
    import solvers
    torch.set_grad_enabled(False)
    net, kw = _ref_net('imagenet64', 31)
    g = torch.Generator().manual_seed(32)
    latents = torch.randn(1, 3, 64, 64, generator=g)
    labels = torch.eye(1000)[torch.randint(1000, (1,), generator=g)]
    t_steps = torch.tensor(CONFIG3_TSTEPS)
    tr = solvers.ipndm_sampler(net, latents, class_labels=labels, num_steps=11, sigma_min=0.002, sigma_max=80., max_order=4,
                               t_steps=t_steps, return_inters=True)
    np.savez_compressed(os.path.join(OUT, 'sampler_imagenet64_ipndm_gits_nfe10_b1.npz'), seed=31, input_seed=32, latents=latents.numpy(),
                        labels=labels.numpy(), t_steps=t_steps.numpy(), traj=tr.numpy())
    print('full3 imagenet64 ipndm-4 GITS-form schedule', tuple(tr.shape), float(tr[-1].abs().max()), flush=True)


def part_fullffhq():
    """The sampler call `bench.py --config ffhq` times (DPM-Solver++(2M), logSNR schedule, NFE = 10) on the full-size FFHQ-64 SongUNet at
    B = 2 by the REAL reference: the GPU test scatters these two latents over the benchmark batch of 128."""
    sys.path.insert(0, os.path.join(REF, 'diff-solvers-main'))
    import solvers
    torch.set_grad_enabled(False)
    net, kw = _ref_net('ffhq', 61)
    latents = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(7))
    out = solvers.dpm_pp_sampler(net, latents, num_steps=11, sigma_min=0.002, sigma_max=80., schedule_type='logsnr', schedule_rho=7,
                                 max_order=2, predict_x0=True, lower_order_final=True)
    np.savez_compressed(os.path.join(OUT, 'sampler_ffhq_dpmpp2m_nfe10_b2.npz'), seed=61, latent_seed=7, latents=latents.numpy(), out=out.numpy())
    print('fullffhq dpmpp2m nfe10 b2', float(out.abs().max()), flush=True)


def part_full4():
    sys.path.insert(0, os.path.join(REF, 'amed-solver-main'))
    import solvers_amed
    from training.networks import AMED_predictor
    torch.set_grad_enabled(False)
    net, kw = _ref_net('ffhq', 41)
    latents = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(42))
    pk = dict(scale_dir=0.01, scale_time=0)
    pred = AMED_predictor(num_steps=4, sampler_stu='amed', sampler_tea='heun', M=1, schedule_type='time_uniform', schedule_rho=1,
                          afs=True, **pk).eval()
    pred.load_state_dict(amed_predictor_params(43, pk['scale_dir'], pk['scale_time']), strict=True)
    tr = solvers_amed.amed_sampler(net, latents, num_steps=4, sigma_min=0.002, sigma_max=80., schedule_type='time_uniform',
                                   schedule_rho=1, afs=True, return_inters=True, AMED_predictor=pred)
    np.savez_compressed(os.path.join(OUT, 'sampler_ffhq_amed_nfe5_b2.npz'), seed=41, input_seed=42, predictor_seed=43,
                        latents=latents.numpy(), traj=tr.numpy(), **pk)
    print('full4 ffhq AMED-Solver nfe5', tuple(tr.shape), float(tr[-1].abs().max()), flush=True)


def part_fullgits():
    sys.path.insert(0, os.path.join(REF, 'gits-main'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29572', RANK='0', WORLD_SIZE='1')
    torch.distributed.init_process_group('gloo', rank=0, world_size=1)
    import gits_utils
    net, kw = _ref_net('cifar10', 31)
    tag, gk = GITS_FULL_CASE
    kwargs = dict(GITS_COMMON); kwargs.update(gk)
    kwargs['solver_kwargs_seed'] = 2000 + len(tag)
    torch.manual_seed(kwargs['solver_kwargs_seed'])
    with torch.no_grad():
        dp_list = gits_utils.get_dp_list(net, torch.device('cpu'), **kwargs)
    np.savez_compressed(os.path.join(OUT, 'gits_cifar10.npz'), seed=31, warmup_seed=kwargs['solver_kwargs_seed'], dp_list=np.array(dp_list))
    print('fullgits', tag, dp_list, flush=True)
    torch.distributed.destroy_process_group()


def part_fullsolv():
    sys.path.insert(0, os.path.join(REF, 'diff-solvers-main'))
    import solvers
    import solver_utils as su
    torch.set_grad_enabled(False)
    net, kw = _ref_net('cifar10', 31)
    latents = torch.randn(4, 3, 32, 32, generator=torch.Generator().manual_seed(5))
    d = dict(seed=31, latent_seed=5)
    for tag, fn, kind, rho, n, extra in FULL_SOLVER_CASES:
        extra = dict(extra)
        ts = torch.tensor(FULL_SOLVER_TSTEPS) if kind is None else su.get_schedule(n, 0.002, 80., schedule_type=kind, schedule_rho=rho)
        if fn == 'deis_sampler':
            with torch.enable_grad():
                extra['coeff_list'] = su.get_deis_coeff_list(ts, extra['max_order'], deis_mode=extra.pop('deis_mode'))
        out = getattr(solvers, fn)(net, latents, num_steps=n, t_steps=ts, **extra)
        d[f'{tag}_t'] = ts.numpy()
        d[f'{tag}_out'] = out.numpy()
        print('fullsolv', tag, float(out.abs().max()), flush=True)
    np.savez_compressed(os.path.join(OUT, 'sampler_cifar10_solvers_nfe10_b4.npz'), **d)


def part_ldmw16():
    """The SD-checkpoint loader's expectation (round 6): public SD-1.x ``.ckpt`` files hold fp16 tensors under ``model.diffusion_model.`` and the
    loader widens them (``sample.py:create_model``: ``v.float()``).  The REAL reference CFGPrecond + UNetModel at full SD-1.5 size with every weight
    rounded through fp16, on the inputs of ``ldm_sd15.npz``: what an fp32 evaluation of such a checkpoint must return."""
    sys.path.insert(0, os.path.join(REF, 'diff-solvers-main'))
    net, unet, kw, spec = _ref_cfg_net('sd15', 23)
    with torch.no_grad():
        for p in unet.parameters():
            p.copy_(p.half().float())
    z = np.load(os.path.join(OUT, 'ldm_sd15.npz'))
    x, sig = torch.from_numpy(z['x']), torch.from_numpy(z['sigma'])
    cond, uncond = torch.from_numpy(z['cond']), torch.from_numpy(z['uncond'])
    with torch.no_grad():
        out = net(x, sig, condition=cond, unconditional_condition=uncond).numpy()
    rel = float(np.abs(out - z['out_vec']).max() / np.abs(z['out_vec']).max())
    np.savez_compressed(os.path.join(OUT, 'ldm_sd15_w16.npz'), seed=23, out_vec=out, rel_to_fp32_weights=rel)
    print('ldm sd15 with fp16-rounded weights: rel distance from the fp32-weight golden', rel, flush=True)


PARTS = dict(ldmw16=part_ldmw16, full5b=part_full5b, gitsldm=part_gitsldm, fullffhq=part_fullffhq, fullgits=part_fullgits, fullsolv=part_fullsolv, full3=part_full3, full4=part_full4, net=part_net, sched=part_sched, samplers=part_samplers, amed=part_amed, gits=part_gits, ldm=part_ldm, full=part_full)

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--part', default=None, choices=list(PARTS))
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    torch.set_grad_enabled(True)
    if args.part:
        PARTS[args.part]()
    else:
        for p in PARTS:   # separate interpreters: the sub-projects reuse module names (solver_utils, models, ...)
            subprocess.check_call([sys.executable, os.path.abspath(__file__), '--part', p])
