"""ORACLE tooling (test infrastructure): case tables and deterministic input/weight generators shared by
``oracle/gen_golden.py`` (which runs the real reference on them) and the tests (which replay them through
the oracle restatement and through the HIP path)."""
import torch

GITS_TSTEPS = [80, 10.9836, 3.8811, 1.584, 0.5666, 0.1698, 0.002]     # diff-solvers-main/launch.sh:122


def make_inputs(kw, B, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, kw['img_channels'], kw['img_resolution'], kw['img_resolution'], generator=g)
    lab = None
    if kw['label_dim']:
        idx = torch.randint(kw['label_dim'], (B,), generator=g)
        lab = torch.eye(kw['label_dim'])[idx]
    return x, lab


SAMPLER_CASES = [
    # tag, solver fn name, schedule kind/rho, num_steps, extra kwargs
    ('euler', 'euler_sampler', 'polynomial', 7, 6, {}),
    ('euler_afs_d0', 'euler_sampler', 'polynomial', 7, 6, dict(afs=True, denoise_to_zero=True)),
    ('heun', 'heun_sampler', 'polynomial', 7, 5, {}),
    ('heun_afs', 'heun_sampler', 'polynomial', 7, 5, dict(afs=True)),
    ('dpm2', 'dpm_2_sampler', 'polynomial', 7, 5, {}),
    ('dpm2_r03_afs', 'dpm_2_sampler', 'logsnr', 7, 5, dict(r=0.3, afs=True)),
    ('ipndm4', 'ipndm_sampler', 'polynomial', 7, 8, dict(max_order=4)),
    ('ipndm3_afs', 'ipndm_sampler', 'polynomial', 7, 6, dict(max_order=3, afs=True)),
    ('ipndm2', 'ipndm_sampler', 'polynomial', 7, 6, dict(max_order=2)),
    ('ipndm4_gits', 'ipndm_sampler', None, None, 7, dict(max_order=4)),
    ('ipndmv4', 'ipndm_v_sampler', 'polynomial', 7, 8, dict(max_order=4)),
    ('ipndmv3_afs', 'ipndm_v_sampler', 'logsnr', 7, 6, dict(max_order=3, afs=True)),
    ('deis_tab3', 'deis_sampler', 'time_uniform', 2, 7, dict(max_order=3, deis_mode='tab')),
    ('deis_tab4_afs', 'deis_sampler', 'time_uniform', 2, 8, dict(max_order=4, deis_mode='tab', afs=True)),
    ('deis_rhoab4', 'deis_sampler', 'polynomial', 7, 8, dict(max_order=4, deis_mode='rhoab')),
    ('dpmpp2m', 'dpm_pp_sampler', 'logsnr', 7, 7, dict(max_order=2, predict_x0=True, lower_order_final=True)),
    ('dpmpp3m', 'dpm_pp_sampler', 'logsnr', 7, 8, dict(max_order=3, predict_x0=True, lower_order_final=True)),
    ('dpmpp3m_nolof_afs', 'dpm_pp_sampler', 'logsnr', 7, 6, dict(max_order=3, predict_x0=True, lower_order_final=False, afs=True)),
    ('dpmpp2m_eps', 'dpm_pp_sampler', 'logsnr', 7, 6, dict(max_order=2, predict_x0=False, lower_order_final=True)),
    ('dpmpp3m_eps_d0', 'dpm_pp_sampler', 'polynomial', 7, 6, dict(max_order=3, predict_x0=False, lower_order_final=True, denoise_to_zero=True)),
    ('unipc3_bh2', 'unipc_sampler', 'logsnr', 7, 7, dict(max_order=3, predict_x0=True, lower_order_final=True, variant='bh2')),
    ('unipc2_bh1_eps', 'unipc_sampler', 'logsnr', 7, 6, dict(max_order=2, predict_x0=False, lower_order_final=True, variant='bh1')),
    ('unipc3_afs', 'unipc_sampler', 'polynomial', 7, 6, dict(max_order=3, predict_x0=True, lower_order_final=False, variant='bh2', afs=True)),
]


# Full-size CIFAR-10 net, NFE = 10 for every solver family north_star names (oracle/gen_golden.py --part fullsolv): tag, solver fn,
# schedule kind (None = the 11-point GITS-form literal), rho, num_steps, extra kwargs
FULL_SOLVER_TSTEPS = [80.0, 31.78, 14.51, 7.42, 3.88, 2.05, 1.06, 0.5666, 0.2531, 0.0631, 0.002]
FULL_SOLVER_CASES = [
    ('heun', 'heun_sampler', 'polynomial', 7, 6, {}),
    ('dpm2', 'dpm_2_sampler', 'polynomial', 7, 6, {}),
    ('ipndm4', 'ipndm_sampler', 'polynomial', 7, 11, dict(max_order=4)),
    ('ipndm4_gits', 'ipndm_sampler', None, None, 11, dict(max_order=4)),
    ('ipndmv3_afs', 'ipndm_v_sampler', 'logsnr', 7, 11, dict(max_order=3, afs=True)),
    ('deis_tab3', 'deis_sampler', 'time_uniform', 2, 11, dict(max_order=3, deis_mode='tab')),
    ('dpmpp3m', 'dpm_pp_sampler', 'logsnr', 7, 11, dict(max_order=3, predict_x0=True, lower_order_final=True)),
    ('dpmpp2m_eps', 'dpm_pp_sampler', 'logsnr', 7, 11, dict(max_order=2, predict_x0=False, lower_order_final=True)),
    ('unipc3_bh2', 'unipc_sampler', 'logsnr', 7, 11, dict(max_order=3, predict_x0=True, lower_order_final=True, variant='bh2')),
]


AMED_CASES = [
    # tag, student sampler, num_steps, schedule, rho, afs, predictor kwargs, sampler kwargs
    ('amed', 'amed', 4, 'time_uniform', 1, True, dict(scale_dir=0.01, scale_time=0), {}),
    ('amed_noafs_st', 'amed', 4, 'polynomial', 7, False, dict(scale_dir=0.01, scale_time=0.05), {}),
    ('amed_euler', 'euler', 4, 'polynomial', 7, False, dict(scale_dir=0.01, scale_time=0), {}),
    ('amed_ipndm', 'ipndm', 4, 'polynomial', 7, True, dict(scale_dir=0.01, scale_time=0), dict(max_order=4)),
    ('amed_ipndm3', 'ipndm', 5, 'polynomial', 7, False, dict(scale_dir=0.05, scale_time=0.05), dict(max_order=3)),
    ('amed_dpm2', 'dpm', 4, 'polynomial', 7, False, dict(scale_dir=0.01, scale_time=0), {}),
    ('amed_dpmpp', 'dpmpp', 4, 'logsnr', 7, True, dict(scale_dir=0.01, scale_time=0), dict(max_order=3, predict_x0=True, lower_order_final=True)),
    ('amed_dpmpp2_eps', 'dpmpp', 4, 'logsnr', 7, False, dict(scale_dir=0.02, scale_time=0), dict(max_order=2, predict_x0=False, lower_order_final=True)),
]


def amed_predictor_params(seed, scale_dir, scale_time):
    """Deterministic AMED_predictor weights keyed like its state_dict (training/networks.py:107-117)."""
    g = torch.Generator().manual_seed(seed)
    shapes = [('map_layer0.weight', (8, 8)), ('map_layer0.bias', (8,)), ('enc_layer0.weight', (128, 64)), ('enc_layer0.bias', (128,)),
              ('enc_layer1.weight', (4, 128)), ('enc_layer1.bias', (4,)), ('fc_r.weight', (1, 20)), ('fc_r.bias', (1,))]
    if scale_dir:
        shapes += [('fc_scale_dir.weight', (1, 20)), ('fc_scale_dir.bias', (1,))]
    if scale_time:
        shapes += [('fc_scale_time.weight', (1, 20)), ('fc_scale_time.bias', (1,))]
    return {k: (torch.randn(s, generator=g) * (0.3 if k.endswith('weight') else 0.1)) for k, s in shapes}




# GITS schedule search (gits-main/gits_utils.py): tag, kwargs of get_dp_list
GITS_CASES = [
    ('dev_ipndm', dict(num_steps=5, num_steps_tea=13, metric='dev', coeff=1.15, afs=False, solver='ipndm', solver_tea='ipndm', max_order=3,
                       schedule_type='polynomial', schedule_rho=7, num_warmup=6, max_batch_size=4)),
    ('l2_euler_afs', dict(num_steps=4, num_steps_tea=10, metric='l2', coeff=1.0, afs=True, solver='euler', solver_tea='heun', max_order=None,
                          schedule_type='logsnr', schedule_rho=7, num_warmup=3, max_batch_size=3)),
    ('l1_dpmpp', dict(num_steps=4, num_steps_tea=9, metric='l1', coeff=0.9, afs=False, solver='dpmpp', solver_tea='dpmpp', max_order=2,
                      schedule_type='time_uniform', schedule_rho=2, num_warmup=2, max_batch_size=2)),
]
# the schedule search on the FULL-size CIFAR-10 net (oracle/gen_golden.py --part fullgits): 21-step iPNDM-4 teacher, 8 warm-up latents
GITS_FULL_CASE = ('dev_ipndm4_cifar10', dict(num_steps=6, num_steps_tea=21, metric='dev', coeff=1.15, afs=False, solver='ipndm', solver_tea='ipndm',
                                              max_order=4, schedule_type='polynomial', schedule_rho=7, num_warmup=8, max_batch_size=8))
GITS_COMMON = dict(dataset_name='cifar10', sigma_min=0.002, sigma_max=80., model_source='edm', prompt=None, guidance_type=None,
                   guidance_rate=None, predict_x0=True, lower_order_final=True, deis_mode='tab', denoise_to_zero=False)


def gits_warmup_latents(seed, rounds, batch, shape):
    """The latents the reference draws (unseeded torch.randn on the default generator, gits_utils.py:85) after
    torch.manual_seed(seed): one [batch, *shape] tensor per accumulation round."""
    torch.manual_seed(seed)
    return [torch.randn([batch] + list(shape)) for _ in range(rounds)]


# GITS schedule search on a latent-diffusion denoiser (gits-main/gits_utils.py:86-108, model_source == 'ldm'): tiny LDM U-Net under
# classifier-free guidance; the text encoder is a shim whose k-th call returns seeded N(0, 1) states (oracle/gen_golden.py --part gitsldm)
GITS_LDM_CASE = ('dev_dpmpp_tiny_ldm', dict(num_steps=4, num_steps_tea=9, metric='dev', coeff=1.1, afs=False, solver='dpmpp', solver_tea='dpmpp',
                                             max_order=2, schedule_type='discrete', schedule_rho=1, num_warmup=3, max_batch_size=2,
                                             dataset_name='ms_coco', model_source='ldm', prompt='a photo', guidance_type='cfg', guidance_rate=7.5,
                                             predict_x0=False))


def gits_ldm_conditions(seed, rounds, batch, ctx_dim):
    """What the shimmed get_learned_conditioning returns, in the reference's call order per round: unconditional first, then the prompts
    (gits_utils.py:97-101)."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(rounds):
        uc = torch.randn(batch, 77, ctx_dim, generator=g)
        c = torch.randn(batch, 77, ctx_dim, generator=g)
        out.append((None, c, uc))
    return out
