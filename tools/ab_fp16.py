"""A/B of the fp16-mode routing knobs on whole sampler calls, alternating variants inside ONE process (one GPU box, one session):

    python tools/ab_fp16.py [--config sd15 imagenet64] [--calls 3] [--rounds 2]

    sd15       default | q / k / v fp32 rows at d = 40 (qkv_f16_min_head = 64) | Downsample on the fp32 kernel | no split-K in the fp16 convolution
               | tile shapes from the library's cost model instead of the planner's measurement (plan.AUTOTUNE off)
    imagenet64 default | no split-K in the fp16 convolution (ds_conv_tune.splits = 1 while the plan is built) | cost-model tile shapes

Prints images/s per variant and round (bench.py's sampler calls: SD-1.5 DPM-Solver++(2M) B = 16 under CFG, ImageNet-64 iPNDM-4 B = 64, NFE = 10)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from diff_sampler_amd import _lib, plan as plan_mod, solvers  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--config', nargs='*', default=['sd15', 'imagenet64'])
ap.add_argument('--calls', type=int, default=3)
ap.add_argument('--rounds', type=int, default=2)
args = ap.parse_args()
dev = torch.device('cuda:0')
NFE = 10


def variants(config):
    if config == 'sd15':
        import diff_sampler_amd.ldm_arch as la
        from diff_sampler_amd.ldm_engine import CFGDenoiser
        spec = la.ldm_unet_spec(**la.NAMED_LDM_CONFIGS['sd15'])
        params = la.init_ldm_params(spec, seed=0)
        mk = lambda **kw: CFGDenoiser(spec, params, dev, guidance_rate=7.5, use_fp16=True, **kw)
        return spec, True, 16, 'dpmpp', [('default', mk, {}), ('qkv fp32 rows at d=40', lambda: mk(qkv_f16_min_head=64), {}),
                                         ('Downsample fp32', lambda: mk(f16_downsample=False), {}), ('no split-K', mk, dict(splits=1)),
                                         ('cost-model tiles', mk, 'noauto')]
    from diff_sampler_amd.engine import EDMDenoiser
    mk = lambda: EDMDenoiser.from_config(config, seed=0, device=dev, use_fp16=True)
    net0 = mk()
    return net0.spec, False, 64, 'ipndm', [('default', lambda: net0, {}), ('no split-K', mk, dict(splits=1)), ('cost-model tiles', mk, 'noauto')]


for config in args.config:
    spec, is_ldm, batch, solver, vs = variants(config)
    g = torch.Generator(device='cpu').manual_seed(4321)
    lat = torch.randn(batch, spec.in_channels, spec.img_resolution, spec.img_resolution, generator=g).to(dev)
    ldm = (torch.randn(batch, 77, spec.context_dim, generator=g).to(dev), torch.randn(batch, 77, spec.context_dim, generator=g).to(dev)) if is_ldm else None
    nets = []
    for name, mk, tune in vs:
        net = mk()
        plan_mod.AUTOTUNE = tune != 'noauto'
        with _lib.tuning(**(tune if isinstance(tune, dict) else {})):       # the plan (and its ds_conv_tune words) is built by the first call
            out = bench.sampler_call(solvers, solver, net, lat, NFE, ldm)
        plan_mod.AUTOTUNE = True
        torch.cuda.synchronize()
        assert torch.isfinite(out).all(), name
        nets.append((name, net, out.clone()))
    ref = nets[0][2]
    for name, _, out in nets[1:]:
        print(f'{config}: max |{name} - default| / max |default| = {float((out - ref).abs().max() / ref.abs().max()):.3e}', flush=True)
    for rnd in range(args.rounds):
        for name, net, _ in nets:
            bench.sampler_call(solvers, solver, net, lat, NFE, ldm)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.calls):
                bench.sampler_call(solvers, solver, net, lat, NFE, ldm)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.calls
            print(f'{config} round {rnd}: {name:28s} {batch / dt:8.2f} images/s  {dt * 1e3:8.2f} ms per call', flush=True)
    del nets
    torch.cuda.empty_cache()
print('# measured tile shapes (taps, stride, n, h, w, c0, ec0, cout, act, out_f16, res_f16, res, cbias, bias, stats, tune.splits) -> (nb, nw); ms per candidate')
for (dev_, key), (nb, nw, times) in sorted(plan_mod.tune_report().items()):
    print(key, '->', (nb, nw), ' '.join(f'{c[0]}/{c[1]}:{ms:.3f}' for c, ms in sorted(times.items())), flush=True)
