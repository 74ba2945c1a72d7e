"""A/B of the fp16 normalisation pass (round 6) on whole sampler calls, alternating variants inside ONE process:

    python tools/ab_norm.py [--config imagenet64 sd15] [--calls 3] [--rounds 3]

    r5      the round-5 form: 8-byte pass (ds_norm_args.tune_variant = 1), statistics always a launch of their own (plan.FOLD_FINALIZE off)
    16B     the 16-byte pass (norm_act16_kernel), statistics a launch of their own
    16B+F   plan.FOLD_FINALIZE on: the 16-byte pass computes the statistics itself on images of at most 32 x 32 pixels (no ds_gn_finalize there)
    (the engines' default is 16B)

Prints the launches of each plan (passes, statistics launches), max |variant - r5| (expected 0: same arithmetic) and images/s per round."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from diff_sampler_amd import _lib, plan as plan_mod, solvers  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--config', nargs='*', default=['imagenet64', 'sd15'])
ap.add_argument('--calls', type=int, default=3)
ap.add_argument('--rounds', type=int, default=3)
args = ap.parse_args()
dev = torch.device('cuda:0')
lib = _lib.load()
NFE = 10

for config in args.config:
    if config == 'sd15':
        import diff_sampler_amd.ldm_arch as la
        from diff_sampler_amd.ldm_engine import CFGDenoiser
        spec = la.ldm_unet_spec(**la.NAMED_LDM_CONFIGS['sd15'])
        params = la.init_ldm_params(spec, seed=0)
        mk, is_ldm, batch, solver = (lambda: CFGDenoiser(spec, params, dev, guidance_rate=7.5, use_fp16=True)), True, 16, 'dpmpp'
    else:
        from diff_sampler_amd.engine import EDMDenoiser
        mk, is_ldm, batch, solver = (lambda: EDMDenoiser.from_config(config, seed=0, device=dev, use_fp16=True)), False, 64, 'ipndm'
        spec = mk().spec
    g = torch.Generator(device='cpu').manual_seed(4321)
    lat = torch.randn(batch, spec.in_channels, spec.img_resolution, spec.img_resolution, generator=g).to(dev)
    ldm = (torch.randn(batch, 77, spec.context_dim, generator=g).to(dev), torch.randn(batch, 77, spec.context_dim, generator=g).to(dev)) if is_ldm else None
    nets = []
    for name, fold, variant in (('r5', False, 1), ('16B', False, 0), ('16B+F', True, 0)):
        plan_mod.FOLD_FINALIZE = fold
        net = mk()
        out = bench.sampler_call(solvers, solver, net, lat, NFE, ldm)            # builds the plans
        torch.cuda.synchronize()
        for P in net.engine._plans.values():
            for op in P.ops:
                if op.fn is lib.ds_norm_act:
                    op.keep[0].tune_variant = variant
            P.close()                                                            # the native copy of the argument structs is rebuilt
        out = bench.sampler_call(solvers, solver, net, lat, NFE, ldm).clone()
        torch.cuda.synchronize()
        P = list(net.engine._plans.values())[-1]
        na = sum(1 for op in P.ops if op.fn is lib.ds_norm_act)
        nf = sum(1 for op in P.ops if op.fn is lib.ds_gn_finalize)
        folded = sum(1 for op in P.ops if op.fn is lib.ds_norm_act and op.keep[0].stats0)
        print(f'{config} {name:6s}: {len(P.ops)} launches per evaluation: {na} passes ({folded} with their own statistics), {nf} ds_gn_finalize', flush=True)
        nets.append((name, net, out))
    plan_mod.FOLD_FINALIZE = False
    for name, _, out in nets[1:]:
        print(f'{config}: max |{name} - r5| = {float((out - nets[0][2]).abs().max()):.3e}', flush=True)
    for rnd in range(args.rounds):
        for name, net, _ in nets:
            bench.sampler_call(solvers, solver, net, lat, NFE, ldm)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.calls):
                bench.sampler_call(solvers, solver, net, lat, NFE, ldm)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.calls
            print(f'{config} round {rnd}: {name:6s} {batch / dt:8.2f} images/s  {dt * 1e3:8.2f} ms per call', flush=True)
    del nets
    torch.cuda.empty_cache()
