"""Eager vs hipGraph replay of one SD-1.5 sampler call (DPM-Solver++(2M), NFE=10 with CFG doubling) at small batch."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sampler_amd import solvers  # noqa: E402
from diff_sampler_amd.graph import GraphedSampler  # noqa: E402
from diff_sampler_amd.ldm_engine import CFGDenoiser  # noqa: E402

net = CFGDenoiser.from_config('sd15', seed=0, guidance_rate=7.5)
for B in [int(b) for b in (sys.argv[1:] or ['1', '2'])]:
    kw = dict(num_steps=6, sigma_min=net.sigma_min, sigma_max=net.sigma_max, schedule_type='discrete', schedule_rho=1, max_order=2,
              predict_x0=False, lower_order_final=True)
    lat = torch.randn(B, 4, 64, 64, device='cuda')
    c, uc = torch.randn(B, 77, 768, device='cuda'), torch.randn(B, 77, 768, device='cuda')
    g = GraphedSampler(solvers.dpm_pp_sampler, net, (B, 4, 64, 64), condition_shape=(B, 77, 768), uncond_shape=(B, 77, 768), **kw)
    for name, fn in (('eager', lambda: solvers.dpm_pp_sampler(net, lat, condition=c, unconditional_condition=uc, **kw)),
                     ('graph', lambda: g(lat, condition=c, unconditional_condition=uc, clone=False))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f'sd15 B={B} {name}: {dt*1e3:.1f} ms per sampler call (5 CFG-doubled evaluations) = {B/dt:.2f} images/s', flush=True)
