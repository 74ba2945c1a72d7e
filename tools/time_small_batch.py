"""Small-batch latency of the headline sampler call (CIFAR-10 net, DPM-Solver++(2M), NFE = 10) by host path:
    python-walk   one ctypes call per launch (Plan.run_python: ~2 000 calls per sampler call)
    native-walk   one ds_plan_run call per network evaluation (the default)
    hipGraph      the whole sampler call replayed from one captured graph (graph.GraphedSampler)
Prints one JSON line {batch: {mode: ms per call}}.  VERDICT r2 item 7: B <= 32 is where the host path shows."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from diff_sampler_amd import solvers, plan as plan_mod
    from diff_sampler_amd.engine import EDMDenoiser
    from diff_sampler_amd.graph import GraphedSampler
    dev = torch.device('cuda')
    kw = dict(num_steps=11, sigma_min=0.002, sigma_max=80., schedule_type='logsnr', schedule_rho=7, max_order=2, predict_x0=True,
              lower_order_final=True)
    out = {}
    for B in (1, 8, 32):
        net = EDMDenoiser.from_config('cifar10', seed=0, device=dev)
        lat = torch.randn(B, 3, 32, 32, device=dev)
        graphed = GraphedSampler(solvers.dpm_pp_sampler, net, tuple(lat.shape), device=dev, **kw)
        native_run = plan_mod.Plan.run

        def timed(step, n=10):
            step(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                step()
            torch.cuda.synchronize()
            return round((time.perf_counter() - t0) / n * 1e3, 3)
        res = {}
        plan_mod.Plan.run = plan_mod.Plan.run_python
        try:
            res['python_walk_ms'] = timed(lambda: solvers.dpm_pp_sampler(net, lat, **kw))
        finally:
            plan_mod.Plan.run = native_run
        res['native_walk_ms'] = timed(lambda: solvers.dpm_pp_sampler(net, lat, **kw))
        res['hipgraph_ms'] = timed(lambda: graphed(lat, clone=False))
        out[f'batch_{B}'] = res
        del graphed, net
    print(json.dumps(out))


if __name__ == '__main__':
    main()
