"""HBM roofline of the fused solver-update kernels at large batch (CIFAR-10 shapes): the Adams-Bashforth family's ds_solver_update
(iPNDM order-4 step, Euler step) and the headline solver's ds_dpmpp_x0_step (DPM-Solver++(2M) data-prediction step: D, dynamic
threshold, combination in one launch; 5 passes = x, F, m1 read, m0, x' written).  `--only-dpmpp B` runs just that kernel at one batch
(a clean row for rocprofv3 --kernel-trace --stats)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sampler_amd import ops  # noqa: E402

C, H = 3, 32
per = C * H * H * 4
only = int(sys.argv[sys.argv.index('--only-dpmpp') + 1]) if '--only-dpmpp' in sys.argv else 0
for B in ([only] if only else [256, 4096, 16384, 65536]):
    dev = 'cuda'
    x = torch.randn(B, C, H, H, device=dev)
    f4 = torch.randn(B, C, H, H, device=dev)        # raw network output, channel-planar (f_ld = 0)
    hist = [torch.randn(B, C, H, H, device=dev) for _ in range(3)]
    xo = torch.empty_like(x); mo = torch.empty_like(x)
    cases = {
        'euler  (x,F -> x\')        3 passes': (ops.make_update_args(x, x, f4, B, C, H, H, xo, raw=True, f_ld=0, hcoefs=[1, -.5, 0, 0, 0, 2., 2., 0]), 3),
        'ipndm4 (x,F,3 hist -> x\',d) 7 passes': (ops.make_update_args(x, x, f4, B, C, H, H, xo, raw=True, f_ld=0, hist=hist, hcoefs=[1, -.5, .1, .2, .3, 2., 2., 0], m_out=mo), 7),
    }
    cases["dpmpp2m (x,F,m1 -> m0,x')   5 passes"] = (ops.make_update_args(x, x, f4, B, C, H, H, xo, raw=True, f_ld=0, hist=hist[:1],
                                                                          hcoefs=[.5, .6, -.1, 0, 0, 2., 2., 0], m_out=mo, store_d=False), 5)
    for name, (a, passes) in cases.items():
        if only and not name.startswith('dpmpp'):
            continue
        launch = (lambda a=a: ops.dpmpp_x0_step(a)) if name.startswith('dpmpp') else (lambda a=a: ops.solver_update(a))
        for _ in range(3):
            launch()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            launch()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        gb = passes * per * B / 1e9
        print(f'B={B:6d} {name}: {ms*1e3:9.1f} us  {gb/ms*1e3:8.1f} GB/s  ({gb/ms*1e3/8000*100:.1f}% of 8 TB/s)')
