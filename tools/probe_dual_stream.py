"""Probe: one denoiser evaluation of 2N images as ONE plan on one stream vs TWO N-image plans on two HIP streams.

Every tile of a convolution launch takes the same time, so all 256 CUs reach their epilogue -- the only HBM-heavy phase of a
tile -- together (profiles/r2_conv_tile_options.txt).  Two half-batch plans on two streams interleave kernels of different layers;
this measures whether the hardware's own interleaving hides those bursts and the launch tails.

    python tools/probe_dual_stream.py [--config cifar10] [--half 128] [--iters 8]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C  # noqa: E402
from diff_sampler_amd.engine import EDMDenoiser  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--config', default='cifar10')
ap.add_argument('--half', type=int, nargs='+', default=[128, 256])
ap.add_argument('--iters', type=int, default=8)
ap.add_argument('--lag', type=int, nargs='+', default=[0, 3, 40], help='stream B is enqueued this many launches behind stream A')
args = ap.parse_args()

net = EDMDenoiser.from_config(args.config, seed=0)
eng = net.engine
spec = net.spec


def fill(plan, B):
    plan.bufs['x'].copy_(torch.randn(B, spec.in_channels, spec.img_resolution, spec.img_resolution, device='cuda'))
    plan.bufs['sigma'].fill_(1.5)


def timed(fn, iters):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


for N in args.half:
    full = eng.plan(2 * N, 1); fill(full, 2 * N)
    pa = eng.plan(N, 1); del eng._plans[(N, 1)]
    pb = eng.plan(N, 1)
    fill(pa, N); fill(pb, N)
    s0 = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    qa, qb = C.c_void_p(sa.cuda_stream), C.c_void_p(sb.cuda_stream)
    t_full = timed(lambda: full.run(s0), args.iters)
    t_half = timed(lambda: pa.run(s0), args.iters)
    print(f'{args.config}: one plan of {2*N}: {t_full:.2f} ms;  one plan of {N}: {t_half:.2f} ms (x2 = {2*t_half:.2f})', flush=True)
    for lag in args.lag:
        def dual():
            ops_a, ops_b = pa.ops, pb.ops
            n = len(ops_a)
            for i in range(n + lag):
                if i < n:
                    o = ops_a[i]; rc = o.fn(*o.args, qa); assert rc == 0, o.name
                if i >= lag:
                    o = ops_b[i - lag]; rc = o.fn(*o.args, qb); assert rc == 0, o.name
        t_dual = timed(dual, args.iters)
        print(f'   two plans of {N} on two streams, B lags {lag:3d} launches: {t_dual:.2f} ms  ({100 * (t_full / t_dual - 1):+.2f} % vs one plan of {2*N})', flush=True)
    # the same two plans back to back on ONE stream (what splitting alone costs)
    t_seq = timed(lambda: (pa.run(s0), pb.run(s0)), args.iters)
    print(f'   two plans of {N} back to back on one stream: {t_seq:.2f} ms  ({100 * (t_full / t_seq - 1):+.2f} %)', flush=True)
    del eng._plans[(2 * N, 1)], full, pa, pb
    eng._plans.pop((N, 1), None)
    torch.cuda.empty_cache()
