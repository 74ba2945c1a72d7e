#!/usr/bin/env python
"""Per-layer A/B of the fused input normalisation (engine.fuse_norm16): for every 3x3 convolution that the fused plan runs on
conv3x3_f16dma_kernel<.., NORM>, the time of [ds_norm_act pass + convolution] in the pass plan against the time of the fused convolution
alone, each launch sequence timed with HIP events over `--reps` repetitions on the plans' own buffers (after one full evaluation, so the
operands are real activations).  Layers are grouped by (image side, input channels, output channels, sources).

    python tools/ab_fuse_norm_layers.py --config imagenet64 --batch 64
    python tools/ab_fuse_norm_layers.py --config sd15 --batch 16"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sampler_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--config', default='imagenet64')
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--reps', type=int, default=10)
args = ap.parse_args()
lib = _lib.load()
dev = torch.device('cuda')
g = torch.Generator().manual_seed(1)
if args.config == 'sd15':
    from diff_sampler_amd.ldm_engine import CFGDenoiser
    net = CFGDenoiser.from_config('sd15', seed=0, guidance_rate=7.5, use_fp16=True)
    x = torch.randn(args.batch, 4, 64, 64, generator=g).to(dev) * 3
    c, uc = torch.randn(args.batch, 77, 768, generator=g).to(dev), torch.randn(args.batch, 77, 768, generator=g).to(dev)
    run = lambda: net(x, 3.0, condition=c, unconditional_condition=uc)
else:
    from diff_sampler_amd.engine import EDMDenoiser
    net = EDMDenoiser.from_config(args.config, seed=0, use_fp16=True)
    R = net.img_resolution
    x = torch.randn(args.batch, 3, R, R, generator=g).to(dev) * 3
    lab = torch.eye(net.label_dim)[torch.randint(net.label_dim, (args.batch,), generator=g)].to(dev) if net.label_dim else None
    run = lambda: net(x, 3.0, class_labels=lab)


def plan_of(fuse):
    net.engine.fuse_norm16 = fuse
    run()
    torch.cuda.synchronize()
    return list(net.engine._plans.values())[-1]


def time_ops(ops):
    st = _lib.stream_ptr()
    for op in ops:
        assert op.fn(*op.args, st) == 0, op.name
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        for op in ops:
            op.fn(*op.args, st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.reps * 1e3          # microseconds


P0, P1 = plan_of(False), plan_of(True)
by0 = {op.name: (i, op) for i, op in enumerate(P0.ops)}
rows = collections.OrderedDict()
for op in P1.ops:
    if not (op.fn is lib.ds_conv2d_nhwc and op.keep[0].in_f16 and op.keep[0].norm_coefs and op.keep[0].taps == 9):
        continue
    a = op.keep[0]
    i0, conv0 = by0[op.name]
    # the pass that feeds this convolution in the pass plan: the nearest preceding ds_norm_act (its activated output is the conv's x0)
    j = i0 - 1
    while j >= 0 and P0.ops[j].fn is not lib.ds_norm_act:
        j -= 1
    npass = P0.ops[j]
    assert npass.keep[0].out == conv0.keep[0].x0, (op.name, npass.name)
    t_pass, t_conv, t_fused = time_ops([npass]), time_ops([conv0]), time_ops([op])
    key = (a.h, a.c0 + a.c1, a.cout, 2 if a.c1 else 1, a.ec0 + a.ec1)
    r = rows.setdefault(key, [0, 0.0, 0.0, 0.0])
    r[0] += 1; r[1] += t_pass; r[2] += t_conv; r[3] += t_fused
print(f'# {args.config} fp16, batch {args.batch}: per layer class (side, cin, cout, sources, skip-projection channels): launches per evaluation, microseconds per evaluation')
print(f'# {"class":34s} {"n":>3s} {"pass":>9s} {"conv":>9s} {"pass+conv":>10s} {"fused":>9s} {"fused - (pass+conv)":>20s}')
tot = [0.0, 0.0, 0.0]
for key, (n, tp, tc, tf) in rows.items():
    print(f'  {str(key):34s} {n:3d} {tp:9.1f} {tc:9.1f} {tp + tc:10.1f} {tf:9.1f} {tf - tp - tc:+20.1f}   {"FUSE" if tf < tp + tc else "pass"}')
    tot[0] += tp; tot[1] += tc; tot[2] += tf
print(f'# total: pass {tot[0]:.0f} + conv {tot[1]:.0f} = {tot[0] + tot[1]:.0f} us; fused {tot[2]:.0f} us; best per class {sum(min(tp + tc, tf) for _, tp, tc, tf in rows.values()):.0f} us')
