#!/usr/bin/env python
"""Per-launch times of one network evaluation INSIDE the network: every op of the engine's plan bracketed by a HIP event pair during whole
evaluations (so every operand is as warm / cold as the sampler finds it), averaged over `--reps` evaluations and grouped by layer signature.
What rocprofv3's per-instantiation table cannot show: which LAYERS a kernel instantiation spends its time on, and each layer's distance to its
own matrix / HBM bound (algorithmic FLOPs and bytes from the launch arguments).

    python tools/time_plan_ops.py --config sd15 --batch 16
    python tools/time_plan_ops.py --config imagenet64 --batch 64 [--fp32]"""
import argparse
import collections
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sampler_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--config', default='imagenet64')
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--reps', type=int, default=5)
ap.add_argument('--fp32', action='store_true')
ap.add_argument('--top', type=int, default=45)
ap.add_argument('--json', default='')
args = ap.parse_args()
lib = _lib.load()
dev = torch.device('cuda')
g = torch.Generator().manual_seed(1)
f16 = not args.fp32
if args.config == 'sd15':
    from diff_sampler_amd.ldm_engine import CFGDenoiser
    net = CFGDenoiser.from_config('sd15', seed=0, guidance_rate=7.5, use_fp16=f16)
    x = torch.randn(args.batch, 4, 64, 64, generator=g).to(dev) * 3
    c, uc = torch.randn(args.batch, 77, 768, generator=g).to(dev), torch.randn(args.batch, 77, 768, generator=g).to(dev)
    run = lambda: net(x, 3.0, condition=c, unconditional_condition=uc)
else:
    from diff_sampler_amd.engine import EDMDenoiser
    net = EDMDenoiser.from_config(args.config, seed=0, use_fp16=f16)
    R = net.img_resolution
    x = torch.randn(args.batch, 3, R, R, generator=g).to(dev) * 3
    lab = torch.eye(net.label_dim)[torch.randint(net.label_dim, (args.batch,), generator=g)].to(dev) if net.label_dim else None
    run = lambda: net(x, 3.0, class_labels=lab)

run(); run()
torch.cuda.synchronize()
plan = list(net.engine._plans.values())[-1]
ops = plan.ops
st = _lib.stream_ptr()
conv_fn, norm_fn, fin_fn = lib.ds_conv2d_nhwc, lib.ds_norm_act, lib.ds_gn_finalize
attn_fns = (lib.ds_attention, lib.ds_attention_f16)


def signature(op):
    """(label, flops, bytes) of a launch from its argument struct."""
    a = op.keep[0] if op.keep else None
    if op.fn is conv_fn:
        m = a.n * a.h * a.w
        s = a.stride if a.stride else 1
        k = a.taps * (a.c0 + a.c1) + a.ec0 + a.ec1
        cout = a.cout
        eb = 2 if a.in_f16 else 4
        ob = 2 if a.out_f16 else 4
        n_eff = cout // 2 if a.act == 2 else cout
        by = m * s * s * (a.c0 + a.c1) * eb + m * (a.ec0 + a.ec1) * eb + m * n_eff * ob + (m * n_eff * (2 if a.res_f16 else 4) if a.res else 0) + k * cout * 2
        lab = f'conv{"3x3" if a.taps == 9 else "1x1"} n={a.n} {a.h}x{a.w} {a.c0}+{a.c1}(+{a.ec0}+{a.ec1})->{cout}' \
              f'{" s2" if s == 2 else ""}{" geglu" if a.act == 2 else ""}{" norm" if a.norm_coefs else ""}{" res" if a.res else ""}{" f16" if a.in_f16 else ""}'
        return lab, 2.0 * m * k * cout, by
    if op.fn is norm_fn:
        m = a.n * a.h * a.w
        cc = a.c0 + a.c1
        by = m * cc * ((2 if a.in_f16 else 4) + (2 if a.out_f16 else 4)) + (m * cc * 2 if a.raw_out else 0)
        return f'norm n={a.n} {a.h}x{a.w} c={a.c0}+{a.c1} rs={a.resample}{" fin" if a.stats0 else ""}{" raw" if a.raw_out else ""}{" f16" if a.in_f16 else ""}', 0.0, by
    if op.fn is fin_fn:
        return f'gn_finalize n={a.n} hw={a.hw} c={a.c0}+{a.c1}', 0.0, 0
    if op.fn in attn_fns:
        fl = 4.0 * a.batch * a.heads * a.sq * a.skv * a.d
        return f'attn b={a.batch} h={a.heads} sq={a.sq} skv={a.skv} d={a.d}{" f16" if op.fn is lib.ds_attention_f16 else ""}', fl, 0
    return op.name.split('.')[-1] if '.' in op.name else op.name, 0.0, 0


sigs = [signature(op) for op in ops]
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in ops]
tot = [0.0] * len(ops)
# the engine's own call path fills the inputs (embedding, stem operands ...) -- then replay the plan op by op with event pairs
for _ in range(args.reps):
    run()
    for (e0, e1), op in zip(ev, ops):
        e0.record()
        rc = op.fn(*op.args, st)
        e1.record()
        assert rc == 0, op.name
    torch.cuda.synchronize()
    for i, (e0, e1) in enumerate(ev):
        tot[i] += e0.elapsed_time(e1) * 1e3
tot = [t / args.reps for t in tot]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(args.reps):
    plan.run(st)
e1.record()
torch.cuda.synchronize()
whole = e0.elapsed_time(e1) * 1e3 / args.reps

groups = collections.OrderedDict()
for (lab, fl, by), t in zip(sigs, tot):
    gq = groups.setdefault(lab, dict(n=0, us=0.0, flops=fl, bytes=by))
    gq['n'] += 1
    gq['us'] += t
rows = sorted(groups.items(), key=lambda kv: -kv[1]['us'])
ssum = sum(tot)
print(f'# {args.config} batch {args.batch} {"fp16" if f16 else "fp32"}: {len(ops)} launches per evaluation, sum of event pairs {ssum/1e3:.2f} ms, '
      f'plan.run {whole/1e3:.2f} ms per evaluation')
print(f'{"layer":78s} {"n":>3s} {"us each":>8s} {"us all":>8s} {"share":>6s} {"TF":>7s} {"GB/s":>7s}')
for lab, gq in rows[:args.top]:
    each = gq['us'] / gq['n']
    tf = gq['flops'] / each / 1e6 if gq['flops'] else 0.0
    gbs = gq['bytes'] / each / 1e3 if gq['bytes'] else 0.0
    print(f'{lab[:78]:78s} {gq["n"]:3d} {each:8.1f} {gq["us"]:8.1f} {gq["us"]/ssum:6.3f} {tf:7.0f} {gbs:7.0f}')
if args.json:
    with open(args.json, 'w') as fh:
        json.dump(dict(config=args.config, batch=args.batch, fp16=f16, launches=len(ops), sum_us=ssum, plan_run_us=whole,
                       layers=[dict(layer=k, **v) for k, v in rows]), fh, indent=0)
