"""Time one denoiser evaluation (raw plan) on the GPU: ms, TFLOP/s; optional per-op breakdown."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diff_sampler_amd.arch as arch  # noqa: E402
from diff_sampler_amd import _lib  # noqa: E402
from diff_sampler_amd.engine import EDMDenoiser  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--config', default='cifar10')
ap.add_argument('--batch', type=int, nargs='+', default=[64, 256])
ap.add_argument('--iters', type=int, default=5)
ap.add_argument('--breakdown', action='store_true')
ap.add_argument('--per-op', action='store_true', help='with --breakdown: one line per convolution launch (geometry, kernel id, time, TFLOP/s)')
ap.add_argument('--fp16', action='store_true', help="the reference's use_fp16 mode")
args = ap.parse_args()
# A/B switches: DS_CONV (= ds_conv_tune.mode) / DS_CONV_VARIANT in the environment apply to every layer of the plans built below (_lib._ENV_TUNE)

import diff_sampler_amd.ldm_arch as ldm_arch  # noqa: E402
if args.config in ldm_arch.NAMED_LDM_CONFIGS:
    from diff_sampler_amd.ldm_engine import CFGDenoiser  # noqa: E402
    net = CFGDenoiser.from_config(args.config, seed=0, guidance_rate=7.5)
    spec = net.spec
    for B in args.batch:
        x = torch.randn(B, spec.in_channels, spec.img_resolution, spec.img_resolution, device='cuda')
        c = torch.randn(B, 77, spec.context_dim, device='cuda'); uc = torch.randn(B, 77, spec.context_dim, device='cuda')
        net.raw(x, 1.5, c, uc)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(args.iters):
            out, plan, _ = net.raw(x, 1.5, c, uc)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / args.iters
        fl = net.engine.flops(2 * B)
        print(f'{args.config} B={B} (CFG: {2*B} U-Net images): {dt*1e3:.2f} ms/eval  {fl/dt/1e12:.1f} TFLOP/s  {B/dt:.1f} img-evals/s  ({len(plan.ops)} launches per evaluation + {len(plan.ctx.ops)} once per context)', flush=True)
        if args.breakdown:
            st = _lib.stream_ptr()
            tot = {}
            # per-evaluation launches; the cross-attention K / V projections of the context (plan.ctx) run once per context tensor and are
            # listed separately under 'ctx:' (they are NOT part of the ms/eval above once the context is cached)
            for op in list(plan.ops) + list(plan.ctx.ops):
                torch.cuda.synchronize()
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                op.fn(*op.args, st)
                e1.record(); torch.cuda.synchronize()
                parts = op.name.split('.')
                kind = parts[-1] if parts[-1] not in ('stats',) else '.'.join(parts[-2:])
                if op in plan.ctx.ops: kind = 'ctx:' + kind
                if 'attn1' in op.name and kind == 'attn1': kind = 'attn1(self)'
                tot[kind] = tot.get(kind, 0.0) + e0.elapsed_time(e1)
            for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
                print(f'   {k:20s} {v:8.2f} ms')
    sys.exit(0)
lib = _lib.load()


def _again(op, st):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); op.fn(*op.args, st); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)


net = EDMDenoiser.from_config(args.config, seed=0, use_fp16=args.fp16)
spec = net.spec
for B in args.batch:
    x = torch.randn(B, spec.in_channels, spec.img_resolution, spec.img_resolution, device='cuda')
    sig = torch.tensor(1.5)
    lab = torch.eye(spec.label_dim, device='cuda')[torch.randint(spec.label_dim, (B,), device='cuda')] if spec.label_dim else None
    net.raw(x, sig, lab)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(args.iters):
        out, plan = net.raw(x, sig, lab)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / args.iters
    fl = arch.flops_per_image(spec) * B
    print(f'{args.config} B={B}: {dt*1e3:.2f} ms/eval  {fl/dt/1e12:.1f} TFLOP/s  {B/dt:.0f} img-evals/s  ({len(plan.ops)} launches)', flush=True)
    if args.breakdown:
        st = _lib.stream_ptr()
        tot = {}
        for op in plan.ops:
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            op.fn(*op.args, st)
            e1.record(); torch.cuda.synchronize()
            kind = op.name.split('.')[-1]
            ms = min(e0.elapsed_time(e1), _again(op, st))
            tot[kind] = tot.get(kind, 0.0) + ms
            if args.per_op and op.fn is lib.ds_norm_act:
                a = op.keep[0]
                up = 4 if a.resample == 2 else 1
                pix = a.n * a.h * a.w
                byts = pix * (a.c0 * (2 if a.in_f16 & 1 else 4) + a.c1 * (2 if a.in_f16 & 2 else 4)) + \
                    pix * up // (4 if a.resample == 1 else 1) * (a.c0 + a.c1) * ((2 if a.out_f16 else 4) + (2 if a.raw_out else 0))
                print(f'      {op.name:34s} {a.h:2d}x{a.w:<2d} norm c={a.c0:4d}+{a.c1:<4d} resample={a.resample} in16={a.in_f16} out16={a.out_f16} raw={int(bool(a.raw_out))} '
                      f'{ms * 1e3:7.1f} us {byts / ms / 1e9:6.0f} GB/s', flush=True)
            if args.per_op and op.fn is lib.ds_conv2d_nhwc:
                import ctypes as C
                a = op.keep[0]
                fl = 2.0 * a.n * a.h * a.w * a.cout * (a.taps * (a.c0 + a.c1) + a.ec0 + a.ec1)
                print(f'      {op.name:34s} {a.h:2d}x{a.w:<2d} taps={a.taps} cin={a.c0 + a.c1:4d}+{a.ec0 + a.ec1:<4d} cout={a.cout:4d} kid={lib.ds_conv_kernel_id(C.byref(a)):4d} '
                      f'{ms * 1e3:7.1f} us {fl / ms / 1e9:6.1f} TF', flush=True)
        for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
            print(f'   {k:16s} {v:8.2f} ms')
