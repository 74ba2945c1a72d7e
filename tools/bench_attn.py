"""Micro-benchmark of the fused attention kernels on the SD-1.5 / ImageNet-64 shapes: fp32 (flash_attn_kernel) vs fp16 operands
(flash_attn_f16_kernel).      python tools/bench_attn.py --images 16 [--iters 5]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sampler_amd import ops  # noqa: E402

SHAPES = [  # (heads, d, sq, skv)
    (8, 40, 4096, 4096), (8, 40, 4096, 77), (8, 80, 1024, 1024), (8, 80, 1024, 77), (8, 160, 256, 256), (8, 160, 64, 64),
    (6, 64, 1024, 1024), (9, 64, 256, 256), (12, 64, 64, 64),
]
ap = argparse.ArgumentParser()
ap.add_argument('--images', type=int, default=16)
ap.add_argument('--iters', type=int, default=5)
ap.add_argument('--only', type=int, nargs='*', help='shape indices')
ap.add_argument('--f16-only', action='store_true')
ap.add_argument('--variant', type=int, default=0, help='ds_attn_args.variant with --f16-rows: 1 = one query block per wave, 2 = two (round 6)')
ap.add_argument('--f16-rows', action='store_true', help='q / k / v given as fp16 tensors (ds_attn_args.in_f16) instead of fp32 rows rounded while staged')
args = ap.parse_args()
B = args.images
for si, (heads, d, sq, skv) in enumerate(SHAPES):
    if args.only and si not in args.only:
        continue
    c = heads * d
    q = torch.randn(B, sq, c, device='cuda')
    kv = torch.randn(B, skv, 2 * c, device='cuda')
    out = torch.zeros(B, sq, c, device='cuda')
    line = f'heads={heads} d={d} sq={sq} skv={skv}:'
    if args.f16_rows:           # fp16 q / k / v tensors (ds_attn_args.in_f16 = 3: what the engines pass where the projections emit fp16 rows)
        import ctypes as C
        from diff_sampler_amd import _lib
        lib = _lib.load()
        q16, kv16, out16 = q.half(), kv.half(), out.half()
        a = _lib.AttnArgs(q16.data_ptr(), kv16.data_ptr(), kv16[:, :, c:].data_ptr(), out16.data_ptr(), c, 2 * c, 2 * c, c, sq * c, skv * 2 * c,
                          skv * 2 * c, sq * c, B, heads, sq, skv, d, d ** -0.5, 1, 3, args.variant)
        run16 = lambda: _lib.check(lib.ds_attention_f16(C.byref(a), _lib.stream_ptr()))
        for _ in range(2):
            run16()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            run16()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        print(line + f'   fp16 rows in / out {ms:8.3f} ms {4.0 * B * heads * sq * skv * d / ms / 1e9:7.1f} TF', flush=True)
        continue
    for f16 in ((True,) if args.f16_only else (False, True)):
        run = lambda: ops.attention(q, kv, kv[:, :, c:], out, batch=B, heads=heads, sq=sq, skv=skv, d=d, ldq=c, ldk=2 * c, ldv=2 * c, ldo=c,
                                    q_bs=sq * c, k_bs=skv * 2 * c, v_bs=skv * 2 * c, o_bs=sq * c, scale=d ** -0.5, f16=f16)
        for _ in range(2):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        line += f'   {"fp16" if f16 else "fp32"} {ms:8.3f} ms {4.0 * B * heads * sq * skv * d / ms / 1e9:7.1f} TF'
    print(line, flush=True)
