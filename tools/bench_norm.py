"""Micro-benchmark of the fp16-mode normalisation pass (ds_norm_act with out_f16) on the shapes of the two fp16 lines: achieved HBM GB/s per
shape for the three forms of the pass --
    8B    norm_act_kernel (rounds 3 - 5: 8 bytes per lane) on planes written by ds_gn_finalize          [finalize + pass timed together]
    16B   norm_act16_kernel (round 6: 16 bytes per lane) on the same planes                            [finalize + pass timed together]
    16B+F norm_act16_kernel<FIN>: the pass computes the statistics itself, no ds_gn_finalize launch    [images of at most 32 x 32 pixels]
Operands are cold (a 512 MiB buffer is written between the timed calls): in a network the tensor was written by another kernel a launch ago.

    python tools/bench_norm.py [--config imagenet64|sd15] [--warm]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sampler_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--config', default='imagenet64')
ap.add_argument('--warm', action='store_true', help='no cache flush between the timed calls')
args = ap.parse_args()
lib = _lib.load()
dev = 'cuda'
# (images, res, c0, c1, raw copy) -- every pass shape of the line at its benchmark batch (ImageNet-64: 64 images; SD-1.5: 16 latents = 32 U-Net images)
SHAPES = {'imagenet64': [(64, 64, 192, 0, 0), (64, 32, 192, 0, 1), (64, 32, 384, 0, 0), (64, 32, 384, 384, 1), (64, 32, 384, 192, 1), (64, 16, 384, 0, 1),
                         (64, 16, 576, 0, 0), (64, 16, 576, 576, 1), (64, 8, 576, 0, 1), (64, 8, 768, 0, 0), (64, 8, 768, 768, 1)],
          'sd15': [(32, 64, 320, 0, 0), (32, 64, 320, 320, 1), (32, 32, 320, 0, 1), (32, 32, 640, 0, 0), (32, 32, 640, 640, 1), (32, 16, 640, 0, 1),
                   (32, 16, 1280, 0, 0), (32, 16, 1280, 1280, 1), (32, 8, 1280, 0, 0), (32, 8, 1280, 1280, 1)]}[args.config]
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
st = _lib.stream_ptr()


def timed(fn, reps=8):
    fn(); torch.cuda.synchronize()
    tot = 0.0
    for _ in range(reps):
        if not args.warm:
            flush.fill_(1)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / reps * 1e3          # us


print(f'# {args.config}: us per (statistics + pass), algorithmic GB/s of the pass bytes; cold operands = {not args.warm}')
sums = {'8B': 0.0, '16B': 0.0, 'best': 0.0}
for B, res, c0, c1, raw in SHAPES:
    Cc, M, HW = c0 + c1, B * res * res, res * res
    x0 = torch.randn(M, c0, device=dev).to(torch.float16)
    x1 = torch.randn(M, c1, device=dev).to(torch.float16) if c1 else None
    gm, bt = torch.rand(Cc, device=dev) + 0.5, torch.randn(Cc, device=dev) * 0.1
    s0 = torch.randn(M // 64, 2, c0, device=dev).abs() * 64
    s1 = torch.randn(M // 64, 2, c1, device=dev).abs() * 64 if c1 else None
    mean, rstd, planes = torch.empty(B * 32, device=dev), torch.empty(B * 32, device=dev), torch.empty(B, 3, Cc, device=dev)
    out = torch.empty(M, Cc, dtype=torch.float16, device=dev)
    rw = torch.empty(M, Cc, dtype=torch.float16, device=dev) if raw else None
    f = _lib.GnFinalizeArgs(s0.data_ptr(), s1.data_ptr() if c1 else None, c0, c1, B, HW, 32, 1e-5, gm.data_ptr(), bt.data_ptr(), None, None, 0, 1,
                            mean.data_ptr(), rstd.data_ptr(), planes.data_ptr())

    def args_for(variant, fin):
        a = ops._norm_args(x0, c0, c0, B, res, res, x1=x1, c1=c1, ld1=c1, groups=32, eps=1e-5, act=1, out=out, out_ld=Cc)
        a.in_f16, a.out_f16, a.tune_variant = (3 if c1 else 1), 1, variant
        if raw:
            a.raw_out, a.raw_ld = C.c_void_p(rw.data_ptr()), Cc
        if fin:
            a.stats0, a.stats1 = C.c_void_p(s0.data_ptr()), (C.c_void_p(s1.data_ptr()) if c1 else None)
            a.gamma, a.beta = C.c_void_p(gm.data_ptr()), C.c_void_p(bt.data_ptr())
        else:
            a.coefs = C.c_void_p(planes.data_ptr())
        return a

    a8, a16, af = args_for(1, False), args_for(0, False), args_for(0, True)
    af1, afn = args_for(2, True), args_for(4, True)           # other workgroup caps of the self-finalising pass: sums <= rows / 4, no cap

    def two(a):
        def go():
            assert lib.ds_gn_finalize(C.byref(f), st) == 0
            assert lib.ds_norm_act(C.byref(a), st) == 0
        return go

    t8, t16 = timed(two(a8)), timed(two(a16))
    tf = timed(lambda: lib.ds_norm_act(C.byref(af), st)) if HW <= 1024 else None
    tf1 = timed(lambda: lib.ds_norm_act(C.byref(af1), st)) if HW <= 1024 else None
    tfn = timed(lambda: lib.ds_norm_act(C.byref(afn), st)) if HW <= 1024 else None
    byts = M * Cc * (2 + 2 + (2 if raw else 0))
    best = min(t16, tf, tf1, tfn) if tf is not None else t16
    sums['8B'] += t8; sums['16B'] += t16; sums['best'] += best
    print(f'{B:3d} x {res:2d}x{res:2d} x {c0:4d}+{c1:<4d} raw={raw}  {byts / 1e6:7.1f} MB   8B {t8:7.1f} us {byts / t8 / 1e3:6.0f} GB/s   16B {t16:7.1f} us {byts / t16 / 1e3:6.0f} GB/s   '
          + (f'16B+F {tf:7.1f} us {byts / tf / 1e3:6.0f} GB/s  (cap 4x: {tf1:6.1f} us, no cap: {tfn:6.1f} us)' if tf is not None else '16B+F     --'), flush=True)
print(f'# sum over the shapes: 8B {sums["8B"]:.0f} us, 16B {sums["16B"]:.0f} us, best of 16B / 16B+F {sums["best"]:.0f} us')
