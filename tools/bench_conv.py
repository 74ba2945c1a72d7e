"""Micro-benchmark of the implicit-GEMM conv kernel on the CIFAR-10 denoiser's shapes (B images).

    python tools/bench_conv.py --batch 256 [--entry ds_conv2d_nhwc] [--check]
    python tools/bench_conv.py --batch 256 --only 0 1 --variants 0 1 28 29 --rounds 7    # interleaved A/B of kernel variants
                                                    (ds_conv_tune.variant; ablations need a -DDS_CONV_ABLATIONS build)
    variant words: include/ds_engine.h (ds_conv_tune.variant); 256 = plain 256 x 256 kernel, 512 = coefficient planes from global memory,
    2048 = four-wave 128 x 128 tiles where the default takes eight half-size waves; 65536 + {4, 16, 20, 28} = ablations of the 256 x 256 tile
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sampler_amd import _lib, ops  # noqa: E402
from diff_sampler_amd._lib import ConvArgs  # noqa: E402

SHAPES = [  # (res, c0, c1, cout, taps, share of FLOPs label)
    (32, 256, 0, 256, 9), (32, 256, 256, 256, 9), (32, 256, 128, 256, 9), (32, 128, 0, 256, 9),
    (16, 256, 0, 256, 9), (16, 256, 256, 256, 9), (8, 256, 0, 256, 9), (8, 256, 256, 256, 9),
    (32, 256, 256, 256, 1), (16, 256, 0, 512, 1), (32, 256, 0, 3, 9),
]

SHAPES_IMAGENET64 = [  # ADM ImageNet-64 (192 / 384 / 576 / 768 channels): bench with --batch 64
    (64, 192, 0, 192, 9), (64, 192, 192, 192, 9), (32, 384, 0, 384, 9), (32, 384, 384, 384, 9),
    (16, 576, 0, 576, 9), (16, 576, 576, 576, 9), (8, 768, 0, 768, 9), (32, 192, 0, 384, 9),
]
SHAPES_FFHQ = [  # FFHQ-64 SongUNet (128 / 256 channels at 64 / 32 / 16 / 8): bench with --batch 128; the 128-channel layers run 128-column tiles
    (64, 128, 0, 128, 9), (64, 128, 128, 128, 9), (64, 256, 128, 128, 9), (32, 128, 0, 256, 9), (8, 256, 0, 256, 9),
]
SHAPES_SD15 = [  # SD-1.5 latent U-Net 3x3 convs (320 / 640 / 1280 channels at 64 / 32 / 16 / 8): bench with --batch 32
    (64, 320, 0, 320, 9), (64, 320, 320, 320, 9), (32, 640, 0, 640, 9), (32, 640, 640, 640, 9), (16, 1280, 0, 1280, 9), (16, 1280, 1280, 1280, 9),
    (8, 1280, 0, 1280, 9), (8, 1280, 1280, 1280, 9),                                       # [6], [7]: the 8x8 stage (64-column tiles)
]

SHAPES_SD15_GEMM = [  # SD-1.5 transformer projections / 1x1s on the image rows (taps = 1): bench with --batch 32 --f16 --dma16 [--f16io]
    (64, 320, 0, 320, 1), (64, 320, 0, 960, 1), (64, 1280, 0, 320, 1), (32, 640, 0, 640, 1), (32, 640, 0, 1920, 1), (32, 2560, 0, 640, 1),
    (16, 1280, 0, 1280, 1), (16, 1280, 0, 3840, 1), (16, 5120, 0, 1280, 1),
    (64, 320, 0, 256, 1), (64, 320, 0, 64, 1), (64, 320, 0, 128, 1), (64, 320, 0, 192, 1),   # [9..12] locality experiments: N = one tile
    (64, 320, 0, 2560, 1), (32, 640, 0, 5120, 1), (16, 1280, 0, 10240, 1),                 # [13..15] the GEGLU projections: bench with --geglu
]

SHAPES_SD15_DOWN = [  # SD-1.5 Downsample convolutions (3x3, stride 2; res = OUTPUT size): bench with --batch 32 [--f16 --dma16 --f16out]
    (32, 320, 0, 320, 9), (16, 640, 0, 640, 9), (8, 1280, 0, 1280, 9),
]

ap = argparse.ArgumentParser()
ap.add_argument('--shapes', default='cifar10', choices=['cifar10', 'imagenet64', 'sd15', 'ffhq', 'sd15gemm', 'sd15down'])
ap.add_argument('--batch', type=int, default=256)
ap.add_argument('--entry', default='ds_conv2d_nhwc')
ap.add_argument('--iters', type=int, default=10)
ap.add_argument('--check', action='store_true')
ap.add_argument('--only', type=int, nargs='*')
ap.add_argument('--variants', type=int, nargs='*', default=None)
ap.add_argument('--rounds', type=int, default=5)
ap.add_argument('--split', action='store_true', help='split-fp16 fp32-emulated kernel (wgt_f16 = 2), 3x3 shapes only')
ap.add_argument('--f16', action='store_true', help='fp16-operand kernel (ds_conv_args.wgt_f16), 3x3 shapes only')
ap.add_argument('--dma16', action='store_true', help='with --f16: fp16 ACTIVATIONS (ds_conv_args.in_f16, csrc/conv3x3_f16dma.hip); single source (c0 + c1 channels), no --norm')
ap.add_argument('--nb', type=int, default=0, help='with --dma16: force the column-tile width (64 * nb)')
ap.add_argument('--lda', type=int, default=0, help='with --dma16: override the leading dimension of the fp16 input (timing experiments on access locality; results are then meaningless)')
ap.add_argument('--nw', type=int, default=0, help='with --dma16, taps = 1 shapes: force the 4- / 8-wave GEMM variant')
ap.add_argument('--ablate', type=int, default=0, help='with --dma16: ds_conv_tune.ablate mask (timing only)')
ap.add_argument('--no-res', action='store_true', help='no residual operand in the epilogue')
ap.add_argument('--geglu', action='store_true', help='taps = 1 shapes with --dma16: the GEGLU gate in the epilogue (DS_ACT_GEGLU: cout / 2 output columns, no residual)')
ap.add_argument('--f16out', action='store_true', help='with --dma16: fp16 output rows only')
ap.add_argument('--f16res', action='store_true', help='with --dma16: fp16 residual rows only')
ap.add_argument('--f16io', action='store_true', help='with --dma16: fp16 output rows and fp16 residual rows (ds_conv_args.out_f16 / res_f16: the fp16 residual stream)')
ap.add_argument('--extra', action='store_true', help='append the fused 1x1 skip projection (ec0 = c0 + c1 raw columns) as conv1 of a block with a skip conv has it')
ap.add_argument('--ws', action='store_true', help='give the launcher a split-K workspace (256 MiB), as the engine plans do (the persistent schedule needs it)')
ap.add_argument('--splits', type=int, default=0, help='ds_conv_tune.splits (with --ws): force the split-K factor, 1 = never split')
ap.add_argument('--norm', action='store_true', help='fused GroupNorm affine + SiLU in the halo loader, as the network uses it')
args = ap.parse_args()
STRIDE2 = args.shapes == 'sd15down'
SHAPES = {'cifar10': SHAPES, 'imagenet64': SHAPES_IMAGENET64, 'sd15': SHAPES_SD15, 'ffhq': SHAPES_FFHQ, 'sd15gemm': SHAPES_SD15_GEMM,
          'sd15down': SHAPES_SD15_DOWN}[args.shapes]

lib = _lib.load()
# DS_CONV / DS_CONV_VARIANT in the environment become ds_conv_tune.mode / .variant of every ConvArgs built below (_lib._ENV_TUNE)
fn = getattr(lib, args.entry)
fn.restype, fn.argtypes = C.c_int, [C.POINTER(ConvArgs), C.c_void_p]
B = args.batch
dev = 'cuda'
tot_fl = tot_t = 0.0
for si, (res, c0, c1, cout, taps) in enumerate(SHAPES):
    if args.only and si not in args.only:
        continue
    M = B * res * res
    MI = 4 * M if STRIDE2 else M                      # input rows (stride 2: the input is 2 res x 2 res)
    x0 = torch.randn(MI, c0, device=dev)
    x1 = torch.randn(M, c1, device=dev) if c1 else None
    w = torch.randn(cout, c0 + c1, 3 if taps == 9 else 1, 3 if taps == 9 else 1, device=dev) / (taps * (c0 + c1)) ** 0.5
    wp = ops.pack_conv_weight(w)
    bias = torch.randn(cout, device=dev)
    res_t = torch.randn(M, cout, device=dev)
    old = 4 if cout < 4 else cout
    out = torch.zeros(M, old, device=dev)
    a = ConvArgs(x0.data_ptr(), x1.data_ptr() if c1 else None, c0, c1, c0, c1, B, res, res, taps, wp.data_ptr(), cout, bias.data_ptr(),
                 None, 0, 1, res_t.data_ptr() if cout >= 4 else None, cout, 0.70710678, 0, out.data_ptr(), old)
    if args.no_res or STRIDE2:
        a.res = None
    if STRIDE2:
        a.stride = 2
    if (args.f16 or args.split) and ((taps != 9 and not args.dma16) or cout < 64 or (args.norm and res < 16)):
        continue
    if args.extra and taps == 9:
        ec = c0 + c1
        e0 = torch.randn(M, ec, device=dev)
        we = torch.randn(cout, ec, 1, 1, device=dev) / ec ** 0.5
        if args.split:
            wp, a.wgt_shift = ops.pack_conv_weight_split(w, we)
        else:
            wp = ops.pack_conv_weight_f16(w, we) if args.f16 else torch.cat([wp, ops.pack_conv_weight(we)], 1).contiguous()
        a.wgt, a.e0, a.ec0, a.eld0 = wp.data_ptr(), e0.data_ptr(), ec, ec
    elif args.split:
        wp, a.wgt_shift = ops.pack_conv_weight_split(w)
        a.wgt = wp.data_ptr()
    elif args.f16:
        wp = ops.pack_conv_weight_f16(w) if taps == 9 else ops.pack_linear_weight_f16(wp)
        a.wgt = wp.data_ptr()
    if args.f16 or args.split:
        a.wgt_f16 = 2 if args.split else 1
    if args.dma16:
        assert args.f16 and not args.norm
        x16 = torch.randn(MI, c0 + c1, device=dev).to(torch.float16)
        a.x0, a.x1, a.c0, a.c1, a.ld0, a.ld1, a.in_f16 = x16.data_ptr(), None, c0 + c1, 0, c0 + c1, 0, 1
        if args.extra and taps == 9:
            e16 = torch.randn(M, c0 + c1, device=dev).to(torch.float16)
            a.e0, a.ec0, a.eld0 = e16.data_ptr(), c0 + c1, c0 + c1
        if args.f16io or args.f16out:
            out16 = torch.zeros(M, cout, device=dev, dtype=torch.float16)
            a.out, a.out_f16 = out16.data_ptr(), 1
        if (args.f16io or args.f16res) and not args.no_res:
            res16 = res_t.to(torch.float16)
            a.res, a.res_f16 = res16.data_ptr(), 1
        if args.lda:
            a.ld0 = args.lda
        a.tune.f16dma_nb, a.tune.f16dma_nw, a.tune.ablate = args.nb, args.nw, args.ablate
        a.tune.splits = args.splits
        if args.geglu:
            assert taps == 1 and cout % 128 == 0
            a.act, a.res, a.res_f16, a.cbias, a.out_ld, a.out_scale = 2, None, 0, None, cout // 2, 1.0
    if args.norm and taps == 9:
        coefs = torch.randn(B, 3, c0 + c1, device=dev) * 0.1 + torch.tensor([0., 1., 0.], device=dev).reshape(1, 3, 1)
        a.norm_coefs, a.norm_act = coefs.data_ptr(), 1
    if args.ws:
        scratch = torch.empty(64 << 20, device=dev)
        a.workspace, a.workspace_floats = scratch.data_ptr(), scratch.numel()
    st = _lib.stream_ptr()
    if args.variants:
        import statistics
        fl = 2.0 * M * (taps * (c0 + c1) + (c0 + c1 if args.extra and taps == 9 else 0)) * cout
        times = {v: [] for v in args.variants}
        for rnd in range(args.rounds + 1):
            for v in args.variants:
                a.tune.variant = v
                fn(C.byref(a), st); torch.cuda.synchronize()
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    fn(C.byref(a), st)
                e1.record(); torch.cuda.synchronize()
                if rnd:
                    times[v].append(e0.elapsed_time(e1) / args.iters)
        a.tune.variant = 0
        print(f'[{si}] {res}x{res} {c0}+{c1}->{cout} taps={taps} M={M} norm={int(args.norm)}: ' +
              '  '.join(f'v{v}: {statistics.median(t):.3f} ms {fl / statistics.median(t) / 1e9:6.1f} TF (min {fl / min(t) / 1e9:6.1f})' for v, t in times.items()), flush=True)
        continue
    rc = fn(C.byref(a), st); assert rc == 0, rc
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        fn(C.byref(a), st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    fl = 2.0 * M * taps * (c0 + c1) * cout
    tot_fl += fl; tot_t += ms
    msg = f'[{si}] {res}x{res} {c0}+{c1}->{cout} taps={taps} M={M}: {ms:8.3f} ms  {fl/ms/1e9:7.1f} TFLOP/s'
    if args.dma16:        # the launch's algorithmic HBM bytes: fp16 activations in, output and residual rows out / in
        byt = MI * (c0 + c1) * 2 + M * cout * (2 if a.out_f16 else 4) + (M * cout * (2 if a.res_f16 else 4) if a.res else 0)
        msg += f'  {byt / ms / 1e6:6.0f} GB/s'
    if args.check:
        xin = torch.cat([x0, x1], 1) if c1 else x0
        ri = 2 * res if STRIDE2 else res
        xin = xin.reshape(B, ri, ri, c0 + c1).permute(0, 3, 1, 2)
        ref = torch.nn.functional.conv2d(xin, w, bias, stride=(2 if STRIDE2 else 1), padding=(1 if taps == 9 else 0))
        ref = ref.permute(0, 2, 3, 1).reshape(M, cout)
        if cout >= 4 and not STRIDE2:
            ref = (ref + res_t) * 0.70710678
        else:
            ref = ref * 0.70710678
        got = out16.float() if args.dma16 and a.out_f16 else out[:, :cout]          # fp16 output rows: within one fp16 ulp of the reference
        if args.dma16:                                                               # the kernel multiplies fp16 operands
            xin16 = x16.float().reshape(B, ri, ri, c0 + c1).permute(0, 3, 1, 2)
            ref = torch.nn.functional.conv2d(xin16, w.half().float(), bias, stride=(2 if STRIDE2 else 1), padding=(1 if taps == 9 else 0)).permute(0, 2, 3, 1).reshape(M, cout)
            if a.res:
                ref = (ref + (res16.float() if a.res_f16 else res_t)) * 0.70710678
            else:
                ref = ref * 0.70710678
        err = float((got - ref).abs().max() / ref.abs().max())
        msg += f'  relerr {err:.2e}'
    print(msg, flush=True)
if tot_t:
    print(f'total {tot_t:.3f} ms, {tot_fl/tot_t/1e9:.1f} TFLOP/s aggregate (unweighted by layer counts)')
