"""Micro-benchmark of the 1x1 / Linear kernels on the SD-1.5 transformer shapes: fp32 (gemm_dma8_kernel) vs fp16 operands
(gemm_f16_kernel).      python tools/bench_gemm.py --images 16 [--iters 10]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sampler_amd import _lib, ops  # noqa: E402
from diff_sampler_amd._lib import ConvArgs  # noqa: E402

SHAPES = [  # (side, K, N, GEGLU?)   rows = images * side^2
    (64, 320, 960, 0), (64, 320, 320, 0), (64, 320, 2560, 1), (64, 1280, 320, 0),
    (32, 640, 1920, 0), (32, 640, 5120, 1), (32, 2560, 640, 0),
    (16, 1280, 3840, 0), (16, 1280, 10240, 1), (16, 5120, 1280, 0),
    (32, 384, 1152, 0), (16, 576, 1728, 0),          # ImageNet-64 ADM qkv
]
ap = argparse.ArgumentParser()
ap.add_argument('--images', type=int, default=16)
ap.add_argument('--iters', type=int, default=10)
args = ap.parse_args()
lib = _lib.load()
dev = 'cuda'
for side, k, n, geglu in SHAPES:
    rows = args.images * side * side
    x = torch.randn(rows, k, device=dev)
    w = ops.pack_linear_weight(torch.randn(n, k, device=dev) / k ** 0.5)
    w16 = ops.pack_linear_weight_f16(w)
    bias = torch.randn(n, device=dev)
    old = n // 2 if geglu else n
    out = torch.zeros(rows, old, device=dev)
    line = f'{side}x{side} rows={rows} K={k} N={n}{" geglu" if geglu else ""}:'
    for mode, wt in ((0, w), (1, w16)):
        a = ConvArgs(x.data_ptr(), None, k, 0, k, 0, rows, 1, 1, 1, wt.data_ptr(), n, bias.data_ptr(), None, 0, 1, None, 0, 1.0,
                     _lib.DS_ACT_GEGLU if geglu else 0, out.data_ptr(), old)
        a.wgt_f16 = mode
        st = _lib.stream_ptr()
        for _ in range(3):
            rc = lib.ds_conv2d_nhwc(C.byref(a), st)
            assert rc == 0, rc
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            lib.ds_conv2d_nhwc(C.byref(a), st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        line += f'   {"fp16" if mode else "fp32"} {ms:7.3f} ms {2.0 * rows * k * n / ms / 1e9:7.1f} TF'
    print(line, flush=True)
