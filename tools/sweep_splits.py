"""Sweep the split-K factor and the halo tile shape of ds_conv2d_nhwc on representative under-filled layers.

    python tools/sweep_splits.py            # prints ms per (shape, tile, splits)
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sampler_amd import _lib, ops  # noqa: E402
from diff_sampler_amd._lib import ConvArgs  # noqa: E402

SHAPES = [  # (label, images, res, cin, cout, taps)
    ('cifar B=256 8x8', 256, 8, 256, 256, 9), ('cifar B=256 8x8 cat', 256, 8, 512, 256, 9),
    ('cifar B=32 16x16', 32, 16, 256, 256, 9), ('cifar B=32 32x32', 32, 32, 256, 256, 9), ('cifar B=8 32x32', 8, 32, 256, 256, 9),
    ('cifar B=8 8x8', 8, 8, 256, 256, 9),
    ('sd N=32 16x16', 32, 16, 1280, 1280, 9), ('sd N=32 32x32', 32, 32, 640, 640, 9), ('sd N=32 32x32 cat', 32, 32, 1280, 640, 9),
    ('sd N=32 8x8', 32, 8, 1280, 1280, 9), ('sd N=8 16x16', 8, 16, 1280, 1280, 9), ('sd N=2 8x8', 2, 8, 1280, 1280, 9),
    ('sd N=2 16x16', 2, 16, 1280, 1280, 9), ('sd N=2 32x32', 2, 32, 640, 640, 9), ('sd N=2 64x64', 2, 64, 320, 320, 9),
    ('sd N=32 ff.proj 16x16', 32 * 256, 1, 1280, 10240, 1), ('sd N=2 ff.proj 8x8', 128, 1, 1280, 10240, 1),
    ('sd N=2 qkv 64x64', 8192, 1, 320, 960, 1),
]
lib = _lib.load()
st = _lib.stream_ptr()
ws = torch.empty(64 << 20, device='cuda')
for label, n, res, cin, cout, taps in SHAPES:
    M = n * res * res
    x = torch.randn(M, cin, device='cuda')
    k = 3 if taps == 9 else 1
    wp = ops.pack_conv_weight(torch.randn(cout, cin, k, k, device='cuda') / (taps * cin) ** 0.5)
    bias = torch.randn(cout, device='cuda')
    out = torch.zeros(M, cout, device='cuda')
    a = ConvArgs(x.data_ptr(), None, cin, 0, cin, 0, n, res, res, taps, wp.data_ptr(), cout, bias.data_ptr(), None, 0, 1, None, 0, 1.0, 0,
                 out.data_ptr(), cout)
    a.workspace, a.workspace_floats = ws.data_ptr(), ws.numel()
    fl = 2.0 * M * taps * cin * cout
    rows = []
    for tile in ((128, 256) if taps == 9 else (0,)):
        for s in (0, 1, 2, 3, 4, 5, 6, 8, 12, 16, 24, 32):
            a.tune.mode, a.tune.splits = tile, (s if s else 0)
            if s == 0 and tile != (128 if taps == 9 else 0):
                continue
            if s == 0:
                a.tune.mode = 0                           # pure heuristic (tile and splits)
            rc = lib.ds_conv2d_nhwc(C.byref(a), st)
            assert rc == 0, rc
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                lib.ds_conv2d_nhwc(C.byref(a), st)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            rows.append((('heur' if s == 0 else f't{tile}/s{s}'), ms))
    best = min(r[1] for r in rows)
    print(f'{label:26s} M={M:6d} {cin}->{cout}: ' + '  '.join(f'{k}:{ms*1e3:.0f}{"*" if ms == best else ""}' for k, ms in rows) +
          f'   [us; best {fl/best/1e9:.0f} TF]', flush=True)
