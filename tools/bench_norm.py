"""Micro-benchmark of the fp16-mode normalisation pass (ds_norm_act with out_f16): achieved HBM GB/s per shape.
    python tools/bench_norm.py [--batch 64]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sampler_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=64)
args = ap.parse_args()
lib = _lib.load()
dev = 'cuda'
B = args.batch
# (res, c0, c1, raw copy, fp16 input)
for res, c0, c1, raw, in16 in [(64, 192, 0, False, False), (64, 192, 192, True, False), (64, 192, 0, False, True), (32, 384, 384, True, False),
                               (32, 384, 0, False, True), (16, 576, 0, False, True), (8, 768, 0, False, False)]:
    Cc = c0 + c1
    M = B * res * res
    x0 = torch.randn(M, c0, device=dev)
    if in16:
        x0 = x0.to(torch.float16)
    x1 = torch.randn(M, c1, device=dev) if c1 else None
    coefs = torch.randn(B, 3, Cc, device=dev) * 0.1 + torch.tensor([0., 1., 0.], device=dev).reshape(1, 3, 1)
    out = torch.empty(M, Cc, dtype=torch.float16, device=dev)
    rw = torch.empty(M, Cc, dtype=torch.float16, device=dev) if raw else None
    a = _lib.NormArgs(x0.data_ptr(), x1.data_ptr() if c1 else None, c0, c1, c0, c1, B, res, res, 32, 1e-5, None, None, None, None, None, None, 0, 1,
                      1, 0, out.data_ptr(), Cc, coefs.data_ptr())
    a.out_f16, a.raw_out, a.raw_ld, a.in_f16 = 1, (rw.data_ptr() if raw else None), Cc, int(in16)
    st = _lib.stream_ptr()
    rc = lib.ds_norm_act(C.byref(a), st); assert rc == 0, rc
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        lib.ds_norm_act(C.byref(a), st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    byts = M * Cc * ((2 if in16 else 4) + 2 + (2 if raw else 0))
    print(f'{res}x{res} {c0}+{c1} raw={int(raw)} in16={int(in16)}: {ms * 1e3:8.1f} us  {byts / ms / 1e6:7.0f} GB/s', flush=True)
