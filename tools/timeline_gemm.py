#!/usr/bin/env python
"""Phase timeline of one launch of the fp16-activation GEMM (gemm_f16dma_kernel), from s_memtime stamps written by the 'timeline' build of the
library (csrc/ds_common.h: DS_TIMELINE; `python -c "from diff_sampler_amd import build; build.build_libs(('timeline',))"`): per workgroup
  0 entry   1 first operands landed (prologue done)   2 main loop done   3 epilogue issued   4 stores acknowledged
plus {HW_ID, XCC_ID}.  Prints the duration of each phase (median / p10 / p90 over workgroups), the launch's span, how many workgroups were in
each phase at sample instants, and -- for CUs that ran two workgroups at once -- the offset between the partners' phases.

    DS_LIB_PATH=diff_sampler_amd/csrc/libdsamd_timeline.so python tools/timeline_gemm.py --m 131072 --k 320 --n 960 [--geglu] [--res] [--nw 4] [--nb 3]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sampler_amd import _lib, ops  # noqa: E402
from diff_sampler_amd._lib import ConvArgs  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--m', type=int, default=131072)
ap.add_argument('--k', type=int, default=320)
ap.add_argument('--n', type=int, default=960)
ap.add_argument('--geglu', action='store_true')
ap.add_argument('--res', action='store_true')
ap.add_argument('--nw', type=int, default=0)
ap.add_argument('--nb', type=int, default=0)
ap.add_argument('--ablate', type=int, default=0)
ap.add_argument('--cold', action='store_true', help='flush L2 / MALL before the stamped launch')
ap.add_argument('--conv', type=int, nargs=4, metavar=('IMAGES', 'SIDE', 'CIN', 'COUT'), help='a 3x3 convolution on fp16 rows (conv3x3_f16dma_kernel) instead of the GEMM')
ap.add_argument('--stats', action='store_true', help='--conv: GroupNorm column sums in the epilogue')
ap.add_argument('--silu', action='store_true')
args = ap.parse_args()
lib = _lib.load()
dev = 'cuda'
if args.conv:
    nimg, side, cin, cout = args.conv
    M, K, N = nimg * side * side, 9 * cin, cout
    x16 = torch.randn(M, cin, device=dev).to(torch.float16)
    wp = ops.pack_conv_weight_f16(torch.randn(cout, cin, 3, 3, device=dev) / (9 * cin) ** 0.5)
    bias = torch.randn(cout, device=dev)
    out16 = torch.zeros(M, cout, device=dev, dtype=torch.float16)
    res16 = torch.randn(M, cout, device=dev).to(torch.float16) if args.res else None
    stats = torch.zeros((M // 64) * 2 * cout, device=dev) if args.stats else None
    ws = torch.zeros(32 << 20, device=dev)
    a = ConvArgs(x16.data_ptr(), None, cin, 0, cin, 0, nimg, side, side, 9, wp.data_ptr(), cout, bias.data_ptr(), None, 0, 1,
                 res16.data_ptr() if res16 is not None else None, cout, 0.70710678 if res16 is not None else 1.0, 1 if args.silu else 0, out16.data_ptr(), cout)
    a.wgt_f16, a.in_f16, a.out_f16, a.res_f16 = 1, 1, 1, 1 if res16 is not None else 0
    if stats is not None:
        a.stats_out = stats.data_ptr()
    a.tune.f16dma_nb = args.nb
    a.tune.splits = 1
    a.workspace, a.workspace_floats = ws.data_ptr(), ws.numel()
    n_out = cout
else:
    M, K, N = args.m, args.k, args.n
    x16 = torch.randn(M, K, device=dev).to(torch.float16)
    w = torch.randn(N, K, 1, 1, device=dev) / K ** 0.5
    wp = ops.pack_linear_weight_f16(ops.pack_conv_weight(w))
    bias = torch.randn(N, device=dev)
    n_out = N // 2 if args.geglu else N
    out16 = torch.zeros(M, n_out, device=dev, dtype=torch.float16)
    res16 = torch.randn(M, N, device=dev).to(torch.float16) if args.res and not args.geglu else None
    ws = torch.zeros(32 << 20, device=dev)                                    # u64[workgroup][8][8]
    a = ConvArgs(x16.data_ptr(), None, K, 0, K, 0, M, 1, 1, 1, wp.data_ptr(), N, bias.data_ptr(), None, 0, 1, res16.data_ptr() if res16 is not None else None,
                 N, 1.0, 2 if args.geglu else 0, out16.data_ptr(), n_out)
    a.wgt_f16, a.in_f16, a.out_f16, a.res_f16 = 1, 1, 1, 1 if res16 is not None else 0
    a.tune.f16dma_nb, a.tune.f16dma_nw = args.nb, args.nw
    a.workspace, a.workspace_floats = ws.data_ptr(), ws.numel()
st = _lib.stream_ptr()
fn = lib.ds_conv2d_nhwc
a.tune.ablate = args.ablate
for _ in range(3):
    assert fn(C.byref(a), st) == 0
torch.cuda.synchronize()
if args.cold:
    flush = torch.empty(768 << 20, dtype=torch.uint8, device=dev)
    flush.fill_(1)
    torch.cuda.synchronize()
a.tune.ablate = args.ablate | 0x8000
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
assert fn(C.byref(a), st) == 0
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
raw = ws.cpu().numpy().view(np.uint64).reshape(-1, 8, 8)
used = np.nonzero(raw[:, 0, 1])[0]
if len(used) == 0:
    sys.exit('no stamps: is DS_LIB_PATH the timeline build?')
T = raw[used][:, 0, :].astype(np.int64)                                   # wave 0 of every workgroup (= tile)
hw = raw[used][:, 0, 7]
hwid, xcc = (hw & 0xffffffff).astype(np.int64), (hw >> 32).astype(np.int64) & 0xf
cu = (xcc << 16) | (((hwid >> 13) & 7) << 8) | (((hwid >> 12) & 1) << 7) | ((hwid >> 8) & 0xf)          # (xcc, se, sh, cu)
# s_memtime counts shader cycles and is NOT one clock across the chip: only differences on one CU are used
print(f'# {"conv3x3 " + "x".join(map(str, args.conv)) + " " if args.conv else ""}M={M} K={K} N={N}{" geglu" if args.geglu else ""}{" res" if res16 is not None else ""}: {len(used)} workgroups on {len(np.unique(cu))} CUs, '
      f'launch {ms*1e3:.1f} us by events; durations in shader cycles (s_memtime)')
names = ['prologue (entry -> first operands landed)', 'main loop', 'epilogue (arithmetic + store issue)', 'store drain (vmcnt 0)', 'whole workgroup']
d = [T[:, 1] - T[:, 0], T[:, 2] - T[:, 1], T[:, 3] - T[:, 2], T[:, 4] - T[:, 3], T[:, 4] - T[:, 0]]
for nm, v in zip(names, d):
    print(f'{nm:45s} median {np.median(v):8.0f}   p10 {np.percentile(v, 10):8.0f}   p90 {np.percentile(v, 90):8.0f}   sum per CU {v.sum() / len(np.unique(cu)):10.0f}')
gaps, partner = [], []
for c in np.unique(cu):
    idx = np.nonzero(cu == c)[0]
    order = idx[np.argsort(T[idx, 0])]
    ends = []
    for i in order:
        s_, e_ = T[i, 0], T[i, 4]
        live = [x for x in ends if x[1] > s_]
        if live:                                                          # a partner is resident: where is it in its own timeline?
            partner.append(int(np.searchsorted(T[live[-1][0], :5], s_, side='right')) - 1)
        prev = [x[1] for x in ends if x[1] <= s_]
        if prev:
            gaps.append(s_ - max(prev))
        ends.append((i, e_))
if gaps:
    print(f'slot turnover (a workgroup ends -> the next one on that CU enters): median {np.median(gaps):.0f} cycles   p90 {np.percentile(gaps, 90):.0f}')
if partner:
    print('when a workgroup enters, its resident partner is in [prologue, main, epilogue, drain, done]:', np.bincount(np.array(partner), minlength=5).tolist())
