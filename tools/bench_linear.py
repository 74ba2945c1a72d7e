"""1x1 / Linear contraction shapes of the SD-1.5 SpatialTransformer (32 U-Net images): 8-wave LDS-DMA kernel vs generic kernel."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sampler_amd import _lib, ops  # noqa: E402
from diff_sampler_amd._lib import ConvArgs  # noqa: E402

SHAPES = [('ff.proj 64x64', 131072, 320, 2560), ('ff.out 64x64', 131072, 1280, 320), ('qkv 64x64', 131072, 320, 960),
          ('to_out 64x64', 131072, 320, 320), ('ff.proj 32x32', 32768, 640, 5120), ('ff.out 32x32', 32768, 2560, 640),
          ('qkv 32x32', 32768, 640, 1920), ('ff.proj 16x16', 8192, 1280, 10240), ('ff.out 16x16', 8192, 5120, 1280),
          ('cifar qkv 16x16 B=256', 65536, 256, 768), ('adm qkv 32x32 B=128', 131072, 384, 1152)]
lib = _lib.load()
st = _lib.stream_ptr()
for label, M, K, N in SHAPES:
    x = torch.randn(M, K, device='cuda')
    wp = ops.pack_linear_weight(torch.randn(N, K, device='cuda') / K ** 0.5)
    bias = torch.randn(N, device='cuda')
    res = torch.randn(M, N, device='cuda')
    out = torch.zeros(M, N, device='cuda')
    a = ConvArgs(x.data_ptr(), None, K, 0, K, 0, M, 1, 1, 1, wp.data_ptr(), N, bias.data_ptr(), None, 0, 1, res.data_ptr(), N, 1.0, 0,
                 out.data_ptr(), N)
    row = []
    outs = []
    for force in (0, 1):
        a.tune.mode = force                   # ds_conv_tune.mode 1: the generic gather kernel
        kid = lib.ds_conv_kernel_id(C.byref(a))
        assert lib.ds_conv2d_nhwc(C.byref(a), st) == 0
        torch.cuda.synchronize()
        outs.append(out.clone())
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            lib.ds_conv2d_nhwc(C.byref(a), st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        row.append(f'{("dma8" if kid == 2561 else "generic") if force == 0 else "generic (forced)"}: {ms*1e3:7.0f} us {2.0*M*K*N/ms/1e9:6.1f} TF')
    a.tune.mode = 0
    err = float((outs[0] - outs[1]).abs().max() / outs[1].abs().max())
    print(f'{label:24s} M={M:6d} K={K:5d} N={N:5d}  ' + '   '.join(row) + f'   rel diff {err:.1e}', flush=True)
