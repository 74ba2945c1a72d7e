"""ds_fid_moments (csrc/fid.hip) at the FID shapes: [rows, 2048] fp32 features -> mu / sigma fp64 in place.
Reports time per launch, fp64 TFLOP/s (2 * rows * dim^2; v_mfma_f64_16x16x4_f64 peak 78.6) and the HBM rate of the sigma read-modify-write
(2 * dim^2 * 8 B per launch), next to the two torch expressions of fid.py:69-71 (rocBLAS fp64 GEMM + the .to(float64) copy)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sampler_amd import _lib  # noqa: E402

lib = _lib.load()
dim = 2048
for rows in (16, 64, 128, 250, 1000):
    f = torch.randn(rows, dim, device='cuda')
    mu = torch.zeros(dim, dtype=torch.float64, device='cuda')
    sg = torch.zeros(dim, dim, dtype=torch.float64, device='cuda')

    def hip():
        _lib.check(lib.ds_fid_moments(C.c_void_p(f.data_ptr()), 0, dim, rows, dim, C.c_void_p(mu.data_ptr()), C.c_void_p(sg.data_ptr()), _lib.stream_ptr()))

    def ref():
        g = f.to(torch.float64)
        mu.add_(g.sum(0))
        sg.add_(g.T @ g)

    res = {}
    for name, fn in (('ds_fid_moments', hip), ('torch (rocBLAS)', ref)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 20
    ms = res['ds_fid_moments']
    print(f'rows={rows:5d}: ds_fid_moments {ms * 1e3:7.1f} us  {2.0 * rows * dim * dim / ms / 1e9:5.1f} TF fp64  sigma RMW {2 * dim * dim * 8 / ms / 1e6:6.0f} GB/s'
          f'   torch {res["torch (rocBLAS)"] * 1e3:7.1f} us', flush=True)
