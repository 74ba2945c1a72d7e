#!/bin/bash
# Round-2 session 32: small-batch check of the half-size-wave tiles (most layers have at most one tile per CU there): B = 8 and B = 32,
# variant 2048 = four-wave kernel, 0 = default.
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s32; mkdir -p $O
timeout 100 python - > $O/small_batch.txt 2>&1 <<'PY'
import sys, time, torch
sys.path.insert(0, '.')
from diff_sampler_amd import _lib, solvers
from diff_sampler_amd.engine import EDMDenoiser
lib = _lib.load()
net = EDMDenoiser.from_config('cifar10', seed=0)
for B in (8, 32, 64):
    lat = torch.randn(B, 3, 32, 32, device='cuda')
    res = {}
    for v in (2048, 0, 2048, 0):
        lib.ds_debug_conv_variant(v)
        net.engine._plans.clear()
        f = lambda: solvers.dpm_pp_sampler(net, lat, num_steps=11, sigma_min=0.002, sigma_max=80., schedule_type='logsnr', max_order=2, predict_x0=True, lower_order_final=True)
        f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): f()
        torch.cuda.synchronize()
        res.setdefault(v, []).append((time.perf_counter() - t0) / 3 * 1e3)
    print(f'B={B}: four-wave kernel {min(res[2048]):.2f} ms per sampler call, half-size waves {min(res[0]):.2f} ms ({100 * (min(res[2048]) / min(res[0]) - 1):+.2f} %)', flush=True)
lib.ds_debug_conv_variant(0)
PY
cat $O/small_batch.txt | grep -v amdgpu.ids
true
