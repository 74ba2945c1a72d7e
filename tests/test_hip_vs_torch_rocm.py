"""GPU cross-check and timing: the oracle's restatement of the reference math executed by PyTorch-ROCm (ATen + MIOpen,
fp32 NCHW -- i.e. what the reference itself would run on this GPU) against the HIP engine, on the bench workload
(CIFAR-10 SongUNet, one denoiser evaluation).  The agreement is asserted; the two timings are written to
``gpurun_out/torch_rocm_vs_hip.json`` for DESIGN.md (a report, not a pass criterion)."""
import json
import os
import sys
import time

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

import diff_sampler_amd.arch as arch  # noqa: E402


def _time(fn, iters):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


@pytest.mark.parametrize('batch', [int(os.environ.get('DS_XCHECK_BATCH', '64'))])
def test_engine_agrees_with_torch_rocm_and_report_timing(batch):
    assert torch.cuda.is_available(), 'needs the MI355X'
    from diff_sampler_amd.engine import EDMDenoiser
    from oracle.edm_net import edm_denoise
    dev = torch.device('cuda')
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    spec = arch.edm_precond_spec(**arch.NAMED_CONFIGS['cifar10'])
    params = arch.init_params(spec, seed=5)
    net = EDMDenoiser.from_config('cifar10', seed=5)
    p_dev = {k: v.to(dev) for k, v in params.items()}
    cfg = dict(arch.NAMED_CONFIGS['cifar10'])
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(batch, 3, 32, 32, generator=g) * 3.0).to(dev)
    sigma = torch.full((batch,), 2.5, device=dev)

    with torch.no_grad():
        ref = edm_denoise(p_dev, cfg, x, sigma)
        out = net(x, sigma)
        torch.cuda.synchronize()
        rel = float((out - ref).abs().max() / ref.abs().max())
        assert rel < 2e-4, rel

        for _ in range(2):
            edm_denoise(p_dev, cfg, x, sigma)
            net(x, sigma)
        t_ref = _time(lambda: edm_denoise(p_dev, cfg, x, sigma), 5)
        t_hip = _time(lambda: net(x, sigma), 5)
    rec = dict(workload='cifar10 SongUNet, one denoiser evaluation, fp32', batch=batch, rel_err=rel,
               torch_rocm_ms=t_ref * 1e3, hip_engine_ms=t_hip * 1e3, torch_rocm_img_per_s_eval=batch / t_ref,
               hip_engine_img_per_s_eval=batch / t_hip, speedup=t_ref / t_hip)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', f'torch_rocm_vs_hip_b{batch}.json'), 'w') as f:
        json.dump(rec, f, indent=1)
    print(rec)
