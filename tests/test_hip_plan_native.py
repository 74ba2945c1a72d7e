"""GPU: the native launch-plan runner of the C ABI (ds_plan_create / ds_plan_add / ds_plan_run / ds_plan_graph_*; include/ds_engine.h
"Native launch plans", SURVEY.md section 8b's ds_unet_forward / ds_graph_capture_step proposal).

A full network evaluation issued by ONE ds_plan_run call must (a) match the real reference's golden, (b) be bit-identical to the same
launches issued one ctypes call at a time, (c) be bit-identical when replayed from the hipGraph the library captures itself."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu
G = os.path.join(ROOT, 'tests', 'golden')


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-6))


def test_full_cifar10_evaluation_through_ds_plan_run_only_matches_reference_golden():
    from diff_sampler_amd import _lib, ops
    from diff_sampler_amd.engine import EDMDenoiser
    lib = _lib.load()
    dev = torch.device('cuda')
    z = np.load(os.path.join(G, 'net_cifar10.npz'))
    net = EDMDenoiser.from_config('cifar10', seed=int(z['seed']))
    x, sig = torch.from_numpy(z['x']).to(dev), torch.from_numpy(z['sigma']).to(dev)
    B = x.shape[0]
    # inputs are written into the plan's input buffers; then the whole forward is ONE C call on a raw stream handle
    plan, emb_rows = net._prepare(x, sig, None)
    h = plan.native()
    assert lib.ds_plan_size(h) == len(plan.ops) == 178
    st = _lib.stream_ptr()
    plan.bufs['out'].zero_()
    assert lib.ds_plan_run(h, st) == 0 and lib.ds_plan_last_failed(h) == -1
    torch.cuda.synchronize()
    f_native = plan.bufs['out'].clone()
    # D = c_skip x + c_out F by the update kernel (what EDMDenoiser.__call__ does after the plan)
    out = torch.empty_like(x)
    a = ops.make_update_args(plan.bufs['x'], plan.bufs['x'], plan.bufs['out'], B, 3, 32, 32, None, raw=True, f_ld=0,
                             coefs=net._sigma_coefs(plan, emb_rows), coef_rows=B, sigma_data=net.sigma_data, m_out=out, store_d=False)
    ops.solver_update(a)
    torch.cuda.synchronize()
    assert _rel(out.cpu(), torch.from_numpy(z['out_vec'])) < 2e-4
    # the same launches one ctypes call at a time: bit-identical
    plan.bufs['out'].zero_()
    plan.run_python(st)
    torch.cuda.synchronize()
    assert torch.equal(plan.bufs['out'], f_native)
    # captured by the library into a hipGraph on a side stream and replayed: bit-identical
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        sp = C.c_void_p(side.cuda_stream)
        plan.run(sp)                                  # warm: every launcher has set its function attributes
        side.synchronize()
        plan.graph_capture(sp)
        plan.bufs['out'].zero_()
        plan.graph_launch(sp)
        side.synchronize()
    assert torch.equal(plan.bufs['out'], f_native)


def test_ldm_plan_runs_natively():
    """The latent-diffusion plan (LayerNorm / GEGLU / timestep-embedding launches carry scalar arguments: recorded through their
    argument structs) gives the same result through ds_plan_run as through per-launch calls."""
    from diff_sampler_amd import _lib
    from diff_sampler_amd.ldm_engine import CFGDenoiser
    dev = torch.device('cuda')
    z = np.load(os.path.join(G, 'ldm_tiny_ldm.npz'))
    net = CFGDenoiser.from_config('tiny_ldm', seed=int(z['seed']), guidance_rate=7.5)
    x, cond, uncond = (torch.from_numpy(z[k]).to(dev) for k in ('x', 'cond', 'uncond'))
    f, plan, _ = net.raw(x, 1.9, cond, uncond)
    torch.cuda.synchronize()
    a = f.clone()
    plan.run_python(_lib.stream_ptr())
    torch.cuda.synchronize()
    assert torch.equal(plan.bufs['out'], a)


def test_failing_launch_is_reported_with_its_index():
    from diff_sampler_amd import _lib
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.ds_plan_create(C.byref(h)) == 0
    ne = _lib.NoiseEmbedArgs(None, 1, None, 16, 0, None, 16)            # null pointers: the entry point rejects it
    assert lib.ds_plan_add(h, _lib.DS_OP_NOISE_EMBED, C.byref(ne), C.sizeof(ne)) == 0
    assert lib.ds_plan_run(h, _lib.stream_ptr()) != 0
    assert lib.ds_plan_last_failed(h) == 0
    lib.ds_plan_destroy(h)


def test_two_threads_run_plans_with_different_kernel_overrides_concurrently():
    """The library keeps no process-wide kernel selection: overrides travel inside each launch's ds_conv_args.tune (include/ds_engine.h).
    Two plans of the same net -- one built under `_lib.tuning(variant=2048)` (four-wave 128 x 128 tiles where the library would take eight
    half-size waves), one with the library's own choices -- run concurrently from two host threads on two streams; every evaluation of
    each must be bit-identical to that plan's single-threaded result, and the two routings must really differ."""
    import threading
    from diff_sampler_amd import _lib
    from diff_sampler_amd.engine import EDMDenoiser
    lib = _lib.load()
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(5)
    x = torch.randn(8, 3, 32, 32, generator=g).to(dev)
    sig = (torch.rand(8, generator=g) * 3 + 0.1).to(dev)
    nets, plans, want = {}, {}, {}
    for name, tune in (('tuned', dict(variant=2048)), ('default', {})):
        with _lib.tuning(**tune):
            nets[name] = EDMDenoiser.from_config('cifar10', seed=3)
            want[name] = nets[name](x, sig).clone()                 # builds the plan inside the override scope
            plans[name] = nets[name].engine.plan(8, 8)
    torch.cuda.synchronize()

    def kernel_ids(plan):
        return [lib.ds_conv_kernel_id(C.byref(op.keep[0])) for op in plan.ops if op.fn is lib.ds_conv2d_nhwc and op.keep[0].taps == 9]
    ids_t, ids_d = kernel_ids(plans['tuned']), kernel_ids(plans['default'])
    # at 8 images most 3x3 layers have at most one 128-pixel tile per CU: 1284 = eight half-size waves by default, 128 = four waves when tuned
    assert len(ids_t) == len(ids_d) > 30 and sum(i == 1284 for i in ids_d) > 30 and not any(i == 1284 for i in ids_t), (ids_t, ids_d)
    assert sum(a != b for a, b in zip(ids_t, ids_d)) > 30
    assert _rel(want['tuned'].cpu(), want['default'].cpu()) < 2e-5           # two exact-fp32 kernels: same result up to summation order
    errors = []

    def worker(name):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for i in range(12):
                    out = nets[name](x, sig)
                    s.synchronize()
                    if not torch.equal(out, want[name]):
                        errors.append((name, i))
        except Exception as e:                                       # noqa: BLE001
            errors.append((name, repr(e)))
    ts = [threading.Thread(target=worker, args=(n,)) for n in ('tuned', 'default')]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
