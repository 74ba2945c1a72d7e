"""GPU parity of the AMED-Solver / AMED-Plugin samplers against trajectories produced by the real reference
(amed-solver-main/solvers_amed.py with a random-init AMED_predictor; tests/golden/sampler_tiny_song_amed*.npz).

Tolerance 1e-3 of the trajectory scale: the predictor outputs (r, scale_dir, scale_time) feed powf/expm1f per sample
on the device in fp32, and t_mid enters the second network evaluation, so rounding differences are amplified a little
more than in the fixed-schedule samplers (observed ~1e-5)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

from oracle import cases  # noqa: E402

G = os.path.join(ROOT, 'tests', 'golden')
TOL = 1e-3


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-6))


@pytest.mark.parametrize('netname', ['tiny_song_amed', 'tiny_song_amed_cond'])
def test_amed_samplers_match_reference(netname):
    from diff_sampler_amd import solvers_amed
    from diff_sampler_amd.engine import EDMDenoiser
    dev = torch.device('cuda')
    z = np.load(os.path.join(G, f'sampler_{netname}.npz'))
    net = EDMDenoiser.from_config(netname, seed=int(z['seed']))
    latents = torch.from_numpy(z['latents']).to(dev)
    lab = torch.from_numpy(z['labels']).to(dev) if z['labels'].size else None
    fns = dict(amed=solvers_amed.amed_sampler, euler=solvers_amed.euler_sampler, ipndm=solvers_amed.ipndm_sampler,
               dpm=solvers_amed.dpm_2_sampler, dpmpp=solvers_amed.dpm_pp_sampler)
    checked = 0
    for tag, stu, n, kind, rho, afs, pk, sk in cases.AMED_CASES:
        if f'{tag}_inters' not in z.files:
            continue
        pp = cases.amed_predictor_params(int(z[f'{tag}_pseed']), pk['scale_dir'], pk['scale_time'])
        pred = solvers_amed.AMEDPredictor(pp, device=dev, num_steps=n, sampler_stu=stu, schedule_type=kind, schedule_rho=rho,
                                          afs=afs, **pk)
        inters = fns[stu](net, latents, class_labels=lab, num_steps=n, sigma_min=0.002, sigma_max=80., schedule_type=kind,
                          schedule_rho=rho, afs=afs, return_inters=True, AMED_predictor=pred, **sk)
        torch.cuda.synchronize()
        gold = torch.from_numpy(z[f'{tag}_inters'])
        assert tuple(inters.shape) == tuple(gold.shape), (tag, inters.shape)
        err = _rel(inters.cpu(), gold)
        assert err < TOL, (netname, tag, err)
        out = fns[stu](net, latents, class_labels=lab, num_steps=n, sigma_min=0.002, sigma_max=80., schedule_type=kind,
                       schedule_rho=rho, afs=afs, AMED_predictor=pred, **sk)
        assert _rel(out.cpu(), gold[-1]) < TOL
        checked += 1
    assert checked >= 2


def test_amed_predictor_kernel_matches_oracle():
    from diff_sampler_amd import solvers_amed
    from oracle import solvers_ref
    dev = torch.device('cuda')
    for sd, st in [(0.01, 0), (0.05, 0.05), (0, 0)]:
        pp = cases.amed_predictor_params(77, sd, st)
        pred = solvers_amed.AMEDPredictor(pp, device=dev, scale_dir=sd, scale_time=st)
        g = torch.Generator().manual_seed(2)
        bott = torch.randn(5, 8, 8, generator=g)
        out = torch.empty(5, 4, device=dev)
        pred.predict(bott.to(dev), 3.3, 0.9, out)
        r, sdir, stime = solvers_ref.amed_predict(pp, dict(scale_dir=sd, scale_time=st), bott, torch.tensor(3.3), torch.tensor(0.9))
        o = out.cpu()
        assert torch.allclose(o[:, 0], r.flatten(), rtol=1e-5, atol=1e-6)
        assert torch.allclose(o[:, 1], sdir.flatten(), rtol=1e-5, atol=1e-6)
        assert torch.allclose(o[:, 2], stime.flatten(), rtol=1e-5, atol=1e-6)
        tm = (torch.tensor(0.9) ** r.flatten()) * (torch.tensor(3.3) ** (1 - r.flatten()))
        assert torch.allclose(o[:, 3], tm, rtol=1e-5)


def test_amed_requires_hip_denoiser():
    from diff_sampler_amd import solvers_amed
    pp = cases.amed_predictor_params(1, 0.01, 0)
    pred = solvers_amed.AMEDPredictor(pp, device='cuda', scale_dir=0.01)
    with pytest.raises(RuntimeError, match='bottleneck tap'):
        solvers_amed.amed_sampler(lambda x, t, class_labels=None: x, torch.zeros(1, 3, 16, 16, device='cuda'), num_steps=3,
                                  AMED_predictor=pred)


def test_block_output_is_the_bottleneck_a_forward_hook_would_see():
    """`EDMDenoiser.block_output('enc.8x8_block3')` (what `persistence_hook` hands to the forward hooks of amed-solver-main's
    `init_hook`, solvers_amed.py:7-18) against the oracle's tap of the same block, and its channel mean against `bottleneck_mean`
    (the kernel the HIP AMED samplers use, solvers_amed.py:24-28)."""
    import diff_sampler_amd.arch as arch
    from diff_sampler_amd.engine import EDMDenoiser
    from oracle.edm_net import OracleNet
    dev = torch.device('cuda')
    kw = dict(arch.NAMED_CONFIGS['tiny_song_amed'])
    params = arch.init_params(arch.edm_precond_spec(**kw), seed=5)
    net = EDMDenoiser(arch.edm_precond_spec(**kw), params)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(3, 3, 16, 16, generator=g) * 2.0
    d = net(x.to(dev), 1.7)
    tap = net.block_output('enc.8x8_block3')
    assert tuple(tap.shape) == (3, 64, 8, 8) and tap.is_contiguous()
    plan, B = net._last
    assert _rel(tap.mean(1).cpu(), net.bottleneck_mean(plan, B, class_cond=False).cpu()) < 1e-5
    ref_net = OracleNet(params, kw)
    with torch.no_grad():
        want_d = ref_net(x, torch.tensor(1.7))
        want_tap = ref_net.last_bottleneck                # the oracle's tap of enc.8x8_block3 (no labels), NCHW
    assert _rel(d.cpu(), want_d) < 2e-4
    assert tuple(want_tap.shape) == tuple(tap.shape) and _rel(tap.cpu(), want_tap) < 2e-4


@pytest.mark.parametrize('netname', ['tiny_song_amed', 'tiny_song_amed_cond'])
def test_init_hook_and_get_amed_prediction_shims_follow_the_reference_helpers(netname):
    """amed-solver-main/solvers_amed.py:7-55: `unet_enc_out, hook = init_hook(net, class_labels)` taps the bottleneck block of the last
    evaluation and `get_amed_prediction` turns it into [B,1,1,1] tensors r / scale_dir / scale_time.  Checked against the oracle: its
    bottleneck (oracle.edm_net.OracleNet.last_bottleneck) and its predictor restatement (oracle.solvers_ref.amed_predict)."""
    import diff_sampler_amd.arch as arch
    from diff_sampler_amd import solvers_amed
    from diff_sampler_amd.engine import EDMDenoiser
    from oracle import solvers_ref
    from oracle.edm_net import OracleNet
    dev = torch.device('cuda')
    kw = dict(arch.NAMED_CONFIGS[netname])
    spec = arch.edm_precond_spec(**kw)
    params = arch.init_params(spec, seed=8)
    net = EDMDenoiser(spec, params)
    g = torch.Generator().manual_seed(4)
    B = 3
    x = torch.randn(B, spec.in_channels, spec.img_resolution, spec.img_resolution, generator=g) * 2.0
    lab = torch.eye(spec.label_dim)[torch.randint(spec.label_dim, (B,), generator=g)] if spec.label_dim else None
    tap, hook = solvers_amed.init_hook(net, class_labels=lab)
    assert len(tap) == 0
    net(x.to(dev), torch.tensor(2.0, device=dev), class_labels=(lab.to(dev) if lab is not None else None))
    onet = OracleNet(params, kw)
    with torch.no_grad():
        onet(x, torch.tensor(2.0), class_labels=lab)
    bott = onet.last_bottleneck                                # [B, C, 8, 8] of the block the reference hooks
    assert len(tap) == 1 and _rel(tap[-1].cpu(), bott) < 2e-4
    pk = dict(scale_dir=0.05, scale_time=0.05)
    pp = cases.amed_predictor_params(91, pk['scale_dir'], pk['scale_time'])
    pred = solvers_amed.AMEDPredictor(pp, device=dev, **pk)
    for use_afs in (False, True):
        r, sd, st = solvers_amed.get_amed_prediction(pred, torch.tensor(2.0), torch.tensor(0.7), net, tap, use_afs, B)
        assert r.shape == sd.shape == st.shape == (B, 1, 1, 1)
        bm = torch.zeros(B, 8, 8) if use_afs else bott.mean(dim=1)
        rr, rsd, rst = solvers_ref.amed_predict(pp, pk, bm, torch.tensor(2.0), torch.tensor(0.7))
        for got, want in ((r, rr), (sd, rsd), (st, rst)):
            assert torch.allclose(got.cpu().flatten(), want.flatten(), rtol=2e-4, atol=1e-5)
    # a plain list of NCHW tensors (what the reference's own hook_fn fills) is accepted too
    r2 = solvers_amed.get_amed_prediction(pred, torch.tensor(2.0), torch.tensor(0.7), net, [tap[-1]], False, B)[0]
    assert torch.equal(r2, solvers_amed.get_amed_prediction(pred, torch.tensor(2.0), torch.tensor(0.7), net, tap, False, B)[0])
    hook.remove()
    with pytest.raises(RuntimeError):
        tap[-1]
