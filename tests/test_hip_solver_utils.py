"""GPU parity of the public helper functions of `solver_utils` (the names a caller of the reference module can import:
solver_utils.py:63,90,102,117,137,174) -- tensors in, tensors out, executed by the fused HIP update kernel -- against the
oracle's tensor-level restatements (oracle/solvers_ref.py: dpmpp_step, _unipc_update).  Tolerance 2e-5 of the output scale."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

from oracle import solvers_ref  # noqa: E402


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-6))


def _data(B=3, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 3, 16, 16, generator=g) * 5
    ms = [torch.randn(B, 3, 16, 16, generator=g) for _ in range(3)]
    ts = [torch.tensor(v) for v in (9.0, 4.1, 1.7)]
    return x, ms, ts, torch.tensor(0.6)


@pytest.mark.parametrize('predict_x0', [True, False])
@pytest.mark.parametrize('order', [1, 2, 3])
def test_dpm_pp_updates_match_oracle(order, predict_x0):
    from diff_sampler_amd import solver_utils as su
    x, ms, ts, t = _data()
    dev = torch.device('cuda')
    ref = solvers_ref.dpmpp_step(x, ms, ts, t, order, predict_x0=predict_x0)
    xd, msd, tsd, td = x.to(dev), [m.to(dev) for m in ms], [v.to(dev) for v in ts], t.to(dev)
    out = su.dpm_pp_update(xd, msd, tsd, td, order, predict_x0=predict_x0)
    assert _rel(out.cpu(), ref) < 2e-5
    fn = {1: lambda: su.dpm_solver_first_update(xd, tsd[-1], td, model_s=msd[-1], predict_x0=predict_x0),
          2: lambda: su.multistep_dpm_solver_second_update(xd, msd, tsd, td, predict_x0=predict_x0),
          3: lambda: su.multistep_dpm_solver_third_update(xd, msd, tsd, td, predict_x0=predict_x0)}[order]
    assert torch.equal(fn(), out)
    with pytest.raises(ValueError):
        su.dpm_pp_update(xd, msd, tsd, td, 4)


@pytest.mark.parametrize('order', [1, 2, 3])
def test_dpm_pp_update_per_sample_times_and_scale(order):
    """The AMED plugins call dpm_pp_update with [B,1,1,1] times and a per-sample scale (amed-solver-main/solvers_amed.py:596-611)."""
    from diff_sampler_amd import solver_utils as su
    x, ms, ts, _ = _data()
    dev = torch.device('cuda')
    B = x.shape[0]
    t = torch.tensor([0.6, 0.9, 0.33]).reshape(B, 1, 1, 1)
    t0 = torch.tensor([1.7, 2.0, 1.2]).reshape(B, 1, 1, 1)
    scale = torch.tensor([1.01, 0.99, 1.0]).reshape(B, 1, 1, 1)
    tsv = [ts[0], ts[1], t0]
    for predict_x0 in (True, False):
        ref = solvers_ref.dpmpp_step(x, ms, tsv, t, order, predict_x0=predict_x0, scale=scale, scaled_form=True)
        out = su.dpm_pp_update(x.to(dev), [m.to(dev) for m in ms], [v.to(dev) for v in tsv], t.to(dev), order, predict_x0=predict_x0,
                               scale=scale.to(dev))
        assert _rel(out.cpu(), ref) < 2e-5


@pytest.mark.parametrize('variant', ['bh1', 'bh2'])
@pytest.mark.parametrize('predict_x0', [True, False])
@pytest.mark.parametrize('order', [1, 2, 3])
def test_unipc_update_matches_oracle(order, predict_x0, variant):
    from diff_sampler_amd import solver_utils as su
    x, ms, ts, t = _data(seed=order)
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(77)
    w = torch.randn(3, 3, generator=g) * 0.3

    def net_cpu(xq, tq, labels=None):            # a stand-in denoiser: any callable net(x, t, class_labels) works
        return torch.einsum('oc,bchw->bohw', w, xq) / (1 + tq)

    def net_gpu(xq, tq, labels=None):
        return torch.einsum('oc,bchw->bohw', w.to(dev), xq) / (1 + tq)

    ev = (lambda xq, tq: solvers_ref.threshold(net_cpu(xq, tq))) if predict_x0 else net_cpu
    for corrector in (True, False):
        rx, rm = solvers_ref._unipc_update(x, ms, ts, t, order, variant, predict_x0, ev, corrector)
        ox, om = su.unipc_update(x.to(dev), [m.to(dev) for m in ms], [v.to(dev) for v in ts], t.to(dev), order, variant=variant,
                                 predict_x0=predict_x0, net=net_gpu, use_corrector=corrector)
        assert _rel(ox.cpu(), rx) < 5e-5, (order, predict_x0, variant, corrector)
        assert (om is None) == (rm is None)
        if om is not None:
            assert _rel(om.cpu(), rm) < 5e-5


def test_expand_dims_and_deis_helper_names():
    from diff_sampler_amd import solver_utils as su
    v = torch.arange(3.)
    assert su.expand_dims(v, 4).shape == (3, 1, 1, 1)
    for name in ('cal_poly', 't2alpha_fn', 'cal_intergrand', 'edm2t', 'get_deis_coeff_list', 'unipc_update', 'get_schedule'):
        assert callable(getattr(su, name))
