"""AMED-Solver and AMED-Plugin samplers (reference: amed-solver-main/solvers_amed.py) on the HIP engine.

The reference hooks the U-Net bottleneck with ``register_forward_hook`` on every step (solvers_amed.py:7-18), feeds its
channel mean to a 9K-parameter predictor (training/networks.py:121-155) and then works on per-sample ``[B,1,1,1]``
tensors ``r, scale_dir, scale_time, t_mid``.  Here the bottleneck is an explicit plan output of the denoiser
(``EDMDenoiser.bottleneck_mean``), the predictor is one kernel (``ds_amed_predict``), the per-sample coefficient rows
of both stages are one kernel each (``ds_amed_coefs``) and each stage's update is the same fused ``ds_solver_update``
launch the other samplers use, reading its scalars per sample from the coefficient rows.

Scope: inference (``train=False``) with an ``EDMDenoiser`` net -- the configuration of BASELINE config 4.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict

import torch

from . import _lib, ops, solvers
from ._lib import AmedCoefArgs, AmedPredictor as _CPred
from .solver_utils import get_schedule, dynamic_thresholding_fn
from .solvers import _Run, _Ring, get_denoised  # noqa: F401  (get_denoised re-exported like the reference)

MODE = dict(amed=0, euler=1, ipndm=2, dpm=3, dpmpp=4)


class AMEDPredictor:
    """Device-resident AMED predictor.  Carries the solver settings as attributes like the reference module does
    (training/networks.py:88-103; amed-solver-main/sample.py:168-185 reads them back)."""

    _ATTRS = ('dataset_name', 'img_resolution', 'num_steps', 'sampler_stu', 'sampler_tea', 'M', 'guidance_type',
              'guidance_rate', 'schedule_type', 'schedule_rho', 'afs', 'scale_dir', 'scale_time', 'max_order',
              'predict_x0', 'lower_order_final')

    def __init__(self, state_dict: Dict[str, torch.Tensor], device='cuda', **settings):
        self.device = torch.device(device)
        for k in self._ATTRS:
            setattr(self, k, settings.get(k, {'afs': False, 'scale_dir': 0, 'scale_time': 0, 'predict_x0': True,
                                              'lower_order_final': True}.get(k)))
        g = lambda k: state_dict[k].detach().to(self.device, torch.float32).contiguous() if k in state_dict else None
        self.w = {k: g(k) for k in ('map_layer0.weight', 'map_layer0.bias', 'enc_layer0.weight', 'enc_layer0.bias',
                                    'enc_layer1.weight', 'enc_layer1.bias', 'fc_r.weight', 'fc_r.bias',
                                    'fc_scale_dir.weight', 'fc_scale_dir.bias', 'fc_scale_time.weight', 'fc_scale_time.bias')}
        p = lambda k: None if self.w[k] is None else C.c_void_p(self.w[k].data_ptr())
        self._c = _CPred(p('map_layer0.weight'), p('map_layer0.bias'), p('enc_layer0.weight'), p('enc_layer0.bias'),
                         p('enc_layer1.weight'), p('enc_layer1.bias'), p('fc_r.weight'), p('fc_r.bias'),
                         p('fc_scale_dir.weight') if self.scale_dir else None, p('fc_scale_dir.bias') if self.scale_dir else None,
                         p('fc_scale_time.weight') if self.scale_time else None, p('fc_scale_time.bias') if self.scale_time else None,
                         self.w['map_layer0.weight'].shape[0], self.w['enc_layer0.weight'].shape[1],
                         self.w['enc_layer0.weight'].shape[0], self.w['enc_layer1.weight'].shape[0],
                         float(self.scale_dir or 0), float(self.scale_time or 0))

    @classmethod
    def from_module(cls, module, device='cuda'):
        m = getattr(module, 'module', module)              # unwrap DDP like the reference does (solvers_amed.py:33)
        return cls(m.state_dict(), device=device, **{k: getattr(m, k, None) for k in cls._ATTRS})

    def predict(self, bott_mean: torch.Tensor, t_cur: float, t_next: float, out: torch.Tensor):
        n = bott_mean.shape[0]
        rc = _lib.load().ds_amed_predict(C.byref(self._c), C.c_void_p(bott_mean.data_ptr()), n, float(t_cur), float(t_next),
                                         C.c_void_p(out.data_ptr()), _lib.stream_ptr())
        _lib.check(rc, 'ds_amed_predict')
        return out


# ---- the two public helpers of the reference module (amed-solver-main/solvers_amed.py:7-55).  The reference's samplers call them on every
# step; here the samplers read the bottleneck from the launch plan directly (_amed_loop), and these shims exist so that code written
# against the reference module -- `from solvers_amed import init_hook, get_amed_prediction` -- keeps working on an engine net. ----------
class _BottleneckTap:
    """What ``init_hook`` hands back as ``unet_enc_out``: the reference appends the hooked block's output on every forward and reads
    ``unet_enc_out[-1]``; the engine keeps the tap of the LAST evaluation as a plan buffer, so ``[-1]`` fetches it (NCHW, fp32)."""

    def __init__(self, net, name):
        self.net, self.name, self.active = net, name, True

    def __getitem__(self, i):
        if i != -1:
            raise IndexError('only the bottleneck of the last evaluation is kept (unet_enc_out[-1])')
        if not self.active:
            raise RuntimeError('the hook was removed')
        return self.net.block_output(self.name)

    def __len__(self):
        return 1 if getattr(self.net, '_last', None) is not None else 0


class _TapHandle:
    """The ``hook`` half of ``init_hook``'s return value: ``.remove()`` like a torch RemovableHandle."""

    def __init__(self, tap):
        self.tap = tap

    def remove(self):
        self.tap.active = False


def init_hook(net, class_labels=None):
    """``unet_enc_out, hook = init_hook(net, class_labels)`` (solvers_amed.py:7-18): the U-Net bottleneck tap -- ``enc['8x8_block2']`` for
    class-conditional EDM nets, ``enc['8x8_block3']`` otherwise.  LDM / 256-pixel ADM nets (``middle_block``) are outside the engine's
    AMED scope and raise."""
    if hasattr(net, 'guidance_type') or getattr(net, 'img_resolution', 0) == 256 or not hasattr(net, 'block_output'):
        raise NotImplementedError('init_hook: the engine exposes the AMED bottleneck of EDM nets (engine.EDMDenoiser) only')
    tap = _BottleneckTap(net, 'enc.8x8_block2' if class_labels is not None else 'enc.8x8_block3')
    return tap, _TapHandle(tap)


def get_amed_prediction(AMED_predictor, t_cur, t_next, net, unet_enc_out, use_afs, batch_size):
    """``r, scale_dir, scale_time`` as ``[B, 1, 1, 1]`` tensors (solvers_amed.py:22-55): the channel mean of the last bottleneck (zeros
    under AFS, :27) through the predictor kernel; an absent head yields ones (:43,48,53-54)."""
    pred = _as_predictor(AMED_predictor, net.device if hasattr(net, 'device') else 'cuda')
    dev = pred.device
    if use_afs:
        bott = torch.zeros(batch_size, 8, 8, dtype=torch.float32, device=dev)
    elif isinstance(unet_enc_out, _BottleneckTap):                      # the plan's own NHWC buffer: no copy
        if not unet_enc_out.active:
            raise RuntimeError('the hook was removed')
        bott = net.bottleneck_mean(net._last[0], batch_size, class_cond=unet_enc_out.name.endswith('block2'))
    else:                                                               # any list of [B, C, 8, 8] tensors, as the reference's hook fills
        t = unet_enc_out[-1].to(dev, torch.float32)
        nhwc = t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).contiguous()
        bott = torch.empty(batch_size, 8, 8, dtype=torch.float32, device=dev)
        ops.channel_mean(nhwc, t.shape[1], t.shape[1], batch_size * 64, bott)
    out = torch.empty(batch_size, 4, dtype=torch.float32, device=dev)
    pred.predict(bott, float(t_cur), float(t_next), out)
    col = lambda j: out[:, j].reshape(-1, 1, 1, 1).clone()
    return col(0), col(1), col(2)


def _as_predictor(p, device):
    if p is None or isinstance(p, AMEDPredictor):
        return p
    return AMEDPredictor.from_module(p, device)


def _amed_loop(mode, net, latents, class_labels, condition, unconditional_condition, num_steps, sigma_min, sigma_max, schedule_type,
               schedule_rho, afs, denoise_to_zero, return_inters, predictor, train, max_order=None, predict_x0=True,
               lower_order_final=True):
    if train:
        raise NotImplementedError('the training branches of solvers_amed are out of scope of the HIP engine (SURVEY.md section 2, row 7)')
    t_steps = get_schedule(num_steps, sigma_min, sigma_max, device=latents.device, schedule_type=schedule_type, schedule_rho=schedule_rho, net=net)
    run = _Run(net, latents, class_labels, condition, unconditional_condition, t_steps, return_inters, False)
    run.fuse_head = False               # the bottleneck is read between an evaluation and its update: evaluations run on their own
    if not run.fused:
        raise RuntimeError('AMED samplers need the bottleneck tap of engine.EDMDenoiser (the reference hooks net.model.enc[...])')
    predictor = _as_predictor(predictor, latents.device)
    lib = _lib.load()
    dev, B = latents.device, run.B
    ts, x = run.ts, run.x
    f32 = dict(dtype=torch.float32, device=dev)
    pred = torch.empty(B, 4, **f32)
    c1, c2 = torch.empty(B, 8, **f32), torch.empty(B, 8, **f32)
    sigma2 = torch.empty(B, **f32)
    thist = torch.zeros(B, 4, **f32)
    zeros_b = None
    in_dim = predictor.w['enc_layer0.weight'].shape[1]
    m = MODE[mode]
    ring = _Ring(run, (max_order - 1) if mode == 'ipndm' else 3)
    total = 2 * num_steps - 1
    xt, d1 = run.new(), run.new()

    def coefs(stage, order, out):
        a = AmedCoefArgs(C.c_void_p(pred.data_ptr()), ts_cur, ts_next, m, stage, order, int(predict_x0),
                         C.c_void_p(thist.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(sigma2.data_ptr()) if stage == 1 else None, B)
        _lib.check(lib.ds_amed_coefs(C.byref(a), _lib.stream_ptr()), 'ds_amed_coefs')

    def pp_order(step_cur):
        if lower_order_final:
            return step_cur if step_cur < max_order else min(max_order, total - step_cur)
        return min(max_order, step_cur)

    def model_step(xe, xb, cf, order, x_out, use_afs):
        """DPM-Solver++ plugin stage: push thresh(D) or d into the ring, then the multistep combination."""
        if predict_x0:
            mbuf = ring.slot()
            run.update(xe=xe, xb=xe, t=1.0, sigma=1.0, cx=0.0, cm=0.0, x_out=None, m_out=mbuf, store_d=False, afs=use_afs, coefs=cf)
            dynamic_thresholding_fn(mbuf, out=mbuf)
            ring.push()
            hs = ring.newest_first()
            run.update(xe=xb, xb=xb, t=1.0, sigma=1.0, cx=0.0, cm=0.0, x_out=x_out, hist=hs[1:order], store_d=False, f=hs[0], raw=False,
                       coefs=cf)
        else:
            hist = ring.newest_first()[:order - 1]
            dbuf = ring.slot()
            run.update(xe=xe, xb=xb, t=1.0, sigma=1.0, cx=0.0, cm=0.0, x_out=x_out, hist=hist, m_out=dbuf, afs=use_afs, coefs=cf)
            ring.push()

    for i in range(len(ts) - 1):
        ts_cur, ts_next = ts[i], ts[i + 1]
        use_afs = (afs and len(ring) == 0) if mode in ('ipndm', 'dpmpp') else (afs and i == 0)
        # ---- stage 1: evaluation at (x, t), learned intermediate time ----------------------------------------------
        if use_afs:
            if zeros_b is None:
                zeros_b = torch.zeros(B, in_dim, **f32)
            bott = zeros_b                                             # solvers_amed.py:27
        else:
            run.evaluate(x, ts_cur)
            bott = net.bottleneck_mean(run._plan, B, class_cond=(class_labels is not None))
        predictor.predict(bott, ts_cur, ts_next, pred)
        if mode == 'ipndm':
            o1 = min(max_order, len(ring) + 1)
            coefs(1, o1, c1)
            hist = ring.newest_first()[:o1 - 1]
            dbuf = ring.slot()
            run.update(xe=x, xb=x, t=1.0, sigma=1.0, cx=0.0, cm=0.0, x_out=xt, hist=hist, m_out=dbuf, afs=use_afs, coefs=c1)
            ring.push()
        elif mode == 'dpmpp':
            coefs(1, pp_order(2 * i + 1), c1)
            model_step(x, x, c1, pp_order(2 * i + 1), xt, use_afs)
        else:
            coefs(1, 1, c1)
            run.update(xe=x, xb=x, t=1.0, sigma=1.0, cx=0.0, cm=0.0, x_out=xt, m_out=d1, afs=use_afs, coefs=c1)
        # ---- stage 2: evaluation at (x~, scale_time * t_mid), step to t_next ------------------------------------------
        run.evaluate(xt, sigma2)
        xn = run.new() if return_inters else x
        if mode == 'amed':
            coefs(2, 1, c2)
            run.update(xe=xt, xb=x, t=1.0, sigma=1.0, cx=0.0, cm=0.0, x_out=xn, coefs=c2)
        elif mode == 'euler':
            coefs(2, 1, c2)
            run.update(xe=xt, xb=xt, t=1.0, sigma=1.0, cx=0.0, cm=0.0, x_out=xn, coefs=c2)
        elif mode == 'dpm':
            coefs(2, 1, c2)
            run.update(xe=xt, xb=x, t=1.0, sigma=1.0, cx=0.0, cm=0.0, x_out=xn, hist=[d1], coefs=c2)
        elif mode == 'ipndm':
            o2 = min(max_order, len(ring) + 1)
            coefs(2, o2, c2)
            hist = ring.newest_first()[:o2 - 1]
            dbuf = ring.slot()
            run.update(xe=xt, xb=xt, t=1.0, sigma=1.0, cx=0.0, cm=0.0, x_out=xn, hist=hist, m_out=dbuf, coefs=c2)
            ring.push()
        else:
            coefs(2, pp_order(2 * i + 2), c2)
            model_step(xt, xt, c2, pp_order(2 * i + 2), xn, False)
        x = xn
        run.record(x)
    return run.finish(x, denoise_to_zero)


def amed_sampler(net, latents, class_labels=None, condition=None, unconditional_condition=None, num_steps=None, sigma_min=0.002,
                 sigma_max=80, schedule_type='polynomial', schedule_rho=7, afs=False, denoise_to_zero=False, return_inters=False,
                 AMED_predictor=None, step_idx=None, train=False, **kwargs):
    """AMED-Solver (solvers_amed.py:69-159)."""
    assert AMED_predictor is not None
    return _amed_loop('amed', net, latents, class_labels, condition, unconditional_condition, num_steps, sigma_min, sigma_max,
                      schedule_type, schedule_rho, afs, denoise_to_zero, return_inters, AMED_predictor, train)


def euler_sampler(net, latents, class_labels=None, condition=None, unconditional_condition=None, num_steps=None, sigma_min=0.002,
                  sigma_max=80, schedule_type='polynomial', schedule_rho=7, afs=False, denoise_to_zero=False, return_inters=False,
                  AMED_predictor=None, step_idx=None, train=False, **kwargs):
    """AMED-Plugin for Euler (solvers_amed.py:163-257); without a predictor it is the plain Euler sampler."""
    if AMED_predictor is None:
        return solvers.euler_sampler(net, latents, class_labels, condition, unconditional_condition, num_steps, sigma_min, sigma_max,
                                     schedule_type, schedule_rho, afs, denoise_to_zero, return_inters)
    return _amed_loop('euler', net, latents, class_labels, condition, unconditional_condition, num_steps, sigma_min, sigma_max,
                      schedule_type, schedule_rho, afs, denoise_to_zero, return_inters, AMED_predictor, train)


def ipndm_sampler(net, latents, class_labels=None, condition=None, unconditional_condition=None, num_steps=None, sigma_min=0.002,
                  sigma_max=80, schedule_type='polynomial', schedule_rho=7, afs=False, denoise_to_zero=False, return_inters=False,
                  AMED_predictor=None, train=False, max_order=4, buffer_model=[], **kwargs):
    """AMED-Plugin for iPNDM (solvers_amed.py:262-396)."""
    assert max_order >= 1 and max_order <= 4
    if AMED_predictor is None:
        return solvers.ipndm_sampler(net, latents, class_labels, condition, unconditional_condition, num_steps, sigma_min, sigma_max,
                                     schedule_type, schedule_rho, afs, denoise_to_zero, return_inters, max_order=max_order)
    return _amed_loop('ipndm', net, latents, class_labels, condition, unconditional_condition, num_steps, sigma_min, sigma_max,
                      schedule_type, schedule_rho, afs, denoise_to_zero, return_inters, AMED_predictor, train, max_order=max_order)


def dpm_2_sampler(net, latents, class_labels=None, condition=None, unconditional_condition=None, num_steps=None, sigma_min=0.002,
                  sigma_max=80, schedule_type='polynomial', schedule_rho=7, afs=False, denoise_to_zero=False, return_inters=False,
                  AMED_predictor=None, step_idx=None, train=False, r=0.5, **kwargs):
    """AMED-Plugin for DPM-Solver-2 (solvers_amed.py:400-494)."""
    if AMED_predictor is None:
        return solvers.dpm_2_sampler(net, latents, class_labels, condition, unconditional_condition, num_steps, sigma_min, sigma_max,
                                     schedule_type, schedule_rho, afs, denoise_to_zero, return_inters, r=r)
    return _amed_loop('dpm', net, latents, class_labels, condition, unconditional_condition, num_steps, sigma_min, sigma_max,
                      schedule_type, schedule_rho, afs, denoise_to_zero, return_inters, AMED_predictor, train)


def dpm_pp_sampler(net, latents, class_labels=None, condition=None, unconditional_condition=None, num_steps=None, sigma_min=0.002,
                   sigma_max=80, schedule_type='polynomial', schedule_rho=7, afs=False, denoise_to_zero=False, return_inters=False,
                   AMED_predictor=None, step_idx=None, train=False, buffer_model=[], buffer_t=[], max_order=3, predict_x0=True,
                   lower_order_final=True, **kwargs):
    """AMED-Plugin for multistep DPM-Solver++ (solvers_amed.py:498-631)."""
    assert max_order >= 1 and max_order <= 3
    if AMED_predictor is None:
        return solvers.dpm_pp_sampler(net, latents, class_labels, condition, unconditional_condition, num_steps, sigma_min, sigma_max,
                                      schedule_type, schedule_rho, afs, denoise_to_zero, return_inters, max_order=max_order,
                                      predict_x0=predict_x0, lower_order_final=lower_order_final)
    return _amed_loop('dpmpp', net, latents, class_labels, condition, unconditional_condition, num_steps, sigma_min, sigma_max,
                      schedule_type, schedule_rho, afs, denoise_to_zero, return_inters, AMED_predictor, train, max_order=max_order,
                      predict_x0=predict_x0, lower_order_final=lower_order_final)


def heun_sampler(net, latents, class_labels=None, condition=None, unconditional_condition=None, num_steps=None, sigma_min=0.002,
                 sigma_max=80, schedule_type='polynomial', schedule_rho=7, afs=False, denoise_to_zero=False, return_inters=False,
                 **kwargs):
    """Teacher sampler of AMED training (solvers_amed.py:635-708) = the Heun sampler."""
    return solvers.heun_sampler(net, latents, class_labels, condition, unconditional_condition, num_steps, sigma_min, sigma_max,
                                schedule_type, schedule_rho, afs, denoise_to_zero, return_inters)
