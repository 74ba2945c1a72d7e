"""ODE samplers with the reference's names and signatures (diff-solvers-main/solvers.py), executed as fused HIP steps.

Each reference sampler is a Python loop issuing ~5-40 ATen elementwise launches per step on 0-dim device scalars
(solvers.py:71-81, :153-168, :242-258, :334-363, :436-488, :564-596, :674-702, :788-812).  Here every sampler is
compiled into a *step program*: per step, the host computes one row of scalars from ``t_steps`` (``solver_utils``
coefficient compilers) and launches ``ds_solver_update`` once -- the kernel applies the EDM preconditioning epilogue,
forms ``d`` (or the AFS direction), writes it to the multistep history ring and produces ``x_next`` in a single pass
over HBM.  With the HIP denoiser (``engine.EDMDenoiser``) the raw network output is consumed directly, so the
``c_skip x + c_out F`` pass of ``EDMPrecond.forward`` disappears; with any other ``net`` callable (the reference
protocol: ``net(x, t, class_labels=...)`` / ``condition=...``) its denoised output is consumed instead.

Contract kept from the reference: signatures and defaults, ``**kwargs`` swallowing, returns
(tensor | trajectory | (trajectory, eps)), ``latents`` not mutated, asserts on ``max_order`` ranges, quirks such as
``ipndm_sampler(max_order=1)`` raising ``IndexError`` and ``dpm_pp_sampler`` needing ``num_steps``.
"""
from __future__ import annotations

import math
import os
from typing import List, Optional

import torch

from . import ops, solver_utils
from .solver_utils import (get_schedule, host_times, dynamic_thresholding_fn, dpmpp_coeffs, ipndm_coeffs, ipndm_v_coeffs,
                           unipc_coeffs, get_deis_coeff_list)

__all__ = ['get_denoised', 'euler_sampler', 'heun_sampler', 'dpm_2_sampler', 'ipndm_sampler', 'ipndm_v_sampler',
           'deis_sampler', 'dpm_pp_sampler', 'unipc_sampler']


def get_denoised(net, x, t, class_labels=None, condition=None, unconditional_condition=None):
    """Same dispatch as the reference (solvers.py:9-14)."""
    if hasattr(net, 'guidance_type'):
        return net(x, t, condition=condition, unconditional_condition=unconditional_condition)
    return net(x, t, class_labels=class_labels)


# The solver update of the linear solvers runs in the network head's epilogue (engine.EDMDenoiser.raw(update=...), csrc/conv3x3_thin.hip)
# instead of in a launch of its own: north_star's "fused back-to-back" taken literally.  Same arithmetic definition for both forms
# (csrc/ds_common.h), so the trajectories are equal bit for bit (tests/test_hip_samplers.py).  DS_FUSE_HEAD=0 keeps the two-launch form.
FUSE_HEAD = os.environ.get('DS_FUSE_HEAD', '1') != '0'
FUSED_UPDATES = [0]        # process-wide count of updates that ran inside a network head (tests, bench.py)


class _Run:
    """State of one sampler call: evaluation, fused update, history ring, trajectory capture."""

    def __init__(self, net, latents, class_labels, condition, unconditional_condition, t_steps, return_inters, return_eps):
        if not latents.is_cuda:
            raise RuntimeError('diff-sampler_amd runs on the MI355X only: latents must be a CUDA/HIP tensor '
                               '(there is no CPU fallback; use the reference for CPU runs)')
        self.net = net
        self.fused = bool(getattr(net, 'edm_raw_output', False))   # engine.EDMDenoiser: raw F consumed by the update kernel
        self.latents = latents.to(torch.float32).contiguous()
        self.B, self.C, self.H, self.W = self.latents.shape
        self.per_sample = self.C * self.H * self.W
        self.cl, self.cond, self.ucond = class_labels, condition, unconditional_condition
        self.ts = host_times(t_steps)
        self.sigma_data = float(getattr(net, 'sigma_data', 0.5))
        self.return_inters, self.return_eps = return_inters, return_eps
        self.inters: List[torch.Tensor] = []
        self.eps: List[torch.Tensor] = []
        self.x = torch.empty_like(self.latents)
        ops.scale(self.latents, self.ts[0], self.x)                # x_0 = latents * t_0 (solvers.py:68)
        if return_inters:
            self.inters.append(self.x.clone())
        self._f = None
        self._raw = False
        self._plan = None
        # head-fused update: evaluate() only NOTES the evaluation; the update() that follows runs the network with the update attached
        self.fuse_head = bool(self.fused and FUSE_HEAD and hasattr(net, 'head_update_ok'))
        self._pending = None
        self.fused_updates = 0              # how many updates ran inside the head (tests, bench.py)

    def _flush(self):
        """Run a noted evaluation on its own (someone needs F, or the update that follows cannot ride in the head)."""
        if self._pending is not None:
            x, sigma = self._pending
            self._pending = None
            self._f, self._plan = self.net.raw(x, sigma, self.cl)
            self._raw = True

    def new(self):
        return torch.empty_like(self.latents)

    # -- network evaluation at (x, sigma): remembers F (raw) or D (denoised) for the next update --------------------
    def evaluate(self, x, sigma: float):
        if self.fused:
            # sigma: Python float (uniform) or a device tensor [B] (per-sample, AMED second stage)
            self._flush()
            if self.fuse_head and self.net.head_update_ok(self.B, sigma, self.cl):
                self._pending = (x, sigma)                       # deferred: see update()
                self._f, self._raw = None, True
                return
            self._f, self._plan = self.net.raw(x, sigma, self.cl)
            self._raw = True
        else:
            # a denoiser that takes sigma as a host float (ldm_engine.CFGDenoiser) gets it as such: no H2D copy, no sync,
            # capturable into a hipGraph; any other callable gets the reference's 0-dim device tensor
            t = sigma if getattr(self.net, 'host_sigma_ok', False) else torch.tensor(sigma, dtype=torch.float32, device=x.device)
            self._f = get_denoised(self.net, x, t, class_labels=self.cl, condition=self.cond,
                                   unconditional_condition=self.ucond).to(torch.float32).contiguous()
            self._raw = False

    def denoised(self, x, sigma: float) -> torch.Tensor:
        """D(x; sigma) as a tensor (thresholding, denoise_to_zero)."""
        self.evaluate(x, sigma)
        if not self._raw:
            return self._f
        out = self.new()
        self.update(xe=x, xb=x, t=1.0, sigma=sigma, cx=0.0, cm=1.0, x_out=None, m_out=out, store_d=False)
        return out

    # -- one fused launch ---------------------------------------------------------------------------------------------
    def update(self, xe, xb, t, sigma, cx, cm, x_out, hist=(), ch=(), m_out=None, store_d=True, afs=False, f=None, raw=None,
               coefs=None):
        """coefs: optional device tensor [B, 8] of per-sample coefficient rows (AMED); overrides t/sigma/cx/cm/ch."""
        hc = [0.0] * 8
        hc[0], hc[1], hc[5], hc[6] = cx, cm, t, sigma
        for i, c in enumerate(ch):
            hc[2 + i] = c
        if self._pending is not None:
            px, ps = self._pending
            if f is None and raw is None and not afs and xe is px and (x_out is not None or m_out is not None):
                # the update rides in the head of the evaluation it consumes: one plan run, no update launch
                a = ops.make_update_args(xe, xb, None, self.B, self.C, self.H, self.W, x_out, raw=True, f_ld=0, hist=list(hist), hcoefs=hc,
                                         afs=False, sigma_data=self.sigma_data, m_out=m_out, store_d=store_d, coefs=coefs,
                                         coef_rows=(self.B if coefs is not None else 1))
                self._pending = None
                self._f, self._plan = self.net.raw(px, ps, self.cl, update=a)
                self._raw = True
                self.fused_updates += 1
                FUSED_UPDATES[0] += 1
                return
            self._flush()
        f = self._f if f is None else f
        raw = self._raw if raw is None else raw
        a = ops.make_update_args(xe, xb, None if afs else f, self.B, self.C, self.H, self.W, x_out, raw=(raw and not afs),
                                 f_ld=0, hist=list(hist), hcoefs=hc, afs=afs, sigma_data=self.sigma_data, m_out=m_out,
                                 store_d=store_d, coefs=coefs, coef_rows=(self.B if coefs is not None else 1))
        ops.solver_update(a)

    def dpmpp_x0_step(self, xe, xb, t, sigma, cx, cm, x_out, m_out, hist=(), ch=(), afs=False):
        """D -> dynamic threshold -> multistep combination of one data-prediction step in ONE launch (ds_dpmpp_x0_step)."""
        self._flush()                       # needs F in memory: the per-sample quantile sits between D and the combination
        hc = [0.0] * 8
        hc[0], hc[1], hc[5], hc[6] = cx, cm, t, sigma
        for i, c in enumerate(ch):
            hc[2 + i] = c
        a = ops.make_update_args(xe, xb, None if afs else self._f, self.B, self.C, self.H, self.W, x_out, raw=(self._raw and not afs),
                                 f_ld=0, hist=list(hist), hcoefs=hc, afs=afs, sigma_data=self.sigma_data, m_out=m_out, store_d=False)
        ops.dpmpp_x0_step(a)

    def record(self, x, d=None):
        if self.return_inters:
            self.inters.append(x.clone())
        if self.return_eps and d is not None:
            self.eps.append(d.clone())

    def finish(self, x, denoise_to_zero):
        if denoise_to_zero:
            x = self.denoised(x, self.ts[-1])
            if self.return_inters:
                self.inters.append(x.clone())
        dev = self.latents.device
        if self.return_inters:
            tr = torch.stack(self.inters, dim=0).to(dev)
            if self.return_eps:
                return tr, torch.stack(self.eps, dim=0).to(dev)
            return tr
        return x


_FUSED_X0_MAX = 37000      # elements per sample the one-launch data-prediction step holds in LDS (ds_dpmpp_x0_step)


def _schedule(t_steps, num_steps, sigma_min, sigma_max, latents, schedule_type, schedule_rho, net):
    if t_steps is None:
        t_steps = get_schedule(num_steps, sigma_min, sigma_max, device=latents.device, schedule_type=schedule_type,
                               schedule_rho=schedule_rho, net=net)
    return t_steps


class _Ring:
    """History of the last ``cap`` model outputs, newest first on read.  Buffers are recycled, never reallocated."""

    def __init__(self, run: _Run, cap: int):
        self.cap = cap
        self.items: List[torch.Tensor] = []     # oldest ... newest
        self.run = run
        self._spare: Optional[torch.Tensor] = None

    def slot(self) -> torch.Tensor:
        """Buffer the next model output should be written to (becomes the newest entry after ``push``)."""
        if self._spare is None:
            self._spare = self.run.new()
        return self._spare

    def push(self):
        if self.cap <= 0:
            [][-1] = None                       # noqa: reproduces the reference IndexError for max_order == 1 (solvers.py:361)
        new = self._spare
        if len(self.items) == self.cap:
            self._spare = self.items.pop(0)
        else:
            self._spare = None
        self.items.append(new)

    def newest_first(self):
        return self.items[::-1]

    def __len__(self):
        return len(self.items)


# ------------------------------------------------------------------------------------------------------------------
@torch.no_grad()
def euler_sampler(net, latents, class_labels=None, condition=None, unconditional_condition=None, num_steps=None,
                  sigma_min=0.002, sigma_max=80, schedule_type='polynomial', schedule_rho=7, afs=False,
                  denoise_to_zero=False, return_inters=False, return_eps=False, t_steps=None, **kwargs):
    """Euler / DDIM (solvers.py:19-96): x' = x + (t' - t) d, one launch per step."""
    t_steps = _schedule(t_steps, num_steps, sigma_min, sigma_max, latents, schedule_type, schedule_rho, net)
    run = _Run(net, latents, class_labels, condition, unconditional_condition, t_steps, return_inters, return_eps)
    ts, x = run.ts, run.x
    for i in range(len(ts) - 1):
        t, tn = ts[i], ts[i + 1]
        use_afs = afs and i == 0
        if not use_afs:
            run.evaluate(x, t)
        xn = run.new() if return_inters else x
        d = run.new() if return_eps else None
        run.update(xe=x, xb=x, t=t, sigma=t, cx=1.0, cm=(tn - t), x_out=xn, m_out=d, afs=use_afs)
        x = xn
        run.record(x, d)
    return run.finish(x, denoise_to_zero)


@torch.no_grad()
def heun_sampler(net, latents, class_labels=None, condition=None, unconditional_condition=None, num_steps=None,
                 sigma_min=0.002, sigma_max=80, schedule_type='polynomial', schedule_rho=7, afs=False,
                 denoise_to_zero=False, return_inters=False, return_eps=False, t_steps=None, **kwargs):
    """Heun (solvers.py:101-183): Euler predictor + trapezoid corrector, two launches per step."""
    t_steps = _schedule(t_steps, num_steps, sigma_min, sigma_max, latents, schedule_type, schedule_rho, net)
    run = _Run(net, latents, class_labels, condition, unconditional_condition, t_steps, return_inters, return_eps)
    ts, x = run.ts, run.x
    d1, xt = run.new(), run.new()
    for i in range(len(ts) - 1):
        t, tn = ts[i], ts[i + 1]
        use_afs = afs and i == 0
        if not use_afs:
            run.evaluate(x, t)
        run.update(xe=x, xb=x, t=t, sigma=t, cx=1.0, cm=(tn - t), x_out=xt, m_out=d1, afs=use_afs)
        run.evaluate(xt, tn)
        xn = run.new() if return_inters else x
        run.update(xe=xt, xb=x, t=tn, sigma=tn, cx=1.0, cm=0.5 * (tn - t), x_out=xn, hist=[d1], ch=[0.5 * (tn - t)])
        x = xn
        run.record(x, d1)
    return run.finish(x, denoise_to_zero)


@torch.no_grad()
def dpm_2_sampler(net, latents, class_labels=None, condition=None, unconditional_condition=None, num_steps=None,
                  sigma_min=0.002, sigma_max=80, schedule_type='polynomial', schedule_rho=7, afs=False,
                  denoise_to_zero=False, return_inters=False, return_eps=False, r=0.5, t_steps=None, **kwargs):
    """DPM-Solver-2 (solvers.py:188-273): midpoint at t_mid = t'^r t^(1-r)."""
    t_steps = _schedule(t_steps, num_steps, sigma_min, sigma_max, latents, schedule_type, schedule_rho, net)
    run = _Run(net, latents, class_labels, condition, unconditional_condition, t_steps, return_inters, return_eps)
    ts, x = run.ts, run.x
    d1, xt = run.new(), run.new()
    for i in range(len(ts) - 1):
        t, tn = ts[i], ts[i + 1]
        use_afs = afs and i == 0
        if not use_afs:
            run.evaluate(x, t)
        tm = (tn ** r) * (t ** (1 - r))
        run.update(xe=x, xb=x, t=t, sigma=t, cx=1.0, cm=(tm - t), x_out=xt, m_out=d1, afs=use_afs)
        run.evaluate(xt, tm)
        xn = run.new() if return_inters else x
        run.update(xe=xt, xb=x, t=tm, sigma=tm, cx=1.0, cm=(tn - t) * (1 / (2 * r)), x_out=xn, hist=[d1],
                   ch=[(tn - t) * (1 - 1 / (2 * r))])
        x = xn
        run.record(x, d1)
    return run.finish(x, denoise_to_zero)


def _multistep_d(run, afs_rule, order_coeffs, max_order, denoise_to_zero, return_inters, return_eps):
    """Shared loop of the Adams-Bashforth family (iPNDM, iPNDM_v, DEIS): x' = x + sum_j c_j d_{n-j}."""
    ts, x = run.ts, run.x
    ring = _Ring(run, max_order - 1)
    for i in range(len(ts) - 1):
        t, tn = ts[i], ts[i + 1]
        use_afs = afs_rule(i, len(ring))
        if not use_afs:
            run.evaluate(x, t)
        order = min(max_order, i + 1)
        cs = order_coeffs(i, order)
        hist = ring.newest_first()[:order - 1]
        d = ring.slot()
        xn = run.new() if return_inters else x
        run.update(xe=x, xb=x, t=t, sigma=t, cx=1.0, cm=cs[0], x_out=xn, hist=hist, ch=cs[1:], m_out=d, afs=use_afs)
        x = xn
        run.record(x, d)
        ring.push()
    return run.finish(x, denoise_to_zero)


@torch.no_grad()
def ipndm_sampler(net, latents, class_labels=None, condition=None, unconditional_condition=None, num_steps=None,
                  sigma_min=0.002, sigma_max=80, schedule_type='polynomial', schedule_rho=7, afs=False,
                  denoise_to_zero=False, return_inters=False, return_eps=False, max_order=4, t_steps=None, **kwargs):
    """Improved PNDM (solvers.py:278-374): fixed-step Adams-Bashforth on d, order <= 4."""
    assert max_order >= 1 and max_order <= 4
    t_steps = _schedule(t_steps, num_steps, sigma_min, sigma_max, latents, schedule_type, schedule_rho, net)
    run = _Run(net, latents, class_labels, condition, unconditional_condition, t_steps, return_inters, return_eps)
    ts = run.ts
    return _multistep_d(run, lambda i, nh: afs and i == 0, lambda i, order: ipndm_coeffs(order, ts[i + 1] - ts[i]),
                        max_order, denoise_to_zero, return_inters, return_eps)


@torch.no_grad()
def ipndm_v_sampler(net, latents, class_labels=None, condition=None, unconditional_condition=None, num_steps=None,
                    sigma_min=0.002, sigma_max=80, schedule_type='polynomial', schedule_rho=7, afs=False,
                    denoise_to_zero=False, return_inters=False, return_eps=False, max_order=4, t_steps=None, **kwargs):
    """Variable-step Adams-Bashforth (solvers.py:379-499)."""
    assert max_order >= 1 and max_order <= 4
    t_steps = _schedule(t_steps, num_steps, sigma_min, sigma_max, latents, schedule_type, schedule_rho, net)
    run = _Run(net, latents, class_labels, condition, unconditional_condition, t_steps, return_inters, return_eps)
    ts = run.ts
    return _multistep_d(run, lambda i, nh: afs and nh == 0, lambda i, order: ipndm_v_coeffs(order, ts, i),
                        max_order, denoise_to_zero, return_inters, return_eps)


@torch.no_grad()
def deis_sampler(net, latents, class_labels=None, condition=None, unconditional_condition=None, num_steps=None,
                 sigma_min=0.002, sigma_max=80, schedule_type='polynomial', schedule_rho=7, afs=False,
                 denoise_to_zero=False, return_inters=False, return_eps=False, max_order=4, coeff_list=None, t_steps=None,
                 **kwargs):
    """DEIS (solvers.py:504-607): x' = x + sum_j C[i][j] d_{n-j}; C from ``get_deis_coeff_list``."""
    assert max_order >= 1 and max_order <= 4
    assert coeff_list is not None
    t_steps = _schedule(t_steps, num_steps, sigma_min, sigma_max, latents, schedule_type, schedule_rho, net)
    run = _Run(net, latents, class_labels, condition, unconditional_condition, t_steps, return_inters, return_eps)
    ts = run.ts

    def coeffs(i, order):
        if order == 1:
            return [ts[i + 1] - ts[i]]
        row = coeff_list[i]
        if len(row) != order:      # the reference unpacks exactly `order` values (solvers.py:578-584)
            raise ValueError('too many values to unpack (expected %d)' % order if len(row) > order
                             else 'not enough values to unpack (expected %d, got %d)' % (order, len(row)))
        return [float(c) for c in row]

    return _multistep_d(run, lambda i, nh: afs and nh == 0, coeffs, max_order, denoise_to_zero, return_inters, return_eps)


@torch.no_grad()
def dpm_pp_sampler(net, latents, class_labels=None, condition=None, unconditional_condition=None, num_steps=None,
                   sigma_min=0.002, sigma_max=80, schedule_type='polynomial', schedule_rho=7, afs=False,
                   denoise_to_zero=False, return_inters=False, return_eps=False, max_order=3, predict_x0=True,
                   lower_order_final=True, t_steps=None, **kwargs):
    """Multistep DPM-Solver++ (solvers.py:613-713).  One launch per step in both modes: x0-prediction fuses D -> dynamic
    threshold -> 1/2M/3M update per sample (ds_dpmpp_x0_step; three launches only for samples too large for one workgroup's LDS)."""
    assert max_order >= 1 and max_order <= 3
    t_steps = _schedule(t_steps, num_steps, sigma_min, sigma_max, latents, schedule_type, schedule_rho, net)
    run = _Run(net, latents, class_labels, condition, unconditional_condition, t_steps, return_inters, return_eps)
    ts, x = run.ts, run.x
    ring = _Ring(run, 3)
    t_hist: List[float] = []
    for i in range(len(ts) - 1):
        t, tn = ts[i], ts[i + 1]
        use_afs = afs and i == 0
        if not use_afs:
            run.evaluate(x, t)
        t_hist = (t_hist + [t])[-3:]
        if lower_order_final:
            order = i + 1 if i + 1 < max_order else min(max_order, num_steps - (i + 1))
        else:
            order = min(max_order, i + 1)
        cx, cm = dpmpp_coeffs(t_hist, tn, order, predict_x0)
        xn = run.new() if return_inters else x
        d_rec = run.new() if return_eps else None
        if predict_x0 and run.per_sample <= _FUSED_X0_MAX:
            # one launch: D -> dynamic threshold -> 1/2M/3M combination (m is both the history entry and an operand)
            m = ring.slot()
            if return_eps:
                run.update(xe=x, xb=x, t=t, sigma=t, cx=0.0, cm=0.0, x_out=None, m_out=d_rec, store_d=True, afs=use_afs)
            hs = ring.newest_first()
            run.dpmpp_x0_step(xe=x, xb=x, t=t, sigma=t, cx=cx, cm=cm[0], x_out=xn, m_out=m, hist=hs[:order - 1], ch=cm[1:], afs=use_afs)
            ring.push()
        elif predict_x0:
            m = ring.slot()
            run.update(xe=x, xb=x, t=t, sigma=t, cx=0.0, cm=0.0, x_out=None, m_out=m, store_d=False, afs=use_afs)     # D
            if return_eps:
                run.update(xe=x, xb=x, t=t, sigma=t, cx=0.0, cm=0.0, x_out=None, m_out=d_rec, store_d=True, afs=use_afs)
            dynamic_thresholding_fn(m, out=m)
            ring.push()
            hs = ring.newest_first()
            run.update(xe=x, xb=x, t=1.0, sigma=t, cx=cx, cm=cm[0], x_out=xn, hist=hs[1:order], ch=cm[1:], store_d=False,
                       f=hs[0], raw=False)
        else:
            hist = ring.newest_first()[:order - 1]
            d = ring.slot()
            run.update(xe=x, xb=x, t=t, sigma=t, cx=cx, cm=cm[0], x_out=xn, hist=hist, ch=cm[1:], m_out=d, afs=use_afs)
            d_rec = d
            ring.push()
        x = xn
        run.record(x, d_rec)
    return run.finish(x, denoise_to_zero)


@torch.no_grad()
def unipc_sampler(net, latents, class_labels=None, condition=None, unconditional_condition=None, num_steps=None,
                  sigma_min=0.002, sigma_max=80, schedule_type='polynomial', schedule_rho=7, afs=False,
                  denoise_to_zero=False, return_inters=False, return_eps=False, max_order=3, predict_x0=True,
                  lower_order_final=True, variant='bh2', t_steps=None, **kwargs):
    """UniPC predictor-corrector (solvers.py:718-821; update rule solver_utils.py:174-287)."""
    assert max_order > 0 and max_order < 4
    t_steps = _schedule(t_steps, num_steps, sigma_min, sigma_max, latents, schedule_type, schedule_rho, net)
    run = _Run(net, latents, class_labels, condition, unconditional_condition, t_steps, return_inters, False)
    ts, x = run.ts, run.x

    def model_out(xq, tq, use_afs=False):
        """thresh(D) (x0 form) or d (noise form) at (xq, tq) into a fresh buffer."""
        m = run.new()
        if not use_afs:
            run.evaluate(xq, tq)
        run.update(xe=xq, xb=xq, t=tq, sigma=tq, cx=0.0, cm=0.0, x_out=None, m_out=m, store_d=(not predict_x0), afs=use_afs)
        if predict_x0:
            dynamic_thresholding_fn(m, out=m)
        return m

    ms = [model_out(x, ts[0], use_afs=afs)]       # oldest ... newest
    tsh = [ts[0]]
    for i in range(len(ts) - 1):
        tn = ts[i + 1]
        if i + 1 < max_order:
            order, use_corrector = i + 1, True
        else:
            order = min(max_order, num_steps - i - 1) if lower_order_final else max_order
            use_corrector = not (i == num_steps - 2)
        assert order <= len(ms)
        k = unipc_coeffs(tsh, tn, order, predict_x0=predict_x0, variant=variant, use_corrector=use_corrector)
        hs = ms[::-1]
        xp = run.new()
        run.update(xe=x, xb=x, t=1.0, sigma=1.0, cx=k['cx'], cm=k['pred'][0], x_out=xp, hist=hs[1:order], ch=k['pred'][1:],
                   store_d=False, f=hs[0], raw=False)
        mt = None
        xn = xp
        if use_corrector:
            mt = model_out(xp, tn)
            xn = run.new()
            c = k['corr']
            run.update(xe=x, xb=x, t=1.0, sigma=1.0, cx=k['cx'], cm=c[-1], x_out=xn, hist=hs[:order], ch=c[:order],
                       store_d=False, f=mt, raw=False)
        if i + 1 < max_order:
            ms.append(mt)
            tsh.append(tn)
        else:
            for q in range(max_order - 1):
                ms[q] = ms[q + 1]
                tsh[q] = tsh[q + 1]
            tsh[-1] = tn
            if i < num_steps - 2:
                ms[-1] = mt
        x = xn
        run.record(x)
    return run.finish(x, denoise_to_zero)
