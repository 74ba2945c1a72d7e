"""Host side of the solvers: time schedules and the *coefficient compilers* that turn each ODE solver step into one
row of scalars for the fused HIP update kernel (``ds_solver_update``), plus API-compatible wrappers of the helper
functions other code imports from the reference's ``solver_utils`` (``get_schedule``, ``expand_dims``,
``dynamic_thresholding_fn``, ``dpm_pp_update`` and its three order-specific updates, ``unipc_update``, ``edm2t``, ``cal_poly``,
``t2alpha_fn``, ``cal_intergrand``, ``get_deis_coeff_list``) -- every public name of the reference module exists here.

Reference: diff-solvers-main/solver_utils.py (schedules :6-52, thresholding :77-86, DPM-Solver++ :90-163, UniPC
:174-287, DEIS :297-400), gits-main/solver_utils.py:52-53 (``dp_list``), amed-solver-main/solver_utils.py:90-160
(``scale``).  Design difference: the reference evaluates every scalar as a 0-dim device tensor inside the sampling
loop (dozens of tiny launches per step); here every scalar that depends only on ``t_steps`` is computed once on the
host (float64 from the fp32 schedule values) and handed to the kernel by value, so a step is exactly one launch.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import ops

# ------------------------------------------------------------------------------------------------------------------
# Time schedules.  Computed on the host with the same fp32 operation order as the reference so the values are
# bit-identical to the reference run on CPU, then moved to ``device``.

def _sched_polynomial(n, smin, smax, rho):
    i = torch.arange(n)
    return (smax ** (1 / rho) + i / (n - 1) * (smin ** (1 / rho) - smax ** (1 / rho))) ** rho


def _sched_logsnr(n, smin, smax, rho):
    top = -1 * torch.log(torch.tensor(smin))
    bot = -1 * torch.log(torch.tensor(smax))
    return (-torch.linspace(bot.item(), top.item(), steps=n)).exp()


def _vp_constants(smin=0.002, smax=80, eps_s=1e-3):
    beta_d = 2 * (np.log(torch.tensor(smin).cpu() ** 2 + 1) / eps_s - np.log(torch.tensor(smax).cpu() ** 2 + 1)) / (eps_s - 1)
    beta_min = np.log(torch.tensor(smax).cpu() ** 2 + 1) - 0.5 * beta_d
    return beta_d, beta_min


def _sched_time_uniform(n, smin, smax, rho):
    eps_s = 1e-3
    beta_d, beta_min = _vp_constants(smin, smax, eps_s)
    i = torch.arange(n)
    tau = (1 + i / (n - 1) * (eps_s ** (1 / rho) - 1)) ** rho
    return (np.e ** (0.5 * beta_d * (tau ** 2) + beta_min * tau) - 1) ** 0.5


_SCHEDULES = {'polynomial': _sched_polynomial, 'logsnr': _sched_logsnr, 'time_uniform': _sched_time_uniform}


def get_schedule(num_steps, sigma_min, sigma_max, device=None, schedule_type='polynomial', schedule_rho=7, net=None,
                 dp_list=None):
    """Same contract as the reference ``get_schedule`` (+ GITS ``dp_list``): Tensor[num_steps] on ``device``."""
    if schedule_type in _SCHEDULES:
        t_steps = _SCHEDULES[schedule_type](num_steps, sigma_min, sigma_max, schedule_rho)
    elif schedule_type == 'discrete':
        assert net is not None
        lo = net.sigma_inv(torch.tensor(sigma_min, device=device))
        hi = net.sigma_inv(torch.tensor(sigma_max, device=device))
        i = torch.arange(num_steps, device=device)
        t_steps = net.sigma((hi + i / (num_steps - 1) * (lo ** (1 / schedule_rho) - hi)) ** schedule_rho)
    else:
        raise ValueError("Got wrong schedule type {}".format(schedule_type))
    if dp_list is not None:
        t_steps = t_steps[dp_list]
    return t_steps.to(device)


def host_times(t_steps) -> List[float]:
    """fp32 schedule values as Python floats (one D2H copy per sampler call)."""
    return [float(v) for v in torch.as_tensor(t_steps).detach().to('cpu', torch.float32).tolist()]


# ------------------------------------------------------------------------------------------------------------------
# Dynamic thresholding and generic linear combinations on the device.

def dynamic_thresholding_fn(x0: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """clamp(x0, -s, s) / s with s = max(quantile_0.995(|x0|) per sample, 1) -- HIP radix-select kernel."""
    x0 = x0.contiguous()
    out = torch.empty_like(x0) if out is None else out
    ops.dynamic_threshold(x0, out, x0.shape[0], x0[0].numel(), 0.995)
    return out


def lincomb(cx, x, terms: Sequence, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = cx * x + sum_i c_i * T_i (up to 4 terms) in one pass of the fused update kernel."""
    assert 1 <= len(terms) <= 4
    x = x.contiguous()
    B, C, H, W = x.shape
    out = torch.empty_like(x) if out is None else out
    (c0, t0), rest = terms[0], list(terms[1:])
    hc = [0.0] * 8
    hc[0], hc[1], hc[5] = cx, c0, 1.0
    for i, (c, _) in enumerate(rest):
        hc[2 + i] = c
    a = ops.make_update_args(x, x, t0.contiguous(), B, C, H, W, out, raw=False, hist=[t.contiguous() for _, t in rest],
                             hcoefs=hc, store_d=False)
    ops.solver_update(a)
    return out


# ------------------------------------------------------------------------------------------------------------------
# DPM-Solver++ coefficient compiler (VE form, lambda = -log sigma).

def dpmpp_coeffs(t_hist: Sequence[float], t_next: float, order: int, predict_x0: bool = True, scale: float = 1.0):
    """Scalars (cx, [c_m0, c_m1, c_m2]) such that x' = cx*x + sum_j c_mj * m_{n-j}; m_0 newest.

    Expands solver_utils.py:102-163 (and the ``scale`` variant of amed-solver-main) in closed form."""
    lam = lambda s: -math.log(s)
    t0 = t_hist[-1]
    h = lam(t_next) - lam(t0)
    phi1 = math.expm1(-h) if predict_x0 else math.expm1(h)
    sgn_x = (t_next / t0) if predict_x0 else 1.0
    tf = 1.0 if predict_x0 else t_next           # the noise-prediction form carries a factor t
    if order == 1:
        return sgn_x, [-scale * tf * phi1]
    r0 = (lam(t0) - lam(t_hist[-2])) / h
    if order == 2:
        c0 = -scale * tf * (phi1 + 0.5 * phi1 / r0)
        c1 = scale * tf * (0.5 * phi1 / r0)
        return sgn_x, [c0, c1]
    if order == 3:
        r1 = (lam(t_hist[-2]) - lam(t_hist[-3])) / h
        phi2 = phi1 / h + 1.0 if predict_x0 else phi1 / h - 1.0
        phi3 = phi2 / h - 0.5
        g = r0 / (r0 + r1)
        q = 1.0 / (r0 + r1)
        # x' = sgn_x x - tf phi1 m0 + s2 tf phi2 D1 - tf phi3 D2,  s2 = +1 (x0 form) / -1 (noise form)
        s2 = 1.0 if predict_x0 else -1.0
        A = s2 * phi2 * (1 + g) - phi3 * q        # multiplies D1_0 = (m0 - m1)/r0
        Bq = -s2 * phi2 * g + phi3 * q            # multiplies D1_1 = (m1 - m2)/r1
        c0 = scale * tf * (-phi1 + A / r0)
        c1 = scale * tf * (-A / r0 + Bq / r1)
        c2 = scale * tf * (-Bq / r1)
        return sgn_x, [c0, c1, c2]
    raise ValueError("Solver order must be 1 or 2 or 3, got {}".format(order))


def expand_dims(v, dims):
    """[N] -> [N, 1, ..., 1] with ``dims`` dimensions (solver_utils.py:63-74); a view, no kernel."""
    return v[(...,) + (None,) * (dims - 1)]


def _times(*ts):
    """Host values of solver times given as Python floats, 0-dim tensors or per-sample tensors ([B], [B,1,1,1]).
    Returns (rows, list of lists): rows == 1 when every time is a scalar, else the common per-sample length."""
    vals = [[float(v) for v in torch.as_tensor(t).detach().reshape(-1).to('cpu', torch.float64).tolist()] for t in ts]
    rows = max(len(v) for v in vals)
    assert all(len(v) in (1, rows) for v in vals), 'per-sample times must agree in length'
    return rows, [v * rows if len(v) == 1 else v for v in vals]


def _dpmpp_lincomb(x, models_newest_first, t_hist, t, order, predict_x0, scale):
    """x' = cx x + sum_j c_j m_{n-j} in ONE fused launch.  Scalar times: coefficients by value; per-sample times (the AMED
    plugins pass [B,1,1,1] tensors, amed-solver-main/solvers_amed.py:596-611): a [B, 8] coefficient table, one row per image."""
    rows, vals = _times(*t_hist, t, scale)
    th, tn, sc = vals[:-2], vals[-2], vals[-1]
    if rows == 1:
        cx, cm = dpmpp_coeffs([v[0] for v in th], tn[0], order, predict_x0, sc[0])
        return lincomb(cx, x, [(cm[j], models_newest_first[j]) for j in range(order)])
    assert rows == x.shape[0], 'one time per sample'
    table = torch.zeros(rows, 8, dtype=torch.float32)
    for b in range(rows):
        cx, cm = dpmpp_coeffs([v[b] for v in th], tn[b], order, predict_x0, sc[b])
        table[b, 0], table[b, 5] = cx, 1.0
        for j, c in enumerate(cm):
            table[b, 1 + j] = c
    x = x.contiguous()
    B, C, H, W = x.shape
    out = torch.empty_like(x)
    ms = [m.contiguous() for m in models_newest_first[:order]]
    a = ops.make_update_args(x, x, ms[0], B, C, H, W, out, raw=False, hist=ms[1:], coefs=table.to(x.device), coef_rows=rows,
                             store_d=False)
    ops.solver_update(a)
    return out


def dpm_solver_first_update(x, s, t, model_s=None, predict_x0=True, scale=1):
    """DPM-Solver++ order 1 from time ``s`` to ``t`` (solver_utils.py:102-113; ``scale``: amed-solver-main :102)."""
    return _dpmpp_lincomb(x, [model_s], [s], t, 1, predict_x0, scale)


def multistep_dpm_solver_second_update(x, model_prev_list, t_prev_list, t, predict_x0=True, scale=1):
    """DPM-Solver++(2M) (solver_utils.py:117-133)."""
    return _dpmpp_lincomb(x, [model_prev_list[-1], model_prev_list[-2]], list(t_prev_list[-2:]), t, 2, predict_x0, scale)


def multistep_dpm_solver_third_update(x, model_prev_list, t_prev_list, t, predict_x0=True, scale=1):
    """DPM-Solver++(3M) (solver_utils.py:137-163)."""
    return _dpmpp_lincomb(x, [model_prev_list[-1], model_prev_list[-2], model_prev_list[-3]], list(t_prev_list[-3:]), t, 3,
                          predict_x0, scale)


def dpm_pp_update(x, model_prev_list, t_prev_list, t, order, predict_x0=True, scale=1):
    """API-compatible with the reference ``dpm_pp_update`` (solver_utils.py:90-98; tensors in, tensor out); one fused launch."""
    if order == 1:
        return dpm_solver_first_update(x, t_prev_list[-1], t, model_s=model_prev_list[-1], predict_x0=predict_x0, scale=scale)
    if order == 2:
        return multistep_dpm_solver_second_update(x, model_prev_list, t_prev_list, t, predict_x0=predict_x0, scale=scale)
    if order == 3:
        return multistep_dpm_solver_third_update(x, model_prev_list, t_prev_list, t, predict_x0=predict_x0, scale=scale)
    raise ValueError("Solver order must be 1 or 2 or 3, got {}".format(order))


# ------------------------------------------------------------------------------------------------------------------
# iPNDM (fixed-step Adams-Bashforth) and its variable-step version.

_AB_WEIGHTS = {1: ([1.0], 1.0), 2: ([3.0, -1.0], 2.0), 3: ([23.0, -16.0, 5.0], 12.0), 4: ([55.0, -59.0, 37.0, -9.0], 24.0)}


def ipndm_coeffs(order: int, step: float) -> List[float]:
    w, den = _AB_WEIGHTS[order]
    return [step * v / den for v in w]


def ipndm_v_coeffs(order: int, ts: Sequence[float], i: int) -> List[float]:
    """Variable-step Adams-Bashforth weights (solvers.py:447-476), times the step length."""
    hn = ts[i + 1] - ts[i]
    if order == 1:
        return [hn]
    h1 = ts[i] - ts[i - 1]
    if order == 2:
        return [hn * (2 + hn / h1) / 2, hn * (-(hn / h1) / 2)]
    h2 = ts[i - 1] - ts[i - 2]
    tmp1 = (1 - hn / (3 * (hn + h1)) * (hn * (hn + h1)) / (h1 * (h1 + h2))) / 2
    if order == 3:
        c1 = (2 + hn / h1) / 2 + tmp1
        c2 = -(hn / h1) / 2 - (1 + h1 / h2) * tmp1
        c3 = tmp1 * h1 / h2
        return [hn * c1, hn * c2, hn * c3]
    h3 = ts[i - 2] - ts[i - 3]
    tmp2 = ((1 - hn / (3 * (hn + h1))) / 2 + (1 - hn / (2 * (hn + h1))) * hn / (6 * (hn + h1 + h2))) \
        * (hn * (hn + h1) * (hn + h1 + h2)) / (h1 * (h1 + h2) * (h1 + h2 + h3))
    k = h1 * (h1 + h2) / (h2 * (h2 + h3))
    c1 = (2 + hn / h1) / 2 + tmp1 + tmp2
    c2 = -(hn / h1) / 2 - (1 + h1 / h2) * tmp1 - (1 + h1 / h2 + k) * tmp2
    c3 = tmp1 * h1 / h2 + (h1 / h2 + k * (1 + h2 / h3)) * tmp2
    c4 = -tmp2 * k * h1 / h2
    return [hn * c1, hn * c2, hn * c3, hn * c4]


# ------------------------------------------------------------------------------------------------------------------
# DEIS coefficient tables (host, once per schedule).

def edm2t(edm_steps, epsilon_s=1e-3, sigma_min=0.002, sigma_max=80):
    beta_d, beta_min = _vp_constants(sigma_min, sigma_max, epsilon_s)
    s = torch.as_tensor(edm_steps).detach().cpu()
    t = ((beta_min ** 2 + 2 * beta_d * (s ** 2 + 1).log()).sqrt() - beta_min) / beta_d
    return t, beta_min, beta_d + beta_min


def get_deis_coeff_list(t_steps, max_order, N=10000, deis_mode='tab'):
    """List (one entry per step) of per-order coefficient lists, like the reference (solver_utils.py:335-400).

    'tab': the same N-point rectangle rule on the same fp32 grid, accumulated in float64 with the analytic
    d(log alpha)/d(tau) (the reference differentiates with autograd and sums in fp32; values agree to ~1e-6 rel).
    'rhoab': closed-form polynomial integrals; reproduces the reference's behaviour for i >= 4 (last row reused)."""
    out: list = []
    if deis_mode == 'tab':
        tau, b0, b1 = edm2t(t_steps)
        b0, b1 = float(b0), float(b1)
        tau64 = tau.double().numpy()
        for i in range(len(tau) - 1):
            order = min(i + 1, max_order)
            if order == 1:
                out.append([])
                continue
            grid = torch.linspace(tau[i], tau[i + 1], N).double().numpy()        # fp32 grid values, as the reference
            dtau = (float(tau[i + 1]) - float(tau[i])) / N
            alpha = np.exp(-0.5 * grid ** 2 * (b1 - b0) - grid * b0)
            dlog = -grid * (b1 - b0) - b0
            integrand = -0.5 * dlog / np.sqrt(alpha * (1 - alpha))
            nodes = [tau64[i - k] for k in range(order)]
            row = []
            for j in range(order):
                poly = np.ones_like(grid)
                for k in range(order):
                    if k != j:
                        poly = poly * (grid - nodes[k]) / (nodes[j] - nodes[k])
                row.append(torch.tensor(float(np.sum(integrand * poly) * dtau), dtype=torch.float32))
            out.append(row)
        return out
    if deis_mode == 'rhoab':
        ts = [float(v) for v in torch.as_tensor(t_steps).detach().cpu().double().tolist()]

        def poly_int(nodes, s, e, at):
            # integral over [s, e] of prod_k (t - nodes[k]) / (at - nodes[k])
            coef = np.poly1d([1.0])
            den = 1.0
            for a in nodes:
                coef = coef * np.poly1d([1.0, -a])
                den *= (at - a)
            P = coef.integ()
            return (P(e) - P(s)) / den

        row = None
        for i in range(len(ts) - 1):
            order = min(i, max_order)
            if order == 0:
                out.append([])
                continue
            if order <= 3:
                pts = [ts[i - k] for k in range(order + 1)]
                row = [torch.tensor(poly_int([p for q, p in enumerate(pts) if q != j], ts[i], ts[i + 1], pts[j]), dtype=torch.float32)
                       for j in range(order + 1)]
            out.append(row)
        return out
    raise ValueError(deis_mode)


# ------------------------------------------------------------------------------------------------------------------
# UniPC coefficient compiler (solver_utils.py:174-287): everything except the model evaluation is a function of times.

def unipc_coeffs(t_hist: Sequence[float], t_next: float, order: int, predict_x0=True, variant='bh1', use_corrector=True):
    """Returns dict(cx, pred=[c_m0, c_m1, ...], corr=[c_m0, c_m1, ..., c_model_t] or None).

    predictor : x_p = cx*x + sum_j pred[j] * m_{n-j}
    corrector : x'  = cx*x + sum_j corr[j] * m_{n-j} + corr[-1] * model_t"""
    lam = lambda s: -math.log(s)
    t0 = t_hist[-1]
    h = lam(t_next) - lam(t0)
    rks = [(lam(t_hist[-(i + 1)]) - lam(t0)) / h for i in range(1, order)] + [1.0]
    hh = -h if predict_x0 else h
    h_phi_1 = math.expm1(hh)
    h_phi_k = h_phi_1 / hh - 1
    if variant == 'bh1':
        B_h = hh
    elif variant == 'bh2':
        B_h = math.expm1(hh)
    else:
        raise NotImplementedError()
    R, b, fact = [], [], 1
    for i in range(1, order + 1):
        R.append([rk ** (i - 1) for rk in rks])
        b.append(h_phi_k * fact / B_h)
        fact *= (i + 1)
        h_phi_k = h_phi_k / hh - 1 / fact
    R, b = np.array(R, dtype=np.float64), np.array(b, dtype=np.float64)
    nD = order - 1
    rhos_p = None
    if nD > 0:
        rhos_p = np.array([0.5]) if order == 2 else np.linalg.solve(R[:-1, :-1], b[:-1])
    rhos_c = None
    if use_corrector:
        rhos_c = np.array([0.5]) if order == 1 else np.linalg.solve(R, b)
    cx = (t_next / t0) if predict_x0 else 1.0
    tf = 1.0 if predict_x0 else t_next
    base0 = -tf * h_phi_1                      # coefficient of m0 in x_t_

    def expand(rhos, extra):
        # - tf*B_h * ( sum_k rhos[k] * (m_{k+1} - m0)/rk_k  +  extra * (model_t - m0) )
        c = [base0] + [0.0] * nD
        for k in range(nD):
            if rhos is not None and k < len(rhos):
                w = -tf * B_h * rhos[k] / rks[k]
                c[k + 1] += w
                c[0] -= w
        if extra is not None:
            c[0] -= -tf * B_h * extra
            c.append(-tf * B_h * extra)
        return c

    pred = expand(rhos_p, None)
    corr = expand(rhos_c[:-1] if rhos_c is not None else None, rhos_c[-1]) if use_corrector else None
    return dict(cx=cx, pred=pred, corr=corr)



def unipc_update(x, model_prev_list, t_prev_list, t, order, x_t=None, variant='bh1', predict_x0=True, net=None, class_labels=None,
                 use_corrector=True):
    """API-compatible with the reference ``unipc_update`` (solver_utils.py:174-287): returns ``(x_t, model_t)``.

    Predictor (skipped when ``x_t`` is given), one evaluation ``net(x_t, t, class_labels)`` at the predicted point
    (:260, :278), corrector.  Each of the two combinations is one fused launch with coefficients from ``unipc_coeffs``."""
    assert order <= len(model_prev_list)
    th = [float(tp) for tp in t_prev_list[-order:]]
    tn = float(t)
    k = unipc_coeffs(th, tn, order, predict_x0=predict_x0, variant=variant, use_corrector=use_corrector)
    hs = [m.contiguous() for m in model_prev_list[::-1][:order]]          # newest first
    if x_t is None:
        x_t = lincomb(k['cx'], x, [(k['pred'][j], hs[j]) for j in range(order)])
    model_t = None
    if use_corrector:
        tt = t if torch.is_tensor(t) else torch.tensor(tn, dtype=torch.float32, device=x.device)
        den = net(x_t, tt, class_labels)
        if predict_x0:
            model_t = dynamic_thresholding_fn(den)
        else:
            model_t = lincomb(1.0 / tn, x_t, [(-1.0 / tn, den)])               # (x_t - denoised) / t
        c = k['corr']
        x_t = lincomb(k['cx'], x, [(c[-1], model_t)] + [(c[j], hs[j]) for j in range(order)])
    return x_t, model_t


# ------------------------------------------------------------------------------------------------------------------
# DEIS helper names of the reference module (host-side, CPU tensors; ``get_deis_coeff_list`` above does not need them).

def cal_poly(prev_t, j, taus):
    """Lagrange basis polynomial j over the nodes ``prev_t`` evaluated at ``taus`` (solver_utils.py:307-314)."""
    poly = 1
    for k in range(prev_t.shape[0]):
        if k != j:
            poly = poly * ((taus - prev_t[k]) / (prev_t[j] - prev_t[k]))
    return poly


def t2alpha_fn(beta_0, beta_1, t):
    """alpha(t) of the VP-SDE (solver_utils.py:318-319)."""
    return torch.exp(-0.5 * t ** 2 * (beta_1 - beta_0) - t * beta_0)


def cal_intergrand(beta_0, beta_1, taus):
    """-0.5 dlog(alpha)/dtau / sqrt(alpha (1 - alpha)) (solver_utils.py:323-331).  The reference differentiates log(alpha)
    with autograd; its derivative is -(beta_1 - beta_0) tau - beta_0, evaluated here in closed form."""
    taus = torch.as_tensor(taus)
    alpha = t2alpha_fn(beta_0, beta_1, taus)
    d_log_alpha_dtau = -taus * (beta_1 - beta_0) - beta_0
    return -0.5 * d_log_alpha_dtau / torch.sqrt(alpha * (1 - alpha))
