"""GITS: geometry-inspired time-schedule search (reference: gits-main/gits_utils.py) on the HIP engine.

``get_dp_list`` keeps the reference's contract (same kwargs, returns the list of indices into the teacher schedule
that ``get_schedule(..., dp_list=...)`` consumes) but not its cost: the reference forms the N x N cost matrix with
~N^2/2 Euler jumps, each followed by several full-tensor passes (gits_utils.py:115-130); here one kernel pass over the
teacher trajectory yields six fp64 inner products per point and sample, from which every 'dev' cost follows in closed
form on the host (csrc/gits.hip); 'l1'/'l2' use a direct pair kernel.  The dynamic programme itself is O(K N^2) scalar
work and stays on the host (numpy), with the reference's tie-breaking (first minimiser).
"""
from __future__ import annotations

import copy
import ctypes as C

import numpy as np
import torch

from . import _lib, solvers, solver_utils


def get_sampler_fn(solver, device, dp_list=None, net=None, **kwargs):
    """(sampler function, DEIS coefficient list or None) -- gits_utils.py:15-37."""
    table = dict(euler=solvers.euler_sampler, heun=solvers.heun_sampler, dpm=solvers.dpm_2_sampler, ipndm=solvers.ipndm_sampler,
                 ipndm_v=solvers.ipndm_v_sampler, dpmpp=solvers.dpm_pp_sampler, deis=solvers.deis_sampler)
    if solver not in table:
        raise NotImplementedError(f"Unknown solver: {solver}")
    if solver == 'deis':
        t_steps = solver_utils.get_schedule(kwargs['num_steps_tea'], kwargs['sigma_min'], kwargs['sigma_max'], device=device,
                                            schedule_type=kwargs["schedule_type"], schedule_rho=kwargs["schedule_rho"], net=net,
                                            dp_list=dp_list)
        return table[solver], solver_utils.get_deis_coeff_list(t_steps, kwargs['max_order'], deis_mode=kwargs["deis_mode"])
    return table[solver], None


def trajectory_moments(traj: torch.Tensor, eps=None) -> np.ndarray:
    """[n_pts, batch, 6] fp64 {P, Q, R, S, T, N} (see csrc/gits.hip)."""
    n_pts, B = traj.shape[0], traj.shape[1]
    per = traj[0, 0].numel()
    traj = traj.contiguous()
    eps = eps.contiguous() if eps is not None else None
    out = torch.empty(n_pts, B, 6, dtype=torch.float64, device=traj.device)
    rc = _lib.load().ds_traj_moments(C.c_void_p(traj.data_ptr()), C.c_void_p(eps.data_ptr()) if eps is not None else None, n_pts, B, per,
                                     C.c_void_p(out.data_ptr()), _lib.stream_ptr())
    _lib.check(rc, 'ds_traj_moments')
    return out.cpu().numpy()


def cal_deviation(traj, ch=None, r=None, bs=1):
    """Deviation of the intermediate trajectory points from the start->end chord, [bs, n_pts-2] (gits_utils.py:237-255)."""
    m = trajectory_moments(traj)                                   # [n, B, 6]
    P, R, N = m[1:-1, :, 0], m[1:-1, :, 2], m[1:-1, :, 5]
    dev = np.sqrt(np.maximum(R - P * P / N, 0.0))
    return torch.from_numpy(dev.T.copy()).to(torch.float32).to(traj.device)


def _cost_matrix_round(traj, eps, t_host, metric):
    """Sum over the batch of the per-sample costs of every Euler jump i -> j, [N, N] fp64 (zero where j <= i)."""
    n, B = traj.shape[0], traj.shape[1]
    per = traj[0, 0].numel()
    if metric == 'dev':
        m = trajectory_moments(traj, eps)
        P, Q, R, S, T, N = (m[:, :, k] for k in range(6))
        dev_tea = np.sqrt(np.maximum(R[1:-1] - P[1:-1] ** 2 / N[1:-1], 0.0)).mean(axis=1)       # [n-2], mean over batch
        dev_tea = np.concatenate([dev_tea, [0.0]])
        t = np.asarray(t_host, dtype=np.float64)
        cost = np.zeros((n, n))
        for i in range(n - 1):
            D = (t[i + 1:] - t[i])[:, None]                                                    # [n-i-1, 1]
            val = R[i] - 2 * D * S[i] + D * D * T[i] - (P[i] - D * Q[i]) ** 2 / N[i]           # [n-i-1, B]
            dev_stu = np.sqrt(np.maximum(val, 0.0)).mean(axis=1)
            cost[i, i + 1:] = dev_stu - dev_tea[np.arange(i + 1, n) - 1]
        return cost
    if metric in ('l1', 'l2'):
        cost = torch.zeros(n, n, dtype=torch.float64, device=traj.device)
        tdev = torch.tensor(t_host, dtype=torch.float32, device=traj.device)
        rc = _lib.load().ds_traj_pair_cost(C.c_void_p(traj.contiguous().data_ptr()), C.c_void_p(eps.contiguous().data_ptr()),
                                           C.c_void_p(tdev.data_ptr()), n, B, per, 1 if metric == 'l1' else 2,
                                           C.c_void_p(cost.data_ptr()), _lib.stream_ptr())
        _lib.check(rc, 'ds_traj_pair_cost')
        return cost.cpu().numpy() / B
    raise NotImplementedError(f"Unknown metric: {metric}")


def _load_captions(prompt_path):
    """MS-COCO captions for the FID-30k protocol (gits_utils.py:63-73: the csv's 'text' column)."""
    import csv
    if not prompt_path:
        raise FileNotFoundError("GITS on ms_coco without --prompt needs the caption csv (solver_kwargs['prompt_path'])")
    with open(prompt_path, 'r') as f:
        return [row['text'] for row in csv.DictReader(f)]


def _mean_l2(a, b):
    """mean over the batch of || a_n - b_n ||_2 (gits_utils.py:171) from the trajectory-moment kernel: R of the two-point "trajectory"
    (a, b) is |b - a|^2 per sample -- no ATen arithmetic."""
    m = trajectory_moments(torch.stack([a, b], dim=0))              # [2, B, 6]; index 2 = R = |c - x_i|^2 with c = last point
    return torch.tensor(float(np.sqrt(np.maximum(m[0, :, 2], 0.0)).mean()), dtype=torch.float32, device=a.device)


def dp(cost_mat, num_steps, num_steps_tea, coeff, multiple_coeff=False, desc=None, t_steps=None):
    """Dynamic programme of gits_utils.py:185-203: V[j][k] = min_i cost[j][i] + coeff * V[i][k-1]; the path is read
    back with the reference's rule (first j that attains the minimum)."""
    cost = np.asarray(cost_mat, dtype=np.float64)
    N, K = num_steps_tea, num_steps - 1
    V = np.full((N, K + 1), np.inf)
    V[:, 1] = cost[:, -1]
    for k in range(2, K + 1):
        for j in range(N - 1):
            cand = cost[j, j + 1:N - 1] + coeff * V[j + 1:N - 1, k - 1]
            if cand.size:
                V[j, k] = min(V[j, k], cand.min())
    phi, w = [0], 0
    for temp in range(K):
        k = K - temp
        for j in range(w + 1, N):
            if V[w][k] == cost[w][j] + coeff * V[j][k - 1]:
                phi.append(j)
                w = j
                break
    phi.append(N - 1)
    return phi


def _warmup_conditioning(net, device, batch, model_source, dataset_name, solver_kwargs, sample_captions):
    """Labels / text conditions of one warm-up round, by model source (gits_utils.py:86-102).
      'adm'            integer class indices ``randint(label_dim, (B,))`` (the ADM wrapper takes indices, not one-hot rows)
      'ldm' + ms_coco  text conditions: with a reference-style net (``net.model.get_learned_conditioning``) the prompts are encoded as
                       the reference does -- the given ``prompt`` for every sample, or ``random.sample`` of the caption list; an engine
                       net (``CFGDenoiser``: the text encoder is not on the sampling path) gets seeded N(0, 1) states of the CLIP shape
                       ``[B, 77, context_dim]`` and one fixed unconditional row, the convention of ``sample.py`` (BASELINE config 5)
      otherwise        one-hot rows ``eye(label_dim)[randint]`` (EDM nets)
    Returns (class_labels, condition, unconditional_condition)."""
    if not net.label_dim:
        return None, None, None
    if model_source == 'adm':
        return torch.randint(net.label_dim, size=(batch,), device=device), None, None
    if model_source == 'ldm' and dataset_name == 'ms_coco':
        guided = solver_kwargs.get('guidance_rate') != 1.0
        model = getattr(net, 'model', None)
        if model is not None and hasattr(model, 'get_learned_conditioning'):
            if solver_kwargs.get('prompt') is None:
                import random
                prompts = random.sample(sample_captions, batch)
            else:
                prompts = [solver_kwargs['prompt'] for _ in range(batch)]
            uc = model.get_learned_conditioning(batch * [""]) if guided else None
            return None, model.get_learned_conditioning(list(prompts)), uc
        cd = net.spec.context_dim
        c = torch.randn(batch, 77, cd, device=device)
        uc = torch.randn(1, 77, cd, generator=torch.Generator().manual_seed(0)).to(device).expand(batch, -1, -1) if guided else None
        return None, c, uc
    return torch.eye(net.label_dim, device=device)[torch.randint(net.label_dim, size=[batch], device=device)], None, None


def get_dp_list(net, device, warmup_latents=None, warmup_conditions=None, **solver_kwargs):
    """Search the ``num_steps``-point sub-schedule of the ``num_steps_tea``-point teacher schedule (gits_utils.py:42-180), for every
    model source the reference's search handles: 'edm' (one-hot labels), 'adm' (integer labels), 'ldm' (text conditions; classifier-free
    guidance doubles the evaluation inside the denoiser) -- gits_utils.py:86-108.

    warmup_latents / warmup_conditions (extensions, for reproducible tests): per accumulation round a latent tensor and a
    ``(class_labels, condition, unconditional_condition)`` triple; by default every round draws them on ``device`` like the reference."""
    kwargs = copy.copy(solver_kwargs)
    num_warmup, max_batch_size = kwargs['num_warmup'], kwargs['max_batch_size']
    sigma_min, sigma_max = kwargs['sigma_min'], kwargs['sigma_max']
    num_steps, num_steps_tea = kwargs['num_steps'], kwargs['num_steps_tea']
    schedule_type, schedule_rho = kwargs['schedule_type'], kwargs['schedule_rho']
    afs, metric, coeff = kwargs['afs'], kwargs['metric'], kwargs['coeff']
    dist = torch.distributed if (torch.distributed.is_available() and torch.distributed.is_initialized()) else None
    world = dist.get_world_size() if dist else 1

    kwargs['solver'] = solver_kwargs['solver_tea']
    sampler_fn_tea, coeff_list = get_sampler_fn(device=device, net=net, dp_list=list(range(num_steps_tea)), **kwargs)
    t_steps = solver_utils.get_schedule(num_steps_tea, sigma_min, sigma_max, device=device, schedule_type=schedule_type,
                                        schedule_rho=schedule_rho, net=net)
    t_host = solver_utils.host_times(t_steps)
    kwargs['t_steps'] = t_steps.cpu()
    kwargs['coeff_list'] = coeff_list
    kwargs['return_inters'], kwargs['return_eps'] = True, True
    kwargs['num_steps'] = num_steps_tea
    rounds = num_warmup // (max_batch_size + 1) + 1
    batch_gpu = max_batch_size // world
    cost = np.zeros((num_steps_tea, num_steps_tea))
    model_source, dataset_name = kwargs.get('model_source'), kwargs.get('dataset_name')
    sample_captions = None
    if dataset_name == 'ms_coco' and model_source == 'ldm' and kwargs.get('prompt') is None and hasattr(getattr(net, 'model', None), 'get_learned_conditioning'):
        sample_captions = _load_captions(kwargs.get('prompt_path'))               # gits_utils.py:63-73
    for k in ('condition', 'unconditional_condition', 'class_labels'):           # the search supplies its own
        kwargs.pop(k, None)

    def run_sampler(fn, lat, cond3, kw):
        """The teacher / student call of one round; 'ldm': under autocast + ema_scope when the net is a reference-style module
        (gits_utils.py:104-108) -- an engine net carries its precision mode itself."""
        cl, c, uc = cond3
        if model_source == 'ldm':
            model = getattr(net, 'model', None)
            if model is not None and hasattr(model, 'ema_scope'):
                with torch.autocast('cuda'), model.ema_scope():
                    return fn(net, lat, condition=c, unconditional_condition=uc, **kw)
            return fn(net, lat, condition=c, unconditional_condition=uc, **kw)
        return fn(net, lat, class_labels=cl, **kw)

    latents = cond3 = teacher_traj = None
    for r in range(rounds):
        if warmup_latents is not None:
            latents = warmup_latents[r].to(device)
        else:
            latents = torch.randn([batch_gpu, net.img_channels, net.img_resolution, net.img_resolution], device=device)
        if warmup_conditions is not None:
            cond3 = tuple(None if t is None else t.to(device) for t in warmup_conditions[r])
        else:
            cond3 = _warmup_conditioning(net, device, latents.shape[0], model_source, dataset_name, kwargs, sample_captions)
        teacher_traj, eps_traj = run_sampler(sampler_fn_tea, latents, cond3, kwargs)
        cost += _cost_matrix_round(teacher_traj, eps_traj, t_host, metric)
    cost_t = torch.from_numpy(cost).to(torch.float32).to(device)
    if dist:
        dist.all_reduce(cost_t)                                     # gits_utils.py:134
    cost = (cost_t / (world * rounds)).cpu().numpy()

    dp_list = phi = dp(cost, num_steps, num_steps_tea, coeff)
    kwargs['return_inters'] = kwargs['return_eps'] = False
    kwargs['solver'] = solver_kwargs['solver']
    kwargs['num_steps'] = solver_kwargs['num_steps']
    if afs:                                                          # AFS slot search (gits_utils.py:157-178)
        dist_min = 999999
        for k in range(1, phi[1]):
            cand = copy.deepcopy(phi)
            cand.insert(1, k)
            sampler_fn, kwargs['coeff_list'] = get_sampler_fn(device=device, dp_list=cand, net=net, **{**kwargs, 'coeff_list': None})
            kwargs['t_steps'] = solver_utils.get_schedule(num_steps_tea, sigma_min, sigma_max, device='cpu', schedule_type=schedule_type,
                                                          schedule_rho=schedule_rho, net=net, dp_list=cand)
            images_afs = run_sampler(sampler_fn, latents, cond3, kwargs)
            d = _mean_l2(images_afs, teacher_traj[-1])
            if dist:
                dist.all_reduce(d)
                d = d / world
            if d < dist_min:
                dist_min, dp_list = d, cand
    return dp_list
