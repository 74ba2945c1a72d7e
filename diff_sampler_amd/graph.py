"""hipGraph capture of whole sampler calls.

A sampler call on the HIP engine is a fixed sequence of kernel launches whose arguments do not depend on data: every
per-step scalar (sigma, update coefficients) is a host value passed by value, every workspace pointer is owned by the
denoiser plan.  ``GraphedSampler`` therefore captures ONE call -- ~2.8k launches for 10 network evaluations of the
CIFAR-10 net -- into a hipGraph and replays it with a single ``hipGraphLaunch``; the host work per call drops from
~2.8k ctypes launches (~4 us each) to one.  This matters at small batch (B <= 32, where a network evaluation is a few
milliseconds); at B >= 256 the GPU time dominates and eager launches are already hidden.

Capture/replay go through ``torch.cuda.CUDAGraph`` (on ROCm this *is* hipStreamBeginCapture / hipGraphInstantiate /
hipGraphLaunch); torch is used because its caching allocator must be told about the capture (private pool), not for
any computation: all captured nodes are libdsamd kernels plus the d2d copies of trajectory snapshots.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch


class GraphedSampler:
    """``g = GraphedSampler(solvers.ipndm_sampler, net, latents_shape, t_steps=..., max_order=4); images = g(latents)``.

    Fixed at capture time: batch shape, schedule, solver options, class labels / condition tensors (their *contents* may
    change between replays: they are copied into static buffers).  ``t_steps`` must be host-resident (list / CPU tensor): the
    samplers read it on the host and a device->host copy is not capturable.
    """

    def __init__(self, sampler_fn: Callable, net, latents_shape, class_labels_shape=None, condition_shape=None,
                 uncond_shape=None, device='cuda', warmup=1, **solver_kwargs):
        self.fn, self.net, self.kw = sampler_fn, net, dict(solver_kwargs)
        ts = self.kw.get('t_steps')
        if ts is None:
            from .solver_utils import get_schedule
            ts = get_schedule(self.kw['num_steps'], self.kw.get('sigma_min', 0.002), self.kw.get('sigma_max', 80.), device='cpu',
                              schedule_type=self.kw.get('schedule_type', 'polynomial'), schedule_rho=self.kw.get('schedule_rho', 7),
                              net=net)
        self.kw['t_steps'] = torch.as_tensor(ts).detach().to('cpu')
        self.static_in = torch.zeros(*latents_shape, dtype=torch.float32, device=device)
        self.static_labels = None
        if class_labels_shape is not None:
            self.static_labels = torch.zeros(*class_labels_shape, dtype=torch.float32, device=device)
        # latent-diffusion denoisers (ldm_engine.CFGDenoiser): text-encoder states of the batch and the unconditional ones
        if condition_shape is not None:
            self.kw['condition'] = self.static_cond = torch.zeros(*condition_shape, dtype=torch.float32, device=device)
        if uncond_shape is not None:
            self.kw['unconditional_condition'] = self.static_uncond = torch.zeros(*uncond_shape, dtype=torch.float32, device=device)
        # eager warm-up on a side stream: builds the denoiser plan, fills allocator pools, JIT-free by construction
        s = torch.cuda.Stream(device=device)
        s.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(s):
            for _ in range(max(1, warmup)):
                self.fn(self.net, self.static_in, class_labels=self.static_labels, **self.kw)
        torch.cuda.current_stream(device).wait_stream(s)
        torch.cuda.synchronize(device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=s):
            self.static_out = self.fn(self.net, self.static_in, class_labels=self.static_labels, **self.kw)
        torch.cuda.synchronize(device)
        getattr(self.net, 'invalidate_context_cache', lambda: None)()

    def __call__(self, latents: torch.Tensor, class_labels: Optional[torch.Tensor] = None, clone: bool = True, condition=None,
                 unconditional_condition=None):
        self.static_in.copy_(latents)
        if self.static_labels is not None:
            self.static_labels.copy_(class_labels)
        if condition is not None:
            self.static_cond.copy_(condition)
        if unconditional_condition is not None:
            self.static_uncond.copy_(unconditional_condition)
        self.graph.replay()
        # the replay rewrote the denoiser's cross-attention K / V buffers from the static conditions: an eager call that follows must
        # not believe its own cached context is still there (ldm_engine.CFGDenoiser)
        getattr(self.net, 'invalidate_context_cache', lambda: None)()
        out = self.static_out
        if clone:
            out = tuple(o.clone() for o in out) if isinstance(out, tuple) else out.clone()
        return out
