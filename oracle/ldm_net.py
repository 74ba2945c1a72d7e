"""CPU oracle of the latent-diffusion denoiser: ``UNetModel.forward`` + ``CFGPrecond`` as plain functions over a flat
state_dict.  TEST INFRASTRUCTURE ONLY (imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg).

Follows, line by line, the reference modules it restates:
  * ``UNetModel.forward``          diff-solvers-main/models/ldm/modules/diffusionmodules/openaimodel.py:710-742
  * ``ResBlock._forward``          openaimodel.py:250-274     (GroupNorm32 -> SiLU -> conv, + emb, GroupNorm32 -> SiLU -> conv, + skip)
  * ``Downsample`` / ``Upsample``  openaimodel.py:133-160 / :86-118   (3x3 stride-2 conv / nearest x2 then 3x3 conv)
  * ``SpatialTransformer``         ldm/modules/attention.py:218-260
  * ``BasicTransformerBlock``      attention.py:196-215,  ``CrossAttention`` :152-194,  GEGLU ``FeedForward`` :45-72
  * ``timestep_embedding``         ldm/modules/diffusionmodules/util.py:151-171
  * ``CFGPrecond``                 models/networks_edm.py:630-762  (sigma / sigma_inv / interpolate_fn / forward)
The layer structure is discovered from the key names.  Pinned by tests/golden/ldm_*.npz, which oracle/gen_golden.py
produced by running the real reference modules (part 'ldm').
"""
import contextlib
import math

import torch
import torch.nn.functional as F

# Reduced-precision variant (the reference runs this U-Net under torch.autocast, sample.py:293-297): inside ``operands_f16(pred)`` the
# multiplicands of every layer whose state_dict prefix satisfies ``pred`` are rounded to fp16 and multiplied / accumulated in fp32
# (attention layers are named '<transformer block>.attn1' / '.attn2': q, k, v and the softmax weights are rounded).  See oracle/edm_net.py.
# ``stored(prefix)`` (optional) names the layers whose OUTPUT that mode holds as an fp16 tensor -- under autocast every convolution /
# Linear emits fp16, so the residual stream h, the skip stack and the transformer's x are fp16: the value is rounded right after the
# layer's bias / embedding / residual additions (where the product's epilogue stores it); arithmetic on it stays fp32.
_F16_PRED = None
_F16_STORED = None


@contextlib.contextmanager
def operands_f16(pred, stored=None):
    global _F16_PRED, _F16_STORED
    old, _F16_PRED, _F16_STORED = (_F16_PRED, _F16_STORED), pred, stored
    try:
        yield
    finally:
        _F16_PRED, _F16_STORED = old


def _rnd(prefix, *ts):
    if _F16_PRED is not None and _F16_PRED(prefix):
        return tuple(t.to(torch.float16).to(torch.float32) for t in ts)
    return ts


def _stored(prefix, t):
    if _F16_STORED is not None and _F16_STORED(prefix):
        return t.to(torch.float16).to(torch.float32)
    return t


def _gn32(p, prefix, x, eps):
    return F.group_norm(x.float(), 32, p[prefix + '.weight'], p[prefix + '.bias'], eps)


def _lin(p, prefix, x):
    x, w = _rnd(prefix, x, p[prefix + '.weight'])
    return F.linear(x, w, p.get(prefix + '.bias'))


def _conv(p, prefix, x, stride=1):
    x, w = _rnd(prefix, x, p[prefix + '.weight'])
    return F.conv2d(x, w, p[prefix + '.bias'], stride=stride, padding=w.shape[-1] // 2)


def timestep_embedding(t, dim, max_period=10000):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half).to(t.device)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def _res(p, prefix, x, emb):
    h = _conv(p, prefix + '.in_layers.2', F.silu(_gn32(p, prefix + '.in_layers.0', x, 1e-5)))
    h = _stored(prefix + '.in_layers.2', h + _lin(p, prefix + '.emb_layers.1', F.silu(emb))[:, :, None, None])
    h = _conv(p, prefix + '.out_layers.3', F.silu(_gn32(p, prefix + '.out_layers.0', h, 1e-5)))
    skip = _conv(p, prefix + '.skip_connection', x) if (prefix + '.skip_connection.weight') in p else x
    return _stored(prefix + '.out_layers.3', skip + h)


def _attn(p, prefix, x, context, heads):
    q = _stored(prefix + '.to_q', _lin(p, prefix + '.to_q', x))          # fp16 mode: q (and, self-attention, k | v) are stored fp16 tensors
    ctx = x if context is None else context
    k, v = _stored(prefix + '.to_k', _lin(p, prefix + '.to_k', ctx)), _stored(prefix + '.to_v', _lin(p, prefix + '.to_v', ctx))
    b, n, c = q.shape
    d = c // heads
    split = lambda t: t.reshape(b, t.shape[1], heads, d).permute(0, 2, 1, 3).reshape(b * heads, t.shape[1], d)
    q, k, v = split(q), split(k), split(v)
    if _F16_PRED is not None and _F16_PRED(prefix):
        # the fp16-operand attention kernel's arithmetic: q pre-multiplied by d^-1/2 * log2(e) and THEN rounded, k and v rounded,
        # fp32 scores, un-normalised exp2 weights rounded for the P V product, fp32 row sums of the unrounded weights.  (Where the
        # roundings sit matters: on SD-1.5 two placements differ by 3.5e-3 of the output -- as much as either differs from fp32.)
        if _F16_STORED is not None and _F16_STORED(prefix + '.to_q'):
            # q is a stored fp16 tensor (rounded by its projection): the factor multiplies the fp32 scores, k / v are rounded while staged
            # unless they are stored fp16 tensors themselves (rounding is idempotent)
            q, k, v = _rnd(prefix, q, k, v)
            sim = torch.einsum('bid,bjd->bij', q, k) * (d ** -0.5 * math.log2(math.e))
        else:
            q, k, v = _rnd(prefix, q * (d ** -0.5 * math.log2(math.e)), k, v)
            sim = torch.einsum('bid,bjd->bij', q, k)
        pw = torch.exp2(sim - sim.max(dim=-1, keepdim=True).values)
        out = torch.einsum('bij,bjd->bid', _rnd(prefix, pw)[0], v) / pw.sum(-1, keepdim=True)
    else:
        sim = torch.einsum('bid,bjd->bij', q, k) * (d ** -0.5)
        out = torch.einsum('bij,bjd->bid', sim.softmax(dim=-1), v)
    out = out.reshape(b, heads, n, d).permute(0, 2, 1, 3).reshape(b, n, c)
    return _lin(p, prefix + '.to_out.0', out)


def _st(p, prefix, x, context, heads):
    b, c, h, w = x.shape
    x_in = x
    x = _stored(prefix + '.proj_in', _conv(p, prefix + '.proj_in', F.group_norm(x, 32, p[prefix + '.norm.weight'], p[prefix + '.norm.bias'], 1e-6)))
    x = x.permute(0, 2, 3, 1).reshape(b, h * w, c)
    t = prefix + '.transformer_blocks.0'
    ln = lambda name, v: F.layer_norm(v, (c,), p[f'{t}.{name}.weight'], p[f'{t}.{name}.bias'], 1e-5)
    x = _stored(t + '.attn1.to_out.0', _attn(p, t + '.attn1', ln('norm1', x), None, heads) + x)
    x = _stored(t + '.attn2.to_out.0', _attn(p, t + '.attn2', ln('norm2', x), context, heads) + x)
    y, gate = _lin(p, t + '.ff.net.0.proj', ln('norm3', x)).chunk(2, dim=-1)
    x = _stored(t + '.ff.net.2', _lin(p, t + '.ff.net.2', y * F.gelu(gate)) + x)
    x = x.reshape(b, h, w, c).permute(0, 3, 1, 2)
    return _stored(prefix + '.proj_out', _conv(p, prefix + '.proj_out', x) + x_in)


def _block(p, name, h, emb, context, heads, taps):
    j = 0
    while True:
        q = f'{name}.{j}'
        if (q + '.in_layers.0.weight') in p:
            h = _res(p, q, h, emb)
        elif (q + '.proj_in.weight') in p:
            h = _st(p, q, h, context, heads)
        elif (q + '.op.weight') in p:
            h = _stored(q + '.op', _conv(p, q + '.op', h, stride=2))
        elif (q + '.conv.weight') in p:
            h = _stored(q + '.conv', _conv(p, q + '.conv', F.interpolate(h, scale_factor=2, mode='nearest')))
        elif (q + '.weight') in p and p[q + '.weight'].dim() == 4:
            h = _conv(p, q, h)
        else:
            break
        if taps is not None:
            taps[q] = h
        j += 1
    assert j > 0, name
    return h


def unet_forward(p, cfg, x, timesteps, context, taps=None):
    """UNetModel.forward (openaimodel.py:710-742).  cfg: dict(model_channels=, num_heads=)."""
    heads = cfg['num_heads']
    emb = _lin(p, 'time_embed.2', F.silu(_lin(p, 'time_embed.0', timestep_embedding(timesteps, cfg['model_channels']))))
    hs = []
    h = x.float()
    i = 0
    while any(k.startswith(f'input_blocks.{i}.') for k in p):
        h = _block(p, f'input_blocks.{i}', h, emb, context, heads, taps)
        hs.append(h)
        i += 1
    h = _block(p, 'middle_block', h, emb, context, heads, taps)
    i = 0
    while any(k.startswith(f'output_blocks.{i}.') for k in p):
        h = torch.cat([h, hs.pop()], dim=1)
        h = _block(p, f'output_blocks.{i}', h, emb, context, heads, taps)
        i += 1
    assert not hs
    return _conv(p, 'out.2', F.silu(_gn32(p, 'out.0', h, 1e-5)))


def _interp(x, xp, yp):
    """Piecewise-linear y(x) through the keypoints (xp ascending), extended linearly beyond both ends
    (networks_edm.py:716-759 interpolate_fn, same evaluation formula start_y + (x - start_x) * (end_y - start_y) / (end_x - start_x))."""
    K = xp.shape[0]
    i = torch.searchsorted(xp, x.contiguous()).clamp(1, K - 1) - 1
    return yp[i] + (x - xp[i]) * (yp[i + 1] - yp[i]) / (xp[i + 1] - xp[i])


class OracleCFG:
    """CFGPrecond (networks_edm.py:630-762) over the functional U-Net; the attributes samplers read are provided."""

    def __init__(self, params, cfg, alphas_cumprod, guidance_rate=7.5, guidance_type='classifier-free', epsilon_t=1e-3):
        self.params, self.cfg = params, dict(cfg)
        self.guidance_rate, self.guidance_type = guidance_rate, guidance_type
        self.img_resolution, self.img_channels, self.label_dim = cfg['img_resolution'], cfg['in_channels'], True
        log_alphas = 0.5 * torch.log(alphas_cumprod)
        self.M = len(log_alphas)
        self.t_array = torch.linspace(0., 1., self.M + 1)[1:]
        self.log_alpha_array = log_alphas
        self.sigma_min = float(self.sigma(epsilon_t))
        self.sigma_max = float(self.sigma(1))

    def sigma(self, t):
        t = torch.as_tensor(t, dtype=torch.float32).reshape(-1)
        lmc = _interp(t, self.t_array, self.log_alpha_array)
        return (torch.sqrt(1. - torch.exp(2. * lmc)) / torch.exp(lmc)).reshape(-1)

    def sigma_inv(self, sigma):
        sigma = torch.as_tensor(sigma, dtype=torch.float32).reshape(-1)
        lamb = -(sigma.log())
        log_alpha = -0.5 * torch.logaddexp(torch.zeros((1,)), -2. * lamb)
        return _interp(log_alpha, torch.flip(self.log_alpha_array, [0]), torch.flip(self.t_array, [0])).reshape(-1)

    def round_sigma(self, sigma):
        return torch.as_tensor(sigma)

    def eps(self, x, t, cond):
        return unet_forward(self.params, self.cfg, x, t, cond)

    def __call__(self, x, sigma, condition=None, unconditional_condition=None, **kw):
        x = x.to(torch.float32)
        sigma = torch.as_tensor(sigma, dtype=torch.float32).reshape(-1)
        c_in = 1 / (sigma ** 2 + 1).sqrt()
        c_noise = self.M * self.sigma_inv(sigma) - 1.
        if c_noise.shape[0] == 1:
            c_noise = c_noise.expand(x.shape[0])
        xin = c_in.reshape(-1, 1, 1, 1) * x
        if self.guidance_type == 'uncond':
            f = self.eps(xin, c_noise, None)
        elif self.guidance_rate == 1. or unconditional_condition is None:
            f = self.eps(xin, c_noise, condition)
        else:
            nu, nc = self.eps(torch.cat([xin] * 2), torch.cat([c_noise] * 2),
                              torch.cat([unconditional_condition, condition])).chunk(2)
            self.last_eps = (nu, nc)          # the two raw U-Net outputs of this call (tests pin them separately from the guided combination)
            f = nu + self.guidance_rate * (nc - nu)
        return x + (-sigma).reshape(-1, 1, 1, 1) * f
