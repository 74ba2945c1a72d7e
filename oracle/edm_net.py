"""ORACLE (test infrastructure, not product code): CPU restatement of the EDM denoiser.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
file; the shipped path (``diff_sampler_amd/``) never does.

What it restates (all citations relative to /root/reference/diff-solvers-main/):
  * ``EDMPrecond.forward``                models/networks_edm.py:482-496
  * ``SongUNet.forward``                  models/networks_edm.py:312-355
  * ``DhariwalUNet.forward``              models/networks_edm.py:427-453
  * ``UNetBlock.forward``                 models/networks_edm.py:158-179
  * ``Conv2d.forward`` (up/down, [1,1])   models/networks_edm.py:60-82
  * ``GroupNorm`` / ``AttentionOp`` / ``PositionalEmbedding``   :88-98 / :105-110 / :185-198

It is a *functional* restatement over a flat ``{state_dict key: tensor}`` mapping: the layer structure is
discovered from the key names and tensor shapes (the reference builds it from constructor kwargs), so it
shares no structure with either the reference modules or the product's ``arch.py`` plan compiler.
Arithmetic is plain fp32 ATen on CPU -- the same third-party arithmetic the reference itself bottoms out
in -- so with identical weights it reproduces the reference module to fp32 rounding.

Pinning: ``tests/golden/net_*.npz`` hold outputs of the *real* reference modules (generated in the build
container by ``oracle/gen_golden.py``); ``tests/test_oracle_golden.py`` checks this file against them.
"""
from __future__ import annotations

import contextlib
import math
import re
from collections import OrderedDict

import torch
import torch.nn.functional as F

# Reduced-precision variant of the oracle (networks_edm.py:486 `use_fp16`, :79 `w.to(x.dtype)`): inside ``operands_f16(pred)`` the
# multiplicands of every layer whose reference prefix satisfies ``pred`` ('model.enc.32x32_block0.conv0', '...skip', '...qkv',
# '...proj', '...attention') are rounded to fp16 (round to nearest even, like ``.to(torch.float16)``) and multiplied / accumulated in
# fp32 -- the arithmetic of the product's fp16-operand kernels, on the CPU.  Everything else (norms, SiLU, softmax, residuals,
# embedding path, storage) stays fp32.  Outside the context manager the oracle is the pinned fp32 restatement.
# ``stored(prefix)`` (optional) names the layers whose OUTPUT the fp16 mode stores in fp16 -- '...conv0' (the tensor norm1 reads),
# '...conv1' (the block output: the residual stream, networks_edm.py:165-179 run in x.dtype), '...proj' (the attention block's output):
# that value is rounded to fp16 right where the reference holds an fp16 tensor; arithmetic on it stays fp32.
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


def _silu(x):
    return F.silu(x)


def _gn(p, prefix, x, eps):
    c = x.shape[1]
    groups = min(32, c // 4)                      # networks_edm.py:91
    return F.group_norm(x, groups, p[prefix + '.weight'], p[prefix + '.bias'], eps)


def _lin(p, prefix, x):
    y = x @ p[prefix + '.weight'].t()             # networks_edm.py:32
    if prefix + '.bias' in p:
        y = y + p[prefix + '.bias']
    return y


def _resample(x, up, down):
    # resample_filter=[1,1]: f = [[1,1],[1,1]]/4 (networks_edm.py:56-57).
    c = x.shape[1]
    if up:      # conv_transpose2d with 4*f, stride 2 (networks_edm.py:75)
        f = torch.ones(c, 1, 2, 2, dtype=x.dtype, device=x.device)
        x = F.conv_transpose2d(x, f, groups=c, stride=2, padding=0)
    if down:    # conv2d with f, stride 2 (networks_edm.py:77)
        f = torch.full((c, 1, 2, 2), 0.25, dtype=x.dtype, device=x.device)
        x = F.conv2d(x, f, groups=c, stride=2, padding=0)
    return x


def _conv(p, prefix, x, up=False, down=False):
    x = _resample(x, up, down)
    w = p.get(prefix + '.weight')
    if w is not None:
        x, w = _rnd(prefix, x, w)
        x = F.conv2d(x, w, padding=w.shape[-1] // 2)       # networks_edm.py:79
        b = p.get(prefix + '.bias')
        if b is not None:
            x = x + b.reshape(1, -1, 1, 1)
    return x


def _pos_emb(x, num_channels, endpoint, max_positions=10000):
    half = num_channels // 2
    freqs = torch.arange(0, half, dtype=torch.float32, device=x.device)
    freqs = freqs / (half - (1 if endpoint else 0))
    freqs = (1 / max_positions) ** freqs
    x = x.ger(freqs.to(x.dtype))
    return torch.cat([x.cos(), x.sin()], dim=1)


def _block(p, prefix, x, emb, *, up, down, adaptive, skip_scale, eps, heads, taps=None):
    orig = x
    x = _conv(p, prefix + '.conv0', _silu(_gn(p, prefix + '.norm0', x, eps)), up=up, down=down)
    params = _lin(p, prefix + '.affine', emb)[:, :, None, None]
    if adaptive:
        scale, shift = params.chunk(2, dim=1)
        x = _silu(torch.addcmul(shift, _gn(p, prefix + '.norm1', _stored(prefix + '.conv0', x), eps), scale + 1))
    else:
        x = _silu(_gn(p, prefix + '.norm1', _stored(prefix + '.conv0', x + params), eps))
    x = _conv(p, prefix + '.conv1', x)
    has_skip_w = (prefix + '.skip.weight') in p
    if has_skip_w or up or down:
        s = _conv(p, prefix + '.skip', orig, up=up, down=down)
        if not has_skip_w:
            s = _stored(prefix + '.conv1', s)          # a resampled identity skip is itself an fp16 tensor in that mode
    else:
        s = orig
    x = _stored(prefix + '.conv1', (x + s) * skip_scale)
    if heads:
        n, c = x.shape[0], x.shape[1]
        qkv = _stored(prefix + '.qkv', _conv(p, prefix + '.qkv', _gn(p, prefix + '.norm2', x, eps)))
        q, k, v = qkv.reshape(n * heads, c // heads, 3, -1).unbind(2)
        if _F16_PRED is not None and _F16_PRED(prefix + '.attention'):
            # the fp16-operand attention kernel's arithmetic (see oracle/ldm_net.py:_attn): q * d^-1/2 * log2(e) rounded, k, v rounded,
            # un-normalised exp2 weights rounded for the P V product, fp32 row sums of the unrounded weights.  When q | k | v are stored
            # fp16 tensors (the qkv projection's output is rounded above) the factor multiplies the fp32 scores instead
            f = math.log2(math.e) / math.sqrt(k.shape[1])
            if _F16_STORED is not None and _F16_STORED(prefix + '.qkv'):
                sc = torch.einsum('ncq,nck->nqk', q, k) * f
            else:
                q, k, v = _rnd(prefix + '.attention', q * f, k, v)
                sc = torch.einsum('ncq,nck->nqk', q, k)
            pw = torch.exp2(sc - sc.max(dim=2, keepdim=True).values)
            a = torch.einsum('nqk,nck->ncq', _rnd(prefix + '.attention', pw)[0] / pw.sum(2, keepdim=True), v)
        else:
            w = torch.einsum('ncq,nck->nqk', q, k / math.sqrt(k.shape[1])).softmax(dim=2)   # networks_edm.py:108
            a = torch.einsum('nqk,nck->ncq', w, v)
        x = _stored(prefix + '.proj', (_conv(p, prefix + '.proj', a.reshape(*x.shape)) + x) * skip_scale)
    return x


def _layer_names(p, side):
    """Ordered unique 'model.<side>.<name>' prefixes, in state_dict order."""
    seen = OrderedDict()
    pat = re.compile(r'^model\.%s\.([^.]+)\.' % side)
    for k in p.keys():
        m = pat.match(k)
        if m:
            seen[m.group(1)] = True
    return list(seen.keys())


def unet_forward(p, cfg, x, noise_labels, class_labels=None, taps=None):
    """SongUNet / DhariwalUNet forward.  ``taps`` (optional dict) receives every block output by name."""
    song = cfg['model_type'] == 'SongUNet'
    emb_in = p['model.map_layer0.weight'].shape[1]
    if song:
        emb = _pos_emb(noise_labels, emb_in, endpoint=True)
        emb = emb.reshape(emb.shape[0], 2, -1).flip(1).reshape(*emb.shape)          # networks_edm.py:315
        if 'model.map_label.weight' in p:
            emb = emb + _lin(p, 'model.map_label', class_labels * math.sqrt(p['model.map_label.weight'].shape[1]))
        emb = _silu(_lin(p, 'model.map_layer0', emb))
        emb = _silu(_lin(p, 'model.map_layer1', emb))
        kw = dict(adaptive=False, skip_scale=math.sqrt(0.5), eps=1e-6)
    else:
        emb = _pos_emb(noise_labels, emb_in, endpoint=False)
        emb = _silu(_lin(p, 'model.map_layer0', emb))
        emb = _lin(p, 'model.map_layer1', emb)
        if 'model.map_label.weight' in p:
            emb = emb + _lin(p, 'model.map_label', class_labels)
        emb = _silu(emb)
        kw = dict(adaptive=True, skip_scale=1.0, eps=1e-5)
    if taps is not None:
        taps['emb'] = emb

    def heads_of(prefix):
        if (prefix + '.qkv.weight') not in p:
            return 0
        return 1 if song else p[prefix + '.conv1.weight'].shape[0] // 64

    skips = []
    for name in _layer_names(p, 'enc'):
        prefix = 'model.enc.' + name
        if (prefix + '.conv0.weight') in p:
            x = _block(p, prefix, x, emb, up=False, down=name.endswith('_down'), heads=heads_of(prefix), **kw)
        else:
            x = _conv(p, prefix, x)
        skips.append(x)
        if taps is not None:
            taps['enc.' + name] = x
    out = None
    for name in _layer_names(p, 'dec'):
        prefix = 'model.dec.' + name
        if name.endswith('aux_norm'):
            out = _gn(p, prefix, x, 1e-6)
        elif name.endswith('aux_conv'):
            out = _conv(p, prefix, _silu(out))
        else:
            cin = p[prefix + '.conv0.weight'].shape[1]
            if x.shape[1] != cin:
                x = torch.cat([x, skips.pop()], dim=1)
            x = _block(p, prefix, x, emb, up=name.endswith('_up'), down=False, heads=heads_of(prefix), **kw)
            if taps is not None:
                taps['dec.' + name] = x
    if not song:
        out = _conv(p, 'model.out_conv', _silu(_gn(p, 'model.out_norm', x, 1e-5)))
    return out


def edm_denoise(p, cfg, x, sigma, class_labels=None, taps=None):
    """EDMPrecond.forward: D(x; sigma) = c_skip x + c_out F(c_in x, ln(sigma)/4)."""
    x = x.to(torch.float32)
    sigma = torch.as_tensor(sigma, dtype=torch.float32).reshape(-1, 1, 1, 1)
    label_dim = cfg.get('label_dim', 0)
    if label_dim == 0:
        class_labels = None
    elif class_labels is None:
        class_labels = torch.zeros([1, label_dim], device=x.device)
    else:
        class_labels = class_labels.to(torch.float32).reshape(-1, label_dim)
    sd = cfg.get('sigma_data', 0.5)
    c_skip = sd ** 2 / (sigma ** 2 + sd ** 2)
    c_out = sigma * sd / (sigma ** 2 + sd ** 2).sqrt()
    c_in = 1 / (sd ** 2 + sigma ** 2).sqrt()
    c_noise = sigma.log() / 4
    f_x = unet_forward(p, cfg, c_in * x, c_noise.flatten(), class_labels, taps=taps)
    if taps is not None:
        taps['F_x'] = f_x
    return c_skip * x + c_out * f_x


class OracleNet:
    """Callable with the attributes the samplers read off ``net`` (SURVEY.md section 8b 'net protocol')."""

    def __init__(self, params, cfg):
        self.params = params
        self.cfg = dict(cfg)
        self.img_resolution = cfg['img_resolution']
        self.img_channels = cfg.get('img_channels', 3)
        self.label_dim = cfg.get('label_dim', 0)
        self.sigma_min = cfg.get('sigma_min', 0.002)
        self.sigma_max = cfg.get('sigma_max', 80.0)
        self.last_bottleneck = None          # AMED tap (solvers_amed.py:16-17)

    def __call__(self, x, sigma, class_labels=None):
        taps = {}
        out = edm_denoise(self.params, self.cfg, x, sigma, class_labels, taps=taps)
        key = 'enc.8x8_block2' if class_labels is not None else 'enc.8x8_block3'
        self.last_bottleneck = taps.get(key)
        return out
