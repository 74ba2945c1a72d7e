"""Architecture description of the latent-diffusion U-Net (Stable Diffusion v1.x family) behind ``CFGPrecond``.

The reference builds it imperatively in ``UNetModel.__init__`` (diff-solvers-main/models/ldm/modules/diffusionmodules/
openaimodel.py:413-742: ``ResBlock`` :163-274, ``Downsample`` :133-160, ``Upsample`` :86-118, ``SpatialTransformer``
ldm/modules/attention.py:218-260 with ``BasicTransformerBlock`` :196-215, ``CrossAttention`` :152-194, GEGLU
``FeedForward`` :45-72).  As for the EDM nets (``arch.py``) the HIP engine runs a flat plan compiled from a data model:
``LDMUNetSpec`` lists every layer with its channels, resolution and the *reference state_dict key prefix* of its weights,
so a real SD checkpoint (keys ``model.diffusion_model.*`` stripped to ``input_blocks.* ...``) binds by name.

Supported: the ``use_spatial_transformer=True`` family (``v1-inference.yaml``): ``conv_resample=True``,
``resblock_updown=False``, ``use_scale_shift_norm=False``, ``num_classes=None``, ``transformer_depth=1``.  Anything else
raises NotImplementedError.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np
import torch


@dataclass
class LDMLayer:
    kind: str                  # 'stem' | 'res' | 'st' | 'down' | 'up'
    key: str                   # state_dict prefix, e.g. 'input_blocks.4.0'
    cin: int
    cout: int
    res_in: int
    res_out: int
    heads: int = 0             # 'st' only
    skip_conv: bool = False    # 'res' only: 1x1 skip_connection


@dataclass
class LDMBlock:
    name: str                  # 'input_blocks.4' | 'middle_block' | 'output_blocks.7'
    layers: List[LDMLayer]
    pushes_skip: bool = False
    pops_skip: bool = False
    skip_cin: int = 0


@dataclass
class LDMUNetSpec:
    img_resolution: int        # latent resolution (64 for SD 512x512)
    in_channels: int
    out_channels: int
    model_channels: int
    time_embed_dim: int
    context_dim: int
    num_heads: int
    blocks: List[LDMBlock] = field(default_factory=list)
    # CFGPrecond / LatentDiffusion noise schedule (ddpm.py:118-138, v1-inference.yaml)
    linear_start: float = 0.00085
    linear_end: float = 0.0120
    timesteps: int = 1000
    epsilon_t: float = 1e-3
    guidance_rate: float = 7.5
    guidance_type: str = 'classifier-free'

    def res_layers(self):
        return [l for b in self.blocks for l in b.layers if l.kind == 'res']


def ldm_unet_spec(img_resolution=64, in_channels=4, out_channels=4, model_channels=320, attention_resolutions=(4, 2, 1),
                  num_res_blocks=2, channel_mult=(1, 2, 4, 4), num_heads=8, context_dim=768, transformer_depth=1,
                  conv_resample=True, resblock_updown=False, use_scale_shift_norm=False, num_classes=None,
                  use_spatial_transformer=True, **precond) -> LDMUNetSpec:
    """Layer list in the construction order of openaimodel.py:497-692."""
    if not use_spatial_transformer or transformer_depth != 1 or not conv_resample or resblock_updown \
            or use_scale_shift_norm or num_classes is not None:
        raise NotImplementedError('only the spatial-transformer UNetModel of v1-inference.yaml is supported')
    spec = LDMUNetSpec(img_resolution, in_channels, out_channels, model_channels, model_channels * 4, context_dim, num_heads,
                       **precond)
    res = img_resolution
    ch = model_channels
    spec.blocks.append(LDMBlock('input_blocks.0', [LDMLayer('stem', 'input_blocks.0.0', in_channels, ch, res, res)], pushes_skip=True))
    chans = [ch]
    ds = 1
    idx = 1
    for level, mult in enumerate(channel_mult):
        for _ in range(num_res_blocks):
            cout = mult * model_channels
            layers = [LDMLayer('res', f'input_blocks.{idx}.0', ch, cout, res, res, skip_conv=(ch != cout))]
            ch = cout
            if ds in attention_resolutions:
                layers.append(LDMLayer('st', f'input_blocks.{idx}.1', ch, ch, res, res, heads=num_heads))
            spec.blocks.append(LDMBlock(f'input_blocks.{idx}', layers, pushes_skip=True))
            chans.append(ch)
            idx += 1
        if level != len(channel_mult) - 1:
            spec.blocks.append(LDMBlock(f'input_blocks.{idx}', [LDMLayer('down', f'input_blocks.{idx}.0', ch, ch, res, res // 2)],
                                        pushes_skip=True))
            chans.append(ch)
            idx += 1
            ds *= 2
            res //= 2
    spec.blocks.append(LDMBlock('middle_block', [LDMLayer('res', 'middle_block.0', ch, ch, res, res),
                                                 LDMLayer('st', 'middle_block.1', ch, ch, res, res, heads=num_heads),
                                                 LDMLayer('res', 'middle_block.2', ch, ch, res, res)]))
    idx = 0
    for level, mult in list(enumerate(channel_mult))[::-1]:
        for i in range(num_res_blocks + 1):
            ich = chans.pop()
            cout = model_channels * mult
            layers = [LDMLayer('res', f'output_blocks.{idx}.0', ch + ich, cout, res, res, skip_conv=(ch + ich != cout))]
            ch = cout
            if ds in attention_resolutions:
                layers.append(LDMLayer('st', f'output_blocks.{idx}.{len(layers)}', ch, ch, res, res, heads=num_heads))
            if level and i == num_res_blocks:
                layers.append(LDMLayer('up', f'output_blocks.{idx}.{len(layers)}', ch, ch, res, res * 2))
                ds //= 2
                res *= 2
            spec.blocks.append(LDMBlock(f'output_blocks.{idx}', layers, pops_skip=True, skip_cin=ich))
            idx += 1
    assert not chans
    return spec


NAMED_LDM_CONFIGS = {
    # Stable Diffusion v1.5 (BASELINE config 5; models/ldm/configs/stable-diffusion/v1-inference.yaml)
    'sd15': dict(img_resolution=64, in_channels=4, out_channels=4, model_channels=320, attention_resolutions=(4, 2, 1),
                 num_res_blocks=2, channel_mult=(1, 2, 4, 4), num_heads=8, context_dim=768),
    # same topology at test size
    'tiny_ldm': dict(img_resolution=32, in_channels=4, out_channels=4, model_channels=64, attention_resolutions=(4, 2, 1),
                     num_res_blocks=2, channel_mult=(1, 2, 4, 4), num_heads=4, context_dim=96),
    'tiny_ldm_1res': dict(img_resolution=16, in_channels=4, out_channels=4, model_channels=32, attention_resolutions=(2, 1),
                          num_res_blocks=1, channel_mult=(1, 2, 2), num_heads=2, context_dim=64),
}


def _lin(keys, prefix, fin, fout, bias=True):
    keys.append((f'{prefix}.weight', (fout, fin), ('w', fin)))
    if bias:
        keys.append((f'{prefix}.bias', (fout,), ('b',)))


def _conv(keys, prefix, cin, cout, k):
    keys.append((f'{prefix}.weight', (cout, cin, k, k), ('w', cin * k * k)))
    keys.append((f'{prefix}.bias', (cout,), ('b',)))


def _norm(keys, prefix, c):
    keys.append((f'{prefix}.weight', (c,), ('g',)))
    keys.append((f'{prefix}.bias', (c,), ('b',)))


def ldm_param_table(spec: LDMUNetSpec) -> List[Tuple[str, Tuple[int, ...], tuple]]:
    """Every learnable tensor of the UNetModel, keyed like its state_dict."""
    keys: list = []
    E = spec.time_embed_dim
    _lin(keys, 'time_embed.0', spec.model_channels, E)
    _lin(keys, 'time_embed.2', E, E)
    for b in spec.blocks:
        for l in b.layers:
            p = l.key
            if l.kind == 'stem':
                _conv(keys, p, l.cin, l.cout, 3)
            elif l.kind == 'res':
                _norm(keys, f'{p}.in_layers.0', l.cin)
                _conv(keys, f'{p}.in_layers.2', l.cin, l.cout, 3)
                _lin(keys, f'{p}.emb_layers.1', E, l.cout)
                _norm(keys, f'{p}.out_layers.0', l.cout)
                _conv(keys, f'{p}.out_layers.3', l.cout, l.cout, 3)
                if l.skip_conv:
                    _conv(keys, f'{p}.skip_connection', l.cin, l.cout, 1)
            elif l.kind == 'st':
                c = l.cin
                _norm(keys, f'{p}.norm', c)
                _conv(keys, f'{p}.proj_in', c, c, 1)
                t = f'{p}.transformer_blocks.0'
                for a, ctx in (('attn1', c), ('attn2', spec.context_dim)):
                    _lin(keys, f'{t}.{a}.to_q', c, c, bias=False)
                    _lin(keys, f'{t}.{a}.to_k', ctx, c, bias=False)
                    _lin(keys, f'{t}.{a}.to_v', ctx, c, bias=False)
                    _lin(keys, f'{t}.{a}.to_out.0', c, c)
                _lin(keys, f'{t}.ff.net.0.proj', c, 8 * c)
                _lin(keys, f'{t}.ff.net.2', 4 * c, c)
                for n in ('norm1', 'norm2', 'norm3'):
                    _norm(keys, f'{t}.{n}', c)
                _conv(keys, f'{p}.proj_out', c, c, 1)
            elif l.kind == 'down':
                _conv(keys, f'{p}.op', l.cin, l.cout, 3)
            elif l.kind == 'up':
                _conv(keys, f'{p}.conv', l.cin, l.cout, 3)
    _norm(keys, 'out.0', spec.model_channels)
    _conv(keys, 'out.2', spec.model_channels, spec.out_channels, 3)
    return keys


def init_ldm_params(spec: LDMUNetSpec, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Deterministic CPU-generated weights, every tensor carrying signal (weights ~ N(0, 1/fan_in), biases ~ N(0, 0.1^2),
    norm gains 1 + N(0, 0.1^2)).  The reference zero-initialises ``out_layers.3``, ``proj_out`` and ``out.2``
    (``zero_module``), which would make random-init parity vacuous."""
    g = torch.Generator(device='cpu').manual_seed(int(seed))
    out: Dict[str, torch.Tensor] = {}
    for key, shape, rule in ldm_param_table(spec):
        if rule[0] == 'w':
            t = torch.randn(shape, generator=g) * (1.0 / math.sqrt(rule[1]))
        elif rule[0] == 'b':
            t = torch.randn(shape, generator=g) * 0.1
        else:
            t = 1.0 + torch.randn(shape, generator=g) * 0.1
        out[key] = t.to(torch.float32).contiguous()
    return out


def alphas_cumprod(spec: LDMUNetSpec) -> torch.Tensor:
    """float32 ``alphas_cumprod`` of the 'linear' beta schedule (ldm util.py:21-25, ddpm.py:125-138)."""
    betas = np.linspace(spec.linear_start ** 0.5, spec.linear_end ** 0.5, spec.timesteps, dtype=np.float64) ** 2
    return torch.tensor(np.cumprod(1.0 - betas, axis=0), dtype=torch.float32)


def ldm_flops_per_image(spec: LDMUNetSpec, context_len: int = 77) -> float:
    """Algorithmic FLOPs (2 x MAC) of one U-Net forward on one latent: convs, linears, attention."""
    f = 2.0 * (spec.model_channels * spec.time_embed_dim + spec.time_embed_dim ** 2)
    for b in spec.blocks:
        for l in b.layers:
            hw = l.res_out * l.res_out
            if l.kind in ('stem', 'down', 'up'):
                f += 2.0 * hw * 9 * l.cin * l.cout
            elif l.kind == 'res':
                f += 2.0 * hw * 9 * (l.cin * l.cout + l.cout * l.cout) + 2.0 * spec.time_embed_dim * l.cout
                if l.skip_conv:
                    f += 2.0 * hw * l.cin * l.cout
            elif l.kind == 'st':
                c = l.cin
                f += 2.0 * hw * c * c * 2                                   # proj_in, proj_out
                f += 2.0 * hw * c * c * 4 + 4.0 * hw * hw * c               # self-attention
                f += 2.0 * hw * c * c * 2 + 2.0 * context_len * spec.context_dim * c * 2 + 4.0 * hw * context_len * c
                f += 2.0 * hw * c * 8 * c + 2.0 * hw * 4 * c * c            # GEGLU feed-forward
    f += 2.0 * spec.img_resolution ** 2 * 9 * spec.model_channels * spec.out_channels
    return f
