"""Denoiser architecture descriptions (EDM SongUNet / DhariwalUNet).

The reference builds its U-Nets imperatively inside ``torch.nn.Module``
constructors (diff-solvers-main/models/networks_edm.py:221-310 SongUNet,
:364-425 DhariwalUNet, :126-156 UNetBlock).  The HIP engine does not run
modules; it runs a flat plan.  This file is the data model the plan is compiled
from: a ``UNetSpec`` listing every layer with its channel counts, resolution,
resampling, attention heads and the *reference state_dict key prefix* its
weights live under, so that a pickled EDM network (or a random-init one with the
same key names) can be bound by name.

Only what the BASELINE configs use is described: DDPM++ (``embedding_type=
'positional'``, ``encoder_type='standard'``, ``decoder_type='standard'``,
``resample_filter=[1,1]``) and ADM.  NCSN++ options raise NotImplementedError.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import torch


@dataclass
class BlockSpec:
    name: str                 # e.g. 'enc.16x16_block0'
    kind: str                 # 'conv' (plain 3x3 stem) | 'block' (UNetBlock)
    cin: int
    cout: int
    res_in: int               # spatial size of the block input
    res_out: int              # spatial size of the block output
    up: bool = False
    down: bool = False
    heads: int = 0            # 0 = no attention
    skip_conv: bool = False   # 1x1 projection on the skip path
    adaptive_scale: bool = False
    skip_scale: float = 1.0
    eps: float = 1e-5
    pops_skip: bool = False   # decoder block that concatenates an encoder skip
    skip_cin: int = 0         # channels of the popped skip tensor (cin = x.C + skip_cin)
    pushes_skip: bool = False  # encoder layer whose output is pushed on the skip stack


@dataclass
class UNetSpec:
    model_type: str                     # 'SongUNet' | 'DhariwalUNet'
    img_resolution: int
    in_channels: int
    out_channels: int
    label_dim: int
    augment_dim: int
    model_channels: int
    emb_channels: int
    noise_channels: int
    pos_endpoint: bool                  # PositionalEmbedding(endpoint=...)
    pos_max_positions: int
    swap_sincos: bool                   # SongUNet flips [cos|sin] -> [sin|cos] (networks_edm.py:315)
    out_norm: str                       # key of the output GroupNorm, relative to 'model.'
    out_conv: str                       # key of the output conv
    out_eps: float
    blocks: List[BlockSpec] = field(default_factory=list)
    # EDMPrecond (networks_edm.py:460-480)
    sigma_data: float = 0.5
    sigma_min: float = 0.002
    sigma_max: float = 80.0
    use_fp16: bool = False

    @property
    def enc(self) -> List[BlockSpec]:
        return [b for b in self.blocks if b.name.startswith('enc.')]

    @property
    def dec(self) -> List[BlockSpec]:
        return [b for b in self.blocks if b.name.startswith('dec.')]


def song_unet_spec(
    img_resolution, in_channels, out_channels, label_dim=0, augment_dim=0,
    model_channels=128, channel_mult=(1, 2, 2, 2), channel_mult_emb=4, num_blocks=4,
    attn_resolutions=(16,), dropout=0.10, label_dropout=0, embedding_type='positional',
    channel_mult_noise=1, encoder_type='standard', decoder_type='standard',
    resample_filter=(1, 1),
) -> UNetSpec:
    """DDPM++ layer list; mirrors the construction order of networks_edm.py:260-310."""
    if embedding_type != 'positional' or encoder_type != 'standard' or decoder_type != 'standard' \
            or list(resample_filter) != [1, 1]:
        raise NotImplementedError('only the DDPM++ variant of SongUNet (positional/standard/standard/[1,1]) is supported')
    emb_channels = model_channels * channel_mult_emb
    noise_channels = model_channels * channel_mult_noise
    spec = UNetSpec(
        model_type='SongUNet', img_resolution=img_resolution, in_channels=in_channels, out_channels=out_channels,
        label_dim=label_dim, augment_dim=augment_dim, model_channels=model_channels, emb_channels=emb_channels,
        noise_channels=noise_channels, pos_endpoint=True, pos_max_positions=10000, swap_sincos=True,
        out_norm=f'dec.{img_resolution}x{img_resolution}_aux_norm', out_conv=f'dec.{img_resolution}x{img_resolution}_aux_conv',
        out_eps=1e-6,
    )
    common = dict(skip_scale=math.sqrt(0.5), eps=1e-6, adaptive_scale=False)
    cout = in_channels
    skips: List[int] = []
    for level, mult in enumerate(channel_mult):
        res = img_resolution >> level
        if level == 0:
            cin, cout = cout, model_channels
            spec.blocks.append(BlockSpec(f'enc.{res}x{res}_conv', 'conv', cin, cout, res, res, pushes_skip=True))
        else:
            spec.blocks.append(BlockSpec(f'enc.{res}x{res}_down', 'block', cout, cout, res * 2, res, down=True,
                                         skip_conv=True, pushes_skip=True, **common))
        skips.append(cout)
        for idx in range(num_blocks):
            cin, cout = cout, model_channels * mult
            heads = 1 if res in attn_resolutions else 0
            spec.blocks.append(BlockSpec(f'enc.{res}x{res}_block{idx}', 'block', cin, cout, res, res, heads=heads,
                                         skip_conv=(cin != cout), pushes_skip=True, **common))
            skips.append(cout)
    for level, mult in reversed(list(enumerate(channel_mult))):
        res = img_resolution >> level
        if level == len(channel_mult) - 1:
            spec.blocks.append(BlockSpec(f'dec.{res}x{res}_in0', 'block', cout, cout, res, res, heads=1, **common))
            spec.blocks.append(BlockSpec(f'dec.{res}x{res}_in1', 'block', cout, cout, res, res, **common))
        else:
            spec.blocks.append(BlockSpec(f'dec.{res}x{res}_up', 'block', cout, cout, res // 2, res, up=True,
                                         skip_conv=True, **common))
        for idx in range(num_blocks + 1):
            sc = skips.pop()
            cin, cout = cout + sc, model_channels * mult
            heads = 1 if (idx == num_blocks and res in attn_resolutions) else 0
            spec.blocks.append(BlockSpec(f'dec.{res}x{res}_block{idx}', 'block', cin, cout, res, res, heads=heads,
                                         skip_conv=(cin != cout), pops_skip=True, skip_cin=sc, **common))
    assert not skips
    return spec


def dhariwal_unet_spec(
    img_resolution, in_channels, out_channels, label_dim=0, augment_dim=0,
    model_channels=192, channel_mult=(1, 2, 3, 4), channel_mult_emb=4, num_blocks=3,
    attn_resolutions=(32, 16, 8), dropout=0.10, label_dropout=0,
) -> UNetSpec:
    """ADM layer list; mirrors networks_edm.py:387-425."""
    emb_channels = model_channels * channel_mult_emb
    spec = UNetSpec(
        model_type='DhariwalUNet', img_resolution=img_resolution, in_channels=in_channels, out_channels=out_channels,
        label_dim=label_dim, augment_dim=augment_dim, model_channels=model_channels, emb_channels=emb_channels,
        noise_channels=model_channels, pos_endpoint=False, pos_max_positions=10000, swap_sincos=False,
        out_norm='out_norm', out_conv='out_conv', out_eps=1e-5,
    )
    common = dict(skip_scale=1.0, eps=1e-5, adaptive_scale=True)

    def heads_of(c, attn):
        return c // 64 if attn else 0

    cout = in_channels
    skips: List[int] = []
    for level, mult in enumerate(channel_mult):
        res = img_resolution >> level
        if level == 0:
            cin, cout = cout, model_channels * mult
            spec.blocks.append(BlockSpec(f'enc.{res}x{res}_conv', 'conv', cin, cout, res, res, pushes_skip=True))
        else:
            spec.blocks.append(BlockSpec(f'enc.{res}x{res}_down', 'block', cout, cout, res * 2, res, down=True,
                                         pushes_skip=True, **common))
        skips.append(cout)
        for idx in range(num_blocks):
            cin, cout = cout, model_channels * mult
            spec.blocks.append(BlockSpec(f'enc.{res}x{res}_block{idx}', 'block', cin, cout, res, res,
                                         heads=heads_of(cout, res in attn_resolutions), skip_conv=(cin != cout),
                                         pushes_skip=True, **common))
            skips.append(cout)
    for level, mult in reversed(list(enumerate(channel_mult))):
        res = img_resolution >> level
        if level == len(channel_mult) - 1:
            spec.blocks.append(BlockSpec(f'dec.{res}x{res}_in0', 'block', cout, cout, res, res, heads=heads_of(cout, True), **common))
            spec.blocks.append(BlockSpec(f'dec.{res}x{res}_in1', 'block', cout, cout, res, res, **common))
        else:
            spec.blocks.append(BlockSpec(f'dec.{res}x{res}_up', 'block', cout, cout, res // 2, res, up=True, **common))
        for idx in range(num_blocks + 1):
            sc = skips.pop()
            cin, cout = cout + sc, model_channels * mult
            spec.blocks.append(BlockSpec(f'dec.{res}x{res}_block{idx}', 'block', cin, cout, res, res,
                                         heads=heads_of(cout, res in attn_resolutions), skip_conv=(cin != cout),
                                         pops_skip=True, skip_cin=sc, **common))
    assert not skips
    return spec


def edm_precond_spec(img_resolution, img_channels, label_dim=0, use_fp16=False, sigma_min=0.002, sigma_max=80.0,
                     sigma_data=0.5, model_type='DhariwalUNet', **model_kwargs) -> UNetSpec:
    """Keyword-compatible with the reference ``EDMPrecond.__init__`` (networks_edm.py:461-480)."""
    builder = {'SongUNet': song_unet_spec, 'DhariwalUNet': dhariwal_unet_spec}.get(model_type)
    if builder is None:
        raise ValueError(f'unknown model_type {model_type!r}')
    spec = builder(img_resolution=img_resolution, in_channels=img_channels, out_channels=img_channels,
                   label_dim=label_dim, **model_kwargs)
    spec.sigma_data, spec.sigma_min, spec.sigma_max, spec.use_fp16 = sigma_data, sigma_min, sigma_max, use_fp16
    return spec


# Named configurations.  Architecture kwargs are the ones the reference itself states
# (sfd-main/training/training_loop.py:62-76).
NAMED_CONFIGS: Dict[str, dict] = {
    'cifar10': dict(img_resolution=32, img_channels=3, label_dim=0, model_type='SongUNet', embedding_type='positional',
                    encoder_type='standard', decoder_type='standard', channel_mult_noise=1, resample_filter=[1, 1],
                    model_channels=128, channel_mult=[2, 2, 2], dropout=0.13, augment_dim=9),
    'ffhq': dict(img_resolution=64, img_channels=3, label_dim=0, model_type='SongUNet', embedding_type='positional',
                 encoder_type='standard', decoder_type='standard', channel_mult_noise=1, resample_filter=[1, 1],
                 model_channels=128, channel_mult=[1, 2, 2, 2], dropout=0.05, augment_dim=9),
    'afhqv2': dict(img_resolution=64, img_channels=3, label_dim=0, model_type='SongUNet', embedding_type='positional',
                   encoder_type='standard', decoder_type='standard', channel_mult_noise=1, resample_filter=[1, 1],
                   model_channels=128, channel_mult=[1, 2, 2, 2], dropout=0.05, augment_dim=9),
    'imagenet64': dict(img_resolution=64, img_channels=3, label_dim=1000, model_type='DhariwalUNet',
                       model_channels=192, channel_mult=[1, 2, 3, 4]),
    # Reduced nets used by the parity tests (same code paths, seconds on CPU).
    'tiny_song': dict(img_resolution=16, img_channels=3, label_dim=0, model_type='SongUNet', model_channels=32,
                      channel_mult=[1, 2], num_blocks=1, attn_resolutions=[8], augment_dim=9),
    'tiny_song_cond': dict(img_resolution=16, img_channels=3, label_dim=10, model_type='SongUNet', model_channels=32,
                           channel_mult=[1, 2], num_blocks=1, attn_resolutions=[8], augment_dim=0),
    # tiny nets with the '8x8_block2/3' encoder layers the AMED bottleneck tap needs (solvers_amed.py:16-17).
    'tiny_song_amed': dict(img_resolution=16, img_channels=3, label_dim=0, model_type='SongUNet', model_channels=32,
                           channel_mult=[1, 2], num_blocks=4, attn_resolutions=[8], augment_dim=9),
    'tiny_song_amed_cond': dict(img_resolution=16, img_channels=3, label_dim=10, model_type='SongUNet', model_channels=32,
                                channel_mult=[1, 2], num_blocks=4, attn_resolutions=[8], augment_dim=0),
    'tiny_adm': dict(img_resolution=16, img_channels=3, label_dim=10, model_type='DhariwalUNet', model_channels=64,
                     channel_mult=[1, 2], num_blocks=1, attn_resolutions=[8]),
}


# ------------------------------------------------------------------------------------------------
# Parameter table: (state_dict key, shape, init rule).  Key names are the reference's.

def _linear(keys, prefix, fin, fout, bias=True):
    keys.append((f'{prefix}.weight', (fout, fin), ('linear', fin, fout)))
    if bias:
        keys.append((f'{prefix}.bias', (fout,), ('linear_bias', fin, fout)))


def _conv(keys, prefix, cin, cout, k):
    keys.append((f'{prefix}.weight', (cout, cin, k, k), ('conv', cin * k * k, cout * k * k)))
    keys.append((f'{prefix}.bias', (cout,), ('conv_bias', cin * k * k, cout * k * k)))


def _gn(keys, prefix, c):
    keys.append((f'{prefix}.weight', (c,), ('ones',)))
    keys.append((f'{prefix}.bias', (c,), ('zeros',)))


def param_table(spec: UNetSpec) -> List[Tuple[str, Tuple[int, ...], tuple]]:
    """Every learnable tensor of ``EDMPrecond(model=<spec>)`` in state_dict order."""
    keys: list = []
    m = 'model'
    if spec.model_type == 'SongUNet':
        if spec.label_dim:
            _linear(keys, f'{m}.map_label', spec.label_dim, spec.noise_channels)
        if spec.augment_dim:
            _linear(keys, f'{m}.map_augment', spec.augment_dim, spec.noise_channels, bias=False)
        _linear(keys, f'{m}.map_layer0', spec.noise_channels, spec.emb_channels)
        _linear(keys, f'{m}.map_layer1', spec.emb_channels, spec.emb_channels)
    else:
        if spec.augment_dim:
            _linear(keys, f'{m}.map_augment', spec.augment_dim, spec.model_channels, bias=False)
        _linear(keys, f'{m}.map_layer0', spec.model_channels, spec.emb_channels)
        _linear(keys, f'{m}.map_layer1', spec.emb_channels, spec.emb_channels)
        if spec.label_dim:
            _linear(keys, f'{m}.map_label', spec.label_dim, spec.emb_channels, bias=False)
    for b in spec.blocks:
        p = f'{m}.{b.name}'
        if b.kind == 'conv':
            _conv(keys, p, b.cin, b.cout, 3)
            continue
        _gn(keys, f'{p}.norm0', b.cin)
        _conv(keys, f'{p}.conv0', b.cin, b.cout, 3)
        _linear(keys, f'{p}.affine', spec.emb_channels, b.cout * (2 if b.adaptive_scale else 1))
        _gn(keys, f'{p}.norm1', b.cout)
        _conv(keys, f'{p}.conv1', b.cout, b.cout, 3)
        if b.skip_conv:
            _conv(keys, f'{p}.skip', b.cin, b.cout, 1)
        if b.heads:
            _gn(keys, f'{p}.norm2', b.cout)
            _conv(keys, f'{p}.qkv', b.cout, b.cout * 3, 1)
            _conv(keys, f'{p}.proj', b.cout, b.cout, 1)
    last = spec.blocks[-1].cout
    _gn(keys, f'{m}.{spec.out_norm}', last)
    _conv(keys, f'{m}.{spec.out_conv}', last, spec.out_channels, 3)
    return keys


def num_groups(c: int) -> int:
    """GroupNorm group count rule (networks_edm.py:89-91): min(32, C // 4)."""
    return min(32, c // 4)


def init_params(spec: UNetSpec, seed: int = 0, mode: str = 'signal') -> Dict[str, torch.Tensor]:
    """Deterministic CPU-generated weights keyed like the reference state_dict.

    mode='signal': every tensor carries signal (weights ~ N(0, 1/fan_in), biases ~ N(0, 0.1^2), GroupNorm
        gains 1 + N(0, 0.1^2)).  The reference initialisers zero conv1/proj/out_conv (init_zero,
        networks_edm.py:252,384), which would make parity vacuous, so tests and the
        benchmark use this mode.
    mode='reference': the distributions of the reference initialisers (xavier_uniform / kaiming_uniform with
        the init_weight multipliers) — same statistics, not the same random stream.
    """
    g = torch.Generator(device='cpu').manual_seed(int(seed))
    out: Dict[str, torch.Tensor] = {}
    song = spec.model_type == 'SongUNet'
    for key, shape, rule in param_table(spec):
        kind = rule[0]
        if mode == 'signal':
            if kind in ('conv', 'linear'):
                t = torch.randn(shape, generator=g) * (1.0 / math.sqrt(rule[1]))
            elif kind in ('conv_bias', 'linear_bias', 'zeros'):
                t = torch.randn(shape, generator=g) * 0.1
            else:  # 'ones'
                t = 1.0 + torch.randn(shape, generator=g) * 0.1
        elif mode == 'reference':
            leaf = key.rsplit('.', 2)[-2]
            zero_init = leaf in ('conv1', 'proj') or key.startswith(f'model.{spec.out_conv}')
            if kind in ('conv', 'linear', 'conv_bias', 'linear_bias'):
                fin, fout = rule[1], rule[2]
                if song:
                    bound = math.sqrt(6.0 / (fin + fout))
                    mult = 1e-5 if zero_init else (math.sqrt(0.2) if leaf == 'qkv' else 1.0)
                    if kind.endswith('bias'):
                        mult = 0.0
                else:
                    bound = math.sqrt(3.0 / fin)
                    mult = 0.0 if zero_init else math.sqrt(1.0 / 3.0)
                    if leaf in ('map_label',):
                        bound, mult = 1.0, 1.0  # kaiming_normal * sqrt(label_dim): N(0,1)
                if leaf == 'map_label' and not song:
                    t = torch.randn(shape, generator=g)
                else:
                    t = (torch.rand(shape, generator=g) * 2 - 1) * bound * mult
            elif kind == 'ones':
                t = torch.ones(shape)
            else:
                t = torch.zeros(shape)
        else:
            raise ValueError(mode)
        out[key] = t.to(torch.float32).contiguous()
    return out


def flops_per_image(spec: UNetSpec) -> float:
    """Algorithmic FLOPs (2 x MAC) of one denoiser evaluation on one image: convs, 1x1s, attention, linears."""
    f = 0.0
    for b in spec.blocks:
        hw = b.res_out * b.res_out
        if b.kind == 'conv':
            f += 2.0 * hw * 9 * b.cin * b.cout
            continue
        f += 2.0 * hw * 9 * b.cin * b.cout + 2.0 * hw * 9 * b.cout * b.cout
        f += 2.0 * spec.emb_channels * b.cout * (2 if b.adaptive_scale else 1)
        if b.skip_conv:
            f += 2.0 * hw * b.cin * b.cout
        if b.heads:
            f += 2.0 * hw * b.cout * 3 * b.cout + 2.0 * hw * b.cout * b.cout
            f += 2.0 * 2.0 * hw * hw * b.cout
    f += 2.0 * spec.img_resolution ** 2 * 9 * spec.blocks[-1].cout * spec.out_channels
    f += 2.0 * (spec.noise_channels * spec.emb_channels + spec.emb_channels ** 2)
    return f
