"""Denoiser engine: compiles a ``UNetSpec`` + reference-keyed weights into a flat launch plan of libdsamd kernels.

What it replaces: ``EDMPrecond.forward`` -> ``SongUNet/DhariwalUNet.forward`` -> ``UNetBlock.forward``
(diff-solvers-main/models/networks_edm.py:482-496, :312-355, :427-453, :158-179), i.e. the ~600 ATen launches per
network evaluation of the reference, by ~10 hand-written launches per block:

    GN stats -> normalise+SiLU(+resample) -> 3x3 implicit GEMM (+bias +emb) -> GN stats -> normalise+SiLU
      -> 3x3 implicit GEMM (+bias +skip +scale)   [+ attention: GN, packed 1x1 q|k|v, fused softmax(QK^T)V kernel, 1x1 proj]

Activations are NHWC fp32 and live in engine-owned workspaces (allocated once per batch size through torch, which
is only the allocator here); the decoder's ``torch.cat`` is never materialised (both sources are read in place).
The plan is a list of (C function, prebuilt argument struct): running it is a tight ctypes loop, and because no
pointer changes between calls it can be captured in a hipGraph (see ``graph.py``).

``EDMDenoiser`` is the drop-in for the reference ``net`` object: same call signature and attributes
(``img_resolution, img_channels, label_dim, sigma_min, sigma_max``) plus the raw fast path the fused solvers use.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, List

import torch

from . import _lib, arch
from ._lib import AttnArgs, DS_ACT_NONE, DS_ACT_SILU, DS_RESAMPLE_NONE, DS_RESAMPLE_DOWN, DS_RESAMPLE_UP
from .ops import pack_conv_weight, pack_linear_weight, pack_stem_weight


from .plan import Builder, Plan as _Plan, ptr as _ptr  # noqa: E402


# default of UNetEngine.fuse_norm16: '0' never, '1' every eligible layer, 'auto' the layer classes where the per-layer A/B of round 5 found the
# fusion faster than pass + convolution (profiles/r5_conv_f16dma_fused_norm_per_layer.txt): 64x64 images and one column tile (cout <= 192).
# Whole-net A/B of '1' against '0': profiles/r5_conv_f16dma_fused_norm_ab.txt (ImageNet-64 +0.4 %, SD-1.5 -6.8 %).
FUSE_NORM16_DEFAULT = 'auto'


def fuse_norm16_value(text):
    """'0' / '1' / 'auto' (DS_FUSE_NORM16, or the engines' attribute) -> False / True / 'auto'."""
    return {'0': False, '1': True}.get(str(text), 'auto')


def fuse_norm16_here(mode, side, cout):
    """Does the 3x3 convolution of this geometry normalise its own LDS halo (conv3x3_f16dma NORM) under `mode`?"""
    return mode is True or (mode == 'auto' and side == 64 and cout <= 192)



class UNetEngine:
    def __init__(self, spec: arch.UNetSpec, params: Dict[str, torch.Tensor], device='cuda', use_fp16=False, split_fp16=False):
        """split_fp16: fp32 EMULATED on the fp16 matrix pipe in the 3x3 convolutions -- every operand as fp16 hi + lo, three MFMA
        products per multiplication, fp32 accumulation (ds_conv_args.wgt_f16 == 2); 2**-22 relative per product, i.e. inside every
        fp32 tolerance of this engine, at 16/3 of the fp32 matrix rate.  Not a reduced-precision mode: it is tested against the same
        fp32 goldens and tolerances as the exact fp32 path.
        use_fp16: the reference's reduced-precision mode (networks_edm.py:486: the U-Net body on x.to(float16)).  3x3 convolutions,
        1x1 / Linear layers over image rows and attention multiply fp16 operands on the fp16 matrix pipe with fp32 accumulation, and the
        activations between layers -- the activated GroupNorm outputs, conv0 outputs, the residual stream and the skip stack -- are
        stored in fp16 where the fp16-activation kernels take the geometry (plan(): `stream16`); GroupNorm / SiLU / softmax arithmetic,
        q / k / v and the embedding path are fp32.  DESIGN.md section 2 states the bounds, tests/test_hip_fp16.py enforces them."""
        self.spec = spec
        self.device = torch.device(device)
        self.use_fp16 = bool(use_fp16)
        self.split_fp16 = bool(split_fp16) and not self.use_fp16
        # fp16 mode: GroupNorm apply + SiLU inside the fp16-activation convolution's LDS halo instead of a ds_norm_act pass (plan(): fz0 / fz1;
        # bit-identical results).  An attribute, not an argument, so that A/B runs flip it per engine: DS_FUSE_NORM16 sets the default.
        self.fuse_norm16 = fuse_norm16_value(os.environ.get('DS_FUSE_NORM16', FUSE_NORM16_DEFAULT))
        self._w16_cache = {}
        self.conv_mode = 1 if self.use_fp16 else (2 if self.split_fp16 else 0)      # ds_conv_args.wgt_f16 of the eligible 3x3 layers
        self.lib = _lib.load()
        self._plans: Dict[tuple, _Plan] = {}
        self._pack(params)

    # ------------------------------------------------------------------------------------------ weights
    def _pack(self, params):
        spec, dev = self.spec, self.device
        g = lambda k: params[k].detach().to(device=dev, dtype=torch.float32)
        w: Dict[str, torch.Tensor] = {}
        m = 'model'
        song = spec.model_type == 'SongUNet'
        # embedding MLP
        half = spec.noise_channels // 2
        freqs = torch.arange(0, half, dtype=torch.float32)
        freqs = freqs / (half - (1 if spec.pos_endpoint else 0))
        freqs = (1 / spec.pos_max_positions) ** freqs            # exactly networks_edm.py:193-195, on the host
        w['freqs'] = freqs.to(dev)
        w['map0.w'] = pack_linear_weight(g(f'{m}.map_layer0.weight')); w['map0.b'] = g(f'{m}.map_layer0.bias')
        w['map1.w'] = pack_linear_weight(g(f'{m}.map_layer1.weight')); w['map1.b'] = g(f'{m}.map_layer1.bias')
        if spec.label_dim:
            wl = g(f'{m}.map_label.weight')
            if song:
                wl = wl * math.sqrt(spec.label_dim)              # class_labels * sqrt(in_features), networks_edm.py:320
            w['label.w'] = pack_linear_weight(wl)
            w['label.b'] = g(f'{m}.map_label.bias') if f'{m}.map_label.bias' in params else None
        # per-block affine layers, concatenated into one GEMM
        aff_w, aff_b, off = [], [], 0
        self.aff_off: Dict[str, int] = {}
        for b in spec.blocks:
            if b.kind != 'block':
                continue
            p = f'{m}.{b.name}'
            aff_w.append(g(f'{p}.affine.weight')); aff_b.append(g(f'{p}.affine.bias'))
            self.aff_off[b.name] = off
            off += aff_w[-1].shape[0]
        self.aff_total = off
        w['aff.w'] = pack_linear_weight(torch.cat(aff_w, 0)); w['aff.b'] = torch.cat(aff_b, 0).contiguous()
        # blocks
        for b in spec.blocks:
            p = f'{m}.{b.name}'
            if b.kind == 'conv':
                w[f'{b.name}.w'] = pack_stem_weight(g(f'{p}.weight')); w[f'{b.name}.b'] = g(f'{p}.bias')
                continue
            for leaf in ('norm0', 'norm1'):
                w[f'{b.name}.{leaf}.g'] = g(f'{p}.{leaf}.weight'); w[f'{b.name}.{leaf}.b'] = g(f'{p}.{leaf}.bias')
            w[f'{b.name}.conv0.w'] = pack_conv_weight(g(f'{p}.conv0.weight')); w[f'{b.name}.conv0.b'] = g(f'{p}.conv0.bias')
            w[f'{b.name}.conv1.w'] = pack_conv_weight(g(f'{p}.conv1.weight')); w[f'{b.name}.conv1.b'] = g(f'{p}.conv1.bias')
            if self.conv_mode == 1 and b.cin % 64 == 0 and b.cout % 64 == 0:
                from .ops import pack_conv_weight_f16
                w[f'{b.name}.conv0.w16'] = (pack_conv_weight_f16(g(f'{p}.conv0.weight')), 0)
                w[f'{b.name}.conv1.w16'] = (pack_conv_weight_f16(g(f'{p}.conv1.weight'), g(f'{p}.skip.weight') if b.skip_conv else None), 0)
            elif self.conv_mode == 2 and b.cin % 32 == 0 and b.cout % 32 == 0:
                # ineligible layers (channel counts that are not 32-multiples: tiny / custom nets) keep the exact fp32 kernel
                from .ops import pack_conv_weight_split
                w[f'{b.name}.conv0.w16'] = pack_conv_weight_split(g(f'{p}.conv0.weight'))
                w[f'{b.name}.conv1.w16'] = pack_conv_weight_split(g(f'{p}.conv1.weight'), g(f'{p}.skip.weight') if b.skip_conv else None)
            if b.skip_conv:
                # skip projection fused into conv1: [3x3 columns | 1x1 columns] along K, biases summed
                w[f'{b.name}.conv1s.w'] = torch.cat([w[f'{b.name}.conv1.w'], pack_conv_weight(g(f'{p}.skip.weight'))], dim=1).contiguous()
                w[f'{b.name}.conv1s.b'] = (g(f'{p}.conv1.bias') + g(f'{p}.skip.bias')).contiguous()
                del w[f'{b.name}.conv1.w']
            if b.heads:
                c, h = b.cout, b.heads
                ch = c // h
                w[f'{b.name}.norm2.g'] = g(f'{p}.norm2.weight'); w[f'{b.name}.norm2.b'] = g(f'{p}.norm2.bias')
                # reference layout of the 3C output channels: index = (head*ch + c)*3 + {q,k,v}  (networks_edm.py:174)
                wq = g(f'{p}.qkv.weight').reshape(h, ch, 3, c)
                bq = g(f'{p}.qkv.bias').reshape(h, ch, 3)
                # q | k | v column blocks, each head-major (head*ch + c): the fused attention kernel reads them in place
                w[f'{b.name}.qkv.w'] = pack_linear_weight(torch.cat([wq[:, :, i].reshape(c, c) for i in range(3)], 0))
                w[f'{b.name}.qkv.b'] = torch.cat([bq[:, :, i].reshape(c) for i in range(3)], 0).contiguous()
                w[f'{b.name}.proj.w'] = pack_conv_weight(g(f'{p}.proj.weight')); w[f'{b.name}.proj.b'] = g(f'{p}.proj.bias')
        w['out.g'] = g(f'{m}.{spec.out_norm}.weight'); w['out.b'] = g(f'{m}.{spec.out_norm}.bias')
        w['outc.w'] = pack_conv_weight(g(f'{m}.{spec.out_conv}.weight')); w['outc.b'] = g(f'{m}.{spec.out_conv}.bias')
        if self.conv_mode == 2 and g(f'{m}.{spec.out_conv}.weight').shape[1] % 32 == 0:
            from .ops import pack_conv_weight_split
            w['outc.w16'] = pack_conv_weight_split(g(f'{m}.{spec.out_conv}.weight'))
        self.w = w

    # ------------------------------------------------------------------------------------------ plan
    def plan(self, B: int, emb_rows: int) -> _Plan:
        key = (B, emb_rows, self.fuse_norm16)
        if key in self._plans:
            return self._plans[key]
        spec, dev, w, lib = self.spec, self.device, self.w, self.lib
        bd = Builder(dev, conv_mode=self.conv_mode, w16_cache=self._w16_cache)      # owns the plan, its workspaces and the emitters
        P, new = bd.P, bd.new
        R = spec.img_resolution
        Bs = emb_rows
        E, NC = spec.emb_channels, spec.noise_channels
        kpad = -(-9 * spec.in_channels // 32) * 32
        song = spec.model_type == 'SongUNet'

        # ---- inputs / outputs -------------------------------------------------------------------------------
        bufs = P.bufs
        bufs['x'] = new(B, spec.in_channels, R, R)           # NCHW, un-scaled (c_in is applied by the stem)
        bufs['sigma'] = new(B)                                # per-sample sigma (row 0 only when Bs == 1 and scalar)
        bufs['sigma_rows'] = torch.zeros(1, dtype=torch.int32, device=dev)
        bufs['out'] = new(B, spec.out_channels, R, R)         # raw network output F, channel-planar (NCHW) like the user tensors
        if spec.label_dim:
            lpad = -(-spec.label_dim // 32) * 32
            bufs['labels'] = torch.zeros(Bs, lpad, dtype=torch.float32, device=dev)

        # ---- workspace sizing -------------------------------------------------------------------------------
        max_act = kpad * R * R
        max_h = 0
        max_attn = 0
        max_sc = 0
        for b in spec.blocks:
            if b.kind != 'block':
                max_h = max(max_h, b.cout * b.res_out ** 2)
                continue
            hw = b.res_out ** 2
            max_act = max(max_act, b.cin * hw, b.cout * hw)
            max_h = max(max_h, b.cout * hw, b.cin * hw)
            if b.heads:
                max_attn = max(max_attn, 2 * b.cout * hw)
                max_sc = max(max_sc, b.heads * hw * hw)
        act = new(B * max_act)
        hbuf = new(B * max_h)
        sres = new(B * max_h)            # resampled skip-path input
        sproj = new(B * max_h)           # projected skip
        ncoef = new(B * 3 * max(max(b.cin, b.cout) for b in spec.blocks))      # {mu, A, B} planes of the fused GroupNorm
        if self.conv_mode == 1:          # fp16 activated tensors: norm0 output, norm1 output, raw copy for the fused skip projection
            a16_buf, b16_buf, r16_buf, h16_buf = (bd.new16(B * max_act) for _ in range(4))
        if max_attn:
            n2 = new(B * max_attn // 2); qk = new(B * max_attn // 2 * 3); ao = new(B * max_attn // 2)
            if self.conv_mode == 1:
                n2_16, ao_16, qk_16 = bd.new16(B * max_attn // 2), bd.new16(B * max_attn // 2), bd.new16(B * max_attn // 2 * 3)
        bufs.update(act=act, hbuf=hbuf, sres=sres, sproj=sproj)
        # ---- launch emitters: plan.Builder (shared with ldm_engine); thin adapters keep this file's argument names --------------
        f16_level = bd.f16_level
        add = bd.add

        def conv(x0, c0, ld0, n, h, wd, wgt, cout, out, out_ld, taps, name, act_=DS_ACT_NONE, **kw):
            bd.conv(x0, c0, ld0, n, h, wd, wgt, cout, out, out_ld, taps, name, act=act_, **kw)

        def norm(kind, x0, c0, ld0, n, h, wd, name, act_=DS_ACT_NONE, **kw):
            bd.norm(kind, x0, c0, ld0, n, h, wd, name, act=act_, **kw)

        # ---- embedding path (networks_edm.py:314-324 / :429-439) -----------------------------------------------
        pos = new(Bs, NC); e0 = new(Bs, E); emb = new(Bs, E); aff = new(Bs, self.aff_total)
        add(lib.ds_noise_embed, (_ptr(bufs['sigma']), Bs, _ptr(w['freqs']), NC, int(spec.swap_sincos), _ptr(pos), NC), 'noise_embed')
        if song:
            src = pos
            if spec.label_dim:
                pos2 = new(Bs, NC)
                conv(bufs['labels'], bufs['labels'].shape[1], bufs['labels'].shape[1], Bs, 1, 1, w['label.w'], NC, pos2, NC, 1,
                     'map_label', bias=w['label.b'], res=pos, res_ld=NC)
                src = pos2
            conv(src, NC, NC, Bs, 1, 1, w['map0.w'], E, e0, E, 1, 'map_layer0', bias=w['map0.b'], act_=DS_ACT_SILU)
            conv(e0, E, E, Bs, 1, 1, w['map1.w'], E, emb, E, 1, 'map_layer1', bias=w['map1.b'], act_=DS_ACT_SILU)
        else:
            conv(pos, NC, NC, Bs, 1, 1, w['map0.w'], E, e0, E, 1, 'map_layer0', bias=w['map0.b'], act_=DS_ACT_SILU)
            lab = None
            if spec.label_dim:
                lab = new(Bs, E)
                conv(bufs['labels'], bufs['labels'].shape[1], bufs['labels'].shape[1], Bs, 1, 1, w['label.w'], E, lab, E, 1,
                     'map_label', bias=w['label.b'])
            conv(e0, E, E, Bs, 1, 1, w['map1.w'], E, emb, E, 1, 'map_layer1', bias=w['map1.b'], res=lab, res_ld=E,
                 act_=DS_ACT_SILU)
        conv(emb, E, E, Bs, 1, 1, w['aff.w'], self.aff_total, aff, self.aff_total, 1, 'affine_all', bias=w['aff.b'])
        bufs.update(emb=emb, aff=aff)

        # ---- fp16 residual stream -------------------------------------------------------------------------------
        # The reference's fp16 mode keeps EVERY activation of the U-Net body in fp16 (networks_edm.py:486 `x.to(dtype)`, :165-179 run in
        # that dtype): when every block of this plan runs on the fp16-activation kernels, block outputs (the residual stream, the skip
        # stack) are stored as fp16 rows too -- half the bytes of every epilogue and normalisation pass.  Arithmetic on them is fp32
        # (values are widened when loaded).  Otherwise the stream stays fp32 (the mixed layout of smaller / unusual geometries).
        def dma16_ok(b):
            if self.conv_mode != 1 or w.get(f'{b.name}.conv0.w16') is None or w.get(f'{b.name}.conv1.w16') is None:
                return False
            Ho, M_ = b.res_out, B * b.res_out ** 2
            if not (lib.ds_conv_f16dma_supported(B, Ho, Ho, b.cin, 0, b.cout)
                    and lib.ds_conv_f16dma_supported(B, Ho, Ho, b.cout, b.cin if b.skip_conv else 0, b.cout)):
                return False
            return True

        def attn16_ok(b):          # attention block on the fp16 kernels end to end (CIFAR-10's single 256-wide head is not: fp32 kernel)
            M_ = B * b.res_out ** 2
            return bool(self.conv_mode == 1 and lib.ds_attention_f16_supported(b.cout // b.heads)
                        and lib.ds_gemm_f16dma_supported(M_, b.cout, 3 * b.cout) and lib.ds_gemm_f16dma_supported(M_, b.cout, b.cout))
        stream16 = self.conv_mode == 1 and all(dma16_ok(b) for b in spec.blocks if b.kind == 'block')
        P.stream16 = stream16
        # a block whose attention runs on the fp32 kernels keeps fp32 outputs; every consumer reads a tensor in the dtype it has
        blk16 = lambda b: stream16 and (not b.heads or attn16_ok(b))
        P.f16_views = []          # (address, bytes) of fp32-typed storage that some launches use as fp16 rows (tests/test_plan_cpu.py's dtype lint)

        def as16(t):              # decoder ping-pong storage viewed as fp16 rows
            P.f16_views.append((t.data_ptr(), t.numel() * 2))
            return t.view(torch.float16)[:t.numel()]

        # ---- stem ---------------------------------------------------------------------------------------------
        x_cur = None          # (tensor, channels)
        skips: List[tuple] = []
        dec_pp = [None, None]
        dec_i = 0
        for b in spec.blocks:
            n, Hin, Ho = B, b.res_in, b.res_out
            M = B * Ho * Ho
            if b.kind == 'conv':
                add(lib.ds_stem_im2col, (_ptr(bufs['x']), _ptr(bufs['sigma']), Bs, spec.sigma_data, B, spec.in_channels, R, R,
                                         _ptr(act), kpad), 'stem_im2col')
                out = new(M, b.cout)
                conv(act, kpad, kpad, B, R, R, w[f'{b.name}.w'], b.cout, out, b.cout, 1, b.name, bias=w[f'{b.name}.b'], stats=True)
                x_cur = (out, b.cout)
                skips.append(x_cur)
                bufs[b.name] = out
                continue
            # sources of this block's input
            x0, c0 = x_cur
            x1, c1 = (None, 0)
            if b.pops_skip:
                x1, c1 = skips.pop()
                assert c1 == b.skip_cin and c0 + c1 == b.cin
            cin, cout = b.cin, b.cout
            G_in, G_out = arch.num_groups(cin), arch.num_groups(cout)
            rs = DS_RESAMPLE_DOWN if b.down else (DS_RESAMPLE_UP if b.up else DS_RESAMPLE_NONE)
            nm = b.name
            fuse = bool(lib.ds_conv3x3_halo_supported(Ho, Ho))     # GroupNorm+SiLU applied by the conv's halo loader
            w16_0, w16_1 = w.get(f'{nm}.conv0.w16'), w.get(f'{nm}.conv1.w16')
            if w16_0 is not None and f16_level(n, Ho, Ho, cin, 0, 0, 0) == 1:
                fuse = False        # fp16 operands without the fused normalisation (8x8: four images per tile): normalise in a pass
            aoff = self.aff_off[nm]
            cb = dict(cbias=aff[:, aoff:], cbias_ld=self.aff_total, cbias_rows=Bs) if not b.adaptive_scale else {}
            # fp16 mode, the reference's storage (networks_edm.py:486 runs the body on x.to(float16)): norm + SiLU (+ resample, + the
            # decoder's concatenation) are ONE pass that writes the activated tensor in fp16, and the convolution is a pure matrix
            # kernel on fp16 activations (csrc/conv3x3_f16dma.hip).  Norm arithmetic is fp32; the block output is fp16 under `stream16`.
            dma16 = (w16_0 is not None and w16_1 is not None and self.conv_mode == 1
                     and lib.ds_conv_f16dma_supported(n, Ho, Ho, cin, 0, cout)
                     and lib.ds_conv_f16dma_supported(n, Ho, Ho, cout, cin if b.skip_conv else 0, cout))
            if dma16:
                a16, b16 = a16_buf[:M * cin].view(M, cin), b16_buf[:M * cout].view(M, cout)
                h16 = h16_buf[:M * cout].view(M, cout)       # conv0 output: only read by norm1 -> stored in fp16 (networks_edm.py:486)
                # raw (un-normalised) fp16 copy of the block input: operand of the fused 1x1 skip projection, and -- fp16 stream -- the
                # resampled identity skip.  Not needed when the input already is one fp16 tensor of the right geometry.
                direct = stream16 and x1 is None and rs == DS_RESAMPLE_NONE and x0.dtype == torch.float16
                need_raw = (b.skip_conv and not direct) or (stream16 and not b.skip_conv and rs != DS_RESAMPLE_NONE)
                r16 = r16_buf[:M * cin].view(M, cin) if need_raw else None
                # Round 5, `fuse_norm16`: where every source of a convolution is a raw fp16 tensor of the layer's own geometry, the GroupNorm
                # affine + SiLU is applied by the convolution itself to its LDS halo (conv3x3_f16dma NORM: same arithmetic, same bits) and the
                # ds_norm_act pass -- with it the materialised concatenation and the raw copy for the skip projection -- disappears; the
                # statistics launch (ds_gn_finalize over the producers' column sums) stays.  Resampling blocks keep the pass for conv0.
                raw16 = lambda t: t is None or t.dtype == torch.float16
                fz_here = fuse_norm16_here(self.fuse_norm16, Ho, cout)
                fz0 = bool(fz_here and stream16 and rs == DS_RESAMPLE_NONE and raw16(x0) and raw16(x1) and c0 % 64 == 0 and c1 % 64 == 0)
                fz1 = bool(fz_here and stream16)
                norm('stats', x0, c0, c0, n, Hin, Hin, nm + '.norm0.stats', x1=x1, c1=c1, ld1=c1, groups=G_in, eps=b.eps,
                     gamma=w[f'{nm}.norm0.g'], beta=w[f'{nm}.norm0.b'], coefs=ncoef)
                if fz0:
                    r16 = None
                    conv(x0, c0, c0, n, Ho, Ho, w[f'{nm}.conv0.w'], cout, h16, cout, 9, nm + '.conv0', x1=x1, c1=c1, ld1=c1,
                         bias=w[f'{nm}.conv0.b'], stats=True, w16=w16_0, in_f16=True, out_f16=True, norm_coefs=ncoef, norm_act=DS_ACT_SILU, **cb)
                else:
                    norm('apply', x0, c0, c0, n, Hin, Hin, nm + '.norm0', x1=x1, c1=c1, ld1=c1, groups=G_in, eps=b.eps, use_stats=False,
                         act_=DS_ACT_SILU, resample=rs, out=a16, out_ld=cin, out_f16=True, raw_out=r16, raw_ld=cin, coefs=ncoef)
                    conv(a16, cin, cin, n, Ho, Ho, w[f'{nm}.conv0.w'], cout, h16, cout, 9, nm + '.conv0', bias=w[f'{nm}.conv0.b'], stats=True,
                         w16=w16_0, in_f16=True, out_f16=True, **cb)
                ss = dict(scale=aff[:, aoff:], shift=aff[:, aoff + cout:], ss_ld=self.aff_total, ss_rows=Bs) if b.adaptive_scale else {}
                norm('stats', h16, cout, cout, n, Ho, Ho, nm + '.norm1.stats', groups=G_out, eps=b.eps, gamma=w[f'{nm}.norm1.g'],
                     beta=w[f'{nm}.norm1.b'], coefs=ncoef, **ss)             # from conv0's epilogue sums (fp32), incl. the adaptive scale / shift
                if not fz1:
                    norm('apply', h16, cout, cout, n, Ho, Ho, nm + '.norm1', groups=G_out, eps=b.eps, use_stats=False, act_=DS_ACT_SILU,
                         out=b16, out_ld=cout, out_f16=True, in_f16=True, coefs=ncoef)
                if b.skip_conv:          # 1x1 skip projection fused into conv1 as extra K columns on the raw (resampled) fp16 input
                    c1_w, c1_b = w[f'{nm}.conv1s.w'], w[f'{nm}.conv1s.b']
                    if fz0 and fz1:      # both raw sources straight from the stream: nothing was copied
                        c1_skip = dict(e0=x0, ec0=c0, e1=x1, ec1=c1)
                    else:
                        c1_skip = dict(e0=x0 if direct else r16, ec0=cin)
                elif stream16 and rs != DS_RESAMPLE_NONE:
                    c1_w, c1_b = w[f'{nm}.conv1.w'], w[f'{nm}.conv1.b']
                    c1_skip = dict(res=r16, res_ld=cout)          # resampled raw input, already written by the norm0 pass
                else:
                    s0 = x0
                    if rs != DS_RESAMPLE_NONE:
                        norm('apply', x0, c0, c0, n, Hin, Hin, nm + '.skip.resample', x1=x1, c1=c1, ld1=c1, use_stats=False,
                             resample=rs, out=sres, out_ld=cin)
                        s0 = sres
                    assert x1 is None and c0 == cout
                    c1_w, c1_b = w[f'{nm}.conv1.w'], w[f'{nm}.conv1.b']
                    c1_skip = dict(res=s0, res_ld=cout)
                if fz1:
                    c1_in, c1_norm = h16, dict(in_f16=True, norm_coefs=ncoef, norm_act=DS_ACT_SILU)
                else:
                    c1_in, c1_norm = b16, dict(in_f16=True)
            # norm0 + silu (+resample) -> conv0 (+bias, + per-image embedding for the non-adaptive variant)
            elif fuse and rs == DS_RESAMPLE_NONE:
                norm('stats', x0, c0, c0, n, Hin, Hin, nm + '.norm0.stats', x1=x1, c1=c1, ld1=c1, groups=G_in, eps=b.eps,
                     gamma=w[f'{nm}.norm0.g'], beta=w[f'{nm}.norm0.b'], coefs=ncoef)
                conv(x0, c0, c0, n, Ho, Ho, w[f'{nm}.conv0.w'], cout, hbuf, cout, 9, nm + '.conv0', x1=x1, c1=c1, ld1=c1,
                     bias=w[f'{nm}.conv0.b'], norm_coefs=ncoef, norm_act=DS_ACT_SILU, stats=True, w16=w16_0, **cb)
            else:
                norm('stats', x0, c0, c0, n, Hin, Hin, nm + '.norm0.stats', x1=x1, c1=c1, ld1=c1, groups=G_in, eps=b.eps)
                norm('apply', x0, c0, c0, n, Hin, Hin, nm + '.norm0', x1=x1, c1=c1, ld1=c1, groups=G_in, eps=b.eps,
                     gamma=w[f'{nm}.norm0.g'], beta=w[f'{nm}.norm0.b'], act_=DS_ACT_SILU, resample=rs, out=act, out_ld=cin)
                conv(act, cin, cin, n, Ho, Ho, w[f'{nm}.conv0.w'], cout, hbuf, cout, 9, nm + '.conv0', bias=w[f'{nm}.conv0.b'], stats=True,
                     w16=w16_0, **cb)
            # norm1 (+adaptive scale/shift) + silu
            ss = dict(scale=aff[:, aoff:], shift=aff[:, aoff + cout:], ss_ld=self.aff_total, ss_rows=Bs) if b.adaptive_scale else {}
            if dma16:
                pass
            elif fuse:
                norm('stats', hbuf, cout, cout, n, Ho, Ho, nm + '.norm1.stats', groups=G_out, eps=b.eps, gamma=w[f'{nm}.norm1.g'],
                     beta=w[f'{nm}.norm1.b'], coefs=ncoef, **ss)
                c1_in, c1_norm = hbuf, dict(norm_coefs=ncoef, norm_act=DS_ACT_SILU)
            else:
                norm('stats', hbuf, cout, cout, n, Ho, Ho, nm + '.norm1.stats', groups=G_out, eps=b.eps)
                norm('apply', hbuf, cout, cout, n, Ho, Ho, nm + '.norm1', groups=G_out, eps=b.eps, gamma=w[f'{nm}.norm1.g'],
                     beta=w[f'{nm}.norm1.b'], act_=DS_ACT_SILU, out=act, out_ld=cout, **ss)
                c1_in, c1_norm = act, {}
            # skip path: raw (resampled) input, projected by a 1x1 that is fused into conv1 as extra K columns
            s0, sc0, s1, sc1 = x0, c0, x1, c1
            if dma16:
                pass
            elif rs != DS_RESAMPLE_NONE:
                norm('apply', x0, c0, c0, n, Hin, Hin, nm + '.skip.resample', x1=x1, c1=c1, ld1=c1, use_stats=False,
                     resample=rs, out=sres, out_ld=cin)
                s0, sc0, s1, sc1 = sres, cin, None, 0
            if dma16:
                pass
            elif b.skip_conv:
                c1_w, c1_b = w[f'{nm}.conv1s.w'], w[f'{nm}.conv1s.b']
                c1_skip = dict(e0=s0, ec0=sc0, e1=s1, ec1=sc1)
            else:
                assert s1 is None and sc0 == cout
                c1_w, c1_b = w[f'{nm}.conv1.w'], w[f'{nm}.conv1.b']
                c1_skip = dict(res=s0, res_ld=cout)
            # output buffer: encoder outputs are kept for the skip stack, decoder outputs ping-pong
            f16_out = blk16(b)
            if b.pushes_skip:
                out = bd.new16(M, cout) if f16_out else new(M, cout)
            else:
                if dec_pp[dec_i] is None or dec_pp[dec_i].numel() < M * cout:
                    dec_pp[dec_i] = new(B * max_h)
                out = as16(dec_pp[dec_i]) if f16_out else dec_pp[dec_i]
                dec_i ^= 1
            if b.heads:
                mid = out
                out2 = None
            conv(c1_in, cout, cout, n, Ho, Ho, c1_w, cout, out, cout, 9, nm + '.conv1', bias=c1_b, scale=b.skip_scale, stats=True,
                 w16=w16_1, **c1_norm, **c1_skip)
            if b.heads:
                S = Ho * Ho
                hd = b.heads
                ch = cout // hd
                f16_attn = self.conv_mode == 1 and lib.ds_attention_f16_supported(ch)     # networks_edm.py:98-110 with fp16 q / k / v
                # fp16 mode: the GroupNorm output and the attention output are only operands of the qkv / proj 1x1s -> fp16 rows,
                # streamed by the fp16-activation GEMM (csrc/gemm_f16dma.hip)
                a16 = bool(f16_attn and lib.ds_gemm_f16dma_supported(M, cout, 3 * cout) and lib.ds_gemm_f16dma_supported(M, cout, cout))
                n2_, ao_ = (n2_16[:M * cout].view(M, cout), ao_16[:M * cout].view(M, cout)) if a16 else (n2, ao)
                norm('stats', out, cout, cout, n, Ho, Ho, nm + '.norm2.stats', groups=G_out, eps=b.eps)
                norm('apply', out, cout, cout, n, Ho, Ho, nm + '.norm2', groups=G_out, eps=b.eps, gamma=w[f'{nm}.norm2.g'],
                     beta=w[f'{nm}.norm2.b'], out=n2_, out_ld=cout, out_f16=a16)
                qk_ = qk_16 if a16 else qk          # fp16 mode: q | k | v are the fp16 rows the reference's qkv projection emits (networks_edm.py:171-173)
                conv(n2_, cout, cout, n, Ho, Ho, w[f'{nm}.qkv.w'], 3 * cout, qk_, 3 * cout, 1, nm + '.qkv', bias=w[f'{nm}.qkv.b'])
                # softmax(Q K^T / sqrt(ch)) V per (image, head), scores kept on chip (networks_edm.py:171-176)
                at = AttnArgs(_ptr(qk_), _ptr(qk_[cout:]), _ptr(qk_[2 * cout:]), _ptr(ao_), 3 * cout, 3 * cout, 3 * cout, cout,
                              S * 3 * cout, S * 3 * cout, S * 3 * cout, S * cout, B, hd, S, S, ch, 1.0 / math.sqrt(ch))
                at.out_f16 = 1 if a16 else 0
                at.in_f16 = 3 if a16 else 0
                add(lib.ds_attention_f16 if f16_attn else lib.ds_attention, (C.byref(at),), nm + '.attention', keep=(at,))
                if b.pushes_skip:
                    out2 = bd.new16(M, cout) if f16_out else new(M, cout)
                else:
                    if dec_pp[dec_i] is None:
                        dec_pp[dec_i] = new(B * max_h)
                    out2 = as16(dec_pp[dec_i]) if f16_out else dec_pp[dec_i]
                    dec_i ^= 1
                conv(ao_, cout, cout, n, Ho, Ho, w[f'{nm}.proj.w'], cout, out2, cout, 1, nm + '.proj', bias=w[f'{nm}.proj.b'],
                     res=out, res_ld=cout, scale=b.skip_scale, stats=True)
                out = out2
            x_cur = (out, cout)
            bufs[nm] = out
            if b.pushes_skip:
                skips.append(x_cur)
        assert not skips
        # ---- output head ------------------------------------------------------------------------------------------
        xo, co = x_cur
        if lib.ds_conv3x3_halo_supported(R, R) and xo.dtype == torch.float32:
            norm('stats', xo, co, co, B, R, R, 'out.norm.stats', groups=arch.num_groups(co), eps=spec.out_eps, gamma=w['out.g'],
                 beta=w['out.b'], coefs=ncoef)
            conv(xo, co, co, B, R, R, w['outc.w'], spec.out_channels, bufs['out'], 4, 9, 'out.conv', bias=w['outc.b'],
                 norm_coefs=ncoef, norm_act=DS_ACT_SILU, out_nchw=1, w16=w.get('outc.w16'))
        else:
            norm('stats', xo, co, co, B, R, R, 'out.norm.stats', groups=arch.num_groups(co), eps=spec.out_eps)
            norm('apply', xo, co, co, B, R, R, 'out.norm', groups=arch.num_groups(co), eps=spec.out_eps, gamma=w['out.g'],
                 beta=w['out.b'], act_=DS_ACT_SILU, out=act, out_ld=co)
            conv(act, co, co, B, R, R, w['outc.w'], spec.out_channels, bufs['out'], 4, 9, 'out.conv', bias=w['outc.b'], out_nchw=1)
        # ---- the solver update fused into the head (round 5, ds_conv_args.update): the head launch carries a pointer to ONE persistent
        # ds_update_args of this plan; EDMDenoiser.raw(update=...) fills it before a run and clears its outputs afterwards (x_out == m_out ==
        # NULL = plain head).  Possible where the head runs on conv3x3_thin_kernel with a channel-planar output (ds_conv_kernel_id 2570).
        last = P.ops[-1]
        head = last.keep[0] if (last.fn is lib.ds_conv2d_nhwc and last.keep) else None         # the last launch IS the head convolution
        P.head_update = _lib.UpdateArgs()
        P.head_fusable = bool(head is not None and head.out_nchw and not head.wgt_f16 and lib.ds_conv_kernel_id(C.byref(head)) == 2570)
        if P.head_fusable:
            head.update = C.cast(C.pointer(P.head_update), C.c_void_p)
        from .plan import release_tuning_scratch
        release_tuning_scratch()            # the tile measurement's 512 MiB flush buffer does not outlive the plan build
        self._plans[key] = P
        return P


    def flops(self, B):
        return arch.flops_per_image(self.spec) * B


class EDMDenoiser:
    """Drop-in for the reference ``EDMPrecond`` object (networks_edm.py:460-499) backed by the HIP engine.

    ``net(x, sigma, class_labels=None)`` -> denoised NCHW fp32, like the reference.  The fused solvers bypass
    the final ``c_skip x + c_out F`` pass and read the raw output through ``raw()`` instead.
    """
    edm_raw_output = True      # solvers._Run: ds_solver_update applies the EDM preconditioning to the raw output itself

    def __init__(self, spec: arch.UNetSpec, params: Dict[str, torch.Tensor], device='cuda', use_fp16=False, split_fp16=False):
        self.spec = spec
        self.engine = UNetEngine(spec, params, device, use_fp16=use_fp16, split_fp16=split_fp16)
        self.device = self.engine.device
        self.img_resolution = spec.img_resolution
        self.img_channels = spec.in_channels
        self.label_dim = spec.label_dim
        self.sigma_min = spec.sigma_min
        self.sigma_max = spec.sigma_max
        self.sigma_data = spec.sigma_data
        self.use_fp16 = bool(use_fp16)   # read-only after construction (weights are packed for the mode), like net.use_fp16 of the reference
        self._last = None                # (plan, batch) of the last __call__ evaluation (block_output)
        self.bottleneck_name = None      # set by the AMED path: 'enc.8x8_block3' / 'enc.8x8_block2'

    @classmethod
    def from_config(cls, name_or_kwargs, seed=0, mode='signal', device='cuda', use_fp16=False, split_fp16=False):
        kw = arch.NAMED_CONFIGS[name_or_kwargs] if isinstance(name_or_kwargs, str) else name_or_kwargs
        spec = arch.edm_precond_spec(**kw)
        return cls(spec, arch.init_params(spec, seed=seed, mode=mode), device, use_fp16=use_fp16, split_fp16=split_fp16)

    @classmethod
    def from_reference_module(cls, net, device='cuda', use_fp16=None):
        """Build from a live reference ``EDMPrecond`` instance (duck-typed: pickled EDM classes are exec'd from
        source, so ``isinstance`` is useless -- persistence.py:222-233).  ``use_fp16`` None: follow the module's own flag
        (networks_edm.py:472, :486 -- the public ImageNet-64 checkpoint carries use_fp16=True)."""
        spec = spec_from_module(net)
        if use_fp16 is None:
            use_fp16 = bool(getattr(net, 'use_fp16', False))
        return cls(spec, {k: v for k, v in net.state_dict().items() if 'resample_filter' not in k}, device, use_fp16=use_fp16)

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    # -- raw evaluation: fills plan inputs, runs the plan, returns (F_nhwc4, plan) -----------------------------------
    def _prepare(self, x, sigma, class_labels):
        B = x.shape[0]
        lib = self.engine.lib
        st = _lib.stream_ptr()
        host_scalar = isinstance(sigma, (int, float))
        if host_scalar:
            emb_rows = B if self.label_dim else 1
        else:
            sigma = torch.as_tensor(sigma, dtype=torch.float32, device=self.device).reshape(-1).contiguous()
            emb_rows = B if (sigma.numel() > 1 or self.label_dim) else 1
        plan = self.engine.plan(B, emb_rows)
        xb = plan.bufs['x']
        if x.data_ptr() != xb.data_ptr():
            assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
            _lib.check(lib.ds_copy_rows(_ptr(x), x[0].numel(), _ptr(xb), x[0].numel(), B, x[0].numel(), st), 'copy x')
        sb = plan.bufs['sigma']
        if host_scalar:      # no H2D copy: the sigma rows are written by a kernel
            _lib.check(lib.ds_fill(_ptr(sb), float(sigma), emb_rows, st), 'fill sigma')
        elif sigma.data_ptr() != sb.data_ptr():
            if sigma.numel() == 1 and emb_rows > 1:
                sigma = sigma.expand(emb_rows).contiguous()
            _lib.check(lib.ds_copy_rows(_ptr(sigma), 1, _ptr(sb), 1, emb_rows, 1, st), 'copy sigma')
        if self.label_dim:
            lb = plan.bufs['labels']
            if class_labels is None:
                lb.zero_()
            else:
                cl = class_labels.to(torch.float32).reshape(-1, self.label_dim).contiguous()
                if cl.shape[0] == 1 and B > 1:
                    cl = cl.expand(B, -1).contiguous()
                _lib.check(lib.ds_copy_rows(_ptr(cl), self.label_dim, _ptr(lb), lb.shape[1], B, self.label_dim, st), 'copy labels')
        return plan, emb_rows

    def raw(self, x, sigma, class_labels=None, update=None):
        """F(c_in x; c_noise), the raw network output, as an NCHW [B, C, H, W] tensor.  Engine-owned, overwritten by the
        next evaluation at the same batch size.
        update: a filled ``_lib.UpdateArgs`` (raw = 1, f ignored): the network head applies that solver update in its epilogue -- no update
        launch (``head_update_ok(B, ...)`` says whether this plan's head can; csrc/conv3x3_thin.hip).
        A plan is SINGLE-STREAM state: its workspaces, its input buffers and the one ``head_update`` struct the head launch points at are
        written here and read by the launches that follow, so two threads / streams must not evaluate the same (denoiser, batch) plan
        concurrently -- build one denoiser per stream (the reference's module has the same contract: one process, one stream per rank)."""
        plan, _ = self._prepare(x, sigma, class_labels)
        if update is None:
            plan.run(_lib.stream_ptr())
            return plan.bufs['out'], plan
        if not plan.head_fusable:
            raise _lib.DsError('this plan\'s head does not run on the kernel that can fuse the solver update')
        hu = plan.head_update
        C.memmove(C.byref(hu), C.byref(update), C.sizeof(hu))
        try:
            plan.run(_lib.stream_ptr())
        finally:
            hu.x_out, hu.m_out = None, None                      # the next plain evaluation must not update anything
        return plan.bufs['out'], plan

    def head_update_ok(self, B, sigma, class_labels=None):
        """True when an evaluation at this batch / sigma form can carry a fused solver update (solvers._Run asks before it defers)."""
        host_scalar = isinstance(sigma, (int, float))
        emb_rows = (B if self.label_dim else 1) if host_scalar else (B if (torch.as_tensor(sigma).numel() > 1 or self.label_dim) else 1)
        return bool(self.engine.plan(B, emb_rows).head_fusable)

    def __call__(self, x, sigma, class_labels=None, force_fp32=False, **kwargs):
        from . import ops
        B, Cc, H, W = x.shape
        x = x.to(torch.float32).contiguous()
        plan, emb_rows = self._prepare(x, sigma, class_labels)
        plan.run(_lib.stream_ptr())
        self._last = (plan, B)
        out = torch.empty_like(x)
        # D = c_skip x + c_out F, evaluated by the update kernel with cx = 0, cm = 1, store_d = 0 (m = D)
        args = ops.make_update_args(plan.bufs['x'], plan.bufs['x'], plan.bufs['out'], B, Cc, H, W, None, raw=True, f_ld=0,
                                    coefs=self._sigma_coefs(plan, emb_rows), coef_rows=(B if emb_rows > 1 else 1),
                                    sigma_data=self.sigma_data, m_out=out, store_d=False)
        ops.solver_update(args)
        return out

    def _sigma_coefs(self, plan, emb_rows):
        """[rows][8] coefficient rows carrying sigma in slot 6 (and t = 1 in slot 5)."""
        rows = emb_rows
        key = ('sigcoef', rows)
        if key not in plan.bufs:
            plan.bufs[key] = torch.zeros(rows, 8, dtype=torch.float32, device=self.device)
            plan.bufs[key][:, 5] = 1.0
        cf = plan.bufs[key]
        lib = self.engine.lib
        _lib.check(lib.ds_copy_rows(_ptr(plan.bufs['sigma']), 1, _ptr(cf[:, 6:]), 8, rows, 1, _lib.stream_ptr()), 'sigma->coefs')
        return cf

    def block_output(self, name):
        """Output of block ``name`` ('enc.8x8_block3', 'enc.16x16_block0', ...) of the LAST ``net(x, sigma, ...)`` evaluation, as the
        NCHW tensor ``[B, C, h, w]`` a forward hook on that block of the reference module would see (solvers_amed.py:7-18 taps
        ``net.model.enc['8x8_block3']``).  A copy: the plan's own buffer is overwritten by the next evaluation."""
        plan, B = self._last
        t = plan.bufs[name].float()          # fp16 residual stream (fp16 mode): widened for the caller
        hw = t.shape[0] // B
        h = int(round(hw ** 0.5))
        assert h * h == hw, (name, tuple(t.shape), B)
        return t.reshape(B, h, h, t.shape[1]).permute(0, 3, 1, 2).contiguous()

    def bottleneck_mean(self, plan, B, class_cond):
        """Channel mean of the AMED bottleneck tap, [B, 8, 8] (solvers_amed.py:16-17, :24-28)."""
        from . import ops
        name = 'enc.8x8_block2' if class_cond else 'enc.8x8_block3'
        t = plan.bufs[name].float()
        c = t.shape[1]
        out = torch.empty(B, 8, 8, dtype=torch.float32, device=self.device)
        ops.channel_mean(t, c, c, B * 64, out)
        return out


def spec_from_module(net) -> arch.UNetSpec:
    """Recover the UNetSpec of a live reference EDMPrecond by attribute inspection."""
    model = net.model
    names = list(model.enc.keys())
    song = any('aux' in k for k in model.dec.keys())
    first = model.enc[names[0]]
    res0 = int(net.img_resolution)
    mc_emb = model.map_layer0.weight.shape[0]
    blocks = [k for k in names if 'block' in k]
    levels = sorted({int(k.split('x')[0]) for k in names}, reverse=True)
    num_blocks = sum(1 for k in blocks if k.startswith(f'{res0}x{res0}_block'))
    if song:
        model_channels = first.out_channels
        mult = [model.enc[f'{r}x{r}_block0'].out_channels // model_channels for r in levels]
        attn = [r for r in levels if getattr(model.enc[f'{r}x{r}_block0'], 'num_heads', 0)]
        spec = arch.song_unet_spec(res0, net.img_channels, net.img_channels, label_dim=net.label_dim,
                                   augment_dim=(model.map_augment.weight.shape[1] if getattr(model, 'map_augment', None) is not None else 0),
                                   model_channels=model_channels, channel_mult=mult, channel_mult_emb=mc_emb // model_channels,
                                   num_blocks=num_blocks, attn_resolutions=attn,
                                   channel_mult_noise=model.map_layer0.weight.shape[1] // model_channels)
    else:
        model_channels = model.map_layer0.weight.shape[1]
        mult = [model.enc[f'{r}x{r}_block0'].out_channels // model_channels for r in levels]
        attn = [r for r in levels if getattr(model.enc[f'{r}x{r}_block0'], 'num_heads', 0)]
        spec = arch.dhariwal_unet_spec(res0, net.img_channels, net.img_channels, label_dim=net.label_dim,
                                       augment_dim=(model.map_augment.weight.shape[1] if getattr(model, 'map_augment', None) is not None else 0),
                                       model_channels=model_channels, channel_mult=mult, channel_mult_emb=mc_emb // model_channels,
                                       num_blocks=num_blocks, attn_resolutions=attn)
    spec.sigma_data = float(getattr(net, 'sigma_data', 0.5))
    spec.sigma_min = float(getattr(net, 'sigma_min', 0.002))
    spec.sigma_max = float(getattr(net, 'sigma_max', 80.0))
    return spec
