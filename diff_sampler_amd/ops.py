"""Tensor-level wrappers around the C ABI (one Python function per libdsamd entry point).

Each wrapper only marshals: it checks dtype/device/contiguity, fills the ctypes struct with raw device pointers and
launches on the current HIP stream.  No arithmetic happens in Python and nothing falls back to ATen.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import ConvArgs, GemmArgs, NormArgs, UpdateArgs, DS_ACT_NONE, DS_RESAMPLE_NONE


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return None
    assert t.is_cuda and t.dtype in (torch.float32, torch.float16, torch.uint8, torch.int32, torch.int64), (t.device, t.dtype)
    return C.c_void_p(t.data_ptr())


def pack_conv_weight(w: torch.Tensor, row_pad: int = 128, k_pad: int = 32) -> torch.Tensor:
    """[Cout, Cin, kh, kw] -> [Cout_pad, K] in the kernel's K order, rows zero-padded to a multiple of the N tile.

    3x3: K = (chunk*9 + tap)*32 + cc with c = chunk*32 + cc, tap = ky*3 + kx  (chunk-major, tap-minor; Cin % 32 == 0).
    1x1 / linear: K = c, zero-padded to a multiple of 32."""
    cout, cin, kh, kw = w.shape
    taps = kh * kw
    if taps == 1:
        m = w.reshape(cout, cin)
    else:
        assert cin % k_pad == 0, 'use pack_stem_weight for the 3-channel stem'
        m = w.permute(0, 2, 3, 1).reshape(cout, taps, cin // k_pad, k_pad).permute(0, 2, 1, 3).reshape(cout, taps * cin)
    rows = -(-cout // row_pad) * row_pad
    cols = -(-m.shape[1] // k_pad) * k_pad
    out = torch.zeros(rows, cols, dtype=torch.float32, device=w.device)
    out[:cout, :m.shape[1]] = m
    return out.contiguous()


def pack_conv_weight_f16(w: torch.Tensor, extra: Optional[torch.Tensor] = None, row_pad: int = 128) -> torch.Tensor:
    """fp16 weights of the reduced-precision 3x3 convolution (ds_conv_args.wgt_f16): [Cout, Cin, 3, 3] (+ optional 1x1 skip
    projection [Cout, Ce, 1, 1] appended along K) -> [Cout_pad, K] halfs, K = (slab*9 + tap)*64 + cc with c = slab*64 + cc,
    then the extra columns in 64-channel blocks; rows zero-padded to the N tile.  Rounded to nearest even like ``w.to(float16)``
    (networks_edm.py:79: ``w.to(x.dtype)``).  Returned as a float32-typed view of the same bytes (the ABI carries ``const float*``)."""
    cout, cin, kh, kw = w.shape
    assert kh == 3 and kw == 3 and cin % 64 == 0
    m = w.permute(0, 2, 3, 1).reshape(cout, 9, cin // 64, 64).permute(0, 2, 1, 3).reshape(cout, 9 * cin)
    if extra is not None:
        assert extra.shape[0] == cout and extra.shape[1] % 64 == 0
        m = torch.cat([m, extra.reshape(cout, -1)], dim=1)
    rows = -(-cout // row_pad) * row_pad
    out = torch.zeros(rows, m.shape[1], dtype=torch.float16, device=w.device)
    out[:cout] = m.to(torch.float16)
    return out.contiguous().view(torch.float32)


def pack_linear_weight_f16(w_padded: torch.Tensor) -> torch.Tensor:
    """fp16 weights of the reduced-precision 1x1 convolution / Linear (ds_conv_args.wgt_f16 == 1 with taps == 1): the row-padded
    fp32 matrix [Cout_pad, K] of pack_linear_weight / pack_conv_weight rounded like ``w.to(float16)`` (networks_edm.py:79; torch
    autocast casts nn.Linear weights the same way), K order unchanged.  Float32-typed view of the bytes."""
    assert w_padded.dim() == 2 and w_padded.shape[0] % 128 == 0 and w_padded.shape[1] % 64 == 0
    return w_padded.to(torch.float16).contiguous().view(torch.float32)


def pack_conv_weight_split(w: torch.Tensor, extra: Optional[torch.Tensor] = None, row_pad: int = 128):
    """Split-fp16 weights of the fp32-emulated 3x3 convolution (ds_conv_args.wgt_f16 == 2) -> (packed tensor, shift).

    The weights are multiplied by 2**shift (exact; chosen so that the largest magnitude lands in [2**13, 2**14): hi and lo then
    sit in fp16's normal range for every weight down to 2**-14 of the largest), split as hi = fp16(w'), lo = fp16(w' - hi), and
    laid out per 32-channel slab and tap as 32 hi halfs followed by 32 lo halfs: K = (slab*9 + tap)*64 + {cc | 32 + cc}; the 1x1
    extra columns follow in 32-channel blocks of the same [hi | lo] form.  Float32-typed view of the bytes (the ABI carries float*)."""
    cout, cin, kh, kw = w.shape
    assert kh == 3 and kw == 3 and cin % 32 == 0
    wmax = float(w.abs().max())
    if extra is not None:
        wmax = max(wmax, float(extra.abs().max()))
    import math
    shift = 0 if wmax == 0 else max(0, min(24, 13 - math.floor(math.log2(wmax))))
    sc = float(1 << shift)

    def split(m):                      # [cout, nblk, 32] -> [cout, nblk, 64] halfs
        m = m * sc
        hi = m.to(torch.float16)
        lo = (m - hi.to(torch.float32)).to(torch.float16)
        return torch.cat([hi, lo], dim=-1)
    m = w.permute(0, 2, 3, 1).reshape(cout, 9, cin // 32, 32).permute(0, 2, 1, 3).reshape(cout, 9 * (cin // 32), 32)
    parts = [split(m).reshape(cout, -1)]
    if extra is not None:
        assert extra.shape[0] == cout and extra.shape[1] % 32 == 0
        parts.append(split(extra.reshape(cout, -1, 32)).reshape(cout, -1))
    m = torch.cat(parts, dim=1)
    rows = -(-cout // row_pad) * row_pad
    out = torch.zeros(rows, m.shape[1], dtype=torch.float16, device=w.device)
    out[:cout] = m
    return out.contiguous().view(torch.float32), shift


def pack_stem_weight(w: torch.Tensor, row_pad: int = 128, k_pad: int = 32) -> torch.Tensor:
    """Stem conv [Cout, C, 3, 3] for the im2col'd input of ds_stem_im2col: K = tap*C + c, zero-padded to 32."""
    cout, cin, kh, kw = w.shape
    m = w.permute(0, 2, 3, 1).reshape(cout, kh * kw * cin)
    rows = -(-cout // row_pad) * row_pad
    cols = -(-m.shape[1] // k_pad) * k_pad
    out = torch.zeros(rows, cols, dtype=torch.float32, device=w.device)
    out[:cout, :m.shape[1]] = m
    return out.contiguous()


def pack_linear_weight(w: torch.Tensor, row_pad: int = 128, k_pad: int = 32) -> torch.Tensor:
    return pack_conv_weight(w[:, :, None, None], row_pad, k_pad)


def conv2d_nhwc(x0, c0, ld0, n, h, w, wgt, cout, out, out_ld, *, taps=9, x1=None, c1=0, ld1=0, bias=None, cbias=None,
                cbias_ld=0, cbias_rows=1, res=None, res_ld=0, out_scale=1.0, act=DS_ACT_NONE, stride=1):
    a = ConvArgs(_p(x0), _p(x1), c0, c1, ld0, ld1, n, h, w, taps, _p(wgt), cout, _p(bias), _p(cbias), cbias_ld,
                 cbias_rows, _p(res), res_ld, out_scale, act, _p(out), out_ld)
    a.stride = stride
    _lib.check(_lib.load().ds_conv2d_nhwc(C.byref(a), _lib.stream_ptr()), 'ds_conv2d_nhwc')


def gemm_nt_batched(a, lda, b, ldb, c, ldc, m, n, k, *, batch=1, heads=1, a_bs=0, a_hs=0, b_bs=0, b_hs=0, c_bs=0,
                    c_hs=0, alpha=1.0, rowbias=None, colbias=None, act=DS_ACT_NONE):
    g = GemmArgs(_p(a), lda, a_bs, a_hs, _p(b), ldb, b_bs, b_hs, _p(c), ldc, c_bs, c_hs, m, n, k, batch, heads,
                 alpha, _p(rowbias), _p(colbias), act)
    _lib.check(_lib.load().ds_gemm_nt_batched(C.byref(g), _lib.stream_ptr()), 'ds_gemm_nt_batched')


def _norm_args(x0, c0, ld0, n, h, w, *, x1=None, c1=0, ld1=0, groups=1, eps=1e-5, mean=None, rstd=None, gamma=None,
               beta=None, scale=None, shift=None, ss_ld=0, ss_rows=1, act=DS_ACT_NONE, resample=DS_RESAMPLE_NONE,
               out=None, out_ld=0):
    return NormArgs(_p(x0), _p(x1), c0, c1, ld0, ld1, n, h, w, groups, eps, _p(mean), _p(rstd), _p(gamma), _p(beta),
                    _p(scale), _p(shift), ss_ld, ss_rows, act, resample, _p(out), out_ld)


def gn_stats(x0, c0, ld0, n, h, w, groups, eps, mean, rstd, *, x1=None, c1=0, ld1=0, partial=None, counters=None):
    a = _norm_args(x0, c0, ld0, n, h, w, x1=x1, c1=c1, ld1=ld1, groups=groups, eps=eps, mean=mean, rstd=rstd)
    if partial is not None:     # multi-workgroup statistics (small batches): fp64 scratch + zeroed int32 counters
        a.partial, a.counters = C.c_void_p(partial.data_ptr()), C.c_void_p(counters.data_ptr())
    _lib.check(_lib.load().ds_gn_stats(C.byref(a), _lib.stream_ptr()), 'ds_gn_stats')


def norm_act(x0, c0, ld0, n, h, w, out, out_ld, **kw):
    a = _norm_args(x0, c0, ld0, n, h, w, out=out, out_ld=out_ld, **kw)
    _lib.check(_lib.load().ds_norm_act(C.byref(a), _lib.stream_ptr()), 'ds_norm_act')


def softmax_rows(x, y, rows, cols, ld):
    _lib.check(_lib.load().ds_softmax_rows(_p(x), _p(y), rows, cols, ld, _lib.stream_ptr()), 'ds_softmax_rows')


def noise_embed(sigma, bs, freqs, nch, swap, out, out_ld):
    _lib.check(_lib.load().ds_noise_embed(_p(sigma), bs, _p(freqs), nch, int(swap), _p(out), out_ld, _lib.stream_ptr()),
               'ds_noise_embed')


def stem_im2col(x, sigma, sigma_rows, sigma_data, n, c, h, w, out, kpad):
    _lib.check(_lib.load().ds_stem_im2col(_p(x), _p(sigma), sigma_rows, sigma_data, n, c, h, w, _p(out), kpad,
                                          _lib.stream_ptr()), 'ds_stem_im2col')


def make_update_args(xe, xb, f, n, c, h, w, x_out, *, raw=False, f_ld=0, hist: Sequence = (), coefs=None, coef_rows=1,
                     hcoefs=None, afs=False, sigma_data=0.5, m_out=None, store_d=True) -> UpdateArgs:
    a = UpdateArgs()
    a.xe, a.xb, a.f = _p(xe), _p(xb), _p(f)
    a.raw, a.f_ld = int(raw), f_ld
    for i in range(3):
        a.hist[i] = _p(hist[i]) if i < len(hist) and hist[i] is not None else None
    a.coefs, a.coef_rows = _p(coefs), coef_rows
    if hcoefs is not None:
        for i, v in enumerate(hcoefs):
            a.hcoefs[i] = float(v)
    a.afs, a.sigma_data = int(afs), sigma_data
    a.m_out, a.store_d, a.x_out = _p(m_out), int(store_d), _p(x_out)
    a.n, a.c, a.h, a.w = n, c, h, w
    return a


def solver_update(args: UpdateArgs):
    _lib.check(_lib.load().ds_solver_update(C.byref(args), _lib.stream_ptr()), 'ds_solver_update')


def dpmpp_x0_step(args: UpdateArgs, p=0.995):
    _lib.check(_lib.load().ds_dpmpp_x0_step(C.byref(args), float(p), _lib.stream_ptr()), 'ds_dpmpp_x0_step')


def table_select(table, row_floats, step, advance, dst):
    _lib.check(_lib.load().ds_table_select(_p(table), row_floats, _p(step), int(advance), _p(dst), _lib.stream_ptr()),
               'ds_table_select')


def dynamic_threshold(x0, out, n, per, p=0.995):
    _lib.check(_lib.load().ds_dynamic_threshold(_p(x0), _p(out), n, per, p, _lib.stream_ptr()), 'ds_dynamic_threshold')


def scale(x, a, y):
    _lib.check(_lib.load().ds_scale(_p(x), float(a), _p(y), x.numel(), _lib.stream_ptr()), 'ds_scale')


def quantize_u8_nhwc(x, out, n, c, h, w):
    _lib.check(_lib.load().ds_quantize_u8_nhwc(_p(x), _p(out), n, c, h, w, _lib.stream_ptr()), 'ds_quantize_u8_nhwc')


def channel_mean(x, ld, c, rows, out):
    _lib.check(_lib.load().ds_channel_mean(_p(x), ld, c, rows, _p(out), _lib.stream_ptr()), 'ds_channel_mean')


def copy_rows(src, src_ld, dst, dst_ld, rows, cols):
    _lib.check(_lib.load().ds_copy_rows(_p(src), src_ld, _p(dst), dst_ld, rows, cols, _lib.stream_ptr()), 'ds_copy_rows')


def fill(dst, value, count=None):
    _lib.check(_lib.load().ds_fill(_p(dst), float(value), dst.numel() if count is None else count, _lib.stream_ptr()), 'ds_fill')


def attention(q, k, v, out, *, batch, heads, sq, skv, d, ldq, ldk, ldv, ldo, q_bs, k_bs, v_bs, o_bs, scale, f16=False, variant=None):
    """f16=True: fp16 operands (ds_attention_f16, the reference's fp16 / autocast attention); fails where the head size is not covered.
    variant: ds_attn_args.variant (None = the library's choice)."""
    a = _lib.AttnArgs(_p(q), _p(k), _p(v), _p(out), ldq, ldk, ldv, ldo, q_bs, k_bs, v_bs, o_bs, batch, heads, sq, skv, d, scale)
    if variant is not None:
        a.variant = int(variant)
    lib = _lib.load()
    if f16:
        _lib.check(lib.ds_attention_f16(C.byref(a), _lib.stream_ptr()), 'ds_attention_f16')
    else:
        _lib.check(lib.ds_attention(C.byref(a), _lib.stream_ptr()), 'ds_attention')


def layernorm_rows(x, ldx, gamma, beta, eps, y, ldy, rows, cols):
    _lib.check(_lib.load().ds_layernorm_rows(_p(x), ldx, _p(gamma), _p(beta), eps, _p(y), ldy, rows, cols, _lib.stream_ptr()),
               'ds_layernorm_rows')


def geglu(x, ldx, y, ldy, rows, inner):
    _lib.check(_lib.load().ds_geglu(_p(x), ldx, _p(y), ldy, rows, inner, _lib.stream_ptr()), 'ds_geglu')


def cfg_denoise(x, f, f_ld, sigma, sigma_rows, guidance, doubled, n, c, h, w, out):
    _lib.check(_lib.load().ds_cfg_denoise(_p(x), _p(f), f_ld, _p(sigma), sigma_rows, float(guidance), int(doubled), n, c, h, w,
                                          _p(out), _lib.stream_ptr()), 'ds_cfg_denoise')


def philox_threads_total(n: int, device) -> int:
    """ATen's execution policy for an n-element distribution kernel on this device (DistributionTemplates.h: calc_execution_policy)."""
    prop = torch.cuda.get_device_properties(device)
    blocks = min(prop.multi_processor_count * (prop.max_threads_per_multi_processor // 256), -(-n // 256))
    return 256 * blocks


def philox_offset_step(n: int, threads_total: int) -> int:
    return ((n - 1) // (threads_total * 4) + 1) * 4


def philox_randn(seeds, offset, out, n):
    """seeds: int64 tensor [B] on the device (values < 2**32 as sample.py seeds them); out: float32 [B, n]."""
    tt = philox_threads_total(n, out.device)
    _lib.check(_lib.load().ds_philox_randn(_p(seeds), int(offset), _p(out), seeds.numel(), n, tt, _lib.stream_ptr()), 'ds_philox_randn')
    return philox_offset_step(n, tt)


def philox_randint(seeds, offset, high, out):
    _lib.check(_lib.load().ds_philox_randint(_p(seeds), int(offset), int(high), _p(out), seeds.numel(), _lib.stream_ptr()), 'ds_philox_randint')
    return 4
