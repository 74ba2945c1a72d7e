"""GPU unit parity of the individual C-ABI kernels against plain fp32 ATen on the CPU (the third-party arithmetic the
reference bottoms out in).  Both convolution kernels (LDS-halo and generic gather) are exercised on the same cases,
including the fused input GroupNorm+SiLU, the dual-source (concat-free) K loop and the appended 1x1 skip columns."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

TOL = 2e-5


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-6))


def _nhwc(x):      # [B,C,H,W] -> [B*H*W, C]
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous()


CONV_CASES = [
    # B, H, c0, c1, cout, taps, extras(ec0, ec1), norm
    (3, 16, 64, 0, 64, 9, (0, 0), False),
    (3, 16, 64, 32, 96, 9, (0, 0), True),
    (2, 32, 32, 32, 128, 9, (64, 32), True),
    (5, 8, 64, 64, 160, 9, (64, 0), True),       # 8x8: tiles span two images (per-slot coefficient path)
    (1, 8, 32, 0, 32, 9, (0, 0), True),          # M = 64 < tile
    (2, 64, 32, 0, 64, 9, (32, 0), False),       # W = 64
    (3, 16, 64, 32, 96, 1, (0, 0), False),
    (7, 1, 128, 0, 40, 1, (0, 0), False),        # Linear: h = w = 1
    (2, 32, 64, 0, 3, 9, (0, 0), True),          # ragged N (output conv), out_ld = 4
    (1, 8, 256, 256, 320, 9, (256, 256), True),  # small M, long K: split-K when a workspace is given (SD-1.5 8x8 stage)
    (2, 16, 256, 0, 64, 1, (0, 0), False),       # 1x1 with 8 K tiles
    (3, 4, 128, 0, 96, 9, (0, 0), False),        # 4x4: generic 3x3 fallback, 36 K tiles
    (2, 8, 160, 0, 7, 9, (0, 0), True),          # split-K with a ragged N
    (16, 32, 640, 0, 1024, 1, (0, 0), False),    # 1x1 with a long K loop and many rows (gemm_dma8)
    (17, 32, 320, 320, 1000, 1, (0, 0), False),  # same, dual source, ragged M and N
    (16, 32, 64, 0, 1024, 1, (0, 0), False),     # short K, 512 tiles of 256x128: 8-wave LDS-DMA kernel (gemm_dma8)
    (17, 32, 32, 32, 1000, 1, (0, 0), False),    # same (544 tiles), dual source, ragged M and N
]


@pytest.mark.parametrize('ws', [False, True])
@pytest.mark.parametrize('force', [0, 1, 2, 128, 256])
@pytest.mark.parametrize('case', CONV_CASES)
def test_conv2d_nhwc_matches_aten(case, force, ws):
    from diff_sampler_amd import _lib, ops
    B, H, c0, c1, cout, taps, (ec0, ec1), use_norm = case
    lib = _lib.load()
    if use_norm and force == 1:
        pytest.skip('fused input normalisation exists only in the halo kernel')
    if use_norm and not lib.ds_conv3x3_halo_supported(H, H):
        pytest.skip('geometry not supported by the halo kernel')
    g = torch.Generator().manual_seed(hash(case) % 1000)
    k = 3 if taps == 9 else 1
    x = torch.randn(B, c0 + c1, H, H, generator=g)
    e = torch.randn(B, ec0 + ec1, H, H, generator=g) if ec0 else None
    w = torch.randn(cout, c0 + c1, k, k, generator=g) / (taps * (c0 + c1)) ** 0.5
    we = torch.randn(cout, ec0 + ec1, 1, 1, generator=g) / (ec0 + ec1) ** 0.5 if ec0 else None
    bias = torch.randn(cout, generator=g)
    cb = torch.randn(B, cout, generator=g)
    res = torch.randn(B, cout, H, H, generator=g)
    mu = torch.randn(B, c0 + c1, generator=g) * 0.3
    ga = 1 + 0.2 * torch.randn(B, c0 + c1, generator=g)
    be = 0.2 * torch.randn(B, c0 + c1, generator=g)
    # reference
    xin = F.silu((x - mu[:, :, None, None]) * ga[:, :, None, None] + be[:, :, None, None]) if use_norm else x
    ref = F.conv2d(xin, w, padding=k // 2)
    if ec0:
        ref = ref + F.conv2d(e, we)
    ref = (ref + bias[None, :, None, None] + cb[:, :, None, None] + res) * 0.7071
    ref = F.silu(ref) if cout % 2 == 0 else ref
    # device
    dev = 'cuda'
    xn = _nhwc(x).to(dev)
    x0 = xn[:, :c0].contiguous()
    x1 = xn[:, c0:].contiguous() if c1 else None
    en = _nhwc(e).to(dev) if ec0 else None
    e0 = en[:, :ec0].contiguous() if ec0 else None
    e1 = en[:, ec0:].contiguous() if ec1 else None
    wp = ops.pack_conv_weight(w.to(dev))
    if ec0:
        wp = torch.cat([wp, ops.pack_conv_weight(we.to(dev))], 1).contiguous()
    coefs = torch.stack([mu, ga, be], 1).contiguous().to(dev) if use_norm else None     # [B][3][C]
    old = cout if cout % 4 == 0 else -(-cout // 4) * 4
    out = torch.full((B * H * H, old), float('nan'), device=dev)
    a = _lib.ConvArgs(x0.data_ptr(), x1.data_ptr() if c1 else None, c0, c1, c0, c1, B, H, H, taps, wp.data_ptr(), cout,
                      bias.to(dev).data_ptr(), cb.to(dev).data_ptr(), cout, B, _nhwc(res).to(dev).data_ptr(), cout, 0.7071,
                      1 if cout % 2 == 0 else 0, out.data_ptr(), old, coefs.data_ptr() if use_norm else None, 1,
                      e0.data_ptr() if ec0 else None, e1.data_ptr() if ec1 else None, ec0, ec1, ec0, ec1)
    keep = [x0, x1, e0, e1, wp, coefs]     # noqa: F841  (raw pointers above)
    if ws:                                 # split-K scratch: the launcher splits the K loop of under-filled layers
        scratch = torch.full((8 << 20,), float('nan'), device=dev)
        a.workspace, a.workspace_floats = scratch.data_ptr(), scratch.numel()
    biasd, cbd, resd = bias.to(dev), cb.to(dev), _nhwc(res).to(dev)
    a.bias, a.cbias, a.res = biasd.data_ptr(), cbd.data_ptr(), resd.data_ptr()
    import ctypes as C
    a.tune.mode = force                    # ds_conv_tune: 1 = generic gather kernel, 128 / 256 = forced halo M tile (per call, no global state)
    rc = lib.ds_conv2d_nhwc(C.byref(a), _lib.stream_ptr())
    torch.cuda.synchronize()
    assert rc == 0, lib.ds_error_string(rc)
    got = out[:, :cout].cpu()
    assert _rel(got, _nhwc(ref)) < TOL


def test_conv_argument_errors():
    import ctypes as C
    from diff_sampler_amd import _lib
    lib = _lib.load()
    x = torch.zeros(64, 48, device='cuda')
    w = torch.zeros(128, 48 * 9, device='cuda')
    o = torch.zeros(64, 32, device='cuda')
    a = _lib.ConvArgs(x.data_ptr(), None, 48, 0, 48, 0, 1, 8, 8, 9, w.data_ptr(), 32, None, None, 0, 1, None, 0, 1.0, 0, o.data_ptr(), 32)
    assert lib.ds_conv2d_nhwc(C.byref(a), None) == -3          # c0 % 32 != 0 -> DS_E_SHAPE
    a.c0, a.ld0, a.taps = 32, 48, 5
    assert lib.ds_conv2d_nhwc(C.byref(a), None) == -1          # taps must be 1 or 9
    a.taps, a.ld0 = 9, 47
    assert lib.ds_conv2d_nhwc(C.byref(a), None) == -2          # ld % 4 != 0 -> DS_E_ALIGN


@pytest.mark.parametrize('C_,G,HW', [(64, 16, 16), (96, 24, 8), (192, 32, 16), (384, 32, 8), (320, 32, 64), (32, 8, 32)])
def test_groupnorm_stats_and_apply(C_, G, HW):
    from diff_sampler_amd import ops
    from diff_sampler_amd._lib import DS_ACT_SILU, DS_RESAMPLE_DOWN, DS_RESAMPLE_UP
    g = torch.Generator().manual_seed(C_)
    B = 3
    x = torch.randn(B, C_, HW, HW, generator=g) * 2 + 0.7
    gamma, beta = 1 + 0.1 * torch.randn(C_, generator=g), 0.1 * torch.randn(C_, generator=g)
    c0 = 64 if C_ > 64 else C_
    xn = _nhwc(x).cuda()
    x0, x1 = xn[:, :c0].contiguous(), (xn[:, c0:].contiguous() if C_ > c0 else None)
    mean, rstd = torch.empty(B * G, device='cuda'), torch.empty(B * G, device='cuda')
    ops.gn_stats(x0, c0, c0, B, HW, HW, G, 1e-5, mean, rstd, x1=x1, c1=C_ - c0, ld1=C_ - c0)
    # same statistics gathered by several workgroups per image (small-batch path), twice to exercise the counter reset
    from diff_sampler_amd._lib import DS_GN_MAX_CHUNKS
    part = torch.full((B * DS_GN_MAX_CHUNKS * 128,), float('nan'), dtype=torch.float64, device='cuda')
    cnt = torch.zeros(B, dtype=torch.int32, device='cuda')
    for _ in range(2):
        mean2, rstd2 = torch.full_like(mean, float('nan')), torch.full_like(rstd, float('nan'))
        ops.gn_stats(x0, c0, c0, B, HW, HW, G, 1e-5, mean2, rstd2, x1=x1, c1=C_ - c0, ld1=C_ - c0, partial=part, counters=cnt)
        torch.cuda.synchronize()
        assert torch.allclose(mean2, mean, rtol=1e-6, atol=1e-7) and torch.allclose(rstd2, rstd, rtol=1e-6, atol=1e-7)
    ref = F.silu(F.group_norm(x, G, gamma, beta, 1e-5))
    for rs, refr in [(0, ref), (DS_RESAMPLE_DOWN, F.avg_pool2d(ref, 2)), (DS_RESAMPLE_UP, F.interpolate(ref, scale_factor=2, mode='nearest'))]:
        ho = refr.shape[-1]
        out = torch.empty(B * ho * ho, C_, device='cuda')
        ops.norm_act(x0, c0, c0, B, HW, HW, out, C_, x1=x1, c1=C_ - c0, ld1=C_ - c0, groups=G, eps=1e-5, mean=mean, rstd=rstd,
                     gamma=gamma.cuda(), beta=beta.cuda(), act=DS_ACT_SILU, resample=rs)
        torch.cuda.synchronize()
        assert _rel(out.cpu(), _nhwc(refr)) < TOL, rs


def test_batched_gemm_and_softmax():
    from diff_sampler_amd import ops
    g = torch.Generator().manual_seed(4)
    Bz, Hd, S, Ch = 3, 2, 64, 64
    q = torch.randn(Bz, S, Hd * Ch, generator=g)
    k = torch.randn(Bz, S, Hd * Ch, generator=g)
    qd, kd = q.cuda().contiguous(), k.cuda().contiguous()
    sc = torch.empty(Bz * Hd, S, S, device='cuda')
    ops.gemm_nt_batched(qd, Hd * Ch, kd, Hd * Ch, sc, S, S, S, Ch, batch=Bz, heads=Hd, a_bs=S * Hd * Ch, a_hs=Ch,
                        b_bs=S * Hd * Ch, b_hs=Ch, c_bs=Hd * S * S, c_hs=S * S, alpha=0.125)
    ref = torch.einsum('bqhc,bkhc->bhqk', q.reshape(Bz, S, Hd, Ch), k.reshape(Bz, S, Hd, Ch)) * 0.125
    torch.cuda.synchronize()
    assert _rel(sc.cpu().reshape(Bz, Hd, S, S), ref) < TOL
    ops.softmax_rows(sc, sc, Bz * Hd * S, S, S)
    torch.cuda.synchronize()
    assert _rel(sc.cpu().reshape(Bz, Hd, S, S), ref.softmax(-1)) < TOL


def test_solver_update_bandwidth_kernel_semantics():
    """x' = cx*xb + cm*m + sum ch*hist with m = d or D, raw NHWC network output and AFS -- against the formulas."""
    from diff_sampler_amd import ops
    g = torch.Generator().manual_seed(8)
    B, Cc, H = 5, 3, 16
    x = torch.randn(B, Cc, H, H, generator=g) * 3
    xb = torch.randn(B, Cc, H, H, generator=g)
    Fraw = torch.randn(B, Cc, H, H, generator=g)
    h0, h1 = torch.randn(B, Cc, H, H, generator=g), torch.randn(B, Cc, H, H, generator=g)
    t, sig, sd = 1.7, 2.1, 0.5
    cskip = sd ** 2 / (sig ** 2 + sd ** 2); cout = sig * sd / (sig ** 2 + sd ** 2) ** 0.5
    D = cskip * x + cout * Fraw
    d = (x - D) / t
    f4 = torch.zeros(B * H * H, 4); f4[:, :Cc] = _nhwc(Fraw)
    xo, mo = torch.empty(B, Cc, H, H, device='cuda'), torch.empty(B, Cc, H, H, device='cuda')
    hc = [0.9, -0.4, 0.3, 0.2, 0.0, t, sig, 0.0]
    a = ops.make_update_args(x.cuda(), xb.cuda(), f4.cuda(), B, Cc, H, H, xo, raw=True, f_ld=4, hist=[h0.cuda(), h1.cuda()], hcoefs=hc,
                             sigma_data=sd, m_out=mo, store_d=True)
    ops.solver_update(a); torch.cuda.synchronize()
    assert _rel(mo.cpu(), d) < TOL
    assert _rel(xo.cpu(), 0.9 * xb - 0.4 * d + 0.3 * h0 + 0.2 * h1) < TOL
    # the same step with the raw network output given channel-planar (f_ld = 0: what the engine's output conv writes)
    xo.fill_(float('nan')); mo.fill_(float('nan'))
    a = ops.make_update_args(x.cuda(), xb.cuda(), Fraw.cuda().contiguous(), B, Cc, H, H, xo, raw=True, f_ld=0, hist=[h0.cuda(), h1.cuda()],
                             hcoefs=hc, sigma_data=sd, m_out=mo, store_d=True)
    ops.solver_update(a); torch.cuda.synchronize()
    assert _rel(mo.cpu(), d) < TOL
    assert _rel(xo.cpu(), 0.9 * xb - 0.4 * d + 0.3 * h0 + 0.2 * h1) < TOL
    a = ops.make_update_args(x.cuda(), x.cuda(), None, B, Cc, H, H, xo, hcoefs=[1.0, 0.5, 0, 0, 0, t, t, 0], afs=True, m_out=mo, store_d=False)
    ops.solver_update(a); torch.cuda.synchronize()
    dafs = x / (1 + t * t) ** 0.5
    assert _rel(mo.cpu(), x - t * dafs) < TOL and _rel(xo.cpu(), x + 0.5 * (x - t * dafs)) < TOL


@pytest.mark.parametrize('d,heads,sq,skv', [(40, 8, 256, 256), (40, 8, 1024, 77), (64, 3, 64, 64), (80, 2, 160, 77),
                                            (160, 2, 64, 64), (256, 1, 256, 256), (8, 4, 96, 33), (32, 2, 128, 1000),
                                            (16, 1, 40, 5), (96, 1, 32, 64), (128, 2, 200, 130)])
def test_fused_attention_matches_softmax_qk_v(d, heads, sq, skv):
    """ds_attention against softmax(q k^T * scale) v in fp64; q/k/v are read in place from packed projections (column
    offsets h*d inside wider rows), cross-attention lengths included."""
    from diff_sampler_amd import ops, _lib
    assert _lib.load().ds_attention_supported(d)
    g = torch.Generator().manual_seed(d * 1000 + sq)
    Bz, C_ = 2, heads * d
    qkv = torch.randn(Bz, sq, 3 * C_ + 4, generator=g)           # q | (unused k) | junk, ld not a multiple of C
    kv = torch.randn(Bz, skv, 2 * C_, generator=g) * 1.5
    q = qkv[:, :, :C_]
    k, v = kv[:, :, :C_], kv[:, :, C_:]
    scale = d ** -0.5
    qd, kvd = qkv.cuda().contiguous(), kv.cuda().contiguous()
    out = torch.full((Bz, sq, C_), float('nan'), device='cuda')
    ops.attention(qd, kvd, kvd[:, :, C_:], out, batch=Bz, heads=heads, sq=sq, skv=skv, d=d, ldq=3 * C_ + 4, ldk=2 * C_, ldv=2 * C_,
                  ldo=C_, q_bs=sq * (3 * C_ + 4), k_bs=skv * 2 * C_, v_bs=skv * 2 * C_, o_bs=sq * C_, scale=scale)
    torch.cuda.synchronize()
    qh = q.reshape(Bz, sq, heads, d).double()
    kh = k.reshape(Bz, skv, heads, d).double()
    vh = v.reshape(Bz, skv, heads, d).double()
    w = (torch.einsum('bqhd,bkhd->bhqk', qh, kh) * scale).softmax(-1)
    ref = torch.einsum('bhqk,bkhd->bqhd', w, vh).reshape(Bz, sq, C_).float()
    assert _rel(out.cpu(), ref) < TOL


@pytest.mark.parametrize('d,bz,heads,sq,skv', [(256, 8, 1, 256, 256), (256, 3, 1, 64, 64), (128, 2, 2, 200, 130), (256, 2, 1, 100, 77), (128, 1, 3, 32, 1000)])
def test_fused_attention_channel_split_block_agrees_with_the_query_split_kernel(d, bz, heads, sq, skv):
    """Round 6 (csrc/attention.hip, DSPLIT): for head sizes that are multiples of 128 a small launch takes 32-query blocks whose four waves split the
    CHANNELS (partial S^T blocks summed through LDS) instead of 128-query blocks whose waves split the queries.  Both against fp64, and against each
    other: the scores are summed in another order, so fp32 rounding apart (observed 1 - 2e-6 of the output scale; bound 5e-6), not bit for bit; the library's own choice at
    these sizes is the channel split (equal bits with variant 2), and three repeats of it are identical (the LDS exchange is ordered by barriers)."""
    from diff_sampler_amd import ops
    g = torch.Generator().manual_seed(d + bz * 7 + sq)
    C_ = heads * d
    q = torch.randn(bz, sq, C_, generator=g)
    kv = torch.randn(bz, skv, 2 * C_, generator=g) * 1.5
    scale = d ** -0.5
    qd, kvd = q.cuda().contiguous(), kv.cuda().contiguous()
    outs = {}
    for variant in (1, 2, None, 2, 2):
        out = torch.full((bz, sq, C_), float('nan'), device='cuda')
        ops.attention(qd, kvd, kvd[:, :, C_:], out, batch=bz, heads=heads, sq=sq, skv=skv, d=d, ldq=C_, ldk=2 * C_, ldv=2 * C_, ldo=C_,
                      q_bs=sq * C_, k_bs=skv * 2 * C_, v_bs=skv * 2 * C_, o_bs=sq * C_, scale=scale, variant=variant)
        torch.cuda.synchronize()
        outs.setdefault(variant, []).append(out.cpu())
    qh = q.reshape(bz, sq, heads, d).double()
    kh = kv[:, :, :C_].reshape(bz, skv, heads, d).double()
    vh = kv[:, :, C_:].reshape(bz, skv, heads, d).double()
    w = (torch.einsum('bqhd,bkhd->bhqk', qh, kh) * scale).softmax(-1)
    ref = torch.einsum('bhqk,bkhd->bqhd', w, vh).reshape(bz, sq, C_).float()
    assert _rel(outs[1][0], ref) < TOL and _rel(outs[2][0], ref) < TOL
    assert _rel(outs[2][0], outs[1][0]) < 5e-6
    assert torch.equal(outs[None][0], outs[2][0]) and torch.equal(outs[2][1], outs[2][0]) and torch.equal(outs[2][2], outs[2][0])


@pytest.mark.parametrize('rows,k,cout,act', [(1, 512, 8448, 0), (1, 128, 512, 1), (2, 1280, 20160, 0), (4, 320, 1280, 1), (3, 768, 200, 0), (1, 64, 64, 1)])
def test_linear_on_at_most_four_rows_streams_the_weights(rows, k, cout, act):
    """Round 6 (csrc/gemm_conv.hip: gemv_rows_kernel, kernel id 2573): the embedding path's Linear layers see one row per sampler call (up to four:
    a handful of sigma rows); a wave per output column reads its weight row once instead of a 128-row matrix tile walking K.  Against fp64 and
    against the matrix kernel the same call takes with ds_conv_tune.mode = 1 (fp32 rounding apart); five rows take the matrix kernel."""
    import ctypes as C
    from diff_sampler_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(rows * 31 + k + cout)
    x = torch.randn(rows, k, generator=g)
    wt = torch.randn(cout, k, generator=g) / k ** 0.5
    bias = torch.randn(cout, generator=g)
    ref = (x.double() @ wt.double().t() + bias.double()) * 0.75
    if act:
        ref = F.silu(ref)
    xd, wp, bd = x.cuda(), ops.pack_linear_weight(wt.cuda()), bias.cuda()
    outs = []
    for mode in (0, 1):
        out = torch.full((rows, cout), float('nan'), device='cuda')
        a = _lib.ConvArgs(xd.data_ptr(), None, k, 0, k, 0, rows, 1, 1, 1, wp.data_ptr(), cout, bd.data_ptr(), None, 0, 1, None, 0, 0.75, act,
                          out.data_ptr(), cout)
        a.tune.mode = mode
        assert (lib.ds_conv_kernel_id(C.byref(a)) == 2573) == (mode == 0)
        assert lib.ds_conv2d_nhwc(C.byref(a), _lib.stream_ptr()) == 0
        torch.cuda.synchronize()
        outs.append(out.cpu())
        assert _rel(outs[-1], ref.float()) < TOL, mode
    assert _rel(outs[0], outs[1]) < 5e-6
    x5 = torch.randn(5, k, generator=g).cuda()
    out5 = torch.empty(5, cout, device='cuda')
    a = _lib.ConvArgs(x5.data_ptr(), None, k, 0, k, 0, 5, 1, 1, 1, wp.data_ptr(), cout, bd.data_ptr(), None, 0, 1, None, 0, 0.75, act, out5.data_ptr(), cout)
    assert lib.ds_conv_kernel_id(C.byref(a)) != 2573


def test_layernorm_geglu_cfg_and_timestep_embedding():
    from diff_sampler_amd import ops
    g = torch.Generator().manual_seed(9)
    for rows, cols in [(77, 320), (130, 1280), (5, 64)]:
        x = torch.randn(rows, cols + 8, generator=g) * 3 + 1
        ga, be = torch.randn(cols, generator=g), torch.randn(cols, generator=g)
        y = torch.empty(rows, cols, device='cuda')
        ops.layernorm_rows(x.cuda(), cols + 8, ga.cuda(), be.cuda(), 1e-5, y, cols, rows, cols)
        torch.cuda.synchronize()
        assert _rel(y.cpu(), F.layer_norm(x[:, :cols], (cols,), ga, be, 1e-5)) < TOL
    x = torch.randn(50, 2 * 1280, generator=g) * 2
    y = torch.empty(50, 1280, device='cuda')
    ops.geglu(x.cuda(), 2560, y, 1280, 50, 1280)
    torch.cuda.synchronize()
    assert _rel(y.cpu(), x[:, :1280] * F.gelu(x[:, 1280:])) < TOL
    # CFG epilogue
    n, c, h, w = 3, 4, 8, 8
    xi = torch.randn(n, c, h, w, generator=g)
    f = torch.randn(2 * n * h * w, 4, generator=g)
    sig = torch.tensor([3.0, 0.5, 11.0])
    out = torch.empty(n, c, h, w, device='cuda')
    ops.cfg_denoise(xi.cuda(), f.cuda(), 4, sig.cuda(), n, 7.5, True, n, c, h, w, out)
    fu = f[:n * h * w].reshape(n, h, w, c).permute(0, 3, 1, 2)
    fc = f[n * h * w:].reshape(n, h, w, c).permute(0, 3, 1, 2)
    ref = xi - sig.reshape(-1, 1, 1, 1) * (fu + 7.5 * (fc - fu))
    torch.cuda.synchronize()
    assert _rel(out.cpu(), ref) < TOL
    ops.cfg_denoise(xi.cuda(), f.cuda(), 4, sig[:1].cuda(), 1, 1.0, False, n, c, h, w, out)
    torch.cuda.synchronize()
    assert _rel(out.cpu(), xi - 3.0 * fu) < TOL
    # timestep embedding with the argument given directly (flag 2), [cos | sin]
    tt = torch.tensor([999.0, 0.25, 500.5])
    half = 160
    freqs = torch.exp(-torch.log(torch.tensor(10000.0)) * torch.arange(half, dtype=torch.float32) / half)
    emb = torch.empty(3, 320, device='cuda')
    ops.noise_embed(tt.cuda(), 3, freqs.cuda(), 320, 2, emb, 320)
    torch.cuda.synchronize()
    args = tt[:, None] * freqs[None]
    assert float((emb.cpu() - torch.cat([args.cos(), args.sin()], -1)).abs().max()) < 2e-4   # sin/cos of arguments up to 1e3


@pytest.mark.parametrize('B,Ho,cin,cout', [(2, 16, 64, 96), (3, 8, 32, 32), (1, 32, 320, 320)])
def test_stride2_conv_matches_aten(B, Ho, cin, cout):
    from diff_sampler_amd import ops
    g = torch.Generator().manual_seed(B * 100 + Ho)
    x = torch.randn(B, cin, 2 * Ho, 2 * Ho, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    bias = torch.randn(cout, generator=g)
    out = torch.empty(B * Ho * Ho, cout, device='cuda')
    ops.conv2d_nhwc(_nhwc(x).cuda(), cin, cin, B, Ho, Ho, ops.pack_conv_weight(wt).cuda(), cout, out, cout, taps=9, bias=bias.cuda(),
                    stride=2)
    torch.cuda.synchronize()
    ref = F.conv2d(x, wt, bias, stride=2, padding=1)
    assert _rel(out.cpu(), _nhwc(ref)) < TOL


@pytest.mark.parametrize('B,H,cin,cout,ws', [(3, 16, 64, 3, False), (2, 8, 256, 3, True), (2, 32, 32, 5, False)])
def test_conv_planar_output(B, H, cin, cout, ws):
    """out_nchw: the few-channel output conv writes NCHW planes directly (with and without the split-K reduce)."""
    import ctypes as C
    from diff_sampler_amd import _lib, ops
    g = torch.Generator().manual_seed(B * 7 + H)
    x = torch.randn(B, cin, H, H, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    bias = torch.randn(cout, generator=g)
    xn, wp, bd = _nhwc(x).cuda(), ops.pack_conv_weight(wt).cuda(), bias.cuda()
    out = torch.full((B, cout, H, H), float('nan'), device='cuda')
    a = _lib.ConvArgs(xn.data_ptr(), None, cin, 0, cin, 0, B, H, H, 9, wp.data_ptr(), cout, bd.data_ptr(), None, 0, 1, None, 0, 1.0, 0,
                      out.data_ptr(), 4)
    a.out_nchw = 1
    if ws:
        scratch = torch.empty(4 << 20, device='cuda')
        a.workspace, a.workspace_floats = scratch.data_ptr(), scratch.numel()
    lib = _lib.load()
    assert lib.ds_conv2d_nhwc(C.byref(a), _lib.stream_ptr()) == 0
    torch.cuda.synchronize()
    assert _rel(out.cpu(), F.conv2d(x, wt, bias, padding=1)) < TOL


THIN_CASES = [
    # B, H, W, cin, cout, norm (0 none / 1 affine / 2 affine + SiLU), planar output, bias
    (2, 32, 32, 256, 3, 2, True, True),          # EDM CIFAR-10 head: GroupNorm + SiLU fused, NCHW output
    (1, 64, 64, 192, 3, 2, True, True),          # ImageNet-64 head
    (2, 64, 64, 320, 4, 0, False, True),         # SD-1.5 head on the normalised tensor, rows with out_ld = 4
    (3, 8, 8, 64, 1, 1, False, False),           # one output, 8x8 images (a wave = one image), no bias
    (2, 16, 32, 96, 2, 2, True, False),          # non-square
    (5, 16, 16, 32, 4, 0, True, True),           # one 32-channel step, five images (M = 1 280: the last workgroup is ragged)
    # rounds per wave (round 6: small launches take fewer than eight, launch_conv3x3_thin): the cases above all run one round
    (130, 32, 32, 32, 3, 2, True, True),         # M = 133 120: eight rounds, 520 workgroups
    (40, 32, 32, 32, 4, 0, False, True),         # M = 40 960: two rounds
    (70, 32, 32, 32, 3, 1, True, False),         # M = 71 680: four rounds
]


@pytest.mark.parametrize('case', THIN_CASES)
def test_conv_thin_output_kernel_matches_the_matrix_kernels(case):
    """csrc/conv3x3_thin.hip (kernel id 2570): 3x3 layers with at most four output channels -- the network heads -- as VALU work on 8 pixels x
    32 channels per wave instruction instead of a 64- / 128-column matrix tile.  Against ATen in fp64 (the fp32 bound of the other kernels)
    and against the matrix kernel the same call takes with ds_conv_tune.mode = 8."""
    import ctypes as C
    from diff_sampler_amd import _lib, ops
    B, H, W, cin, cout, norm, planar, with_bias = case
    lib = _lib.load()
    dev = 'cuda'
    g = torch.Generator().manual_seed(sum(case[:5]))
    x = torch.randn(B, cin, H, W, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5
    bias = torch.randn(cout, generator=g)
    mu = torch.randn(B, cin, generator=g) * 0.3
    ga = 1 + 0.2 * torch.randn(B, cin, generator=g)
    be = 0.2 * torch.randn(B, cin, generator=g)
    xin = x.double()
    if norm:
        xin = (xin - mu.double()[:, :, None, None]) * ga.double()[:, :, None, None] + be.double()[:, :, None, None]
        if norm == 2:
            xin = F.silu(xin)
    ref = F.conv2d(xin, wt.double(), bias.double() if with_bias else None, padding=1) * 0.5
    xn, wp, bd = _nhwc(x).to(dev), ops.pack_conv_weight(wt).to(dev), bias.to(dev)
    coefs = torch.stack([mu, ga, be], 1).contiguous().to(dev)
    outs = []
    for mode in (0, 8):
        out = torch.full((B, cout, H, W) if planar else (B * H * W, 4), float('nan'), device=dev)
        a = _lib.ConvArgs(xn.data_ptr(), None, cin, 0, cin, 0, B, H, W, 9, wp.data_ptr(), cout, bd.data_ptr() if with_bias else None, None, 0, 1,
                          None, 0, 0.5, 0, out.data_ptr(), 4, coefs.data_ptr() if norm else None, 1 if norm == 2 else 0)
        a.out_nchw = int(planar)
        a.tune.mode = mode
        kid = lib.ds_conv_kernel_id(C.byref(a))
        assert (kid == 2570) == (mode == 0), kid
        rc = lib.ds_conv2d_nhwc(C.byref(a), _lib.stream_ptr())
        torch.cuda.synchronize()
        if mode == 8 and rc != 0 and norm:
            continue                              # (fused normalisation outside the halo kernel's geometries: no matrix kernel to compare with)
        assert rc == 0, lib.ds_error_string(rc)
        got = out.cpu() if planar else out[:, :cout].cpu()
        assert torch.isfinite(got).all()
        want = ref.float() if planar else _nhwc(ref.float())
        assert _rel(got, want) < TOL, mode
        outs.append(got)
    if len(outs) == 2:
        assert _rel(outs[0], outs[1]) < TOL


@pytest.mark.parametrize('B,H,cin,cout,taps,ws', [(3, 16, 64, 128, 9, False), (2, 8, 256, 320, 9, True), (2, 32, 32, 192, 9, False),
                                                   (16, 32, 640, 1024, 1, False), (4, 8, 64, 64, 1, False), (2, 64, 32, 64, 9, False)])
def test_conv_epilogue_statistics_feed_groupnorm(B, H, cin, cout, taps, ws):
    """stats_out + ds_gn_finalize reproduce ds_gn_stats / F.group_norm statistics of the convolution's output (128- and
    64-column halo tiles, the split-K reduce path, the generic and the 8-wave LDS-DMA 1x1 kernels)."""
    import ctypes as C
    from diff_sampler_amd import _lib, ops
    g = torch.Generator().manual_seed(B * 11 + H + cout)
    k = 3 if taps == 9 else 1
    x = torch.randn(B, cin, H, H, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (taps * cin) ** 0.5
    bias = torch.randn(cout, generator=g)
    xn, wp, bd = _nhwc(x).cuda(), ops.pack_conv_weight(wt).cuda(), bias.cuda()
    M = B * H * H
    out = torch.empty(M, cout, device='cuda')
    stats = torch.full((-(-M // 64) * 2 * cout,), float('nan'), device='cuda')
    a = _lib.ConvArgs(xn.data_ptr(), None, cin, 0, cin, 0, B, H, H, taps, wp.data_ptr(), cout, bd.data_ptr(), None, 0, 1, None, 0, 1.0, 0,
                      out.data_ptr(), cout)
    a.stats_out = stats.data_ptr()
    if ws:
        scratch = torch.empty(8 << 20, device='cuda')
        a.workspace, a.workspace_floats = scratch.data_ptr(), scratch.numel()
    lib = _lib.load()
    assert lib.ds_conv2d_nhwc(C.byref(a), _lib.stream_ptr()) == 0
    G_ = 32
    mean, rstd = torch.empty(B * G_, device='cuda'), torch.empty(B * G_, device='cuda')
    coefs = torch.empty(B * 3 * cout, device='cuda')
    gamma, beta = (1 + 0.1 * torch.randn(cout, generator=g)).cuda(), (0.1 * torch.randn(cout, generator=g)).cuda()
    f = _lib.GnFinalizeArgs(stats.data_ptr(), None, cout, 0, B, H * H, G_, 1e-5, gamma.data_ptr(), beta.data_ptr(), None, None, 0, 1,
                            mean.data_ptr(), rstd.data_ptr(), coefs.data_ptr())
    assert lib.ds_gn_finalize(C.byref(f), _lib.stream_ptr()) == 0
    torch.cuda.synchronize()
    ref = F.conv2d(x, wt, bias, padding=k // 2)
    assert _rel(out.cpu(), _nhwc(ref)) < TOL
    r = ref.reshape(B, G_, -1).double()
    assert torch.allclose(mean.cpu(), r.mean(-1).float().reshape(-1), rtol=1e-4, atol=1e-5)
    assert torch.allclose(rstd.cpu(), (1.0 / (r.var(-1, unbiased=False) + 1e-5).sqrt()).float().reshape(-1), rtol=1e-4, atol=1e-5)
    cf = coefs.cpu().reshape(B, 3, cout)
    cpg = cout // G_
    assert torch.allclose(cf[:, 1], (rstd.cpu().reshape(B, G_).repeat_interleave(cpg, 1) * gamma.cpu()), rtol=1e-5)


@pytest.mark.parametrize('B,H,W,cin,cout', [(2, 8, 16, 64, 128), (3, 16, 8, 32, 64), (1, 32, 64, 32, 128), (2, 4, 32, 64, 64)])
def test_conv_non_square_images(B, H, W, cin, cout):
    """The ABI takes h and w separately: non-square NHWC images through the halo / generic kernels, stride 1 and 2, with the
    epilogue statistics."""
    import ctypes as C
    from diff_sampler_amd import _lib, ops
    g = torch.Generator().manual_seed(H * 100 + W)
    x = torch.randn(B, cin, H, W, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    bias = torch.randn(cout, generator=g)
    xn, wp, bd = x.permute(0, 2, 3, 1).reshape(-1, cin).contiguous().cuda(), ops.pack_conv_weight(wt).cuda(), bias.cuda()
    lib = _lib.load()
    for stride in (1, 2):
        Ho, Wo = H // stride, W // stride
        out = torch.full((B * Ho * Wo, cout), float('nan'), device='cuda')
        a = _lib.ConvArgs(xn.data_ptr(), None, cin, 0, cin, 0, B, Ho, Wo, 9, wp.data_ptr(), cout, bd.data_ptr(), None, 0, 1, None, 0, 1.0, 0,
                          out.data_ptr(), cout)
        a.stride = stride
        assert lib.ds_conv2d_nhwc(C.byref(a), _lib.stream_ptr()) == 0
        torch.cuda.synchronize()
        ref = F.conv2d(x, wt, bias, stride=stride, padding=1).permute(0, 2, 3, 1).reshape(-1, cout)
        assert _rel(out.cpu(), ref) < TOL, stride


@pytest.mark.parametrize('rows,k,inner', [(300, 64, 128), (16384, 640, 1280), (77, 320, 1280)])
def test_geglu_fused_into_projection(rows, k, inner):
    """DS_ACT_GEGLU: x W^T + b with the gate applied in the epilogue (generic and 8-wave LDS-DMA kernels) == value * gelu(gate)."""
    import ctypes as C
    from diff_sampler_amd import _lib, ops
    g = torch.Generator().manual_seed(rows + k)
    x = torch.randn(rows, k, generator=g)
    wt = torch.randn(2 * inner, k, generator=g) / k ** 0.5
    bias = torch.randn(2 * inner, generator=g) * 0.3
    perm = torch.arange(2 * inner).reshape(-1, 2, 32)
    perm = (perm[:, 0] // 64 * 32 + perm[:, 0] % 32).reshape(-1, 1, 32).repeat(1, 2, 1)
    perm[:, 1] += inner
    perm = perm.reshape(-1)
    xd, wp, bd = x.cuda(), ops.pack_linear_weight(wt[perm].cuda()), bias[perm].contiguous().cuda()
    out = torch.full((rows, inner), float('nan'), device='cuda')
    a = _lib.ConvArgs(xd.data_ptr(), None, k, 0, k, 0, rows, 1, 1, 1, wp.data_ptr(), 2 * inner, bd.data_ptr(), None, 0, 1, None, 0, 1.0,
                      _lib.DS_ACT_GEGLU, out.data_ptr(), inner)
    assert _lib.load().ds_conv2d_nhwc(C.byref(a), _lib.stream_ptr()) == 0
    torch.cuda.synchronize()
    y = F.linear(x, wt, bias)
    assert _rel(out.cpu(), y[:, :inner] * F.gelu(y[:, inner:])) < TOL


HALO2_CASES = [
    # B, H(=W), c0, c1, cout, (ec0, ec1), norm, act
    (1, 16, 32, 0, 128, (0, 0), False, False),        # one tile, one slab
    (2, 16, 64, 32, 128, (0, 0), True, True),          # dual source
    (1, 32, 64, 0, 256, (0, 0), True, True),           # 4 row tiles x 2 column tiles per image
    (2, 32, 32, 32, 128, (64, 32), True, True),        # fused 1x1 skip projection slabs (dual extra source)
    (1, 32, 96, 0, 192, (32, 0), True, False),         # ragged channel count: 128-column tile here + 64-column tail on kernel 1
    (1, 64, 32, 0, 128, (0, 0), True, True),           # W = 64 (7 halo slots per thread)
    (1, 64, 32, 32, 128, (32, 0), False, False),
    (3, 16, 288, 0, 128, (0, 0), True, True),          # 9 slabs (weight-buffer parity flips every slab)
]


def _run_halo_case(case, variant, workspace=False, tile=256, want_kid=None):
    """One fused 3x3 layer (dual source, fused normalisation + SiLU, 1x1 skip columns, bias, per-image bias, residual, scale,
    epilogue statistics) through ds_conv2d_nhwc with the 256-pixel tile forced and the given kernel variant; checked against ATen."""
    import ctypes as C
    from diff_sampler_amd import _lib, ops
    B, H, c0, c1, cout, (ec0, ec1), use_norm, act = case
    lib = _lib.load()
    g = torch.Generator().manual_seed(sum(case[:5]) + 7)
    x = torch.randn(B, c0 + c1, H, H, generator=g)
    e = torch.randn(B, ec0 + ec1, H, H, generator=g) if ec0 else None
    w = torch.randn(cout, c0 + c1, 3, 3, generator=g) / (9 * (c0 + c1)) ** 0.5
    we = torch.randn(cout, ec0 + ec1, 1, 1, generator=g) / (ec0 + ec1) ** 0.5 if ec0 else None
    bias = torch.randn(cout, generator=g)
    cb = torch.randn(B, cout, generator=g)
    res = torch.randn(B, cout, H, H, generator=g)
    mu = torch.randn(B, c0 + c1, generator=g) * 0.3
    ga = 1 + 0.2 * torch.randn(B, c0 + c1, generator=g)
    be = 0.2 * torch.randn(B, c0 + c1, generator=g)
    xin = x
    if use_norm:
        xin = (x - mu[:, :, None, None]) * ga[:, :, None, None] + be[:, :, None, None]
        xin = F.silu(xin) if act else xin
    ref = F.conv2d(xin, w, padding=1)
    if ec0:
        ref = ref + F.conv2d(e, we)
    ref = (ref + bias[None, :, None, None] + cb[:, :, None, None] + res) * 0.7071
    dev = 'cuda'
    xn = _nhwc(x).to(dev)
    x0 = xn[:, :c0].contiguous()
    x1 = xn[:, c0:].contiguous() if c1 else None
    en = _nhwc(e).to(dev) if ec0 else None
    e0 = en[:, :ec0].contiguous() if ec0 else None
    e1 = en[:, ec0:].contiguous() if ec1 else None
    wp = ops.pack_conv_weight(w.to(dev))
    if ec0:
        wp = torch.cat([wp, ops.pack_conv_weight(we.to(dev))], 1).contiguous()
    coefs = torch.stack([mu, ga, be], 1).contiguous().to(dev) if use_norm else None
    out = torch.full((B * H * H, cout), float('nan'), device=dev)
    stats = torch.zeros(B * H * H // 64 * 2 * cout, device=dev)
    biasd, cbd, resd = bias.to(dev), cb.to(dev), _nhwc(res).to(dev)
    a = _lib.ConvArgs(x0.data_ptr(), x1.data_ptr() if c1 else None, c0, c1, c0, c1, B, H, H, 9, wp.data_ptr(), cout, biasd.data_ptr(),
                      cbd.data_ptr(), cout, B, resd.data_ptr(), cout, 0.7071, 0, out.data_ptr(), cout,
                      coefs.data_ptr() if use_norm else None, 1 if act else 0,
                      e0.data_ptr() if ec0 else None, e1.data_ptr() if ec1 else None, ec0, ec1, ec0, ec1)
    a.stats_out = stats.data_ptr()
    if workspace:                          # split-K scratch
        scratch = torch.full((20 << 20,), float('nan'), device=dev)
        a.workspace, a.workspace_floats = scratch.data_ptr(), scratch.numel()
    a.tune.mode, a.tune.variant = tile, variant
    kid = lib.ds_conv_kernel_id(C.byref(a))
    rc = lib.ds_conv2d_nhwc(C.byref(a), _lib.stream_ptr())
    torch.cuda.synchronize()
    assert rc == 0, lib.ds_error_string(rc)
    if variant == 3:
        assert kid == 2560, 'the layer was not routed to the second-generation kernel'
    if want_kid is not None:
        assert kid == want_kid, (kid, want_kid)
    if (variant & 31) == 6:
        assert kid == 2565, 'the layer was not routed to the 256 x 256-tile kernel'
    if variant & 16384:
        assert kid == 2568, 'the layer was not routed to the 256 x 192-tile kernel'
    want = _nhwc(ref)
    assert _rel(out.cpu(), want) < TOL
    if cout % 64 == 0:       # the epilogue's per-(64-row block, channel) sums feed the consumer's GroupNorm
        st = stats.cpu().reshape(-1, 2, cout)
        blocks = want.reshape(-1, 64, cout)
        assert _rel(st[:, 0], blocks.sum(1)) < 1e-4 and _rel(st[:, 1], (blocks ** 2).sum(1)) < 1e-4


def _need_experiments():
    """The kernel variants kept as A/B records (conv3x3_halo2 modes 0 / 1, conv3x3_f16dmah) are compiled only with DS_BUILD_EXPERIMENTS=1
    (diff_sampler_amd/build.py); the default library -- what the engines run -- does not hold them."""
    from diff_sampler_amd import _lib
    if not _lib.load().ds_build_experiments() & 1:
        pytest.skip('experimental kernel variant: build with DS_BUILD_EXPERIMENTS=1')



@pytest.mark.parametrize('case', HALO2_CASES)
def test_conv_halo2_kernel_matches_aten(case):
    """Second-generation 256 x 128 halo kernel (conv3x3_halo2.hip), routed by ds_conv_tune.variant = 3 with the 256-pixel tile
    forced; the routing itself is asserted through the launch counter."""
    _need_experiments()
    _run_halo_case(case, 3)


WIDE_N_CASES = [
    # B, H(=W), c0, c1, cout, (ec0, ec1), norm, act
    (1, 32, 32, 0, 256, (0, 0), False, False),         # one 256-column tile per row tile
    (2, 16, 64, 32, 256, (0, 0), True, True),          # dual source, fused normalisation
    (1, 32, 64, 0, 512, (32, 0), True, True),          # two 256-column tiles, fused 1x1 skip columns
    (2, 32, 32, 32, 384, (64, 32), True, False),       # 384 = 256 + 128: second launch of 128-column tiles from column 256
    (1, 16, 96, 0, 320, (0, 0), True, True),           # 320 = 256 + 64: 64-column tail tiles from column 256
    (3, 16, 288, 0, 576, (0, 0), False, False),        # 576 = 2 x 256 + 64, nine slabs
    (1, 64, 32, 32, 320, (32, 0), True, True),         # W = 64 (7 halo slots per thread), SD-1.5's 320 channels
]


@pytest.mark.parametrize('case', WIDE_N_CASES)
def test_conv_wide_n_tiles_match_aten(case):
    """conv3x3_halo_kernel<4, NT = 4>: 256-pixel x 256-channel tiles (64 x 128 per wave) for the 256-multiples of the channel
    count, the remainder on 128- / 64-column tiles of the same layer (ds_conv_tune.variant = 6 forces the shape at test sizes)."""
    _run_halo_case(case, 6)


WIDE192_CASES = [
    # B, H(=W), c0, c1, cout, (ec0, ec1), norm, act          -- ADM channel counts: 192-multiples that are not 256-multiples
    (1, 32, 32, 0, 192, (0, 0), False, False),
    (2, 16, 64, 32, 384, (0, 0), True, True),          # two 192-column tiles, dual source, fused normalisation
    (1, 64, 32, 32, 192, (32, 0), True, True),         # W = 64 (7 halo slots per thread), fused 1x1 skip columns
    (3, 16, 288, 0, 576, (0, 0), False, False),        # three 192-column tiles, nine slabs
]


@pytest.mark.parametrize('case', WIDE192_CASES)
def test_conv_192_column_tiles_match_aten(case):
    """conv3x3_halo_kernel<4, NT = 3>: 256-pixel x 192-channel tiles (64 x 96 per wave) for channel counts that are multiples of 192 but not
    of 256 (ADM: 192 / 384 / 576) -- all columns of the layer in one launch instead of 256- / 128-column tiles plus a 64-column tail
    (ds_conv_tune.variant bit 14 forces the shape at test sizes)."""
    _run_halo_case(case, 16384)


@pytest.mark.parametrize('variant', [6 | 256])
@pytest.mark.parametrize('case', [WIDE_N_CASES[1], WIDE_N_CASES[2], WIDE_N_CASES[3], WIDE_N_CASES[6]])
def test_conv_wide_n_tile_options_match_aten(case, variant):
    """The 256 x 256 tile's default kernel has VAR_LEAN | VAR_NTEPI (scalar-addressed weight DMA, non-temporal epilogue; covered by
    the test above); here the plain kernel (variant bit 8, kept for A/B runs)."""
    _run_halo_case(case, variant)


@pytest.mark.parametrize('variant', [0, 512])
@pytest.mark.parametrize('case', [(5, 8, 64, 64, 128, (64, 0), True, True), (3, 8, 96, 0, 192, (0, 0), True, False)])
def test_conv_multi_image_tiles_coefficient_planes(case, variant):
    """8x8 layers: a 256-pixel tile holds four images (the last tile here: one or three of them, the rest past the batch), each halo
    slot normalises with its OWN image's {mu, A, B}.  Default: the planes of the tile's images staged in LDS; variant bit 9: read from
    global memory per slot (the launcher's choice when they do not fit)."""
    _run_halo_case(case, variant)


HALF_WAVE_CASES = [
    # B, H(=W), c0, c1, cout, (ec0, ec1), norm, act
    (16, 8, 64, 64, 256, (64, 0), True, True),         # 8x8: two images per 128-pixel tile, coefficient planes in LDS, skip-projection slab
    (5, 8, 96, 0, 128, (0, 0), True, False),           # odd image count: the last tile's second image is past the batch
    (2, 16, 64, 32, 256, (0, 0), True, True),          # 16x16, dual source
    (1, 32, 64, 0, 384, (32, 0), False, False),        # 32x32 (4 halo slots), three column tiles
    (1, 64, 32, 0, 128, (0, 0), True, True),           # W = 64: five halo slots per thread
    (1, 8, 256, 256, 256, (256, 256), True, True),     # long K on one row tile: split-K partial tiles through the same epilogue
]


@pytest.mark.parametrize('variant', [0, 2048])
@pytest.mark.parametrize('case', HALF_WAVE_CASES)
def test_conv_half_wave_tiles_match_aten(case, variant):
    """conv3x3_halo_kernel<2, true, 4, 0, 1>: a layer with at most one 128 x 128 tile per CU runs that tile on EIGHT waves of 64 x 32
    (two per SIMD) instead of four of 64 x 64 (kernel id 1284; variant bit 11 switches back to the four-wave kernel, id 128)."""
    _run_halo_case(case, variant, workspace=True, tile=128, want_kid=1284 if variant == 0 else 128)

F16_CASES = [
    # B, H(=W), c0, c1, cout, (ec0, ec1), norm, act
    (1, 16, 64, 0, 128, (0, 0), False, False),
    (2, 16, 128, 64, 128, (0, 0), True, True),          # dual source, 3 slabs
    (1, 32, 64, 0, 256, (0, 0), True, True),
    (2, 32, 64, 64, 192, (128, 64), True, True),        # skip-projection slabs; ragged 192 = one full + one half-empty tile
    (1, 64, 64, 0, 128, (64, 0), True, False),          # W = 64
    (4, 8, 128, 0, 128, (0, 0), False, False),          # 8x8: four images per tile, raw input
    (8, 8, 64, 64, 320, (64, 0), False, False),
    (3, 16, 576, 0, 64, (0, 0), True, True),            # 9 slabs, cout below one tile
]


@pytest.mark.parametrize('case', F16_CASES)
def test_conv_f16_operands_matches_fp16_rounded_reference(case):
    """Reduced-precision mode (ds_conv_args.wgt_f16): fp16 operands on v_mfma_f32_32x32x16_f16, fp32 accumulation.  The reference
    is the SAME arithmetic on the CPU: input normalised/activated in fp32, rounded to fp16; weights rounded to fp16; products summed
    in fp64.  Tolerance 2e-3 of the output scale (a device exp that differs in the last bit can move an operand by one fp16 ulp);
    the distance to the pure fp32 convolution is reported against the fp16 rounding bound 3e-3."""
    import ctypes as C
    from diff_sampler_amd import _lib, ops
    _need_experiments()
    B, H, c0, c1, cout, (ec0, ec1), use_norm, act = case
    lib = _lib.load()
    sup = lib.ds_conv_f16_supported(B, H, H, c0, c1, ec0, ec1)
    assert sup >= (2 if use_norm else 1)
    g = torch.Generator().manual_seed(sum(case[:5]) + 11)
    x = torch.randn(B, c0 + c1, H, H, generator=g)
    e = torch.randn(B, ec0 + ec1, H, H, generator=g) if ec0 else None
    w = torch.randn(cout, c0 + c1, 3, 3, generator=g) / (9 * (c0 + c1)) ** 0.5
    we = torch.randn(cout, ec0 + ec1, 1, 1, generator=g) / (ec0 + ec1) ** 0.5 if ec0 else None
    bias = torch.randn(cout, generator=g)
    cb = torch.randn(B, cout, generator=g)
    res = torch.randn(B, cout, H, H, generator=g)
    mu = torch.randn(B, c0 + c1, generator=g) * 0.3
    ga = 1 + 0.2 * torch.randn(B, c0 + c1, generator=g)
    be = 0.2 * torch.randn(B, c0 + c1, generator=g)
    xin = x
    if use_norm:
        xin = (x - mu[:, :, None, None]) * ga[:, :, None, None] + be[:, :, None, None]
        xin = F.silu(xin) if act else xin
    h16 = lambda t: t.to(torch.float16).to(torch.float64)
    ref16 = F.conv2d(h16(xin), h16(w), padding=1)
    ref32 = F.conv2d(xin, w, padding=1)
    if ec0:
        ref16 = ref16 + F.conv2d(h16(e), h16(we))
        ref32 = ref32 + F.conv2d(e, we)
    tail = (bias[None, :, None, None] + cb[:, :, None, None] + res)
    ref16 = ((ref16 + tail.double()) * 0.7071).float()
    ref32 = (ref32 + tail) * 0.7071
    dev = 'cuda'
    xn = _nhwc(x).to(dev)
    x0 = xn[:, :c0].contiguous()
    x1 = xn[:, c0:].contiguous() if c1 else None
    en = _nhwc(e).to(dev) if ec0 else None
    e0 = en[:, :ec0].contiguous() if ec0 else None
    e1 = en[:, ec0:].contiguous() if ec1 else None
    wp = ops.pack_conv_weight_f16(w.to(dev), we.to(dev) if ec0 else None)
    coefs = torch.stack([mu, ga, be], 1).contiguous().to(dev) if use_norm else None
    old = cout if cout % 4 == 0 else -(-cout // 4) * 4
    out = torch.full((B * H * H, old), float('nan'), device=dev)
    biasd, cbd, resd = bias.to(dev), cb.to(dev), _nhwc(res).to(dev)
    a = _lib.ConvArgs(x0.data_ptr(), x1.data_ptr() if c1 else None, c0, c1, c0, c1, B, H, H, 9, wp.data_ptr(), cout, biasd.data_ptr(),
                      cbd.data_ptr(), cout, B, resd.data_ptr(), cout, 0.7071, 0, out.data_ptr(), old,
                      coefs.data_ptr() if use_norm else None, 1 if act else 0,
                      e0.data_ptr() if ec0 else None, e1.data_ptr() if ec1 else None, ec0, ec1, ec0, ec1)
    a.wgt_f16 = 1
    assert lib.ds_conv_kernel_id(C.byref(a)) == 2562          # conv3x3_halo2_kernel<W, fp16 operands>
    rc = lib.ds_conv2d_nhwc(C.byref(a), _lib.stream_ptr())
    torch.cuda.synchronize()
    assert rc == 0, lib.ds_error_string(rc)
    got = out[:, :cout].cpu()
    assert _rel(got, _nhwc(ref16)) < 2e-3
    assert _rel(got, _nhwc(ref32)) < 3e-3


def test_conv_f16_unsupported_geometry_fails_loudly():
    import ctypes as C
    from diff_sampler_amd import _lib
    lib = _lib.load()
    if not lib.ds_build_experiments() & 1:                                   # default build: no fp32-activation fp16 kernel at all
        assert lib.ds_conv_f16_supported(4, 8, 8, 64, 0, 0, 0) == 0 and lib.ds_conv_f16_supported(1, 16, 16, 64, 64, 64, 0) == 0
        x = torch.zeros(4 * 64, 64, device='cuda'); w = torch.zeros(128, 64 * 9 // 2, device='cuda'); o = torch.zeros(4 * 64, 64, device='cuda')
        a = _lib.ConvArgs(x.data_ptr(), None, 64, 0, 64, 0, 4, 8, 8, 9, w.data_ptr(), 64, None, None, 0, 1, None, 0, 1.0, 0, o.data_ptr(), 64)
        a.wgt_f16 = 1
        assert lib.ds_conv2d_nhwc(C.byref(a), None) == -3                   # DS_E_SHAPE, no silent fp32 fallback
        return
    assert lib.ds_conv_f16_supported(1, 32, 32, 96, 0, 0, 0) == 0          # 96 channels: not a multiple of 64
    assert lib.ds_conv_f16_supported(3, 8, 8, 64, 0, 0, 0) == 0            # 8x8 needs whole tiles of four images
    assert lib.ds_conv_f16_supported(1, 4, 4, 64, 0, 0, 0) == 0
    assert lib.ds_conv_f16_supported(4, 8, 8, 64, 0, 0, 0) == 1 and lib.ds_conv_f16_supported(1, 16, 16, 64, 64, 64, 0) == 2
    x = torch.zeros(3 * 64, 64, device='cuda'); w = torch.zeros(128, 64 * 9 // 2, device='cuda'); o = torch.zeros(3 * 64, 64, device='cuda')
    a = _lib.ConvArgs(x.data_ptr(), None, 64, 0, 64, 0, 3, 8, 8, 9, w.data_ptr(), 64, None, None, 0, 1, None, 0, 1.0, 0, o.data_ptr(), 64)
    a.wgt_f16 = 1
    assert lib.ds_conv2d_nhwc(C.byref(a), None) == -3                       # DS_E_SHAPE, no silent fp32 fallback


SPLIT_CASES = [
    (1, 16, 32, 0, 128, (0, 0), False, False),
    (2, 16, 64, 32, 128, (0, 0), True, True),
    (1, 32, 96, 0, 256, (0, 0), True, True),
    (2, 32, 32, 32, 192, (64, 32), True, True),
    (1, 64, 32, 0, 128, (32, 0), True, False),
    (4, 8, 64, 0, 128, (0, 0), False, False),
    (3, 16, 288, 0, 3, (0, 0), True, True),            # output-conv shape (3 channels), 9 slabs
]


@pytest.mark.parametrize('case', SPLIT_CASES)
def test_conv_split_fp16_emulates_fp32_within_the_fp32_tolerance(case):
    """wgt_f16 == 2: fp32 emulated by split fp16 hi/lo operands (three MFMA products).  Same reference and the SAME tolerance as the
    exact fp32 kernel's test (TOL = 2e-5 of the output scale): this mode is only acceptable if it is indistinguishable from fp32."""
    import ctypes as C
    from diff_sampler_amd import _lib, ops
    B, H, c0, c1, cout, (ec0, ec1), use_norm, act = case
    lib = _lib.load()
    assert lib.ds_conv_split_supported(B, H, H, c0, c1, ec0, ec1) >= (2 if use_norm else 1)
    g = torch.Generator().manual_seed(sum(case[:5]) + 13)
    x = torch.randn(B, c0 + c1, H, H, generator=g)
    e = torch.randn(B, ec0 + ec1, H, H, generator=g) if ec0 else None
    w = torch.randn(cout, c0 + c1, 3, 3, generator=g) / (9 * (c0 + c1)) ** 0.5
    w[0, 0, 0, 0] = 3e-6                                   # a weight 2**-15 of the largest: lands in fp16's subnormal range after scaling
    we = torch.randn(cout, ec0 + ec1, 1, 1, generator=g) / (ec0 + ec1) ** 0.5 if ec0 else None
    bias = torch.randn(cout, generator=g)
    cb = torch.randn(B, cout, generator=g)
    res = torch.randn(B, cout, H, H, generator=g)
    mu = torch.randn(B, c0 + c1, generator=g) * 0.3
    ga = 1 + 0.2 * torch.randn(B, c0 + c1, generator=g)
    be = 0.2 * torch.randn(B, c0 + c1, generator=g)
    xin = x
    if use_norm:
        xin = (x - mu[:, :, None, None]) * ga[:, :, None, None] + be[:, :, None, None]
        xin = F.silu(xin) if act else xin
    ref = F.conv2d(xin.double(), w.double(), padding=1)
    if ec0:
        ref = ref + F.conv2d(e.double(), we.double())
    ref = ((ref + (bias[None, :, None, None] + cb[:, :, None, None] + res).double()) * 0.7071).float()
    dev = 'cuda'
    xn = _nhwc(x).to(dev)
    x0 = xn[:, :c0].contiguous()
    x1 = xn[:, c0:].contiguous() if c1 else None
    en = _nhwc(e).to(dev) if ec0 else None
    e0 = en[:, :ec0].contiguous() if ec0 else None
    e1 = en[:, ec0:].contiguous() if ec1 else None
    wp, shift = ops.pack_conv_weight_split(w.to(dev), we.to(dev) if ec0 else None)
    coefs = torch.stack([mu, ga, be], 1).contiguous().to(dev) if use_norm else None
    old = cout if cout % 4 == 0 else -(-cout // 4) * 4
    out = torch.full((B * H * H, old), float('nan'), device=dev)
    biasd, cbd, resd = bias.to(dev), cb.to(dev), _nhwc(res).to(dev)
    if cout < 4:
        resd = torch.cat([resd, torch.zeros(resd.shape[0], old - cout, device=dev)], 1).contiguous()
    a = _lib.ConvArgs(x0.data_ptr(), x1.data_ptr() if c1 else None, c0, c1, c0, c1, B, H, H, 9, wp.data_ptr(), cout, biasd.data_ptr(),
                      cbd.data_ptr(), cout, B, resd.data_ptr(), old, 0.7071, 0, out.data_ptr(), old,
                      coefs.data_ptr() if use_norm else None, 1 if act else 0,
                      e0.data_ptr() if ec0 else None, e1.data_ptr() if ec1 else None, ec0, ec1, ec0, ec1)
    a.wgt_f16, a.wgt_shift = 2, shift
    assert lib.ds_conv_kernel_id(C.byref(a)) == 2563          # conv3x3_halo2_kernel<W, split fp16 operands>
    rc = lib.ds_conv2d_nhwc(C.byref(a), _lib.stream_ptr())
    torch.cuda.synchronize()
    assert rc == 0, lib.ds_error_string(rc)
    assert _rel(out[:, :cout].cpu(), _nhwc(ref)) < TOL


GEMM_F16_CASES = [
    # rows, c0, c1, cout, flavour
    (256, 64, 0, 128, 'plain'),                # one tile, one tap
    (512, 128, 0, 128, 'bias_res'),            # two taps (both register sets)
    (1024, 192, 0, 320, 'bias_res'),           # odd tap count; ragged 320 = two full + one half-empty column tile
    (768, 128, 64, 192, 'dual'),               # dual source
    (4096, 320, 0, 2560, 'geglu'),             # SD-1.5 ff.proj_geglu shape at 64x64
    (2048, 1280, 0, 320, 'bias_res'),          # K = 20 taps
    (256, 576, 0, 1728, 'bias_res'),           # ImageNet-64 qkv at 8x8 x 4 images
]


@pytest.mark.parametrize('case', GEMM_F16_CASES)
def test_gemm_f16_operands_matches_fp16_rounded_reference(case):
    """1x1 / Linear in the reduced-precision mode (wgt_f16 == 1, taps == 1 -> gemm_f16_kernel): the reference is the same arithmetic
    on the CPU (operands rounded to fp16, products summed in fp64).  Tolerance 1e-5 of the output scale against that (the kernel's
    only other error is the fp32 accumulation order); against the pure fp32 layer the fp16 rounding bound 3e-3."""
    import ctypes as C
    from diff_sampler_amd import _lib, ops
    rows, c0, c1, cout, kind = case
    lib = _lib.load()
    assert lib.ds_gemm_f16_supported(rows, c0, c1) == 1
    k = c0 + c1
    g = torch.Generator().manual_seed(rows + k + cout)
    x = torch.randn(rows, k, generator=g)
    wt = torch.randn(cout, k, generator=g) / k ** 0.5
    bias = torch.randn(cout, generator=g) * 0.3
    h16 = lambda t: t.to(torch.float16).to(torch.float64)
    y16 = h16(x) @ h16(wt).T + bias.double()
    y32 = F.linear(x, wt, bias)
    dev = 'cuda'
    x0 = x[:, :c0].contiguous().to(dev)
    x1 = x[:, c0:].contiguous().to(dev) if c1 else None
    if kind == 'geglu':
        inner = cout // 2
        perm = torch.arange(cout).reshape(-1, 2, 32)
        perm = (perm[:, 0] // 64 * 32 + perm[:, 0] % 32).reshape(-1, 1, 32).repeat(1, 2, 1)
        perm[:, 1] += inner
        perm = perm.reshape(-1)
        wp = ops.pack_linear_weight_f16(ops.pack_linear_weight(wt[perm].to(dev)))
        bd = bias[perm].contiguous().to(dev)
        out = torch.full((rows, inner), float('nan'), device=dev)
        a = _lib.ConvArgs(x0.data_ptr(), None, c0, 0, c0, 0, rows, 1, 1, 1, wp.data_ptr(), cout, bd.data_ptr(), None, 0, 1, None, 0, 1.0,
                          _lib.DS_ACT_GEGLU, out.data_ptr(), inner)
        ref16 = (y16[:, :inner] * F.gelu(y16[:, inner:])).float()
        ref32 = y32[:, :inner] * F.gelu(y32[:, inner:])
        ncol = inner
    else:
        wp = ops.pack_linear_weight_f16(ops.pack_linear_weight(wt.to(dev)))
        bd = bias.to(dev) if kind != 'plain' else None
        res = torch.randn(rows, cout, generator=g)
        resd = res.to(dev) if kind != 'plain' else None
        out = torch.full((rows, cout), float('nan'), device=dev)
        sc = 0.7071 if kind != 'plain' else 1.0
        a = _lib.ConvArgs(x0.data_ptr(), x1.data_ptr() if c1 else None, c0, c1, c0, c1, rows, 1, 1, 1, wp.data_ptr(), cout,
                          bd.data_ptr() if bd is not None else None, None, 0, 1, resd.data_ptr() if resd is not None else None, cout, sc,
                          0, out.data_ptr(), cout)
        if kind == 'plain':
            ref16, ref32 = (y16 - bias.double()).float(), y32 - bias
        else:
            ref16, ref32 = ((y16 + res.double()) * sc).float(), (y32 + res) * sc
        ncol = cout
    a.wgt_f16 = 1
    assert lib.ds_conv_kernel_id(C.byref(a)) == 2564
    rc = lib.ds_conv2d_nhwc(C.byref(a), _lib.stream_ptr())
    torch.cuda.synchronize()
    assert rc == 0, lib.ds_error_string(rc)
    got = out[:, :ncol].cpu()
    assert _rel(got, ref16) < 1e-5
    assert _rel(got, ref32) < 3e-3


def test_gemm_f16_rejects_unsupported_shapes():
    import ctypes as C
    from diff_sampler_amd import _lib
    lib = _lib.load()
    assert lib.ds_gemm_f16_supported(77, 768, 0) == 0          # ragged rows (the text-context projection)
    assert lib.ds_gemm_f16_supported(256, 96, 0) == 0          # K not a multiple of 64
    assert lib.ds_gemm_f16_supported(256, 96, 32) == 0
    x = torch.zeros(77, 64, device='cuda'); w = torch.zeros(128, 32, device='cuda'); out = torch.zeros(77, 128, device='cuda')
    a = _lib.ConvArgs(x.data_ptr(), None, 64, 0, 64, 0, 77, 1, 1, 1, w.data_ptr(), 128, None, None, 0, 1, None, 0, 1.0, 0, out.data_ptr(), 128)
    a.wgt_f16 = 1
    assert lib.ds_conv2d_nhwc(C.byref(a), _lib.stream_ptr()) == -3         # DS_E_SHAPE: no silent fp32 fallback


@pytest.mark.parametrize('d,bz,heads,sq,skv', [(40, 1, 8, 1024, 1024), (40, 2, 8, 300, 77), (64, 3, 3, 256, 256), (64, 2, 6, 64, 64),
                                               (80, 1, 2, 1024, 1000), (160, 2, 2, 256, 256), (32, 1, 2, 128, 65), (96, 1, 1, 40, 200),
                                               (128, 1, 2, 520, 130), (40, 1, 2, 4096, 4096)])
def test_fused_attention_f16_operands(d, bz, heads, sq, skv):
    """ds_attention_f16 (fp16 operands, fp32 scores / statistics / accumulation) against softmax(q k^T * scale) v in fp64:
      * with q * scale, k, v rounded to fp16 first -- the kernel's own operand rounding; what remains is the fp16 rounding of the
        softmax weights and the fp32 accumulation order: 1e-3 of the output scale;
      * against the unrounded fp64 result: the fp16 operand bound, 4e-3 (logits of magnitude ~ 5 move by ~ 5 * 2**-11 each).
    Non-symmetric random inputs, strided q/k/v inside wider rows, ragged lengths, (image, head) counts off the 8-XCD grid multiple."""
    from diff_sampler_amd import ops, _lib
    assert _lib.load().ds_attention_f16_supported(d)
    g = torch.Generator().manual_seed(d * 1000 + sq + skv)
    C_ = heads * d
    qkv = torch.randn(bz, sq, 3 * C_ + 4, generator=g)
    kv = torch.randn(bz, skv, 2 * C_, generator=g) * 1.5
    q = qkv[:, :, :C_]
    k, v = kv[:, :, :C_], kv[:, :, C_:]
    scale = d ** -0.5
    qd, kvd = qkv.cuda().contiguous(), kv.cuda().contiguous()
    out = torch.full((bz, sq, C_), float('nan'), device='cuda')
    ops.attention(qd, kvd, kvd[:, :, C_:], out, batch=bz, heads=heads, sq=sq, skv=skv, d=d, ldq=3 * C_ + 4, ldk=2 * C_, ldv=2 * C_,
                  ldo=C_, q_bs=sq * (3 * C_ + 4), k_bs=skv * 2 * C_, v_bs=skv * 2 * C_, o_bs=sq * C_, scale=scale, f16=True)
    torch.cuda.synchronize()

    def ref(qq, kk, vv, sc):
        qh, kh, vh = (t.reshape(bz, -1, heads, d).double() for t in (qq, kk, vv))
        w = (torch.einsum('bqhd,bkhd->bhqk', qh, kh) * sc).softmax(-1)
        return torch.einsum('bhqk,bkhd->bqhd', w, vh).reshape(bz, sq, C_).float()

    h16 = lambda t: t.to(torch.float16).to(torch.float32)
    log2e = 1.4426950408889634
    ref16 = ref(h16(q * (scale * log2e)), h16(k), h16(v), 1.0 / log2e)
    got = out.cpu()
    assert torch.isfinite(got).all()
    assert _rel(got, ref16) < 1e-3
    assert _rel(got, ref(q, k, v, scale)) < 4e-3


def test_attention_f16_rejects_uncovered_head_sizes():
    import ctypes as C
    from diff_sampler_amd import _lib
    lib = _lib.load()
    assert lib.ds_attention_f16_supported(256) == 0 and lib.ds_attention_f16_supported(8) == 0
    x = torch.zeros(64, 3 * 256, device='cuda'); o = torch.zeros(64, 256, device='cuda')
    a = _lib.AttnArgs(x.data_ptr(), x.data_ptr(), x.data_ptr(), o.data_ptr(), 768, 768, 768, 256, 0, 0, 0, 0, 1, 1, 64, 64, 256, 1.0)
    assert lib.ds_attention_f16(C.byref(a), _lib.stream_ptr()) == -3


F16DMA_CASES = [
    # B, H(=W), cin, cout, ec0, forced nb (0 = cost model), with stats
    (1, 16, 64, 64, 0, 0, False),            # one slab, NB = 1
    (1, 16, 128, 128, 0, 2, True),
    (2, 32, 64, 192, 0, 3, True),            # 192-column tiles (ADM channel counts)
    (1, 32, 192, 256, 64, 4, True),          # 256-column tiles + one appended 1x1 slab
    (1, 32, 128, 320, 128, 0, False),        # 320 = 256 + 64 / 192 + 128: mixed column tiling, two 1x1 slabs
    (1, 64, 192, 192, 192, 3, True),         # W = 64: seven DMA rounds per halo, three 1x1 slabs back to back
    (1, 64, 64, 128, 0, 1, False),
    (4, 8, 128, 192, 0, 3, True),            # 8x8: four images per tile
    (8, 8, 64, 64, 64, 0, False),
    (1, 16, 576, 576, 0, 0, True),           # nine slabs; few pixel tiles -> the cost model narrows the column tiles
    (2, 16, 64, 256, 0, 4, False),
    # 8x8 images of a batch that is not a multiple of four (round 6): the last 256-pixel tile has empty image slots
    (2, 8, 128, 192, 0, 0, True),            # half a tile (the SD-1.5 8x8 stage at one latent: two U-Net images)
    (3, 8, 64, 64, 64, 1, True),
    (5, 8, 64, 128, 0, 2, True),             # one full tile + one image
    (1, 8, 192, 256, 0, 0, False),
]


@pytest.mark.parametrize('nw', [8, 4])
@pytest.mark.parametrize('case', F16DMA_CASES)
def test_conv_f16_activations_dma_kernel(case, nw):
    """ds_conv2d_nhwc with in_f16 (csrc/conv3x3_f16dma.hip): the input is an fp16 NHWC tensor, both operands go to LDS by DMA, column tiles
    of 64 / 128 / 192 / 256 channels.  Reference = the same arithmetic on the CPU (fp16 operands, products summed in fp64): 2e-5 of the
    output scale -- only the fp32 accumulation order differs; the GroupNorm column sums the epilogue leaves are checked too.
    nw = 8: the eight-wave kernel on 256-pixel tiles (kernel id 2566); nw = 4: the four-wave half-slab variant on 128-pixel tiles, two
    workgroups per CU (csrc/conv3x3_f16dmah.hip, kernel id 2569) -- forced per call through ds_conv_tune.f16dma_nw."""
    import ctypes as C
    from diff_sampler_amd import _lib, ops
    B, H, cin, cout, ec0, nb, with_stats = case
    lib = _lib.load()
    if nw == 4:
        _need_experiments()
    assert lib.ds_conv_f16dma_supported(B, H, H, cin, ec0, cout) == 1
    g = torch.Generator().manual_seed(sum(case[:5]) + 5)
    x = torch.randn(B, cin, H, H, generator=g).to(torch.float16)
    e = torch.randn(B, ec0, H, H, generator=g).to(torch.float16) if ec0 else None
    w = torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5
    we = torch.randn(cout, ec0, 1, 1, generator=g) / ec0 ** 0.5 if ec0 else None
    bias = torch.randn(cout, generator=g)
    cb = torch.randn(B, cout, generator=g)
    res = torch.randn(B, cout, H, H, generator=g)
    h16 = lambda t: t.to(torch.float16).to(torch.float64)
    ref = F.conv2d(x.double(), h16(w), padding=1)
    if ec0:
        ref = ref + F.conv2d(e.double(), h16(we))
    ref = ((ref + (bias[None, :, None, None] + cb[:, :, None, None] + res).double()) * 0.7071).float()
    dev = 'cuda'
    xn = x.permute(0, 2, 3, 1).reshape(-1, cin).contiguous().to(dev)
    en = e.permute(0, 2, 3, 1).reshape(-1, ec0).contiguous().to(dev) if ec0 else None
    wp = ops.pack_conv_weight_f16(w.to(dev), we.to(dev) if ec0 else None)
    M = B * H * H
    out = torch.full((M, cout), float('nan'), device=dev)
    stats = torch.full((-(-M // 64) * 2 * cout,), float('nan'), device=dev) if with_stats else None
    biasd, cbd, resd = bias.to(dev), cb.to(dev), _nhwc(res).to(dev)
    a = _lib.ConvArgs(xn.data_ptr(), None, cin, 0, cin, 0, B, H, H, 9, wp.data_ptr(), cout, biasd.data_ptr(), cbd.data_ptr(), cout, B,
                      resd.data_ptr(), cout, 0.7071, 0, out.data_ptr(), cout, None, 0, en.data_ptr() if ec0 else None, None, ec0, 0, ec0, 0)
    a.wgt_f16, a.in_f16 = 1, 1
    if with_stats:
        a.stats_out = stats.data_ptr()
    a.tune.f16dma_nb, a.tune.f16dma_nw = nb, nw
    assert lib.ds_conv_kernel_id(C.byref(a)) == (2566 if nw == 8 else 2569)
    rc = lib.ds_conv2d_nhwc(C.byref(a), _lib.stream_ptr())
    torch.cuda.synchronize()
    assert rc == 0, lib.ds_error_string(rc)
    got = out.cpu()
    assert torch.isfinite(got).all()
    assert _rel(got, _nhwc(ref)) < 2e-5
    if with_stats:
        st = stats.cpu().reshape(-1, 2, cout)
        blocks = got.reshape(-1, 64, cout)
        assert _rel(st[:, 0], blocks.sum(1)) < 1e-5 and _rel(st[:, 1], (blocks * blocks).sum(1)) < 1e-5


F16DMA_SPLIT_CASES = [
    # B, H(=W), cin, cout, ec0, forced nb, tune.splits (0 = the library's choice), fp16 residual + output rows, splits expected (None = any > 1)
    (4, 8, 256, 192, 0, 0, 2, False, 2),             # 36 taps: two splits of two slabs
    (8, 8, 768, 768, 0, 0, 0, False, None),          # ADM 8x8 layer at a small batch: 8 tiles -> the chooser splits (108 taps -> 6)
    (4, 8, 256, 128, 512, 0, 2, False, 2),           # 44 taps: [two 3x3 slabs | two 3x3 slabs + eight 1x1 slabs]
    (4, 8, 128, 192, 2560, 3, 3, True, 3),           # 58 taps: the middle and the last split are 1x1 slabs only (prologue of a one-tap first slab)
    (2, 16, 512, 256, 0, 4, 2, False, 2),            # 256-column tiles through the staged epilogue
    (4, 8, 512, 192, 0, 0, 3, True, 3),              # fp16 residual stream and fp16 rows out of the reduce kernel
    (4, 8, 128, 64, 0, 0, 4, False, 1),              # 18 taps: too short to split -- forced count clamped to 1
    (2, 8, 1280, 1280, 0, 0, 0, True, None),         # round 6: SD-1.5's 8x8 stage at one latent (half a tile of rows), the library's own split
    (6, 8, 256, 192, 0, 0, 2, False, 2),             # one full tile + two images
]


@pytest.mark.parametrize('case', F16DMA_SPLIT_CASES)
def test_conv_f16_activations_split_k(case):
    """Split-K of the fp16-activation convolution (csrc/conv3x3_f16dma.hip + splitk_reduce_f16_kernel): S workgroups per tile contract
    contiguous slab ranges, a reduce launch applies the epilogue.  Same reference and tolerances as the unsplit kernel; the split count the
    launcher takes is read back through the partial planes (plane S - 1 written, plane S untouched)."""
    import ctypes as C
    from diff_sampler_amd import _lib, ops
    B, H, cin, cout, ec0, nb, splits, f16io, expect = case
    lib = _lib.load()
    assert lib.ds_conv_f16dma_supported(B, H, H, cin, ec0, cout) == 1
    g = torch.Generator().manual_seed(sum(case[:7]) + 11)
    x = torch.randn(B, cin, H, H, generator=g).to(torch.float16)
    e = torch.randn(B, ec0, H, H, generator=g).to(torch.float16) if ec0 else None
    w = torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5
    we = torch.randn(cout, ec0, 1, 1, generator=g) / ec0 ** 0.5 if ec0 else None
    bias = torch.randn(cout, generator=g)
    cb = torch.randn(B, cout, generator=g)
    res = torch.randn(B, cout, H, H, generator=g)
    if f16io:
        res = res.to(torch.float16).float()
    h16 = lambda t: t.to(torch.float16).to(torch.float64)
    ref = F.conv2d(x.double(), h16(w), padding=1)
    if ec0:
        ref = ref + F.conv2d(e.double(), h16(we))
    ref = _nhwc(((ref + (bias[None, :, None, None] + cb[:, :, None, None] + res).double()) * 0.7071).float())
    dev = 'cuda'
    xn = x.permute(0, 2, 3, 1).reshape(-1, cin).contiguous().to(dev)
    en = e.permute(0, 2, 3, 1).reshape(-1, ec0).contiguous().to(dev) if ec0 else None
    wp = ops.pack_conv_weight_f16(w.to(dev), we.to(dev) if ec0 else None)
    M = B * H * H
    out = torch.full((M, cout), float('nan'), dtype=torch.float16 if f16io else torch.float32, device=dev)
    stats = torch.full((-(-M // 64) * 2 * cout,), float('nan'), device=dev)
    planes = 17
    ws = torch.full((planes * M * cout,), float('nan'), device=dev)
    biasd, cbd = bias.to(dev), cb.to(dev)
    resd = _nhwc(res).to(dev).to(torch.float16 if f16io else torch.float32).contiguous()
    a = _lib.ConvArgs(xn.data_ptr(), None, cin, 0, cin, 0, B, H, H, 9, wp.data_ptr(), cout, biasd.data_ptr(), cbd.data_ptr(), cout, B,
                      resd.data_ptr(), cout, 0.7071, 0, out.data_ptr(), cout, None, 0, en.data_ptr() if ec0 else None, None, ec0, 0, ec0, 0)
    a.wgt_f16, a.in_f16, a.out_f16, a.res_f16 = 1, 1, int(f16io), int(f16io)
    a.stats_out = stats.data_ptr()
    a.workspace, a.workspace_floats = ws.data_ptr(), ws.numel()
    a.tune.f16dma_nb, a.tune.splits = nb, splits
    rc = lib.ds_conv2d_nhwc(C.byref(a), _lib.stream_ptr())
    torch.cuda.synchronize()
    assert rc == 0, lib.ds_error_string(rc)
    written = torch.isfinite(ws.reshape(planes, -1)).all(dim=1).cpu()
    used = int(written.sum())
    assert bool(written[:used].all()) and not bool(torch.isfinite(ws.reshape(planes, -1)[used:]).any())     # whole planes 0 .. S - 1, nothing else
    S = used if used else 1
    assert (S == expect) if expect is not None else (S > 1), S
    got = out.float().cpu()
    assert torch.isfinite(got).all()
    if f16io:
        assert _rel(got, ref.to(torch.float16).float()) < 1.5e-3
    else:
        assert _rel(got, ref) < 2e-5
    st = stats.cpu().reshape(-1, 2, cout)
    blocks = got.reshape(-1, 64, cout)
    assert _rel(st[:, 0], blocks.sum(1)) < 1e-5 and _rel(st[:, 1], (blocks * blocks).sum(1)) < 1e-5
    # tune.splits = 1 = never: the unsplit kernel, same result class, no partial plane touched
    ws.fill_(float('nan')); out.fill_(float('nan'))
    a.tune.splits = 1
    assert lib.ds_conv2d_nhwc(C.byref(a), _lib.stream_ptr()) == 0
    torch.cuda.synchronize()
    assert not bool(torch.isfinite(ws).any())
    one = out.float().cpu()
    assert _rel(one, got) < (1.5e-3 if f16io else 2e-5)


def test_conv_f16_activations_rejects_what_it_does_not_cover():
    import ctypes as C
    from diff_sampler_amd import _lib
    lib = _lib.load()
    assert lib.ds_conv_f16dma_supported(1, 16, 16, 96, 0, 64) == 0          # channels not a multiple of 64
    assert lib.ds_conv_f16dma_supported(1, 16, 16, 64, 0, 96) == 0
    assert lib.ds_conv_f16dma_supported(3, 8, 8, 64, 0, 64) == 1            # (round 6: 8x8 images need no whole tile of four any more)
    assert lib.ds_conv_f16dma_supported(3, 16, 8, 64, 0, 64) == 0           # non-square
    assert lib.ds_conv_f16dma_supported(1, 4, 4, 64, 0, 64) == 0
    x = torch.zeros(256, 64, dtype=torch.float16, device='cuda')
    w = torch.zeros(128, 9 * 64 // 2, device='cuda')
    out = torch.zeros(256, 64, device='cuda')
    a = _lib.ConvArgs(x.data_ptr(), None, 64, 0, 64, 0, 1, 16, 16, 9, w.data_ptr(), 64, None, None, 0, 1, None, 0, 1.0, 0, out.data_ptr(), 64)
    a.in_f16 = 1                                                             # fp16 input without fp16 weights
    assert lib.ds_conv2d_nhwc(C.byref(a), _lib.stream_ptr()) == -1
    a.wgt_f16, a.ld0 = 1, 68                                                 # leading dimension not a multiple of 8 halfs
    assert lib.ds_conv2d_nhwc(C.byref(a), _lib.stream_ptr()) == -2


def test_conv_f16_output_rows_and_norm_pass_on_fp16_tensors():
    """The fp16-mode chain of one block: ds_norm_act(out_f16 [+ raw copy]) -> ds_conv2d_nhwc(in_f16, out_f16) -> ds_norm_act(in_f16, out_f16).
    fp16 tensors must equal the fp32 results rounded to nearest even (the conv output within the accumulation-order tolerance)."""
    import ctypes as C
    from diff_sampler_amd import _lib, ops
    lib = _lib.load()
    dev = 'cuda'
    B, H, cin, cout = 2, 16, 128, 192
    M = B * H * H
    g = torch.Generator().manual_seed(77)
    x = torch.randn(M, cin, generator=g).to(dev)
    mean, rstd = (torch.randn(B * 32, generator=g) * 0.1).to(dev), (1 + 0.1 * torch.rand(B * 32, generator=g)).to(dev)
    gamma, beta = (1 + 0.1 * torch.randn(cin, generator=g)).to(dev), (0.1 * torch.randn(cin, generator=g)).to(dev)
    a16 = torch.zeros(M, cin, dtype=torch.float16, device=dev)
    r16 = torch.zeros(M, cin, dtype=torch.float16, device=dev)
    na = ops._norm_args(x, cin, cin, B, H, H, groups=32, eps=1e-5, mean=mean, rstd=rstd, gamma=gamma, beta=beta, act=1, out=a16, out_ld=cin)
    na.out = C.c_void_p(a16.data_ptr())
    na.out_f16, na.raw_out, na.raw_ld = 1, C.c_void_p(r16.data_ptr()), cin
    assert lib.ds_norm_act(C.byref(na), _lib.stream_ptr()) == 0
    xr = x.reshape(B, H * H, 32, cin // 32)
    y = (xr - mean.reshape(B, 1, 32, 1)) * rstd.reshape(B, 1, 32, 1)
    y = F.silu(y.reshape(M, cin) * gamma + beta)
    torch.cuda.synchronize()
    assert torch.equal(r16, x.to(torch.float16))
    assert _rel(a16.float().cpu(), y.to(torch.float16).float().cpu()) < 1e-3       # device exp differs by an ulp here and there
    # conv on the fp16 tensor, fp16 output rows
    w = torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5
    wp = ops.pack_conv_weight_f16(w.to(dev))
    bias = torch.randn(cout, generator=g).to(dev)
    h16 = torch.zeros(M, cout, dtype=torch.float16, device=dev)
    stats = torch.zeros(-(-M // 64) * 2 * cout, device=dev)
    ca = _lib.ConvArgs(a16.data_ptr(), None, cin, 0, cin, 0, B, H, H, 9, wp.data_ptr(), cout, bias.data_ptr(), None, 0, 1, None, 0, 1.0, 0,
                       h16.data_ptr(), cout)
    ca.wgt_f16, ca.in_f16, ca.out_f16, ca.stats_out = 1, 1, 1, stats.data_ptr()
    assert lib.ds_conv2d_nhwc(C.byref(ca), _lib.stream_ptr()) == 0
    torch.cuda.synchronize()
    xin = a16.double().cpu().reshape(B, H, H, cin).permute(0, 3, 1, 2)
    ref = F.conv2d(xin, w.to(torch.float16).double(), padding=1) + bias.double().cpu()[None, :, None, None]
    ref = ref.permute(0, 2, 3, 1).reshape(M, cout).float()
    assert _rel(h16.float().cpu(), ref.to(torch.float16).float()) < 1.5e-3          # one fp16 ulp where the fp32 sums straddle a rounding boundary
    st = stats.cpu().reshape(-1, 2, cout)
    stored = h16.float().cpu().reshape(-1, 64, cout)                               # column sums / sums of squares of the STORED (rounded) tensor
    assert _rel(st[:, 0], stored.sum(1)) < 1e-5 and _rel(st[:, 1], (stored * stored).sum(1)) < 1e-5
    # second norm pass reads the fp16 tensor
    b16 = torch.zeros(M, cout, dtype=torch.float16, device=dev)
    nb = ops._norm_args(x, cout, cout, B, H, H, groups=32, eps=1e-5, mean=mean, rstd=rstd, act=1, out=b16, out_ld=cout)
    nb.x0, nb.out = C.c_void_p(h16.data_ptr()), C.c_void_p(b16.data_ptr())
    nb.in_f16, nb.out_f16 = 1, 1
    assert lib.ds_norm_act(C.byref(nb), _lib.stream_ptr()) == 0
    torch.cuda.synchronize()
    hr = h16.float().reshape(B, H * H, 32, cout // 32)
    y2 = F.silu(((hr - mean.reshape(B, 1, 32, 1)) * rstd.reshape(B, 1, 32, 1)).reshape(M, cout))
    assert _rel(b16.float().cpu(), y2.to(torch.float16).float().cpu()) < 1e-3


GEMM_F16DMA_CASES = [
    # rows, k, cout, forced nb, epilogue ('plain' | 'res' | 'geglu' | 'f16out' | 'stats')
    (256, 64, 64, 0, 'plain'),
    (1000, 320, 960, 0, 'plain'),            # ragged row count, 960 = 3 x 256 + 192 / 5 x 192
    (512, 320, 320, 3, 'res'),
    (4096, 640, 640, 4, 'stats'),
    (300, 1280, 1280, 2, 'res'),
    (2048, 320, 2560, 4, 'geglu'),           # SD-1.5 ff.net.0.proj with the gate in the epilogue
    (1024, 320, 2560, 2, 'geglu_f16out'),
    (777, 384, 1152, 0, 'f16out'),
    (128, 768, 768, 1, 'res'),
]


@pytest.mark.parametrize('case', GEMM_F16DMA_CASES)
def test_gemm_f16_activations_dma_kernel(case):
    """1x1 / Linear on an fp16 tensor (ds_conv_args.in_f16 with taps == 1, csrc/gemm_f16dma.hip) against the same arithmetic on the CPU
    (fp16 operands, fp64 sums): 2e-5 of the output scale; fp16 output rows equal the rounded fp32 result up to one fp16 ulp."""
    import ctypes as C
    from diff_sampler_amd import _lib, ops
    rows, k, cout, nb, mode = case
    lib = _lib.load()
    assert lib.ds_gemm_f16dma_supported(rows, k, cout) == 1
    g = torch.Generator().manual_seed(rows + k + cout)
    x = torch.randn(rows, k, generator=g).to(torch.float16)
    wt = torch.randn(cout, k, generator=g) / k ** 0.5
    bias = torch.randn(cout, generator=g)
    res = torch.randn(rows, cout, generator=g)
    dev = 'cuda'
    geglu, f16out = mode.startswith('geglu'), mode.endswith('f16out')
    h16 = lambda t: t.to(torch.float16).to(torch.float64)
    y = x.double() @ h16(wt).t() + bias.double()
    if geglu:
        inner = cout // 2
        val = torch.arange(inner).reshape(-1, 32)
        perm = torch.stack([val, val + inner], 1).reshape(-1)                              # [32 values | their 32 gates] per 64-row block
        wp = ops.pack_linear_weight_f16(ops.pack_linear_weight(wt[perm].to(dev)))
        bp = bias[perm].contiguous()
        ref = (y[:, :inner] * F.gelu(y[:, inner:])).float()
        ocols = inner
    else:
        wp = ops.pack_linear_weight_f16(ops.pack_linear_weight(wt.to(dev)))
        bp = bias
        ref = ((y + res.double()) * 0.7071).float() if mode == 'res' else y.float()
        ocols = cout
    out = torch.full((rows, ocols), float('nan'), dtype=torch.float16 if f16out else torch.float32, device=dev)
    xd, bd_, resd = x.to(dev), bp.to(dev), res.to(dev)
    stats = torch.full((-(-rows // 64) * 2 * cout,), float('nan'), device=dev) if mode == 'stats' else None
    a = _lib.ConvArgs(xd.data_ptr(), None, k, 0, k, 0, rows, 1, 1, 1, wp.data_ptr(), cout, bd_.data_ptr(), None, 0, 1,
                      resd.data_ptr() if mode == 'res' else None, cout, 0.7071 if mode == 'res' else 1.0, 2 if geglu else 0, out.data_ptr(), ocols)
    a.wgt_f16, a.in_f16, a.out_f16 = 1, 1, int(f16out)
    if stats is not None:
        a.stats_out = stats.data_ptr()
    assert lib.ds_conv_kernel_id(C.byref(a)) == 2567
    a.tune.f16dma_nb = nb
    rc = lib.ds_conv2d_nhwc(C.byref(a), _lib.stream_ptr())
    torch.cuda.synchronize()
    assert rc == 0, lib.ds_error_string(rc)
    got = out.float().cpu()
    assert torch.isfinite(got).all()
    if f16out:
        assert _rel(got, ref.to(torch.float16).float()) < 1.5e-3
    else:
        assert _rel(got, ref) < 2e-5
    if stats is not None and rows % 64 == 0:
        st = stats.cpu().reshape(-1, 2, cout)
        assert _rel(st[:, 0], got.reshape(-1, 64, cout).sum(1)) < 1e-5


CONV_F16_STRIDE2_CASES = [
    # images, OUTPUT side, cin, cout, forced nb, epilogue ('f16out' | 'f32out' | 'staged')
    (1, 8, 64, 64, 0, 'f16out'),             # 64 output rows: less than one row tile (tail rows read the zero page)
    (3, 8, 128, 192, 0, 'f16out'),           # 192 rows, 192 columns (NB = 3)
    (2, 16, 320, 320, 0, 'f16out'),          # SD-1.5 input_blocks.3 at a small batch: 320 = 256 + 64 columns
    (5, 16, 64, 128, 1, 'f32out'),           # 1 280 rows = five row tiles, fp32 output rows (staged epilogue)
    (2, 32, 64, 256, 4, 'staged'),           # NB = 4, fp16 rows through the staged epilogue (tune.ablate bit 12)
    (4, 4, 640, 640, 2, 'f16out'),           # 4x4 output images: every output pixel touches a border
]


@pytest.mark.parametrize('case', CONV_F16_STRIDE2_CASES)
def test_conv3x3_stride2_f16_activations_gather_kernel(case):
    """The latent-diffusion Downsample (3x3, stride 2, pad 1, openaimodel.py:146-148) on fp16 rows: ds_conv_args.in_f16 with taps == 9,
    stride == 2 = the fp16-activation GEMM with a gathered A tile (csrc/gemm_f16dma.hip, GATHER) against F.conv2d on the same fp16 operands in
    fp64: 2e-5 of the output scale for fp32 rows, one fp16 ulp for fp16 rows; the GroupNorm column sums are those of the stored tensor."""
    import ctypes as C
    from diff_sampler_amd import _lib, ops
    n, side, cin, cout, nb, mode = case
    lib = _lib.load()
    assert lib.ds_conv_f16dma_stride2_supported(n, side, side, cin, cout) == 1
    g = torch.Generator().manual_seed(n * 1000 + side * 100 + cin + cout)
    x = torch.randn(n, cin, 2 * side, 2 * side, generator=g).to(torch.float16)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5
    bias = torch.randn(cout, generator=g)
    ref = F.conv2d(x.double(), wt.to(torch.float16).double(), bias.double(), stride=2, padding=1)         # [n, cout, side, side]
    ref = ref.permute(0, 2, 3, 1).reshape(n * side * side, cout).float()
    dev = 'cuda'
    xd = x.permute(0, 2, 3, 1).contiguous().reshape(-1, cin).to(dev)                                      # NHWC fp16 rows of the INPUT
    wp = ops.pack_conv_weight_f16(wt.to(dev))
    bd_ = bias.to(dev)
    M = n * side * side
    f16out = mode != 'f32out'
    out = torch.full((M, cout), float('nan'), dtype=torch.float16 if f16out else torch.float32, device=dev)
    stats = torch.full((-(-M // 64) * 2 * cout,), float('nan'), device=dev)
    a = _lib.ConvArgs(xd.data_ptr(), None, cin, 0, cin, 0, n, side, side, 9, wp.data_ptr(), cout, bd_.data_ptr(), None, 0, 1,
                      None, 0, 1.0, 0, out.data_ptr(), cout)
    a.stride = 2
    a.wgt_f16, a.in_f16, a.out_f16 = 1, 1, int(f16out)
    a.stats_out = stats.data_ptr()
    assert lib.ds_conv_kernel_id(C.byref(a)) == 2571
    a.tune.f16dma_nb = nb
    if mode == 'staged':
        a.tune.ablate = 4096
    rc = lib.ds_conv2d_nhwc(C.byref(a), _lib.stream_ptr())
    torch.cuda.synchronize()
    assert rc == 0, lib.ds_error_string(rc)
    got = out.float().cpu()
    assert torch.isfinite(got).all()
    if f16out:
        assert _rel(got, ref.to(torch.float16).float()) < 1.5e-3
    else:
        assert _rel(got, ref) < 2e-5
    if M % 64 == 0:
        st = stats.cpu().reshape(-1, 2, cout)
        blocks = got.reshape(-1, 64, cout)
        assert _rel(st[:, 0], blocks.sum(1)) < 1e-5 and _rel(st[:, 1], (blocks * blocks).sum(1)) < 1e-5
    # a residual operand is not part of this path: refused, not ignored
    a.res, a.res_ld = out.data_ptr(), cout
    assert lib.ds_conv2d_nhwc(C.byref(a), _lib.stream_ptr()) != 0


def test_layernorm_and_attention_fp16_outputs():
    """ds_layernorm_rows_f16 and ds_attention_f16 with out_f16: the fp32 results rounded to nearest even."""
    import ctypes as C
    from diff_sampler_amd import _lib, ops
    lib = _lib.load()
    dev = 'cuda'
    g = torch.Generator().manual_seed(4)
    x = torch.randn(300, 320, generator=g).to(dev)
    gm, bt = (1 + 0.1 * torch.randn(320, generator=g)).to(dev), (0.1 * torch.randn(320, generator=g)).to(dev)
    y32, y16 = torch.empty(300, 320, device=dev), torch.empty(300, 320, dtype=torch.float16, device=dev)
    ops.layernorm_rows(x, 320, gm, bt, 1e-5, y32, 320, 300, 320)
    rc = lib.ds_layernorm_rows_f16(x.data_ptr(), 320, gm.data_ptr(), bt.data_ptr(), 1e-5, y16.data_ptr(), 320, 300, 320, _lib.stream_ptr())
    torch.cuda.synchronize()
    assert rc == 0 and torch.equal(y16, y32.to(torch.float16))
    B, S, hd, d = 2, 256, 4, 40
    c = hd * d
    qkv = torch.randn(B * S, 3 * c, generator=g).to(dev)
    o32, o16 = torch.empty(B * S, c, device=dev), torch.empty(B * S, c, dtype=torch.float16, device=dev)
    kw = dict(batch=B, heads=hd, sq=S, skv=S, d=d, ldq=3 * c, ldk=3 * c, ldv=3 * c, ldo=c, q_bs=S * 3 * c, k_bs=S * 3 * c, v_bs=S * 3 * c,
              o_bs=S * c, scale=d ** -0.5)
    ops.attention(qkv, qkv[:, c:], qkv[:, 2 * c:], o32, f16=True, **kw)
    a = _lib.AttnArgs(qkv.data_ptr(), qkv[:, c:].data_ptr(), qkv[:, 2 * c:].data_ptr(), o16.data_ptr(), 3 * c, 3 * c, 3 * c, c, S * 3 * c, S * 3 * c,
                      S * 3 * c, S * c, B, hd, S, S, d, d ** -0.5)
    a.out_f16 = 1
    assert lib.ds_attention_f16(C.byref(a), _lib.stream_ptr()) == 0
    torch.cuda.synchronize()
    assert torch.equal(o16, o32.to(torch.float16))


@pytest.mark.parametrize('resample', [0, 1, 2])
def test_norm_passes_read_fp16_sources_like_their_fp32_copies(resample):
    """ds_gn_stats / ds_norm_act with in_f16 (bit 0: x0, bit 1: x1 -- tensors of the fp16 residual stream, ds_engine.h): values are widened
    before any arithmetic, so the results equal those of the same call on fp32 copies of the tensors, bit for bit; the decoder's mixed
    concatenation [fp16 stream | fp32 stem output] included."""
    import ctypes as C
    from diff_sampler_amd import _lib, ops
    lib = _lib.load()
    dev = 'cuda'
    B, H, c0, c1 = 4, 16, 128, 64
    g = torch.Generator().manual_seed(20 + resample)
    x0 = torch.randn(B * H * H, c0, generator=g).to(dev).to(torch.float16)
    x1h = torch.randn(B * H * H, c1, generator=g).to(dev).to(torch.float16)
    gm, bt = (1 + 0.1 * torch.randn(c0 + c1, generator=g)).to(dev), (0.1 * torch.randn(c0 + c1, generator=g)).to(dev)
    OH = {0: H, 1: H // 2, 2: H * 2}[resample]             # DS_RESAMPLE_NONE / DOWN / UP
    for mask, x1 in ((3, x1h), (1, x1h.float())):          # both fp16; fp16 stream + fp32 second source
        res = []
        for use16 in (True, False):
            a0 = x0 if use16 else x0.float()
            a1 = x1 if use16 else x1.float()
            mean, rstd = torch.empty(B * 32, device=dev), torch.empty(B * 32, device=dev)
            st = ops._norm_args(a0, c0, c0, B, H, H, x1=a1, c1=c1, ld1=c1, groups=32, eps=1e-5, mean=mean, rstd=rstd)
            st.in_f16 = mask if use16 else 0
            assert lib.ds_gn_stats(C.byref(st), _lib.stream_ptr()) == 0
            out = torch.empty(B * OH * OH, c0 + c1, device=dev)
            ap = ops._norm_args(a0, c0, c0, B, H, H, x1=a1, c1=c1, ld1=c1, groups=32, eps=1e-5, mean=mean, rstd=rstd, gamma=gm, beta=bt, act=1,
                                resample=resample, out=out, out_ld=c0 + c1)
            ap.in_f16 = mask if use16 else 0
            assert lib.ds_norm_act(C.byref(ap), _lib.stream_ptr()) == 0
            torch.cuda.synchronize()
            res.append((mean.clone(), rstd.clone(), out))
        for t16, t32 in zip(*res):
            assert torch.equal(t16, t32)
    bad = ops._norm_args(x0, c0, c0, B, H, H, groups=32, mean=mean, rstd=rstd)
    bad.in_f16 = 2                                          # bit 1 without a second source
    assert lib.ds_gn_stats(C.byref(bad), _lib.stream_ptr()) != 0


NORM16_CASES = [
    # B, H, c0, c1, silu, raw copy, adaptive scale / shift (ADM), rows of scale / shift
    (8, 8, 768, 0, True, False, True, 8),            # ImageNet-64 8x8 stage: one row block per image, adaptive scale / shift per image
    (4, 16, 576, 0, True, False, True, 4),           # 16x16: four row blocks
    (3, 32, 384, 0, True, False, False, 1),          # 32x32: sixteen row blocks (the largest image the pass finalises itself)
    (2, 16, 320, 320, True, True, False, 1),         # SD-1.5 decoder concatenation 320 | 320 with the raw copy for the skip projection
    (2, 8, 1280, 1280, True, True, False, 1),        # 2 560 channels: 320 octets -> 512-thread workgroups
    (5, 16, 192, 64, False, False, False, 1),        # affine only, ragged octet count (32 octets, 8 pixel lanes)
    (2, 64, 192, 0, True, False, False, 1),          # 64x64: 16-byte kernel, but the statistics stay a launch of their own (not folded)
]


@pytest.mark.parametrize('case', NORM16_CASES)
def test_norm_pass_16_byte_kernel_and_folded_finalize_equal_the_two_launch_form(case):
    """Round 6 (csrc/norm_act.hip): norm_act16_kernel -- the fp16 pass at 16 bytes per lane -- against norm_act_kernel (8 bytes per lane,
    ds_norm_args.tune_variant = 1) on the same {mu, A, B} planes, and norm_act16_kernel<FIN> -- the pass that computes the GroupNorm
    statistics itself from the producers' per-(64-row block, channel) sums (ds_norm_args.stats0 / stats1) -- against ds_gn_finalize + pass:
    EQUAL bits, output rows and raw copy, incl. the ADM adaptive scale / shift; plus an fp64 reference of the whole normalisation."""
    import ctypes as C
    from diff_sampler_amd import _lib, ops
    B, H, c0, c1, silu, raw, adaptive, ss_rows = case
    lib = _lib.load()
    dev = 'cuda'
    Ct, HW, M = c0 + c1, H * H, B * H * H
    g = torch.Generator().manual_seed(sum(case[:4]))
    x = (torch.randn(M, Ct, generator=g) * 1.5 + 0.3).to(torch.float16).to(dev)
    x0 = x[:, :c0].contiguous()
    x1 = x[:, c0:].contiguous() if c1 else None
    gm, bt = (1 + 0.1 * torch.randn(Ct, generator=g)).to(dev), (0.1 * torch.randn(Ct, generator=g)).to(dev)
    sc = (0.2 * torch.randn(ss_rows, Ct, generator=g)).to(dev) if adaptive else None
    sh = (0.2 * torch.randn(ss_rows, Ct, generator=g)).to(dev) if adaptive else None
    # the producers' column sums: per 64-row block and channel {sum, sum of squares} of the stored (fp16) values, fp32 -- what a convolution's
    # epilogue leaves behind (ds_conv_args.stats_out), one tensor per source
    def colsums(t, c):
        v = t.float().reshape(M // 64, 64, c)
        return torch.stack([v.sum(1), (v * v).sum(1)], 1).contiguous()          # [M / 64][2][c]
    s0 = colsums(x0, c0)
    s1 = colsums(x1, c1) if c1 else None
    mean, rstd = torch.empty(B * 32, device=dev), torch.empty(B * 32, device=dev)
    planes = torch.empty(B, 3, Ct, device=dev)
    f = _lib.GnFinalizeArgs(s0.data_ptr(), s1.data_ptr() if c1 else None, c0, c1, B, HW, 32, 1e-5, gm.data_ptr(), bt.data_ptr(),
                            sc.data_ptr() if adaptive else None, sh.data_ptr() if adaptive else None, Ct, ss_rows, mean.data_ptr(), rstd.data_ptr(),
                            planes.data_ptr())
    assert lib.ds_gn_finalize(C.byref(f), _lib.stream_ptr()) == 0

    def run(variant, fin):
        out = torch.full((M, Ct), float('nan'), dtype=torch.float16, device=dev)
        rw = torch.full((M, Ct), float('nan'), dtype=torch.float16, device=dev) if raw else None
        a = ops._norm_args(x0, c0, c0, B, H, H, x1=x1, c1=c1, ld1=c1, groups=32, eps=1e-5, act=(1 if silu else 0), out=out, out_ld=Ct)
        a.in_f16, a.out_f16 = (3 if c1 else 1), 1
        if raw:
            a.raw_out, a.raw_ld = C.c_void_p(rw.data_ptr()), Ct
        a.tune_variant = variant
        if fin:
            a.stats0, a.stats1 = C.c_void_p(s0.data_ptr()), (C.c_void_p(s1.data_ptr()) if c1 else None)
            a.gamma, a.beta = C.c_void_p(gm.data_ptr()), C.c_void_p(bt.data_ptr())
            if adaptive:
                a.scale, a.shift, a.ss_ld, a.ss_rows = C.c_void_p(sc.data_ptr()), C.c_void_p(sh.data_ptr()), Ct, ss_rows
        else:
            a.coefs = C.c_void_p(planes.data_ptr())
        rc = lib.ds_norm_act(C.byref(a), _lib.stream_ptr())
        torch.cuda.synchronize()
        return rc, out, rw

    rc8, o8, r8 = run(1, False)
    rc16, o16, r16 = run(0, False)
    assert rc8 == 0 and rc16 == 0
    assert torch.isfinite(o16.float()).all() and torch.equal(o16, o8)
    if raw:
        assert torch.equal(r16, r8) and torch.equal(r16, x)
    # the mean / rstd form (no planes: the attention blocks' norm2, the SpatialTransformer's GroupNorm): coefficients formed in the prologue
    def run_mr(variant):
        out = torch.full((M, Ct), float('nan'), dtype=torch.float16, device=dev)
        a = ops._norm_args(x0, c0, c0, B, H, H, x1=x1, c1=c1, ld1=c1, groups=32, eps=1e-5, mean=mean, rstd=rstd, gamma=gm, beta=bt,
                           scale=sc, shift=sh, ss_ld=Ct, ss_rows=ss_rows, act=(1 if silu else 0), out=out, out_ld=Ct)
        a.in_f16, a.out_f16, a.tune_variant = (3 if c1 else 1), 1, variant
        rc = lib.ds_norm_act(C.byref(a), _lib.stream_ptr())
        torch.cuda.synchronize()
        assert rc == 0
        return out
    m8, m16 = run_mr(1), run_mr(0)
    assert torch.equal(m16, m8) and torch.equal(m16, o8)            # ... and they are the planes' coefficients (gn_coefs in all three places)
    rcf, of, rf = run(0, True)
    if HW <= 1024:
        assert rcf == 0, lib.ds_error_string(rcf)
        assert torch.equal(of, o8)
        if raw:
            assert torch.equal(rf, x)
    else:
        assert rcf != 0                                     # images above 32 x 32: the statistics stay a launch of their own
    # fp64 reference: group statistics of the stored values, affine, SiLU, RNE to fp16
    xd = x.double().cpu().reshape(B, HW, 32, Ct // 32)
    mu_ = xd.mean((1, 3), keepdim=True)
    var = (xd * xd).mean((1, 3), keepdim=True) - mu_ * mu_
    y = ((xd - mu_) / torch.sqrt(var + 1e-5)).reshape(B, HW, Ct)
    if adaptive:
        scd, shd = sc.double().cpu(), sh.double().cpu()
        scd = scd.expand(B, Ct) if ss_rows == 1 else scd
        shd = shd.expand(B, Ct) if ss_rows == 1 else shd
        y = y * (gm.double().cpu()[None, None] * (1 + scd[:, None])) + (bt.double().cpu()[None, None] * (1 + scd[:, None]) + shd[:, None])
    else:
        y = y * gm.double().cpu()[None, None] + bt.double().cpu()[None, None]
    y = F.silu(y) if silu else y
    assert _rel(o16.float().cpu().reshape(B, HW, Ct), y.float()) < 1.5e-3


@pytest.mark.parametrize('resample,B,H,c0,c1,raw,planes', [(1, 4, 32, 192, 0, True, True), (2, 3, 16, 384, 0, True, True), (1, 2, 64, 192, 0, True, True),
                                                          (2, 2, 8, 768, 0, False, False), (1, 2, 16, 320, 0, True, False), (2, 2, 16, 128, 64, True, True)])
def test_norm_pass_16_byte_kernel_resamples_like_the_8_byte_kernel(resample, B, H, c0, c1, raw, planes):
    """norm_act16_kernel<., RS>: the 2x2 box filter down (four activated pixels averaged in fp32, rounded once; the raw copy filtered the same
    way) and nearest-neighbour x2 up of the resampling blocks (networks_edm.py:74-77) at 16 bytes per lane, against norm_act_kernel
    (tune_variant = 1): EQUAL bits, output and raw copy, planes or mean / rstd coefficients, one or two sources."""
    import ctypes as C
    from diff_sampler_amd import _lib, ops
    lib = _lib.load()
    dev = 'cuda'
    Ct, M = c0 + c1, B * H * H
    OH = H // 2 if resample == 1 else H * 2
    g = torch.Generator().manual_seed(resample * 100 + H + Ct)
    x = (torch.randn(M, Ct, generator=g) * 1.5 + 0.3).to(torch.float16).to(dev)
    x0 = x[:, :c0].contiguous()
    x1 = x[:, c0:].contiguous() if c1 else None
    gm, bt = (1 + 0.1 * torch.randn(Ct, generator=g)).to(dev), (0.1 * torch.randn(Ct, generator=g)).to(dev)
    mean, rstd = (0.2 * torch.randn(B * 32, generator=g)).to(dev), (1 + 0.2 * torch.rand(B * 32, generator=g)).to(dev)
    pl = torch.stack([0.3 * torch.randn(B, Ct, generator=g), 1 + 0.2 * torch.randn(B, Ct, generator=g), 0.2 * torch.randn(B, Ct, generator=g)], 1).contiguous().to(dev)
    outs = []
    for variant in (1, 0):
        out = torch.full((B * OH * OH, Ct), float('nan'), dtype=torch.float16, device=dev)
        rw = torch.full((B * OH * OH, Ct), float('nan'), dtype=torch.float16, device=dev) if raw else None
        if planes:
            a = ops._norm_args(x0, c0, c0, B, H, H, x1=x1, c1=c1, ld1=c1, groups=32, eps=1e-5, act=1, resample=resample, out=out, out_ld=Ct)
            a.coefs = C.c_void_p(pl.data_ptr())
        else:
            a = ops._norm_args(x0, c0, c0, B, H, H, x1=x1, c1=c1, ld1=c1, groups=32, eps=1e-5, mean=mean, rstd=rstd, gamma=gm, beta=bt, act=1,
                               resample=resample, out=out, out_ld=Ct)
        a.in_f16, a.out_f16, a.tune_variant = (3 if c1 else 1), 1, variant
        if raw:
            a.raw_out, a.raw_ld = C.c_void_p(rw.data_ptr()), Ct
        assert lib.ds_norm_act(C.byref(a), _lib.stream_ptr()) == 0
        torch.cuda.synchronize()
        outs.append((out, rw))
    assert torch.isfinite(outs[1][0].float()).all()
    assert torch.equal(outs[1][0], outs[0][0])
    if raw:
        assert torch.equal(outs[1][1], outs[0][1])
        xr = x.float().reshape(B, H, H, Ct)
        want = xr.reshape(B, H // 2, 2, H // 2, 2, Ct).permute(0, 1, 3, 2, 4, 5) if resample == 1 else None
        if resample == 2:
            assert torch.equal(outs[1][1].reshape(B, OH, OH, Ct), xr.repeat_interleave(2, 1).repeat_interleave(2, 2).to(torch.float16))
        else:
            ref = ((want[..., 0, 0, :] + want[..., 0, 1, :]) + (want[..., 1, 0, :] + want[..., 1, 1, :])) * 0.25
            assert torch.equal(outs[1][1].reshape(B, OH, OH, Ct), ref.to(torch.float16))


def test_layernorm_rows_on_fp16_rows():
    """ds_layernorm_rows_f16io (fp16 in, fp16 out: LayerNorm of a tensor of the fp16 residual stream) == ds_layernorm_rows_f16 on the widened
    rows, for every lanes-per-row variant of the kernel (320 / 640 / 1280 columns) and a ragged row count."""
    from diff_sampler_amd import _lib
    lib = _lib.load()
    dev = 'cuda'
    for rows, cols in ((1001, 320), (515, 640), (130, 1280), (7, 2048)):
        g = torch.Generator().manual_seed(cols)
        x16 = (3 * torch.randn(rows, cols, generator=g)).to(dev).to(torch.float16)
        gm, bt = (1 + 0.1 * torch.randn(cols, generator=g)).to(dev), (0.1 * torch.randn(cols, generator=g)).to(dev)
        ya, yb = torch.empty(rows, cols, dtype=torch.float16, device=dev), torch.empty(rows, cols, dtype=torch.float16, device=dev)
        x32 = x16.float()
        assert lib.ds_layernorm_rows_f16io(x16.data_ptr(), cols, gm.data_ptr(), bt.data_ptr(), 1e-5, ya.data_ptr(), cols, rows, cols, _lib.stream_ptr()) == 0
        assert lib.ds_layernorm_rows_f16(x32.data_ptr(), cols, gm.data_ptr(), bt.data_ptr(), 1e-5, yb.data_ptr(), cols, rows, cols, _lib.stream_ptr()) == 0
        torch.cuda.synchronize()
        assert torch.equal(ya, yb)
        ref = F.layer_norm(x32.double(), (cols,), gm.double(), bt.double(), 1e-5).float()
        assert _rel(ya.float(), ref.to(torch.float16).float()) < 1.5e-3
    assert lib.ds_layernorm_rows_f16io(x16.data_ptr(), 2046, gm.data_ptr(), bt.data_ptr(), 1e-5, ya.data_ptr(), 2046, 7, 2046, _lib.stream_ptr()) != 0


@pytest.mark.parametrize('kind,nw', [('gemm', 4), ('gemm', 8), ('conv', 0)])
def test_f16_activation_kernels_with_fp16_residual_and_output_rows(kind, nw):
    """The epilogue of the fp16 residual stream (ds_conv_args.res_f16 + out_f16, csrc/igemm_common.h:epilogue_pipe): the residual operand
    is an fp16 tensor, the sum is formed in fp32 and stored as fp16 rows; every column-tile width, both wave counts of the GEMM, ragged
    rows; the GroupNorm column sums are those of the stored (rounded) tensor.  Reference: fp16 operands, fp64 sums, on the CPU."""
    import ctypes as C
    from diff_sampler_amd import _lib, ops
    lib = _lib.load()
    dev = 'cuda'
    shapes = [(1000, 320, 320), (640, 640, 192), (256, 64, 448)] if kind == 'gemm' else [(4, 16, 192, 192), (2, 32, 128, 320), (8, 8, 256, 64)]
    for shp in shapes:
        for nb in (0, 1, 2, 3, 4):
            g = torch.Generator().manual_seed(sum(shp) + nb)
            if kind == 'gemm':
                rows, k, cout = shp
                n, h, taps, cin = rows, 1, 1, k
                x = torch.randn(rows, k, generator=g).to(torch.float16)
                wt = torch.randn(cout, k, generator=g) / k ** 0.5
                wp = ops.pack_linear_weight_f16(ops.pack_linear_weight(wt.to(dev)))
                y = x.double() @ wt.to(torch.float16).double().t()
            else:
                n, h, cin, cout = shp
                rows, taps = n * h * h, 9
                if not lib.ds_conv_f16dma_supported(n, h, h, cin, 0, cout):
                    continue
                x = torch.randn(rows, cin, generator=g).to(torch.float16)
                wt = torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5
                wp = ops.pack_conv_weight_f16(wt.to(dev))
                y = F.conv2d(x.double().reshape(n, h, h, cin).permute(0, 3, 1, 2), wt.to(torch.float16).double(), padding=1)
                y = y.permute(0, 2, 3, 1).reshape(rows, cout)
            bias = torch.randn(cout, generator=g)
            res16 = torch.randn(rows, cout, generator=g).to(torch.float16)
            ref = ((y + bias.double() + res16.double()) * 0.7071).float().to(torch.float16)
            out = torch.full((rows, cout), float('nan'), dtype=torch.float16, device=dev)
            stats = torch.full((-(-rows // 64) * 2 * cout,), float('nan'), device=dev)
            xd, bd_, rd = x.to(dev), bias.to(dev), res16.to(dev)
            a = _lib.ConvArgs(xd.data_ptr(), None, cin, 0, cin, 0, n, h, h, taps, wp.data_ptr(), cout, bd_.data_ptr(), None, 0, 1, rd.data_ptr(), cout,
                              0.7071, 0, out.data_ptr(), cout)
            a.wgt_f16, a.in_f16, a.out_f16, a.res_f16, a.stats_out = 1, 1, 1, 1, stats.data_ptr()
            a.tune.f16dma_nb, a.tune.f16dma_nw = nb, nw
            rc = lib.ds_conv2d_nhwc(C.byref(a), _lib.stream_ptr())
            torch.cuda.synchronize()
            assert rc == 0, (shp, nb, lib.ds_error_string(rc))
            got = out.float().cpu()
            assert torch.isfinite(got).all(), (shp, nb)
            assert _rel(got, ref.float()) < 1.5e-3, (shp, nb)                    # one fp16 ulp where the fp32 sum straddles a rounding boundary
            if rows % 64 == 0:
                st = stats.cpu().reshape(-1, 2, cout)
                assert _rel(st[:, 0], got.reshape(-1, 64, cout).sum(1)) < 1e-5 and _rel(st[:, 1], (got * got).reshape(-1, 64, cout).sum(1)) < 1e-5
    # misuse fails loudly: fp16 residual rows need the fp16-activation kernels and 16-byte rows
    a.in_f16 = 0
    assert lib.ds_conv2d_nhwc(C.byref(a), _lib.stream_ptr()) != 0
    a.in_f16, a.res_ld = 1, cout + 4
    assert lib.ds_conv2d_nhwc(C.byref(a), _lib.stream_ptr()) != 0


DIRECT_EPILOGUE_CASES = [
    # kind, wave count (GEMM), shape, forced nb, epilogue
    ('gemm', 4, (1000, 320, 320), 0, 'bias'),                # ragged rows
    ('gemm', 4, (512, 320, 960), 3, 'silu'),
    ('gemm', 8, (640, 640, 1280), 4, 'res'),
    ('gemm', 4, (300, 1280, 320), 2, 'res'),
    ('gemm', 8, (777, 64, 64), 1, 'none'),
    ('gemm', 8, (2048, 320, 2560), 4, 'geglu'),
    ('gemm', 4, (1000, 320, 2560), 2, 'geglu'),
    ('gemm', 8, (4096, 640, 640), 4, 'stats'),               # column sums on the GEMM: falls back to the staged epilogue, still correct
    ('conv', 0, (4, 16, 192, 192), 3, 'cbias_silu_stats'),
    ('conv', 0, (2, 32, 128, 320), 0, 'res_stats'),
    ('conv', 0, (2, 32, 64, 256), 4, 'cbias_stats'),
    ('conv', 0, (8, 8, 256, 64), 1, 'res'),
    ('conv', 0, (1, 64, 64, 128), 2, 'bias_stats'),
    ('conv', 0, (2, 16, 128, 256), 4, 'bcast_stats'),        # one per-image bias row for every image (cbias_rows = 1)
]


@pytest.mark.parametrize('case', DIRECT_EPILOGUE_CASES)
def test_f16_epilogue_without_the_lds_transpose_equals_the_staged_one(case):
    """csrc/epi_direct.h (round 4): conv3x3_f16dma / gemm_f16dma multiply with swapped MFMA operands (lane = output row, registers = channels)
    and store fp16 rows straight from the accumulators -- no staging through LDS.  Same operations in the same order on the same values as
    epilogue_pipe: the output rows must be BIT-IDENTICAL to the staged epilogue's (ds_conv_tune.ablate bit 12 forces it), the GroupNorm
    column sums equal up to the order of the fp32 additions, and both agree with the CPU reference (fp16 operands, fp64 sums)."""
    import ctypes as C
    from diff_sampler_amd import _lib, ops
    kind, nw, shp, nb, mode = case
    lib = _lib.load()
    dev = 'cuda'
    g = torch.Generator().manual_seed(sum(shp) + nb + len(mode))
    if kind == 'gemm':
        rows, k, cout = shp
        n, h, taps, cin = rows, 1, 1, k
        x = torch.randn(rows, k, generator=g).to(torch.float16)
        wt = torch.randn(cout, k, generator=g) / k ** 0.5
        y = x.double() @ wt.to(torch.float16).double().t()
    else:
        n, h, cin, cout = shp
        rows, taps = n * h * h, 9
        assert lib.ds_conv_f16dma_supported(n, h, h, cin, 0, cout) == 1
        x = torch.randn(rows, cin, generator=g).to(torch.float16)
        wt = torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5
        y = F.conv2d(x.double().reshape(n, h, h, cin).permute(0, 3, 1, 2), wt.to(torch.float16).double(), padding=1)
        y = y.permute(0, 2, 3, 1).reshape(rows, cout)
    bias = torch.randn(cout, generator=g)
    geglu = mode == 'geglu'
    if geglu:
        inner = cout // 2
        val = torch.arange(inner).reshape(-1, 32)
        perm = torch.stack([val, val + inner], 1).reshape(-1)
        wp = ops.pack_linear_weight_f16(ops.pack_linear_weight(wt[perm].to(dev)))
        biasd = bias[perm].contiguous().to(dev)
        yb = y + bias.double()
        ref = yb[:, :inner] * F.gelu(yb[:, inner:])
        ocols = inner
    else:
        wp = ops.pack_linear_weight_f16(ops.pack_linear_weight(wt.to(dev))) if kind == 'gemm' else ops.pack_conv_weight_f16(wt.to(dev))
        biasd = bias.to(dev)
        ref = y + bias.double() if mode != 'none' else y
        ocols = cout
    cb = res16 = None
    scale = 1.0
    if 'cbias' in mode or 'bcast' in mode:
        cb = torch.randn(1 if 'bcast' in mode else n, cout, generator=g)
        ref = ref + (cb.double()[:, None, :].expand(-1, h * h, -1).reshape(-1, cout) if 'bcast' not in mode else cb.double())
    if 'res' in mode:
        res16 = torch.randn(rows, cout, generator=g).to(torch.float16)
        ref, scale = (ref + res16.double()) * 0.7071, 0.7071
    if 'silu' in mode:
        ref = F.silu(ref)
    ref = ref.float().to(torch.float16).float()
    xd = x.to(dev)
    cbd = cb.to(dev) if cb is not None else None
    rd = res16.to(dev) if res16 is not None else None
    with_stats = 'stats' in mode
    outs, sts = [], []
    for ablate in (0, 4096):
        out = torch.full((rows, ocols), float('nan'), dtype=torch.float16, device=dev)
        stats = torch.full((-(-rows // 64) * 2 * cout,), float('nan'), device=dev) if with_stats else None
        a = _lib.ConvArgs(xd.data_ptr(), None, cin, 0, cin, 0, n, h, h, taps, wp.data_ptr(), cout, biasd.data_ptr() if mode != 'none' else None,
                          cbd.data_ptr() if cb is not None else None, cout if cb is not None else 0, cb.shape[0] if cb is not None else 1,
                          rd.data_ptr() if rd is not None else None, cout, scale, 2 if geglu else (1 if 'silu' in mode else 0), out.data_ptr(), ocols)
        a.wgt_f16, a.in_f16, a.out_f16, a.res_f16 = 1, 1, 1, int(rd is not None)
        if with_stats:
            a.stats_out = stats.data_ptr()
        a.tune.f16dma_nb, a.tune.f16dma_nw, a.tune.ablate = nb, nw, ablate
        assert lib.ds_conv_kernel_id(C.byref(a)) == (2567 if kind == 'gemm' else 2566)
        rc = lib.ds_conv2d_nhwc(C.byref(a), _lib.stream_ptr())
        torch.cuda.synchronize()
        assert rc == 0, lib.ds_error_string(rc)
        outs.append(out.cpu())
        sts.append(stats.cpu() if with_stats else None)
    assert torch.isfinite(outs[0].float()).all()
    assert torch.equal(outs[0], outs[1])
    assert _rel(outs[0].float(), ref) < 1.5e-3
    if with_stats and rows % 64 == 0:
        got = outs[0].float()
        for st in sts:
            st = st.reshape(-1, 2, cout)
            assert _rel(st[:, 0], got.reshape(-1, 64, cout).sum(1)) < 1e-5 and _rel(st[:, 1], (got * got).reshape(-1, 64, cout).sum(1)) < 1e-5


@pytest.mark.parametrize('d,heads,sq,skv,mask', [(40, 8, 1024, 1024, 3), (64, 6, 256, 256, 3), (40, 8, 300, 77, 1), (160, 2, 64, 64, 3), (80, 4, 512, 77, 1),
                                                  (64, 3, 130, 130, 2)])
def test_fused_attention_reads_fp16_q_k_v(d, heads, sq, skv, mask):
    """ds_attn_args.in_f16 (bit 0: q, bit 1: k and v are fp16 tensors -- the qkv projection's fp16 rows): the operands are staged as they
    are and `scale` multiplies the fp32 scores.  Against fp64 softmax(q k^T scale) v on the same fp16 values with the softmax weights
    rounded to fp16 as the kernel does: 1e-3 (the bound of the fp32-input path); self- and cross-attention layouts, ragged lengths."""
    import ctypes as C
    from diff_sampler_amd import _lib
    lib = _lib.load()
    dev = 'cuda'
    B, c = 2, heads * d
    g = torch.Generator().manual_seed(d + sq + mask)
    q = torch.randn(B * sq, c, generator=g).to(torch.float16)
    kv = torch.randn(B * skv, 2 * c, generator=g).to(torch.float16)
    qd = q.to(dev) if mask & 1 else q.float().to(dev)
    kvd = kv.to(dev) if mask & 2 else kv.float().to(dev)
    out = torch.full((B * sq, c), float('nan'), device=dev)
    a = _lib.AttnArgs(qd.data_ptr(), kvd.data_ptr(), kvd[:, c:].data_ptr(), out.data_ptr(), c, 2 * c, 2 * c, c, sq * c, skv * 2 * c, skv * 2 * c,
                      sq * c, B, heads, sq, skv, d, d ** -0.5)
    a.in_f16 = mask
    rc = lib.ds_attention_f16(C.byref(a), _lib.stream_ptr())
    torch.cuda.synchronize()
    assert rc == 0, lib.ds_error_string(rc)
    qq = q.double().reshape(B, sq, heads, d).permute(0, 2, 1, 3)
    kk = kv[:, :c].double().reshape(B, skv, heads, d).permute(0, 2, 1, 3)
    vv = kv[:, c:].double().reshape(B, skv, heads, d).permute(0, 2, 1, 3)
    ref = (torch.softmax(qq @ kk.transpose(-1, -2) * d ** -0.5, -1) @ vv).permute(0, 2, 1, 3).reshape(B * sq, c).float()
    got = out.cpu()
    assert torch.isfinite(got).all()
    assert _rel(got, ref) < 1e-3, _rel(got, ref)
    # the fp32 kernel does not take fp16 tensors, and misaligned fp16 rows are refused
    assert lib.ds_attention(C.byref(a), _lib.stream_ptr()) != 0
    if mask & 1:
        a.ldq = c + 4
        assert lib.ds_attention_f16(C.byref(a), _lib.stream_ptr()) != 0


@pytest.mark.parametrize('d,heads,sq,skv,mask,out16', [(40, 8, 1024, 1024, 3, 1), (40, 4, 4096, 4096, 3, 1), (64, 6, 1024, 1024, 3, 1), (64, 9, 256, 256, 3, 0),
                                                        (40, 8, 300, 77, 1, 0), (32, 3, 130, 130, 2, 0), (64, 2, 64, 200, 0, 1), (40, 2, 257, 515, 3, 1)])
def test_fused_attention_two_query_blocks_per_wave_equals_the_one_block_kernel(d, heads, sq, skv, mask, out16):
    """flash_attn_f16x2_kernel (round 6: a wave owns two 32-query blocks, skewed by half a phase; csrc/attention_f16.hip) against
    flash_attn_f16_kernel (one block per wave) through ds_attn_args.variant = 2 / 1: the same MFMA operands in the same order and the same
    softmax expressions per query => EQUAL bits, fp16 or fp32 operand tensors, fp16 or fp32 output rows, ragged query / key counts
    (partial last query block, partial last key tile, a workgroup whose later waves have no queries).  The library's own choice (variant 0)
    is the one-block kernel: the two-block form measured 5 - 8 % slower (profiles/r6_attn_f16_two_blocks_ab.txt) and is kept as that record;
    head sizes above 64 refuse variant 2."""
    import ctypes as C
    from diff_sampler_amd import _lib
    lib = _lib.load()
    dev = 'cuda'
    B, c = 2, heads * d
    g = torch.Generator().manual_seed(d + sq + skv)
    q = torch.randn(B * sq, c, generator=g).to(torch.float16)
    kv = torch.randn(B * skv, 2 * c, generator=g).to(torch.float16)
    qd = q.to(dev) if mask & 1 else q.float().to(dev)
    kvd = kv.to(dev) if mask & 2 else kv.float().to(dev)
    outs = []
    for variant in (1, 2, 0):
        out = torch.full((B * sq, c), float('nan'), device=dev, dtype=(torch.float16 if out16 else torch.float32))
        a = _lib.AttnArgs(qd.data_ptr(), kvd.data_ptr(), kvd[:, c:].data_ptr(), out.data_ptr(), c, 2 * c, 2 * c, c, sq * c, skv * 2 * c, skv * 2 * c,
                          sq * c, B, heads, sq, skv, d, d ** -0.5)
        a.in_f16, a.out_f16, a.variant = mask, out16, variant
        rc = lib.ds_attention_f16(C.byref(a), _lib.stream_ptr())
        torch.cuda.synchronize()
        assert rc == 0, (variant, lib.ds_error_string(rc))
        outs.append(out)
    assert torch.isfinite(outs[1].float()).all()
    assert torch.equal(outs[1], outs[0]) and torch.equal(outs[2], outs[0])
    qq = q.double().reshape(B, sq, heads, d).permute(0, 2, 1, 3)
    kk = kv[:, :c].double().reshape(B, skv, heads, d).permute(0, 2, 1, 3)
    vv = kv[:, c:].double().reshape(B, skv, heads, d).permute(0, 2, 1, 3)
    ref = (torch.softmax(qq @ kk.transpose(-1, -2) * d ** -0.5, -1) @ vv).permute(0, 2, 1, 3).reshape(B * sq, c).float()
    assert _rel(outs[1].float().cpu(), ref) < 1.5e-3
    a.variant, a.d = 2, 80
    assert lib.ds_attention_f16(C.byref(a), _lib.stream_ptr()) != 0


FUSED_NORM_CASES = [
    # B, H(=W), c0, c1, cout, ec ('none' | 'same': the skip projection reads the same raw sources), forced nb, silu, forced splits
    (1, 64, 192, 0, 192, 'none', 0, True, 0),        # ImageNet-64 64x64 layer: three slabs, 192-column tile, seven DMA rounds
    (1, 64, 64, 64, 128, 'same', 2, True, 0),        # two sources of one slab each + two appended 1x1 slabs from two sources
    (2, 32, 128, 64, 192, 'same', 3, True, 0),       # decoder concatenation 128 | 64, skip projection on both
    (1, 32, 64, 0, 64, 'none', 1, False, 0),         # affine only (norm_act NONE), one slab: everything normalised in the prologue
    (3, 16, 192, 0, 384, 'none', 0, True, 0),        # 16-column images: per-round swizzle
    (2, 16, 320, 320, 320, 'same', 2, True, 0),      # SD-1.5 decoder shape class 640 -> 320
    (8, 8, 256, 128, 256, 'same', 2, True, 0),       # four images per tile: per-image coefficient rows, 8-column swizzle
    (4, 8, 768, 0, 768, 'none', 0, True, 4),         # split-K: every split normalises its own slabs
    (4, 8, 512, 256, 256, 'same', 1, True, 3),       # split-K across the two sources and the appended slabs
    (2, 8, 256, 128, 256, 'same', 2, True, 0),       # round 6: half a tile of 8x8 images (empty image slots: zero pages, clamped coefficient rows)
    (7, 8, 192, 0, 192, 'none', 0, True, 2),         # one full tile + three images, split-K
]


@pytest.mark.parametrize('case', FUSED_NORM_CASES)
def test_conv_f16_activations_fused_input_normalisation_equals_the_two_launch_form(case):
    """ds_conv2d_nhwc(in_f16, norm_coefs) -- conv3x3_f16dma_kernel<.., NORM>: the RAW fp16 sources go to LDS by DMA and the kernel rewrites
    its halo in place with silu((x - mu) * A + B) -- against the two-launch form it replaces: ds_norm_act(out_f16, coefs [, raw copy]) writing
    the activated (and concatenated) fp16 tensor, then the same convolution on that tensor.  Same arithmetic, same rounding, same order of the
    fp32 sums (tile width and split count forced equal) => EQUAL bits, output and GroupNorm column sums.  Also bounded against an fp64
    reference on the CPU (1.5e-3 of the output scale: fp16 operands, one ulp of the device exp here and there)."""
    import ctypes as C
    from diff_sampler_amd import _lib, ops
    B, H, c0, c1, cout, ec, nb, silu, splits = case
    lib = _lib.load()
    dev = 'cuda'
    cin, M = c0 + c1, B * H * H
    g = torch.Generator().manual_seed(sum(case[:5]) + 3)
    x = torch.randn(M, cin, generator=g).to(torch.float16)
    planes = torch.stack([0.3 * torch.randn(B, cin, generator=g), 1 + 0.2 * torch.randn(B, cin, generator=g), 0.2 * torch.randn(B, cin, generator=g)], 1).contiguous()
    w = torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5
    we = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5 if ec == 'same' else None
    bias = torch.randn(cout, generator=g)
    xd = x.to(dev)
    x0 = xd[:, :c0].contiguous()
    x1 = xd[:, c0:].contiguous() if c1 else None
    pl = planes.to(dev)
    wp = ops.pack_conv_weight_f16(w.to(dev), we.to(dev) if we is not None else None)
    biasd = bias.to(dev)
    ws = torch.zeros(16 * M * cout, device=dev) if splits else None
    # ---- two-launch form: norm pass (concatenation materialised, raw copy for the skip projection), then the convolution
    a16 = torch.zeros(M, cin, dtype=torch.float16, device=dev)
    r16 = torch.zeros(M, cin, dtype=torch.float16, device=dev)
    na = ops._norm_args(x0, c0, c0, B, H, H, x1=x1, c1=c1, ld1=c1, act=(1 if silu else 0), out=a16, out_ld=cin)
    na.out, na.coefs, na.in_f16 = C.c_void_p(a16.data_ptr()), C.c_void_p(pl.data_ptr()), 3 if c1 else 1
    na.out_f16, na.raw_out, na.raw_ld = 1, C.c_void_p(r16.data_ptr()), cin
    assert lib.ds_norm_act(C.byref(na), _lib.stream_ptr()) == 0

    def run(fused):
        out = torch.full((M, cout), float('nan'), dtype=torch.float16, device=dev)
        stats = torch.full((-(-M // 64) * 2 * cout,), float('nan'), device=dev)
        if fused:
            a = _lib.ConvArgs(x0.data_ptr(), x1.data_ptr() if c1 else None, c0, c1, c0, c1, B, H, H, 9, wp.data_ptr(), cout, biasd.data_ptr(), None, 0, 1,
                              None, 0, 0.7071, 0, out.data_ptr(), cout, pl.data_ptr(), 1 if silu else 0,
                              x0.data_ptr() if we is not None else None, x1.data_ptr() if (we is not None and c1) else None,
                              c0 if we is not None else 0, c1 if we is not None else 0, c0 if we is not None else 0, c1 if we is not None else 0)
        else:
            a = _lib.ConvArgs(a16.data_ptr(), None, cin, 0, cin, 0, B, H, H, 9, wp.data_ptr(), cout, biasd.data_ptr(), None, 0, 1,
                              None, 0, 0.7071, 0, out.data_ptr(), cout, None, 0,
                              r16.data_ptr() if we is not None else None, None, cin if we is not None else 0, 0, cin if we is not None else 0, 0)
        a.wgt_f16, a.in_f16, a.out_f16, a.stats_out = 1, 1, 1, stats.data_ptr()
        a.tune.f16dma_nb = nb if nb else 3                 # the same column tiles in both forms (the fused kernel has no 256-column tile)
        if splits:
            a.workspace, a.workspace_floats, a.tune.splits = ws.data_ptr(), ws.numel(), splits
        else:
            a.tune.splits = 1
        assert lib.ds_conv_kernel_id(C.byref(a)) == (2572 if fused else 2566)
        rc = lib.ds_conv2d_nhwc(C.byref(a), _lib.stream_ptr())
        torch.cuda.synchronize()
        assert rc == 0, lib.ds_error_string(rc)
        return out, stats

    o_ref, s_ref = run(False)
    o_fus, s_fus = run(True)
    assert torch.isfinite(o_fus.float()).all()
    assert torch.equal(o_fus, o_ref), float((o_fus.float() - o_ref.float()).abs().max())
    assert torch.equal(s_fus, s_ref)
    # fp64 reference of the whole thing
    xf = x.double().reshape(B, H * H, cin)
    t = (xf - planes[:, 0].double()[:, None]) * planes[:, 1].double()[:, None] + planes[:, 2].double()[:, None]
    t = F.silu(t) if silu else t
    xin = t.to(torch.float16).double().reshape(B, H, H, cin).permute(0, 3, 1, 2)
    ref = F.conv2d(xin, w.to(torch.float16).double(), padding=1)
    if we is not None:
        ref = ref + F.conv2d(x.double().reshape(B, H, H, cin).permute(0, 3, 1, 2), we.to(torch.float16).double())
    ref = ((ref + bias.double()[None, :, None, None]) * 0.7071).permute(0, 2, 3, 1).reshape(M, cout).float()
    assert _rel(o_fus.float().cpu(), ref) < 1.5e-3


@pytest.mark.parametrize('case', [
    # B, H, C_in, cout, fused norm + SiLU, history tensors, per-sample coefficient rows, separate base point, store_d
    (3, 16, 64, 3, True, 0, False, False, True),          # Euler-like on a SongUNet head
    (2, 32, 128, 3, True, 3, False, False, True),         # iPNDM order 4
    (4, 16, 96, 3, False, 1, True, True, True),           # per-sample rows (AMED form), xb != xe (second stage of a two-stage solver), plain input
    (1, 64, 192, 3, False, 2, False, False, False),       # m = D (denoised), 64x64
])
def test_network_head_with_fused_solver_update_equals_conv_then_update(case):
    """ds_conv2d_nhwc(update = &ds_update_args) on a network head (conv3x3_thin_kernel<., UPD>): F, x' and the history entry m must equal, bit
    for bit, what the same head followed by ds_solver_update(raw = 1, f = F planar) writes.  Through the C ABI; also: the struct is read at call
    time (NULL outputs = plain head), and a layer that is not a head refuses the update."""
    import ctypes as C
    from diff_sampler_amd import _lib, ops
    B, H, cin, cout, norm, nh, rows, sep_xb, store_d = case
    lib = _lib.load()
    dev = 'cuda'
    M = B * H * H
    g = torch.Generator().manual_seed(sum(case[:4]) + 17)
    a_in = torch.randn(M, cin, generator=g).to(dev)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5
    wp = ops.pack_conv_weight(w.to(dev))
    bias = torch.randn(cout, generator=g).to(dev)
    planes = torch.stack([0.3 * torch.randn(B, cin, generator=g), 1 + 0.2 * torch.randn(B, cin, generator=g), 0.2 * torch.randn(B, cin, generator=g)], 1).contiguous().to(dev)
    xe = (torch.randn(B, cout, H, H, generator=g) * 3).to(dev)
    xb = (torch.randn(B, cout, H, H, generator=g) * 3).to(dev) if sep_xb else xe
    hist = [torch.randn(B, cout, H, H, generator=g).to(dev) for _ in range(nh)]
    hc = [1.0, -0.37, 0.21, -0.11, 0.05, 1.7, 1.7, 0.0]
    coefs = None
    if rows:
        coefs = torch.tensor(hc).repeat(B, 1)
        coefs[:, 1] += 0.01 * torch.arange(B)
        coefs[:, 5] += 0.1 * torch.arange(B)
        coefs[:, 6] = coefs[:, 5]
        coefs = coefs.contiguous().to(dev)

    def head(update):
        F_ = torch.full((B, cout, H, H), float('nan'), device=dev)
        a = _lib.ConvArgs(a_in.data_ptr(), None, cin, 0, cin, 0, B, H, H, 9, wp.data_ptr(), cout, bias.data_ptr(), None, 0, 1, None, 0, 1.0, 0,
                          F_.data_ptr(), 4, planes.data_ptr() if norm else None, 1 if norm else 0)
        a.out_nchw = 1
        if update is not None:
            a.update = C.cast(C.pointer(update), C.c_void_p)
        assert lib.ds_conv_kernel_id(C.byref(a)) == 2570
        rc = lib.ds_conv2d_nhwc(C.byref(a), _lib.stream_ptr())
        torch.cuda.synchronize()
        return rc, F_

    def upd_args(f, x_out, m_out):
        return ops.make_update_args(xe, xb, f, B, cout, H, H, x_out, raw=True, f_ld=0, hist=hist, hcoefs=hc, sigma_data=0.5, m_out=m_out,
                                    store_d=store_d, coefs=coefs, coef_rows=(B if rows else 1))
    # two launches
    rc, F0 = head(None)
    assert rc == 0
    x0, m0 = torch.full_like(xe, float('nan')), torch.full_like(xe, float('nan'))
    ops.solver_update(upd_args(F0, x0, m0))
    torch.cuda.synchronize()
    # one launch
    x1, m1 = torch.full_like(xe, float('nan')), torch.full_like(xe, float('nan'))
    u = upd_args(None, x1, m1)
    rc, F1 = head(u)
    assert rc == 0, lib.ds_error_string(rc)
    assert torch.equal(F1, F0) and torch.isfinite(x1).all()
    assert torch.equal(x1, x0), float((x1 - x0).abs().max())
    assert torch.equal(m1, m0), float((m1 - m0).abs().max())
    # read at call time: the same struct with its outputs cleared is a plain head
    u.x_out, u.m_out = None, None
    x1.fill_(7.0)
    rc, F2 = head(u)
    assert rc == 0 and torch.equal(F2, F0) and bool((x1 == 7.0).all())
    # a layer that cannot fuse refuses loudly (64 output channels: a matrix kernel)
    wide = torch.zeros(M, 64, device=dev)
    wp64 = ops.pack_conv_weight((torch.randn(64, cin, 3, 3, generator=g) / 30).to(dev))
    a = _lib.ConvArgs(a_in.data_ptr(), None, cin, 0, cin, 0, B, H, H, 9, wp64.data_ptr(), 64, None, None, 0, 1, None, 0, 1.0, 0, wide.data_ptr(), 64)
    u2 = upd_args(None, x1, m1)
    a.update = C.cast(C.pointer(u2), C.c_void_p)
    assert lib.ds_conv2d_nhwc(C.byref(a), _lib.stream_ptr()) == -1          # DS_E_ARG
