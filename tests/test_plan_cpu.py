"""CPU: launch plans are pure host logic (argument structs over workspaces) -- build them without a GPU and check their shape.

No kernel is launched: UNetEngine(device='cpu').plan() only marshals ds_* argument structs and asks the library's host-side
support queries (ds_conv_f16_supported, ds_gemm_f16_supported, ds_attention_f16_supported)."""
import ctypes as C
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import diff_sampler_amd.arch as arch  # noqa: E402
from diff_sampler_amd import _lib  # noqa: E402
from diff_sampler_amd.engine import UNetEngine  # noqa: E402


def _engine(name, **kw):
    spec = arch.edm_precond_spec(**dict(arch.NAMED_CONFIGS[name]))
    return spec, UNetEngine(spec, arch.init_params(spec, seed=1), device='cpu', **kw)


def test_cifar10_plan_is_178_launches_with_fused_norms():
    """178 launches per evaluation; GroupNorm statistics come from the conv epilogues (ds_gn_finalize), the
    3x3 convolutions carry the normalisation planes, the network output is written channel-planar."""
    lib = _lib.load()
    spec, eng = _engine('cifar10')
    P = eng.plan(64, 1)
    assert len(P.ops) == 178
    convs = [op.keep[0] for op in P.ops if op.fn is lib.ds_conv2d_nhwc]
    assert sum(1 for a in convs if a.taps == 9 and a.norm_coefs) >= 60
    assert sum(1 for op in P.ops if op.fn is lib.ds_gn_finalize) >= 60
    assert sum(1 for a in convs if a.out_nchw) == 1
    assert all(a.wgt_f16 == 0 for a in convs)
    assert sum(1 for op in P.ops if op.fn is lib.ds_attention) == 6 and not any(op.fn is lib.ds_attention_f16 for op in P.ops)
    assert eng.plan(64, 1) is P                                  # cached per (batch, emb_rows)


@pytest.mark.parametrize('B', [4, 64])
def test_fp16_mode_routes_imagenet64_to_the_fp16_kernels(B):
    """use_fp16 (networks_edm.py:486): every 3x3 conv with 64-multiple channels, every 1x1 over image rows and every attention
    layer of the ADM net is emitted for the fp16-operand kernels; the embedding Linears (one row per image) stay fp32."""
    lib = _lib.load()
    spec, eng = _engine('imagenet64', use_fp16=True)
    P = eng.plan(B, B)
    convs = [(op.name, op.keep[0]) for op in P.ops if op.fn is lib.ds_conv2d_nhwc]
    c3 = [a for _, a in convs if a.taps == 9]
    assert sum(1 for a in c3 if a.wgt_f16 == 1) >= len(c3) - 2               # all but the 3-channel stem / output convolutions
    qkv = [a for n, a in convs if n.endswith('.qkv') or n.endswith('.proj')]
    assert qkv and all(a.wgt_f16 == 1 for a in qkv)
    emb = [a for n, a in convs if n.startswith('map_') or n == 'affine_all']
    assert emb and all(a.wgt_f16 == 0 for a in emb)
    n_attn = sum(1 for op in P.ops if op.fn is lib.ds_attention_f16)
    assert n_attn > 0 and not any(op.fn is lib.ds_attention for op in P.ops)


def test_split_mode_marks_the_3x3_convolutions_only():
    lib = _lib.load()
    spec, eng = _engine('cifar10', split_fp16=True)
    P = eng.plan(64, 1)
    convs = [op.keep[0] for op in P.ops if op.fn is lib.ds_conv2d_nhwc]
    assert sum(1 for a in convs if a.taps == 9 and a.wgt_f16 == 2) >= 60
    assert all(a.wgt_f16 == 0 for a in convs if a.taps == 1)                   # 1x1 / Linear stay exact fp32
    assert all(a.wgt_shift >= 0 for a in convs)


def test_headline_plan_kernel_routing_at_the_benchmark_batch():
    """Which kernel every 3x3 convolution of the CIFAR-10 plan goes to at B = 256 (`ds_conv_kernel_id` is host logic: tile shape,
    split-K and kernel choice of the launcher, no GPU): the 32x32 and 16x16 layers with 256 output channels on the 256 x 256 tile
    (id 2565: 42 launches), the 8x8 layers on 128-pixel tiles -- eight half-size waves (1284) where the launch is at
    most one workgroup per CU, four waves (128) where split-K already puts two on a CU -- and nothing on the generic gather kernel."""
    import ctypes as C
    from collections import Counter
    lib = _lib.load()
    spec, eng = _engine('cifar10')
    P = eng.plan(256, 1)
    by_res = {}
    for op in P.ops:
        if op.fn is lib.ds_conv2d_nhwc:
            a = op.keep[0]
            if a.taps == 9 and (a.stride or 1) == 1 and a.cout >= 128:
                by_res.setdefault(a.h, Counter())[lib.ds_conv_kernel_id(C.byref(a))] += 1
    assert set(by_res) == {32, 16, 8}
    assert sum(by_res[32].values()) + sum(by_res[16].values()) - by_res[32][256] - by_res[16][256] == by_res[32][2565] + by_res[16][2565]
    assert by_res[32][2565] + by_res[16][2565] == 42 and by_res[32][2565] >= 20 and by_res[16][2565] >= 15
    assert set(by_res[8]) <= {128, 1284} and by_res[8][1284] >= 8 and sum(by_res[8].values()) == 24
    # small batch: every layer has at most one tile per CU -> no four-wave 128 x 128 tiles without split-K left
    P8 = eng.plan(8, 1)
    ids = Counter()
    for op in P8.ops:
        if op.fn is lib.ds_conv2d_nhwc:
            a = op.keep[0]
            if a.taps == 9 and (a.stride or 1) == 1 and a.cout >= 128:
                ids[lib.ds_conv_kernel_id(C.byref(a))] += 1
    assert ids[1284] >= 30 and ids[0] == 0 and ids[2565] == 0, ids


def _dtype_lint(P, lib):
    """Every pointer a launch reads or writes as activation rows must point into a plan-owned tensor of the dtype the launch's flags
    claim (ds_conv_args.in_f16 / out_f16 / res_f16, ds_norm_args.in_f16 bits / out_f16, the LayerNorm entry point, ds_attn_args.in_f16 / out_f16):
    a mismatch would not crash, it would silently reinterpret fp16 bytes as fp32.  Returns the number of (pointer, flag) pairs checked."""
    import torch
    spans = []
    for t in list(P.keep) + [v for v in P.bufs.values() if isinstance(v, torch.Tensor)]:
        spans.append((t.data_ptr(), t.data_ptr() + t.numel() * t.element_size(), t.dtype))
    wild = [(a, a + n) for a, n in getattr(P, 'f16_views', [])]

    def dtype_at(ptr):
        if any(a <= ptr < b for a, b in wild):
            return None                                   # fp32-sized ping-pong storage that blocks use in either dtype
        hits = {d for a, b, d in spans if a <= ptr < b}
        assert len(hits) == 1, (hex(ptr), hits)
        return hits.pop()

    n = 0

    def check(ptr, is16, what):
        nonlocal n
        if not ptr:
            return
        d = dtype_at(int(ptr))
        if d is not None:
            assert d == (torch.float16 if is16 else torch.float32), what
            n += 1

    for op in P.ops:
        if op.fn is lib.ds_conv2d_nhwc:
            a = op.keep[0]
            # second sources exist on fp16-activation launches only with the fused input normalisation (round 5): fp16 like the first
            assert not (a.in_f16 and (a.c1 or a.ec1)) or a.norm_coefs, op.name
            check(a.x0, a.in_f16, op.name + '.x0'); check(a.x1, a.in_f16 and a.c1, op.name + '.x1'); check(a.e0, a.in_f16 and a.ec0, op.name + '.e0')
            check(a.e1, a.in_f16 and a.ec1, op.name + '.e1')
            check(a.res, a.res_f16, op.name + '.res'); check(a.out, a.out_f16, op.name + '.out')
            assert not (a.res_f16 or a.out_f16) or a.in_f16, op.name
        elif op.fn in (lib.ds_norm_act, lib.ds_gn_stats):
            a = op.keep[0]
            check(a.x0, a.in_f16 & 1, op.name + '.x0'); check(a.x1, a.in_f16 & 2, op.name + '.x1')
            if op.fn is lib.ds_norm_act:
                check(a.out, a.out_f16, op.name + '.out'); check(a.raw_out, True, op.name + '.raw_out')
        elif op.fn in (lib.ds_layernorm_rows, lib.ds_layernorm_rows_f16, lib.ds_layernorm_rows_f16io):
            check(op.args[0].value, op.fn is lib.ds_layernorm_rows_f16io, op.name + '.x')
            check(op.args[5].value, op.fn is not lib.ds_layernorm_rows, op.name + '.y')
        elif op.fn in (lib.ds_attention, lib.ds_attention_f16):
            a = op.keep[0]
            check(a.q, a.in_f16 & 1, op.name + '.q'); check(a.k, a.in_f16 & 2, op.name + '.k'); check(a.v, a.in_f16 & 2, op.name + '.v')
            check(a.out, a.out_f16, op.name + '.out')
            assert not a.in_f16 or op.fn is lib.ds_attention_f16, op.name
    return n


@pytest.mark.parametrize('name,B,kw', [('cifar10', 4, dict(use_fp16=True)), ('imagenet64', 4, dict(use_fp16=True)), ('imagenet64', 2, dict(use_fp16=True)),
                                       ('ffhq', 8, dict(use_fp16=True)), ('cifar10', 8, dict()), ('cifar10', 8, dict(split_fp16=True))])
def test_plan_flags_match_the_dtypes_of_the_tensors_they_point_to(name, B, kw):
    """The fp16 residual stream is dtype-driven (plan.Builder sets in_f16 / res_f16 / out_f16 from the tensors it is handed): lint the
    whole plan, including the mixed cases -- the fp32 stem output concatenated with an fp16 stream tensor, CIFAR-10's fp32 attention
    blocks inside an fp16 stream, batches whose 8x8 layers have no fp16-activation kernel (no fp16 stream then)."""
    lib = _lib.load()
    spec, eng = _engine(name, **kw)
    P = eng.plan(B, B)
    n = _dtype_lint(P, lib)
    assert n > 100
    n16 = sum(1 for op in P.ops if op.fn is lib.ds_conv2d_nhwc and op.keep[0].res_f16)
    if kw.get('use_fp16') and P.stream16:
        assert n16 >= 5, n16
    else:
        assert n16 == 0


@pytest.mark.parametrize('N', [2, 4])
def test_ldm_plan_flags_match_the_dtypes_of_the_tensors_they_point_to(N):
    """The same lint on the latent-diffusion plan (SD-1.5 layer structure at reduced width): the fp16 stream through ResBlocks,
    transformer blocks, upsampling and -- round 4 -- the three strided Downsample convolutions (the gather form of the fp16-activation GEMM:
    fp16 rows in and out, no widened copy); no widened fp32 copies at all since round 6 (the 8x8 layers of a batch that is not a multiple of four
    take the fp16-activation convolution with empty image slots in their last tile)."""
    import diff_sampler_amd.ldm_arch as la
    from diff_sampler_amd.ldm_engine import LDMUNetEngine
    lib = _lib.load()
    kw = dict(la.NAMED_LDM_CONFIGS['sd15'])
    spec = la.ldm_unet_spec(**kw)
    eng = LDMUNetEngine(spec, la.init_ldm_params(spec, seed=0), device='cpu', use_fp16=True)
    P = eng.plan(N, 1, 77)
    assert P.stream16
    n = _dtype_lint(P, lib)
    assert n > 300
    widen = [op.name for op in P.ops if op.name.endswith('.widen')]
    assert len(widen) == 0, widen
    down = [op.keep[0] for op in P.ops if op.name.endswith('.op')]
    assert len(down) == 3 and all(a.stride == 2 and a.in_f16 == 1 and a.wgt_f16 == 1 and a.out_f16 == 1 and a.taps == 9 for a in down)
    assert all(lib.ds_conv_kernel_id(C.byref(a)) == 2571 for a in down)
    for part in (P.ctx,):
        assert all(op.name.endswith('.attn2.kv') for op in part.ops)


def test_sd15_fp16_routing_is_the_autocast_layer_set_minus_the_stated_exceptions():
    """Routing-independent pin of WHICH layers multiply fp16 operands in the latent-diffusion engine's fp16 mode (the fp16 oracle's predicate is
    read off the plan, tests/_f16_names.py: that checks the kernels' arithmetic on a given routing, not the routing).  The reference samples
    this U-Net under torch.autocast (sample.py:293-297), which casts the operands of EVERY nn.Conv2d / nn.Linear: the set of all 4-d / 2-d
    `.weight` tensors of the state_dict.  The plan's fp16 set must be exactly that set minus what the engine keeps in fp32 on purpose
    (ldm_engine.LDMUNetEngine docstring): the time embedding (time_embed.*, every ResBlock's emb_layers.1), the context key / value
    projections (attn2.to_k / to_v: fp32 states, once per context) and the 4-channel head out.2 -- each of them stricter than the reference."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import _f16_names
    import diff_sampler_amd.ldm_arch as la
    from diff_sampler_amd.ldm_engine import LDMUNetEngine
    spec = la.ldm_unet_spec(**la.NAMED_LDM_CONFIGS['sd15'])
    params = la.init_ldm_params(spec, seed=0)
    eng = LDMUNetEngine(spec, params, device='cpu', use_fp16=True)
    for N in (2, 4, 32):                               # 1 / 2 / 16 latents under guidance: the 8x8 stage has its fp16 kernels at all of them (round 6: also at
                                                       # two U-Net images, half a 256-pixel tile)
        P = eng.plan(N, 1, 77)
        routed = _f16_names.ldm_prefixes(P)
        attn = {n for n in routed if n.endswith('.attn1') or n.endswith('.attn2')}
        assert len(attn) == 32                         # 16 transformer blocks x (self + cross attention) on the fp16 attention kernel
        routed = {n for n in routed - attn if (n + '.weight') in params}      # (the map names a skip_connection for blocks that have none)
        autocast = {k[:-len('.weight')] for k, v in params.items() if k.endswith('.weight') and v.dim() in (2, 4)}
        kept_fp32 = {n for n in autocast if n.startswith('time_embed.') or n.endswith('.emb_layers.1') or n.endswith('.attn2.to_k')
                     or n.endswith('.attn2.to_v') or n == 'out.2'}
        assert len(kept_fp32) == 2 + 22 + 32 + 1
        assert routed == autocast - kept_fp32, (sorted(autocast - kept_fp32 - routed)[:5], sorted(routed - (autocast - kept_fp32))[:5])


@pytest.mark.parametrize('name', ['imagenet64', 'cifar10', 'ffhq'])
def test_edm_fp16_routing_is_every_convolution_of_the_body_but_stem_and_head(name):
    """The same pin for the EDM nets: with use_fp16 the reference runs the whole U-Net body on x.to(float16) (networks_edm.py:486; every Conv2d
    multiplies w.to(x.dtype), :79), while the embedding MLP and the affine Linear layers see fp32 inputs.  The plan must route exactly the
    4-d weights of the state_dict to the fp16-operand kernels, except the 3-channel stem and the 3-channel head (out_conv / aux_conv), which the
    engine keeps in fp32 (stricter than the reference)."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import _f16_names
    from diff_sampler_amd.engine import EDMDenoiser
    cfg = dict(arch.NAMED_CONFIGS[name])
    spec = arch.edm_precond_spec(**cfg)
    params = arch.init_params(spec, seed=9)
    net = EDMDenoiser(spec, params, device='cpu', use_fp16=True)
    convs = {k[:-len('.weight')] for k, v in params.items() if k.endswith('.weight') and v.dim() == 4}
    kept_fp32 = {n for n in convs if n.endswith('_conv') and (n.startswith('model.enc.') or n.endswith('aux_conv') or n == 'model.out_conv')}
    assert len(kept_fp32) == 2, sorted(kept_fp32)
    for B in ((4, 2) if name == 'imagenet64' else (4,)):                          # (round 6: the ADM net also at a batch whose 8x8 images do not fill a 256-pixel
                                                                                  # tile; the SongUNet nets' 8x8 attention projections read fp32 rows and need 256 of them)
        routed = _f16_names.edm_prefixes(net.engine.plan(B, B))
        routed = {n for n in routed if (n + '.weight') in params}                 # drops the attention launches and skip names of blocks without a skip conv
        assert routed == convs - kept_fp32, (B, sorted(convs - kept_fp32 - routed)[:5], sorted(routed - (convs - kept_fp32))[:5])


def test_planner_tile_measurement_is_off_on_the_cpu_and_for_launches_whose_bits_could_move():
    """plan.Builder._autotune: nothing is measured for a plan built on the CPU (tune words stay zero), and plan._tile_neutral refuses the
    launches whose GroupNorm column sums leave through the staged epilogue (1x1 / Linear layers with statistics, fp32 rows, fp32 residual)."""
    from diff_sampler_amd import plan as plan_mod
    import diff_sampler_amd.ldm_arch as la
    from diff_sampler_amd.ldm_engine import LDMUNetEngine
    lib = _lib.load()
    spec = la.ldm_unet_spec(**la.NAMED_LDM_CONFIGS['sd15'])
    P = LDMUNetEngine(spec, la.init_ldm_params(spec, seed=0), device='cpu', use_fp16=True).plan(4, 1, 77)
    convs = [op.keep[0] for op in P.ops if op.fn is lib.ds_conv2d_nhwc and op.keep[0].in_f16]
    assert len(convs) > 150 and all(a.tune.f16dma_nb == 0 and a.tune.f16dma_nw == 0 for a in convs)
    neutral = [plan_mod._tile_neutral(a) for a in convs]
    assert 0.8 * len(convs) < sum(neutral) < len(convs)
    for a, ok in zip(convs, neutral):
        if a.taps == 1 and a.stats_out:
            assert not ok                              # proj_out: column sums through the staged epilogue
        if not a.stats_out:
            assert ok


def test_fuse_norm16_auto_fuses_exactly_the_measured_layer_class():
    """engine.fuse_norm16 (round 5): 'auto' -- the default -- moves GroupNorm apply + SiLU into the fp16 convolution's LDS halo on the layer
    class where the per-layer A/B found it faster (profiles/r5_conv_f16dma_fused_norm_per_layer.txt: 64x64 images, one column tile), nowhere
    else; True / False fuse every eligible layer / none.  Host logic: the plans are built on the CPU."""
    import ctypes as C
    import diff_sampler_amd.ldm_arch as la
    from diff_sampler_amd import engine
    from diff_sampler_amd.ldm_engine import LDMUNetEngine
    lib = _lib.load()
    assert engine.fuse_norm16_value('0') is False and engine.fuse_norm16_value('1') is True and engine.fuse_norm16_value('auto') == 'auto'
    assert engine.fuse_norm16_here('auto', 64, 192) and not engine.fuse_norm16_here('auto', 64, 384) and not engine.fuse_norm16_here('auto', 32, 192)
    assert engine.fuse_norm16_here(True, 8, 768) and not engine.fuse_norm16_here(False, 64, 192)

    def fused(P):
        return [op.keep[0] for op in P.ops if op.fn is lib.ds_conv2d_nhwc and op.keep[0].in_f16 and op.keep[0].norm_coefs]

    spec, eng = _engine('imagenet64', use_fp16=True)
    assert eng.fuse_norm16 == 'auto'                       # the default (DS_FUSE_NORM16 unset)
    f = fused(eng.plan(64, 64))
    assert len(f) == 12 and all(a.h == 64 and a.cout == 192 and a.taps == 9 and lib.ds_conv_kernel_id(C.byref(a)) == 2572 for a in f)
    assert sum(1 for a in f if a.c1) == 3 and sum(1 for a in f if a.ec1) == 3        # the decoder's concatenations are read in place, skip projection too
    eng.fuse_norm16 = False
    assert fused(eng.plan(64, 64)) == []
    eng.fuse_norm16 = True
    assert len(fused(eng.plan(64, 64))) == 64
    lspec = la.ldm_unet_spec(**dict(la.NAMED_LDM_CONFIGS['sd15']))
    leng = LDMUNetEngine(lspec, la.init_ldm_params(lspec, seed=0), device='cpu', use_fp16=True)
    assert leng.fuse_norm16 == 'auto' and fused(leng.plan(32, 1, 77)) == []          # SD-1.5 has no layer of that class (its 64x64 layers have 320 channels)
