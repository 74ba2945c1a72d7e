"""CPU: launch plans are pure host logic (argument structs over workspaces) -- build them without a GPU and check their shape.

No kernel is launched: UNetEngine(device='cpu').plan() only marshals ds_* argument structs and asks the library's host-side
support queries (ds_conv_f16_supported, ds_gemm_f16_supported, ds_attention_f16_supported)."""
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
