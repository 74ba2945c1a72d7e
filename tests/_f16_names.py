"""Test helper: which layers of a launch plan multiply fp16-rounded operands, as reference state_dict prefixes.

The oracle's reduced-precision variant (oracle.edm_net.operands_f16 / oracle.ldm_net.operands_f16) rounds the multiplicands of the
layers a predicate names; this module derives that predicate from the product's own plan (ds_conv_args.wgt_f16 == 1 on a convolution /
Linear launch, ds_attention_f16 for attention), so that the CPU arithmetic follows the routing the engine actually chose."""
from diff_sampler_amd import _lib


def f16_ops(plan):
    """(names of the conv / Linear launches with fp16 operands, names of the attention launches on the fp16 kernel)"""
    lib = _lib.load()
    convs = {op.name for op in plan.ops if op.fn is lib.ds_conv2d_nhwc and op.keep[0].wgt_f16 == 1}
    attn = {op.name for op in plan.ops if op.fn is lib.ds_attention_f16}
    return convs, attn


def edm_prefixes(plan):
    """Reference prefixes ('model.enc.32x32_block0.conv0', ...) of the EDM engine's fp16-operand layers.  The skip projection is fused
    into conv1 as extra K columns (engine.py), so it follows conv1."""
    convs, attn = f16_ops(plan)
    out = set()
    for n in convs:
        out.add('model.' + n)
        if n.endswith('.conv1'):
            out.add('model.' + n[:-len('.conv1')] + '.skip')
    for n in attn:
        assert n.endswith('.attention'), n
        out.add('model.' + n)
    return out


def edm_stored_prefixes(plan):
    """Reference prefixes of the layers whose output the EDM engine stores in fp16 (ds_conv_args.out_f16 == 1): '...conv0', '...conv1'
    (block outputs: the fp16 residual stream), '...proj' -- the `stored` predicate of oracle.edm_net.operands_f16."""
    lib = _lib.load()
    return {'model.' + op.name for op in plan.ops if op.fn is lib.ds_conv2d_nhwc and op.keep[0].out_f16 == 1}


_LDM_MAP = {'.in_layers': ['.in_layers.2'], '.out_layers': ['.out_layers.3', '.skip_connection'], '.proj_in': ['.proj_in'],
            '.proj_out': ['.proj_out'], '.op': ['.op'], '.conv': ['.conv'],
            '.attn1.qkv': ['.transformer_blocks.0.attn1.to_q', '.transformer_blocks.0.attn1.to_k', '.transformer_blocks.0.attn1.to_v'],
            '.attn1.to_out': ['.transformer_blocks.0.attn1.to_out.0'], '.attn2.q': ['.transformer_blocks.0.attn2.to_q'],
            '.attn2.kv': ['.transformer_blocks.0.attn2.to_k', '.transformer_blocks.0.attn2.to_v'],
            '.attn2.to_out': ['.transformer_blocks.0.attn2.to_out.0'], '.ff.proj_geglu': ['.transformer_blocks.0.ff.net.0.proj'],
            '.ff.out': ['.transformer_blocks.0.ff.net.2']}


def ldm_prefixes(plan):
    """Reference prefixes of the latent-diffusion engine's fp16-operand layers (ldm_engine.py names -> state_dict prefixes)."""
    convs, attn = f16_ops(plan)
    out = set()
    for n in convs:
        for suf in sorted(_LDM_MAP, key=len, reverse=True):
            if n.endswith(suf):
                out.update(n[:-len(suf)] + t for t in _LDM_MAP[suf])
                break
        else:
            out.add(n)
    for n in attn:
        assert n.endswith('.attn1') or n.endswith('.attn2'), n
        out.add(n[:-len('.attn1')] + '.transformer_blocks.0' + n[-len('.attn1'):])
    return out


_LDM_STORED = {'.in_layers': ['.in_layers.2'], '.out_layers': ['.out_layers.3'], '.proj_in': ['.proj_in'], '.proj_out': ['.proj_out'], '.conv': ['.conv'],
               '.op': ['.op'],
               '.attn1.to_out': ['.transformer_blocks.0.attn1.to_out.0'], '.attn2.to_out': ['.transformer_blocks.0.attn2.to_out.0'],
               '.ff.out': ['.transformer_blocks.0.ff.net.2'],
               '.attn1.qkv': ['.transformer_blocks.0.attn1.to_q', '.transformer_blocks.0.attn1.to_k', '.transformer_blocks.0.attn1.to_v'],
               '.attn2.q': ['.transformer_blocks.0.attn2.to_q']}


def ldm_stored_prefixes(plan):
    """State_dict prefixes of the layers whose output the latent-diffusion engine stores in fp16 (ds_conv_args.out_f16 == 1 on the launch):
    the `stored` predicate of oracle.ldm_net.operands_f16.  (The GEGLU product is an fp16 tensor too, but it is only the operand of
    ff.net.2, whose operand rounding already covers it.)"""
    lib = _lib.load()
    out = set()
    for op in plan.ops:
        if op.fn is lib.ds_conv2d_nhwc and op.keep[0].out_f16 == 1 and not op.name.endswith('.ff.proj_geglu'):
            for suf in sorted(_LDM_STORED, key=len, reverse=True):
                if op.name.endswith(suf):
                    out.update(op.name[:-len(suf)] + t for t in _LDM_STORED[suf])
                    break
            else:
                raise KeyError(op.name)
    return out
