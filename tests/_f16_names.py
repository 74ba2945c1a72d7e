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


# ---- host mirrors of the tile / split-K rules of csrc/conv3x3_f16dma.hip (tiling, conv3x3_f16dma_plan, conv3x3_f16dma_splits): what a
# ---- stride-1 fp16-activation 3x3 launch `a` (ds_conv_args) will run as, so that a parity test can assert WHICH tilings it exercised ----
def _f16dma_tiling(M, N, nb0):
    out, col, cost = [], 0, 0
    for w in range(nb0, 0, -1):
        if col >= N:
            break
        t = (N - col) // (64 * w)
        if t > 0:
            out.append((col, t, w))
            col += t * 64 * w
            cost += -(-(M // 256) * t // 256) * (1 + w)
    return out, cost


def f16dma_splits(a):
    """Split-K factor of the launch (1 = none): conv3x3_f16dma_splits."""
    M, N = a.n * a.h * a.w, a.cout
    if not a.workspace or a.tune.splits == 1:
        return 1
    wide, _ = _f16dma_tiling(M, N, 4 if (a.w in (16, 32) and not a.norm_coefs) else 3)
    tiles = sum((M // 256) * t for _, t, _ in wide)
    s = a.tune.splits if a.tune.splits > 1 else (256 // tiles if tiles <= 128 else 1)
    kt_all = ((a.c0 + a.c1) // 64) * 9 + (a.ec0 + a.ec1) // 64
    s = min(s, 16, kt_all // 18, a.workspace_floats // (M * N))
    return s if s >= 2 else 1


def f16dma_tile_widths(a):
    """Column-tile widths (in 64-channel units) the launch uses when no width is forced or measured (tune.f16dma_nb == 0): the widest tiling
    for a split layer, else the cost model's (conv3x3_f16dma_plan)."""
    M, N = a.n * a.h * a.w, a.cout
    cap = 4 if (a.w in (16, 32) and not a.norm_coefs) else 3          # max_nb: no 256-column tile with the fused input normalisation
    if a.tune.f16dma_nb > 0:
        return [w for _, _, w in _f16dma_tiling(M, N, min(a.tune.f16dma_nb, cap))[0]]
    if f16dma_splits(a) > 1:
        return [w for _, _, w in _f16dma_tiling(M, N, cap)[0]]
    best = None
    for nb in range(cap, 0, -1):
        til, cost = _f16dma_tiling(M, N, nb)
        if best is None or cost < best[0] or (cost == best[0] and len(til) < best[1]):
            best = (cost, len(til), til)
    return [w for _, _, w in best[2]]
