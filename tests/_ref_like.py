"""A duck-typed stand-in for an unpickled reference ``EDMPrecond`` (test infrastructure; written from scratch, no reference code).

What a real EDM snapshot gives ``sample.py`` (diff-solvers-main/sample.py:81-82: ``pickle.load(f)['ema'].to(device)``) is a
``torch.nn.Module`` tree.  ``/root/reference`` does not exist on the GPU box, so the GPU tests of the real-checkpoint route
(``EDMDenoiser.from_reference_module``, ``engine.spec_from_module``, ``persistence_hook.route_class``) need an object that exposes
exactly what those readers touch -- and nothing that computes:

  * ``img_resolution / img_channels / label_dim / use_fp16 / sigma_min / sigma_max / sigma_data`` (networks_edm.py:473-479),
  * ``.model.enc`` / ``.model.dec``: ``ModuleDict``s keyed ``'<res>x<res>_conv|_down|_block<i>|_in<i>|_up|_aux_norm|_aux_conv'`` whose
    blocks carry ``in_channels / out_channels / num_heads`` (networks_edm.py:137-141) and register forward hooks like any module
    (the AMED bottleneck tap, amed-solver-main/solvers_amed.py:7-18),
  * ``.model.map_layer0 / map_layer1 / map_label / map_augment`` (+ ``out_norm / out_conv`` on the ADM net),
  * a ``state_dict()`` whose keys, order, shapes AND the ``resample_filter`` buffers of the up / down convolutions
    (networks_edm.py:56-57) equal the reference's -- ``tests/test_ref_like_cpu.py`` asserts that against the real class in the build
    container for cifar10 / ffhq / imagenet64 and the tiny nets.

The tree is generated from ``arch.UNetSpec`` (the engine's own layer list), i.e. the constructor order of the reference expressed as
data; weights come from ``arch.init_params``.  ``forward`` raises: a call that reaches it was NOT routed to the engine.

The class is named ``EDMPrecond`` on purpose: ``persistence_hook.hook`` patches pickled module sources that define a class of that name, and
the GPU test feeds THIS file's source through the hook + ``exec`` (what ``persistence._src_to_module`` does, persistence.py:222-233).
"""
import torch


class _Leaf(torch.nn.Module):
    """One parameterised layer: ``weight`` [+ ``bias``] [+ the 2x2 ``resample_filter`` buffer of an up / down convolution]."""

    def __init__(self, wshape=None, bias=True, resample=False, **attrs):
        super().__init__()
        for k, v in attrs.items():
            setattr(self, k, v)
        self.weight = torch.nn.Parameter(torch.zeros(wshape)) if wshape is not None else None
        self.bias = torch.nn.Parameter(torch.zeros(wshape[0])) if (wshape is not None and bias) else None
        self.register_buffer('resample_filter', torch.full((1, 1, 2, 2), 0.25) if resample else None)


class _Norm(torch.nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.ones(c))
        self.bias = torch.nn.Parameter(torch.zeros(c))


class _Block(torch.nn.Module):
    """The attribute / child layout of a U-Net block: norm0, conv0, affine, norm1, conv1, [skip], [norm2, qkv, proj]."""

    def __init__(self, b, emb_channels, song):
        super().__init__()
        rs = bool(b.up or b.down)
        self.in_channels, self.out_channels, self.emb_channels = b.cin, b.cout, emb_channels
        self.num_heads = b.heads
        self.skip_scale, self.adaptive_scale = b.skip_scale, b.adaptive_scale
        self.norm0 = _Norm(b.cin)
        self.conv0 = _Leaf((b.cout, b.cin, 3, 3), resample=rs, in_channels=b.cin, out_channels=b.cout, up=b.up, down=b.down)
        self.affine = _Leaf((b.cout * (2 if b.adaptive_scale else 1), emb_channels))
        self.norm1 = _Norm(b.cout)
        self.conv1 = _Leaf((b.cout, b.cout, 3, 3), in_channels=b.cout, out_channels=b.cout)
        self.skip = None
        if b.cin != b.cout or rs:
            # DDPM++ projects on every resampling block (resample_proj); ADM's resampling skip has no weights, only the filter buffer
            self.skip = _Leaf((b.cout, b.cin, 1, 1) if b.skip_conv else None, resample=rs, in_channels=b.cin, out_channels=b.cout)
        if b.heads:
            self.norm2 = _Norm(b.cout)
            self.qkv = _Leaf((3 * b.cout, b.cout, 1, 1), in_channels=b.cout, out_channels=3 * b.cout)
            self.proj = _Leaf((b.cout, b.cout, 1, 1), in_channels=b.cout, out_channels=b.cout)

    def forward(self, *a, **k):
        raise NotImplementedError('duck-typed block: no arithmetic (the engine evaluates the network)')


class _UNet(torch.nn.Module):
    def __init__(self, spec):
        super().__init__()
        song = spec.model_type == 'SongUNet'
        E, NC, MC = spec.emb_channels, spec.noise_channels, spec.model_channels
        if song:
            self.map_label = _Leaf((NC, spec.label_dim)) if spec.label_dim else None
            self.map_augment = _Leaf((NC, spec.augment_dim), bias=False) if spec.augment_dim else None
            self.map_layer0 = _Leaf((E, NC))
            self.map_layer1 = _Leaf((E, E))
        else:
            self.map_augment = _Leaf((MC, spec.augment_dim), bias=False) if spec.augment_dim else None
            self.map_layer0 = _Leaf((E, MC))
            self.map_layer1 = _Leaf((E, E))
            self.map_label = _Leaf((E, spec.label_dim), bias=False) if spec.label_dim else None
        self.enc, self.dec = torch.nn.ModuleDict(), torch.nn.ModuleDict()
        for b in spec.blocks:
            side, key = b.name.split('.', 1)
            mod = (_Leaf((b.cout, b.cin, 3, 3), in_channels=b.cin, out_channels=b.cout) if b.kind == 'conv' else _Block(b, E, song))
            (self.enc if side == 'enc' else self.dec)[key] = mod
        last = spec.blocks[-1].cout
        if song:
            self.dec[spec.out_norm.split('.', 1)[1]] = _Norm(last)
            self.dec[spec.out_conv.split('.', 1)[1]] = _Leaf((spec.out_channels, last, 3, 3), in_channels=last, out_channels=spec.out_channels)
        else:
            self.out_norm = _Norm(last)
            self.out_conv = _Leaf((spec.out_channels, last, 3, 3), in_channels=last, out_channels=spec.out_channels)


class EDMPrecond(torch.nn.Module):
    """Attributes and module tree of the reference object of the same name; ``forward`` has no arithmetic."""

    def __init__(self, img_resolution, img_channels, label_dim=0, use_fp16=False, sigma_min=0.002, sigma_max=80.0, sigma_data=0.5,
                 model_type='DhariwalUNet', **model_kwargs):
        super().__init__()
        import diff_sampler_amd.arch as arch
        self.img_resolution, self.img_channels, self.label_dim, self.use_fp16 = img_resolution, img_channels, label_dim, use_fp16
        self.sigma_min, self.sigma_max, self.sigma_data = sigma_min, sigma_max, sigma_data
        spec = arch.edm_precond_spec(img_resolution, img_channels, label_dim=label_dim, model_type=model_type, **model_kwargs)
        self.model = _UNet(spec)

    def forward(self, x, sigma, class_labels=None, force_fp32=False, **model_kwargs):
        raise NotImplementedError('duck-typed EDMPrecond: this call was not routed to the HIP engine')

    def round_sigma(self, sigma):
        return torch.as_tensor(sigma)


def build(name_or_kwargs, seed=0, mode='signal', use_fp16=None, cls=None):
    """A filled duck: weights = ``arch.init_params(spec, seed, mode)`` loaded by state_dict key (what a snapshot's ``['ema']`` holds)."""
    import diff_sampler_amd.arch as arch
    kw = dict(arch.NAMED_CONFIGS[name_or_kwargs] if isinstance(name_or_kwargs, str) else name_or_kwargs)
    if use_fp16 is not None:
        kw['use_fp16'] = use_fp16
    net = (cls or EDMPrecond)(**kw).eval().requires_grad_(False)
    params = arch.init_params(arch.edm_precond_spec(**kw), seed=seed, mode=mode)
    missing, unexpected = net.load_state_dict(params, strict=False)
    assert not unexpected and all('resample_filter' in m for m in missing), (missing, unexpected)
    return net
