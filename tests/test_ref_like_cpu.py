"""`tests/_ref_like.py` (the duck-typed EDMPrecond the GPU tests of the real-checkpoint route are built on) against the REAL reference class:
same state_dict keys in the same order with the same shapes (incl. the `resample_filter` buffers), same attributes, same module-tree
names -- so that what `spec_from_module` / `from_reference_module` / the persistence hook read from the duck on the GPU box is what they
would read from an unpickled `edm-*.pkl`.  Needs /root/reference (build container only); skipped on the GPU box."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
REF = '/root/reference/diff-solvers-main'

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason='/root/reference is not present on this machine')


@pytest.fixture(scope='module')
def RefPrecond():
    sys.path.insert(0, REF)
    try:
        from models.networks_edm import EDMPrecond as cls
    finally:
        sys.path.remove(REF)
    return cls


@pytest.mark.parametrize('name', ['cifar10', 'ffhq', 'imagenet64', 'tiny_song', 'tiny_song_cond', 'tiny_adm', 'tiny_song_amed'])
@pytest.mark.parametrize('use_fp16', [False, True])
def test_duck_equals_the_reference_module_where_the_adapters_look(name, use_fp16, RefPrecond):
    import _ref_like
    import diff_sampler_amd.arch as arch
    from diff_sampler_amd.engine import spec_from_module
    kw = dict(arch.NAMED_CONFIGS[name], use_fp16=use_fp16)
    with torch.device('meta'):
        real = RefPrecond(**kw)
        duck = _ref_like.EDMPrecond(**kw)
    rs, ds = real.state_dict(), duck.state_dict()
    assert list(ds) == list(rs)                                                  # keys AND order (buffers included)
    assert all(tuple(ds[k].shape) == tuple(rs[k].shape) and ds[k].dtype == rs[k].dtype for k in rs)
    assert any('resample_filter' in k for k in rs) == any('resample_filter' in k for k in ds)
    for attr in ('img_resolution', 'img_channels', 'label_dim', 'use_fp16', 'sigma_min', 'sigma_max', 'sigma_data'):
        assert getattr(duck, attr) == getattr(real, attr), attr
    assert list(duck.model.enc.keys()) == list(real.model.enc.keys()) and list(duck.model.dec.keys()) == list(real.model.dec.keys())
    for side in ('enc', 'dec'):
        for key, mod in getattr(real.model, side).items():
            d = getattr(duck.model, side)[key]
            for attr in ('in_channels', 'out_channels', 'num_heads'):
                if hasattr(mod, attr):
                    assert getattr(d, attr) == getattr(mod, attr), (key, attr)
    for leaf in ('map_layer0', 'map_layer1', 'map_label', 'map_augment'):
        r, d = getattr(real.model, leaf, None), getattr(duck.model, leaf, None)
        assert (r is None) == (d is None), leaf
        if r is not None:
            assert tuple(r.weight.shape) == tuple(d.weight.shape)
    assert spec_from_module(duck) == spec_from_module(real)


def test_filled_duck_carries_init_params_under_the_reference_keys():
    import _ref_like
    import diff_sampler_amd.arch as arch
    net = _ref_like.build('tiny_song_cond', seed=4)
    want = arch.init_params(arch.edm_precond_spec(**arch.NAMED_CONFIGS['tiny_song_cond']), seed=4)
    sd = net.state_dict()
    assert all(torch.equal(sd[k], v) for k, v in want.items())
    assert not any(p.requires_grad for p in net.parameters())
    with pytest.raises(NotImplementedError):
        net(torch.zeros(1, 3, 16, 16), torch.tensor(1.0))
