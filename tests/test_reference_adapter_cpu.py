"""The drop-in adapter `engine.spec_from_module` (used by `EDMDenoiser.from_reference_module`, i.e. by `sample.py
--model_path`) against the REAL reference `EDMPrecond` module for every named configuration.  Needs /root/reference
(build container only); skipped on the GPU box."""
import dataclasses
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = '/root/reference/diff-solvers-main'

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason='/root/reference is not present on this machine')


@pytest.fixture(scope='module')
def EDMPrecond():
    sys.path.insert(0, REF)
    try:
        from models.networks_edm import EDMPrecond as cls
    finally:
        sys.path.remove(REF)
    return cls


def _as_dict(spec):
    d = dataclasses.asdict(spec) if dataclasses.is_dataclass(spec) else dict(vars(spec))
    return d


@pytest.mark.parametrize('name', ['tiny_song', 'tiny_song_cond', 'tiny_adm', 'tiny_song_amed', 'cifar10', 'ffhq', 'imagenet64'])
def test_spec_from_module_recovers_the_named_config(name, EDMPrecond):
    import diff_sampler_amd.arch as arch
    from diff_sampler_amd.engine import spec_from_module
    kw = dict(arch.NAMED_CONFIGS[name])
    import torch
    with torch.device('meta'):
        net = EDMPrecond(**kw)
    got, want = _as_dict(spec_from_module(net)), _as_dict(arch.edm_precond_spec(**kw))
    assert got == want


def test_spec_from_module_augment_dim(EDMPrecond):
    import torch
    import diff_sampler_amd.arch as arch
    from diff_sampler_amd.engine import spec_from_module
    kw = dict(arch.NAMED_CONFIGS['cifar10'], augment_dim=9)
    with torch.device('meta'):
        net = EDMPrecond(**kw)
    assert _as_dict(spec_from_module(net)) == _as_dict(arch.edm_precond_spec(**kw))


def test_state_dict_keys_bind_by_name(EDMPrecond):
    """Every tensor the engine packs is addressed by the reference's state_dict key."""
    import torch
    import diff_sampler_amd.arch as arch
    kw = dict(arch.NAMED_CONFIGS['tiny_adm'])
    net = EDMPrecond(**kw)
    ours = arch.init_params(arch.edm_precond_spec(**kw), seed=0)
    ref_keys = {k for k in net.state_dict() if 'resample_filter' not in k}
    assert set(ours) == ref_keys
    for k in ref_keys:
        assert tuple(ours[k].shape) == tuple(net.state_dict()[k].shape), k
