"""`diff_sampler_amd.persistence_hook`: the reference's own extension point (`torch_utils/persistence.py:153-181`) as the
zero-edit route from an unpickled `EDMPrecond` to the HIP engine.  Runs against the REAL reference persistence machinery and
`EDMPrecond` class (build container only: needs /root/reference; skipped on the GPU box)."""
import io
import os
import pickle
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = '/root/reference/diff-solvers-main'

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason='/root/reference is not present on this machine')


@pytest.fixture()
def ref():
    """(persistence module, EDMPrecond class) of the reference, with the hook installed for the duration of the test."""
    import diff_sampler_amd.persistence_hook as H
    sys.path.insert(0, REF)
    try:
        import torch_utils.persistence as persistence
        from models.networks_edm import EDMPrecond
    finally:
        sys.path.remove(REF)
    H.install(persistence)
    H.install(persistence)                                   # idempotent
    assert persistence._import_hooks.count(H.hook) == 1
    yield persistence, EDMPrecond
    H.uninstall(persistence)
    assert H.hook not in persistence._import_hooks


def _snapshot(EDMPrecond, name='tiny_song_cond'):
    import diff_sampler_amd.arch as arch
    torch.manual_seed(3)
    net = EDMPrecond(**arch.NAMED_CONFIGS[name]).eval()
    with torch.no_grad():
        for p in net.parameters():                           # the reference initialisers zero conv1 / proj / the output conv
            p.add_(0.05 * torch.randn_like(p))
    f = io.BytesIO()
    pickle.dump(dict(ema=net), f)                            # what an EDM snapshot holds (sample.py:83-84)
    return net, f.getvalue()


def test_unpickled_precond_is_routed_and_keeps_its_interface(ref):
    import diff_sampler_amd.persistence_hook as H
    persistence, EDMPrecond = ref
    net, blob = _snapshot(EDMPrecond)
    got = pickle.loads(blob)['ema']
    cls = type(got)
    assert cls.__dict__.get('_ds_amd_routed') or any(c.__dict__.get('_ds_amd_routed') for c in cls.__mro__)
    assert hasattr(cls.forward, 'reference_forward')
    # one exec'd module for the outer and inner persistent classes (persistence.py:222-233 caches by source text)
    assert type(got).__module__ == type(got.model).__module__
    assert H.MARK in persistence._module_to_src(sys.modules[type(got).__module__])
    # what sample.py / the AMED code read stays what it was (SURVEY 8b "net protocol")
    for attr in ('img_resolution', 'img_channels', 'label_dim', 'sigma_min', 'sigma_max', 'sigma_data', 'use_fp16'):
        assert getattr(got, attr) == getattr(net, attr), attr
    assert list(got.state_dict()) == list(net.state_dict())
    assert all(torch.equal(a, b) for a, b in zip(got.state_dict().values(), net.state_dict().values()))
    assert list(got.model.enc.keys()) == list(net.model.enc.keys())
    # a CPU call is the reference's own forward, bit for bit
    x = torch.randn(2, net.img_channels, net.img_resolution, net.img_resolution)
    lab = torch.eye(net.label_dim)[[1, 0]] if net.label_dim else None
    with torch.no_grad():
        assert torch.equal(got(x, torch.tensor([0.7, 3.0]), lab), net(x, torch.tensor([0.7, 3.0]), lab))
        assert torch.equal(got.round_sigma(torch.tensor(1.5)), net.round_sigma(torch.tensor(1.5)))


def test_gpu_calls_go_to_the_engine_built_from_the_module(ref, monkeypatch):
    import diff_sampler_amd.persistence_hook as H
    from diff_sampler_amd.engine import spec_from_module
    import diff_sampler_amd.arch as arch
    _, EDMPrecond = ref
    net, blob = _snapshot(EDMPrecond)
    got = pickle.loads(blob)['ema']
    built, calls = [], []

    def fake_engine(module, device, use_fp16):
        built.append((module, str(device), use_fp16))
        # the module the engine is built from is the unpickled one, and the adapter recovers its configuration
        assert spec_from_module(module) == arch.edm_precond_spec(**arch.NAMED_CONFIGS['tiny_song_cond'])
        return lambda x, sigma, class_labels=None: calls.append((x, sigma, class_labels)) or 'from-engine'

    monkeypatch.setattr(H, 'make_engine', fake_engine)

    class OnGpu:                                             # stands in for a tensor on the GPU (no GPU in this container)
        is_cuda, device = True, 'cuda:0'

    x = OnGpu()
    with torch.no_grad():                                    # the samplers' own context (solvers.py decorators, sample.py:294)
        assert got(x, 2.0, class_labels='L') == 'from-engine' and got(x, 1.0) == 'from-engine'
        assert len(built) == 1 and built[0][0] is got and built[0][1:] == ('cuda:0', False)      # built once, cached on the instance
        assert calls[0] == (x, 2.0, 'L') and calls[1] == (x, 1.0, None)
        got.use_fp16 = True                                      # networks_edm.py:486: fp16 body unless force_fp32
        got(x, 1.0); got(x, 1.0, force_fp32=True)
        assert [b[2] for b in built] == [False, True] and len(calls) == 4
        H.invalidate(got)
        got(x, 1.0)
        assert len(built) == 3
        # model_kwargs (augment_labels) are not an engine input: the reference's own forward gets them
        with pytest.raises(AttributeError):                      # ... and fails on the stand-in exactly where real code would run
            got(x, 1.0, augment_labels=None)


def test_calls_that_autograd_would_record_go_to_the_reference_forward(ref, monkeypatch):
    """The route is inference-only (the engine's output has no grad_fn): amed-solver-main's training path differentiates through
    net(x_mid(r), scale_time * t_mid(r)) (solvers_amed.py:141-143 with train=True, training_loop.py:205), so a call with grad mode on and
    an input -- or a parameter of the module -- that requires grad must run the module's ORIGINAL forward, never the engine."""
    import diff_sampler_amd.persistence_hook as H
    _, EDMPrecond = ref
    net, blob = _snapshot(EDMPrecond, 'tiny_song')
    got = pickle.loads(blob)['ema']
    monkeypatch.setattr(H, 'make_engine', lambda *a, **k: (_ for _ in ()).throw(AssertionError('engine must not be built')))
    seen = []
    ref_fwd = type(got).forward.reference_forward
    monkeypatch.setattr(type(got).forward, 'reference_forward', ref_fwd, raising=False)

    class GpuTensor(torch.Tensor):                            # a real (CPU) tensor that claims to live on the GPU
        @property
        def is_cuda(self):
            return True

    x = torch.randn(2, net.img_channels, net.img_resolution, net.img_resolution).as_subclass(GpuTensor)
    got.requires_grad_(False)
    assert H._needs_autograd(got, x, torch.tensor(1.0), None) is False
    with torch.no_grad():
        assert H._needs_autograd(got, x.clone().requires_grad_(True), torch.tensor(1.0), None) is False     # nothing is recorded
    sig = torch.tensor([0.7, 3.0], requires_grad=True)       # the AMED training case: sigma = scale_time * t_mid(r) carries the graph
    assert H._needs_autograd(got, x, sig, None) is True
    out = got(x, sig)                                        # reference forward: differentiable
    assert out.grad_fn is not None
    out.sum().backward()
    assert sig.grad is not None and torch.isfinite(sig.grad).all()
    got.requires_grad_(True)                                 # trainable module in grad mode: also the reference forward
    assert H._needs_autograd(got, x, torch.tensor(1.0), None) is True
    assert got(x, torch.tensor([0.7, 3.0])).grad_fn is not None


def test_amed_bottleneck_hooks_fire_with_the_engines_block_output(ref, monkeypatch):
    """amed-solver-main/solvers_amed.py:7-18 taps `net.model.enc['8x8_block3']` ('8x8_block2' with class labels) through a forward hook;
    the routed forward calls the hooks registered there with `EDMDenoiser.block_output` of the same block."""
    import diff_sampler_amd.persistence_hook as H
    _, EDMPrecond = ref
    for name, key in (('tiny_song_amed', '8x8_block3'), ('tiny_song_amed_cond', '8x8_block2')):
        net, blob = _snapshot(EDMPrecond, name)
        got = pickle.loads(blob)['ema']
        asked = []

        class FakeEngine:
            def __call__(self, x, sigma, class_labels=None):
                return 'D'

            def block_output(self, block):
                asked.append(block)
                return torch.full((2, 64, 8, 8), 7.0)

        monkeypatch.setattr(H, 'make_engine', lambda module, device, use_fp16: FakeEngine())

        class OnGpu:
            is_cuda, device = True, 'cuda:0'

        with torch.no_grad():                                                # the sampler's context (solvers_amed.py decorators)
            assert got(OnGpu(), 1.0) == 'D' and asked == []                # no hook registered: the tap is not even read
            seen = []                                                        # init_hook of the reference, verbatim in behaviour
            handle = got.model.enc[key].register_forward_hook(lambda module, inp, out: seen.append(out.detach()))
            assert got(OnGpu(), 1.0) == 'D'
            assert asked == ['enc.' + key] and len(seen) == 1 and tuple(seen[0].shape) == (2, 64, 8, 8)
            assert torch.mean(seen[-1], dim=1).shape == (2, 8, 8)            # what get_amed_prediction does with it (solvers_amed.py:24-28)
            handle.remove()
            got(OnGpu(), 1.0)
            assert len(seen) == 1


def test_classes_without_a_routed_name_are_left_alone(ref):
    import diff_sampler_amd.persistence_hook as H
    persistence, _ = ref

    class Meta(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__

    m = Meta(type='class', version=6, module_src='class Other:\n    pass\n', class_name='Other', state={})
    assert H.hook(m).module_src == 'class Other:\n    pass\n'
    m2 = Meta(type='class', version=6, module_src='class EDMPrecond:\n    pass\n', class_name='SongUNet', state={})
    once = H.hook(m2).module_src
    assert H.MARK in once and H.hook(m2).module_src == once                                  # appended once
