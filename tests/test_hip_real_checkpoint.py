"""GPU: the REAL-CHECKPOINT route executed on the device: every entry a user with actual weights goes through --

  * ``EDMDenoiser.from_reference_module`` / ``engine.spec_from_module`` on an ``EDMPrecond``-shaped ``torch.nn.Module`` tree (what
    ``pickle.load(f)['ema']`` returns, diff-solvers-main/sample.py:81-82), incl. the "follow the checkpoint's use_fp16" rule
    (networks_edm.py:472,486);
  * ``sample.create_model(dataset, model_path=<pickle>)`` and the CLI body on such a pickle;
  * ``persistence_hook``: the source patch the import hook appends (torch_utils/persistence.py:153-181, :222-233 exec the pickled module
    source), ``route_class`` and the routed ``forward`` on GPU tensors, with the AMED bottleneck forward hooks of
    ``net.model.enc['8x8_block2' | '8x8_block3']`` firing (amed-solver-main/solvers_amed.py:7-18);
  * the Stable-Diffusion ``.ckpt`` loader (``model.diffusion_model.`` prefix, fp16 tensors widened; sample.py:111-116).

``/root/reference`` does not exist on the GPU box: the module tree is ``tests/_ref_like.py`` (duck-typed, written from scratch; its state_dict keys /
order / shapes / attributes are asserted equal to the real class in the build container, tests/test_ref_like_cpu.py).  Expectations: bit-equality
with ``from_config`` on the same seeded weights, and the real reference's goldens (2e-4 per evaluation)."""
import importlib
import os
import pickle
import sys
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

pytestmark = pytest.mark.gpu
G = os.path.join(ROOT, 'tests', 'golden')
TOL = 2e-4


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-6))


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda')


def _golden_inputs(name, dev):
    z = np.load(os.path.join(G, f'net_{name}.npz'))
    x = torch.from_numpy(z['x']).to(dev)
    lab = torch.from_numpy(z['labels']).to(dev) if z['labels'].size else None
    return z, x, torch.from_numpy(z['sigma']).to(dev), lab


@pytest.mark.parametrize('name', ['cifar10', 'imagenet64', 'ffhq', 'tiny_song_cond', 'tiny_adm'])
def test_from_reference_module_equals_from_config_and_the_reference_golden(name, dev):
    import _ref_like
    import diff_sampler_amd.arch as arch
    from diff_sampler_amd.engine import EDMDenoiser, spec_from_module
    z, x, sigma, lab = _golden_inputs(name, dev)
    module = _ref_like.build(name, seed=int(z['seed'])).to(dev)          # sample.py:82: the unpickled module is moved to the device first
    assert any('resample_filter' in k for k in module.state_dict()) or name.startswith('tiny_song')
    assert spec_from_module(module) == arch.edm_precond_spec(**arch.NAMED_CONFIGS[name])
    net = EDMDenoiser.from_reference_module(module, device=dev)
    assert not net.use_fp16 and (net.img_resolution, net.img_channels, net.label_dim) == (module.img_resolution, module.img_channels, module.label_dim)
    out = net(x, sigma, class_labels=lab).clone()
    ref = EDMDenoiser.from_config(name, seed=int(z['seed']), device=dev)(x, sigma, class_labels=lab)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)                                         # same weights under the same keys -> the same plan, bit for bit
    assert _rel(out.cpu(), torch.from_numpy(z['out_vec'])) < TOL         # ... and the REAL reference's output on those weights


def test_module_use_fp16_flag_selects_the_fp16_stream(dev):
    """networks_edm.py:486: the public ImageNet-64 checkpoint carries use_fp16=True and the reference then runs the U-Net body in fp16;
    from_reference_module(use_fp16=None) follows the module's flag, an explicit argument overrides it."""
    import _ref_like
    from diff_sampler_amd.engine import EDMDenoiser
    z, x, sigma, lab = _golden_inputs('imagenet64', dev)
    module = _ref_like.build('imagenet64', seed=int(z['seed']), use_fp16=True).to(dev)
    net = EDMDenoiser.from_reference_module(module, device=dev)
    assert net.use_fp16 and net.engine.conv_mode == 1
    out = net(x.expand(4, -1, -1, -1).contiguous(), sigma.expand(4).contiguous(), class_labels=lab.expand(4, -1).contiguous())
    torch.cuda.synchronize()
    plan = net._last[0]
    assert plan.stream16                                                  # the residual stream is fp16, as the reference's body
    gold = torch.from_numpy(z['out_vec'])
    assert all(_rel(out[i:i + 1].cpu(), gold) < 5e-3 for i in range(4))   # fp16 mode vs the fp32 reference
    assert _rel(out[:1].cpu(), gold) > 1e-5                               # it IS a different arithmetic
    ref16 = EDMDenoiser.from_config('imagenet64', seed=int(z['seed']), device=dev, use_fp16=True)
    assert torch.equal(out, ref16(x.expand(4, -1, -1, -1).contiguous(), sigma.expand(4).contiguous(), class_labels=lab.expand(4, -1).contiguous()))
    net32 = EDMDenoiser.from_reference_module(module, device=dev, use_fp16=False)
    assert not net32.use_fp16 and _rel(net32(x, sigma, class_labels=lab).cpu(), gold) < TOL


def _read_pngs(outdir):
    import PIL.Image
    imgs = {}
    for d, _, files in os.walk(outdir):
        for f in files:
            if f.endswith('.png'):
                imgs[int(f[:-4])] = np.asarray(PIL.Image.open(os.path.join(d, f)))
    return imgs


def test_create_model_and_the_cli_body_on_a_pickled_module(tmp_path, dev):
    """sample.py:76-85 with a real file: ``pickle.load(f)['ema']`` -> from_reference_module -> sampler -> PNG tree.  The pickle holds the
    duck module (same weights as ``--random_init`` seed 0), so the images must be the random-init run's, byte for byte."""
    import _ref_like
    from diff_sampler_amd import sample
    path = str(tmp_path / 'edm-tiny-song.pkl')
    with open(path, 'wb') as f:
        pickle.dump(dict(ema=_ref_like.build('tiny_song', seed=0)), f)
    net, source = sample.create_model('tiny_song', model_path=path, device=dev)
    assert source == 'edm' and (net.sigma_min, net.sigma_max) == (0.002, 80.0) and not net.use_fp16
    kw = dict(max_batch_size=4, seeds='0-5', solver='dpmpp', num_steps=6, max_order=2, schedule_type='logsnr')
    a, n = sample.run('tiny_song', outdir=str(tmp_path / 'ckpt'), model_path=path, **kw)
    b, _ = sample.run('tiny_song', outdir=str(tmp_path / 'rand'), random_init=True, **kw)
    ia, ib = _read_pngs(a), _read_pngs(b)
    assert n == 6 and sorted(ia) == sorted(ib) == list(range(6))
    assert all(np.array_equal(ia[k], ib[k]) for k in ia)
    # a checkpoint whose module says use_fp16 is followed (create_model passes use_fp16=None unless the flag is given)
    path16 = str(tmp_path / 'edm-cifar10-fp16.pkl')
    with open(path16, 'wb') as f:
        pickle.dump(dict(ema=_ref_like.build('cifar10', seed=0, use_fp16=True)), f)
    net16, _ = sample.create_model('cifar10', model_path=path16, device=dev)
    assert net16.use_fp16


@pytest.fixture()
def routed_module():
    """The persistence hook end to end without the reference's pickler: the SOURCE of tests/_ref_like.py plays the pickled module source
    (it defines ``class EDMPrecond``), goes through ``persistence_hook.hook`` and is exec'd into a fresh module, which is what
    ``persistence._src_to_module`` does with a snapshot's ``module_src`` (persistence.py:222-233)."""
    import diff_sampler_amd.persistence_hook as H
    with open(os.path.join(ROOT, 'tests', '_ref_like.py')) as f:
        src = f.read()
    meta = types.SimpleNamespace(type='class', version=6, module_src=src, class_name='EDMPrecond', state={})
    meta = H.hook(meta)
    assert H.MARK in meta.module_src and H.hook(meta).module_src == meta.module_src          # appended once
    name = '_ds_test_unpickled_module_src'
    mod = types.ModuleType(name)
    sys.modules[name] = mod
    try:
        exec(meta.module_src, mod.__dict__)
        yield mod
    finally:
        sys.modules.pop(name, None)


def test_persistence_hook_routes_gpu_calls_to_the_engine(routed_module, dev):
    import _ref_like
    import diff_sampler_amd.persistence_hook as H
    from diff_sampler_amd.engine import EDMDenoiser
    cls = routed_module.EDMPrecond
    assert cls.__dict__.get('_ds_amd_routed') and hasattr(cls.forward, 'reference_forward')
    assert not _ref_like.EDMPrecond.__dict__.get('_ds_amd_routed', False)                   # only the exec'd copy was patched
    z, x, sigma, lab = _golden_inputs('tiny_song_cond', dev)
    net = _ref_like.build('tiny_song_cond', seed=int(z['seed']), cls=cls).to(dev)           # sample.py:82
    keys = list(net.state_dict())
    with torch.no_grad():                                                                   # sample.py:294 / the samplers' decorators
        out = net(x, sigma, lab).clone()                                                    # nn.Module.__call__ -> the routed forward
        out_sc = net(x, torch.tensor(0.6, device=dev), class_labels=lab).clone()
    torch.cuda.synchronize()
    ref = EDMDenoiser.from_config('tiny_song_cond', seed=int(z['seed']), device=dev)
    assert torch.equal(out, ref(x, sigma, class_labels=lab))
    assert _rel(out.cpu(), torch.from_numpy(z['out_vec'])) < TOL and _rel(out_sc.cpu(), torch.from_numpy(z['out_scalar'])) < TOL
    # built once and cached on the instance; the module itself is untouched
    engines = net.__dict__[H._ENGINES]
    assert list(engines) == [(str(x.device), False)]
    assert list(net.state_dict()) == keys and net.label_dim == 10 and net.sigma_max == 80.0
    # what is NOT routed reaches the module's own forward (here: the duck's, which has no arithmetic)
    with torch.no_grad(), pytest.raises(NotImplementedError):
        net(x.cpu(), sigma.cpu(), lab.cpu())
    with pytest.raises(NotImplementedError):
        net(x, sigma.clone().requires_grad_(True), lab)                                     # autograd would record it: reference forward
    # use_fp16 on the module (networks_edm.py:486) -> a second engine; force_fp32 -> the first one again
    net.use_fp16 = True
    with torch.no_grad():
        o16 = net(x, sigma, lab).clone()
        o32 = net(x, sigma, lab, force_fp32=True).clone()
    assert sorted(engines) == [(str(x.device), False), (str(x.device), True)]
    assert torch.equal(o32, out) and _rel(o16.cpu(), torch.from_numpy(z['out_vec'])) < 5e-3
    H.invalidate(net)
    assert H._ENGINES not in net.__dict__


@pytest.mark.parametrize('name,key', [('tiny_song_amed', '8x8_block3'), ('tiny_song_amed_cond', '8x8_block2')])
def test_routed_forward_fires_the_amed_bottleneck_hooks_with_the_engines_block(routed_module, name, key, dev):
    """amed-solver-main/solvers_amed.py:7-18 (``init_hook``) registers a forward hook on the bottleneck block and
    ``get_amed_prediction`` (:22-55) reads ``unet_enc_out[-1]``; under the route the block's module does not execute, its hooks are
    called with ``EDMDenoiser.block_output`` -- checked here against the CPU oracle's tap of the same block."""
    import _ref_like
    import diff_sampler_amd.arch as arch
    from diff_sampler_amd import solvers_amed
    from diff_sampler_amd.engine import EDMDenoiser
    from oracle import cases
    from oracle.edm_net import edm_denoise
    kw = dict(arch.NAMED_CONFIGS[name])
    net = _ref_like.build(name, seed=9, cls=routed_module.EDMPrecond).to(dev)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(3, 3, 16, 16, generator=g) * 2.0
    sig = torch.tensor([1.3, 0.2, 5.0])
    lab = torch.eye(10)[torch.tensor([3, 8, 1])] if kw['label_dim'] else None
    glab = lab.to(dev) if lab is not None else None
    unet_enc_out = []                                                       # the reference's init_hook, in behaviour
    handle = net.model.enc[key].register_forward_hook(lambda module, inp, out: unet_enc_out.append(out.detach()))
    with torch.no_grad():
        d = net(x.to(dev), sig.to(dev), glab)
    torch.cuda.synchronize()
    assert len(unet_enc_out) == 1 and unet_enc_out[0].is_cuda and tuple(unet_enc_out[0].shape[2:]) == (8, 8)
    taps = {}
    with torch.no_grad():
        ref = edm_denoise(arch.init_params(arch.edm_precond_spec(**kw), seed=9), kw, x, sig, lab, taps=taps)
    assert _rel(d.cpu(), ref) < TOL and _rel(unet_enc_out[0].cpu(), taps['enc.' + key]) < TOL
    # the predictor consumes the hooked tensor exactly as it consumes the plan's own tap on an EDMDenoiser
    pred = solvers_amed.AMEDPredictor(cases.amed_predictor_params(5, 0.01, 0), device=dev, num_steps=4, sampler_stu='amed',
                                      schedule_type='polynomial', schedule_rho=7, afs=False, scale_dir=0.01, scale_time=0)
    r_hook = solvers_amed.get_amed_prediction(pred, 2.0, 0.5, types.SimpleNamespace(device=dev), unet_enc_out, False, 3)
    eng = EDMDenoiser.from_config(name, seed=9, device=dev)
    tap, tap_handle = solvers_amed.init_hook(eng, class_labels=glab)
    eng(x.to(dev), sig.to(dev), class_labels=glab)
    r_tap = solvers_amed.get_amed_prediction(pred, 2.0, 0.5, eng, tap, False, 3)
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(r_hook, r_tap))
    tap_handle.remove()
    handle.remove()
    with torch.no_grad():
        net(x.to(dev), sig.to(dev), glab)
    assert len(unet_enc_out) == 1                                           # hook removed: the tap is not read any more


def test_sd_checkpoint_loader_fp16_tensors_under_the_lightning_prefix(tmp_path, dev):
    """sample.py:111-116 loads ``models/ldm/stable-diffusion-v1/v1-5-pruned-emaonly.ckpt``: a ``{'state_dict': ...}`` whose U-Net lives under
    ``model.diffusion_model.`` (ddpm.py:1399) -- in the public files as fp16 tensors -- next to VAE / CLIP entries.  The loader must strip
    the prefix, drop the foreign entries, widen to fp32 and bind every tensor by name.  Expectation: the REAL reference's evaluation
    with every weight rounded through fp16 (tests/golden/ldm_sd15_w16.npz, oracle/gen_golden.py --part ldmw16), at the fp32 tolerance."""
    import diff_sampler_amd.ldm_arch as la
    from diff_sampler_amd import sample
    z = np.load(os.path.join(G, 'ldm_sd15.npz'))
    zw = np.load(os.path.join(G, 'ldm_sd15_w16.npz'))
    spec = la.ldm_unet_spec(**la.NAMED_LDM_CONFIGS['sd15'])
    params = la.init_ldm_params(spec, seed=int(z['seed']))
    sd = {'model.diffusion_model.' + k: v.half() for k, v in params.items()}
    sd['first_stage_model.encoder.conv_in.weight'] = torch.zeros(8, 3, 3, 3, dtype=torch.float16)          # VAE / CLIP entries are ignored
    sd['cond_stage_model.transformer.text_model.embeddings.position_ids'] = torch.arange(77).reshape(1, 77)
    path = str(tmp_path / 'v1-5-like.ckpt')
    torch.save({'state_dict': sd, 'global_step': 1}, path)
    del sd, params
    net, source = sample.create_model('ms_coco', model_path=path, device=dev, guidance_type='cfg', guidance_rate=7.5)
    assert source == 'ldm' and not net.use_fp16
    x, cond, uncond = (torch.from_numpy(z[k]).to(dev) for k in ('x', 'cond', 'uncond'))
    out = net(x, torch.from_numpy(z['sigma']).to(dev), condition=cond, unconditional_condition=uncond)
    torch.cuda.synchronize()
    err = _rel(out.cpu(), torch.from_numpy(zw['out_vec']))
    assert err < TOL, err
    # and it is the rounded weights that were evaluated, not the fp32 ones: the fp32-weight golden is further away than the tolerance
    assert _rel(out.cpu(), torch.from_numpy(z['out_vec'])) > TOL
