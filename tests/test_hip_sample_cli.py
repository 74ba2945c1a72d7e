"""GPU: the sample.py mirror end to end (random-init CIFAR-10-shaped tiny net): seed sharding -> per-seed RNG -> fused
sampler -> uint8 kernel -> PNG tree, and its invariances (batching / sharding must not change any image)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _read(outdir):
    import PIL.Image
    imgs = {}
    for d, _, files in os.walk(outdir):
        for f in files:
            if f.endswith('.png'):
                imgs[int(f[:-4])] = np.asarray(PIL.Image.open(os.path.join(d, f)))
    return imgs


def test_sample_run_writes_reference_output_tree(tmp_path):
    from diff_sampler_amd import sample
    a = tmp_path / 'a'
    outdir, n = sample.run('tiny_song', max_batch_size=5, seeds='0-11,1003', outdir=str(a), solver='ipndm', num_steps=5, max_order=3,
                           random_init=True)
    assert n == 13
    imgs = _read(str(a))
    assert sorted(imgs) == list(range(12)) + [1003]
    assert os.path.exists(a / '000000' / '000007.png') and os.path.exists(a / '001000' / '001003.png')
    assert all(v.shape == (16, 16, 3) and v.dtype == np.uint8 for v in imgs.values())
    # a different batch size regroups the seeds but every seed owns its generator: identical images
    b = tmp_path / 'b'
    sample.run('tiny_song', max_batch_size=3, seeds='0-11,1003', outdir=str(b), solver='ipndm', num_steps=5, max_order=3, random_init=True)
    imgs_b = _read(str(b))
    assert all(np.array_equal(imgs[k], imgs_b[k]) for k in imgs)
    # default outdir naming: samples/<dataset>/<solver>_nfe<N>
    assert sample.compute_nfe('ipndm', 5, False, False, 'tiny_song') == 4


def test_sample_run_t_steps_literal_and_deis(tmp_path):
    from diff_sampler_amd import sample
    out, n = sample.run('tiny_song', max_batch_size=4, seeds='0-3', outdir=str(tmp_path / 'c'), solver='deis', max_order=3,
                        t_steps='[80,10.9836,3.8811,1.584,0.5666,0.1698,0.002]', random_init=True)
    assert n == 4 and len(_read(out)) == 4


def test_sample_run_with_gits_schedule_search(tmp_path):
    from diff_sampler_amd import sample
    out, n = sample.run('tiny_song', max_batch_size=4, seeds='0-3', outdir=str(tmp_path / 'g'), solver='ipndm', max_order=3, num_steps=5,
                        dp=True, metric='dev', coeff=1.15, num_warmup=4, solver_tea='ipndm', num_steps_tea=11, random_init=True)
    assert n == 4 and len(_read(out)) == 4


def test_sample_run_ms_coco_latent_diffusion(tmp_path):
    """BASELINE config 5 through the CLI body: SD-1.5 latent U-Net (random init), classifier-free guidance, DPM-Solver++(2M)
    noise prediction on the discrete schedule; writes one latent per seed (the VAE decode is not on this path)."""
    import numpy as np
    from diff_sampler_amd import sample
    out, n = sample.run('ms_coco', max_batch_size=2, seeds='0-1', outdir=str(tmp_path / 'sd'), solver='dpmpp', max_order=2,
                        num_steps=3, predict_x0=False, lower_order_final=True, schedule_type='discrete', schedule_rho=1,
                        guidance_type='cfg', guidance_rate=7.5, random_init=True)
    assert n == 2
    z = np.load(os.path.join(out, '000000', '000001.npy'))
    assert z.shape == (4, 64, 64) and np.isfinite(z).all()


def test_sample_run_amed_cli_config4_ffhq64(tmp_path):
    """BASELINE config 4 the way the reference runs it (amed-solver-main/sample.py): only `--predictor_path` (+ seeds / batch) on
    the command line, every solver setting read back from the predictor; images equal a direct `amed_sampler` call with the
    same predictor and per-seed latents."""
    import PIL.Image
    from diff_sampler_amd import sample, solvers_amed
    from diff_sampler_amd.engine import EDMDenoiser
    from oracle import cases
    dev = torch.device('cuda')
    pp = cases.amed_predictor_params(43, 0.01, 0)
    settings = dict(dataset_name='ffhq', num_steps=4, sampler_stu='amed', sampler_tea='heun', M=1, guidance_type=None, guidance_rate=None,
                    schedule_type='time_uniform', schedule_rho=1, afs=True, scale_dir=0.01, scale_time=0, max_order=None,
                    predict_x0=True, lower_order_final=True)
    path = str(tmp_path / 'predictor.pt')
    torch.save(dict(state_dict=pp, **settings), path)
    out, n = sample.run(predictor_path=path, max_batch_size=2, seeds='5-7', outdir=str(tmp_path / 'amed'), random_init=True,
                        num_steps=99, solver='euler', schedule_type='polynomial')       # CLI values must be overridden by the predictor
    assert n == 3
    imgs = _read(out)
    assert sorted(imgs) == [5, 6, 7] and all(v.shape == (64, 64, 3) for v in imgs.values())
    # default outdir naming uses the AMED NFE rule: 2*(4-1)-1 = 5
    assert sample.AMED_SOLVER_FNS['amed'] == 'amed_sampler'
    net = EDMDenoiser.from_config('ffhq', seed=0)
    rnd = sample.StackedRandomGenerator(dev, [5, 6, 7])
    lat = rnd.randn([3, 3, 64, 64], device=dev)
    pred = solvers_amed.AMEDPredictor(pp, device=dev, **settings)
    ref = solvers_amed.amed_sampler(net, lat, num_steps=4, sigma_min=0.002, sigma_max=80., schedule_type='time_uniform', schedule_rho=1,
                                    afs=True, AMED_predictor=pred)
    u8 = (ref * 127.5 + 128).clip(0, 255).to(torch.uint8).permute(0, 2, 3, 1).cpu().numpy()
    for i, seed in enumerate([5, 6, 7]):
        assert np.abs(imgs[seed].astype(int) - u8[i].astype(int)).max() <= 1      # identical up to the batch-size-dependent split-K order


def test_sample_run_amed_random_predictor_plugin(tmp_path):
    """`--predictor_path random:<seed> --random_init True`: a seeded predictor built from the CLI options (AMED-Plugin on iPNDM)."""
    from diff_sampler_amd import sample
    out, n = sample.run('tiny_song_amed', predictor_path='random:7', max_batch_size=4, seeds='0-3', outdir=str(tmp_path / 'p'),
                        random_init=True, solver='ipndm', num_steps=4, max_order=3, afs=False, schedule_type='polynomial', schedule_rho=7,
                        scale_dir=0.01, scale_time=0)
    assert n == 4 and len(_read(out)) == 4
    with pytest.raises(ValueError):
        sample.run('tiny_song_amed', predictor_path='random:7', seeds='0-1', outdir=str(tmp_path / 'q'), random_init=False, solver='amed')


def test_sample_run_use_fp16_flag_selects_the_fp16_kernels(tmp_path):
    """--use_fp16=True (left unwired in the reference, sample.py:188-189): the CIFAR-10 net on the fp16-operand kernels; images equal
    the fp32 run up to fp16 operand rounding (a few grey levels on a handful of pixels)."""
    from diff_sampler_amd import sample
    kw = dict(max_batch_size=4, seeds='0-3', solver='ipndm', num_steps=6, max_order=3, random_init=True)
    a, _ = sample.run('cifar10', outdir=str(tmp_path / 'f32'), **kw)
    b, _ = sample.run('cifar10', outdir=str(tmp_path / 'f16'), use_fp16=True, **kw)
    ia, ib = _read(a), _read(b)
    assert sorted(ia) == sorted(ib) == [0, 1, 2, 3]
    diff = np.stack([np.abs(ia[k].astype(np.int32) - ib[k].astype(np.int32)) for k in ia])
    assert diff.max() <= 8 and (diff > 1).mean() < 0.05, (diff.max(), (diff > 1).mean())
    assert diff.max() > 0                       # it is a different arithmetic, not the same kernels again
