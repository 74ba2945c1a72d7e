"""CPU-only tests of the host logic: seed sharding (property test + 2 gloo ranks), CLI helpers, coefficient compilers
against the oracle's tensor arithmetic, FID moments + all-reduce on gloo (world_size 2)."""
import math
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from hypothesis import given, settings, strategies as st  # noqa: E402

from diff_sampler_amd import sample as S, solver_utils as U, fid as F  # noqa: E402
from oracle import solvers_ref  # noqa: E402


# ---------------------------------------------------------------------------------------------------- sharding
@settings(max_examples=200, deadline=None)
@given(n=st.integers(1, 3000), batch=st.integers(1, 700), world=st.integers(1, 9))
def test_shard_seeds_partitions_every_seed_exactly_once(n, batch, world):
    seeds = list(range(100, 100 + n))
    got = []
    for r in range(world):
        for b in S.shard_seeds(seeds, batch, r, world):
            assert len(b) <= batch
            got.extend(b.tolist())
    assert sorted(got) == seeds


def test_shard_seeds_is_the_reference_rule():
    seeds = list(range(0, 50000))
    for world in (1, 2, 4, 8):
        nb = ((len(seeds) - 1) // (64 * world) + 1) * world
        allb = torch.as_tensor(seeds).tensor_split(nb)
        for r in range(world):
            mine = S.shard_seeds(seeds, 64, r, world)
            ref = allb[r::world]
            assert len(mine) == len(ref) and all(torch.equal(a, b) for a, b in zip(mine, ref))


def test_cli_helpers():
    assert S.parse_int_list('1,2,5-10') == [1, 2, 5, 6, 7, 8, 9, 10]
    assert S.parse_int_list([3, 4]) == [3, 4]
    assert S.compute_nfe('euler', 11, False, False, 'cifar10') == 10
    assert S.compute_nfe('heun', 6, True, False, 'cifar10') == 9
    assert S.compute_nfe('dpm', 6, False, True, 'cifar10') == 11
    assert S.compute_nfe('ipndm', 6, True, False, 'ms_coco') == 8
    # gits-main/sample.py:240-242 doubles only for guidance rates outside {0, 1}; diff-solvers-main/sample.py:218 always
    assert S.compute_nfe('ipndm', 6, False, False, 'ms_coco', guidance_rate=1.0) == 5
    assert S.compute_nfe('ipndm', 6, False, False, 'ms_coco', guidance_rate=7.5) == 10
    assert S.compute_nfe('ipndm', 6, False, False, 'ms_coco') == 10
    assert set(S.SOLVER_FNS) == {'euler', 'ipndm', 'ipndm_v', 'heun', 'dpm', 'dpmpp', 'deis', 'unipc'}


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _rank_shard(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    seeds = list(range(1000, 1000 + 777))
    mine = torch.cat(list(S.shard_seeds(seeds, 50, rank, world))).to(torch.int64)
    # FID moments: every rank accumulates its shard, then the reference's two all-reduces
    g = torch.Generator().manual_seed(0)
    feats = torch.randn(777, 32, generator=g, dtype=torch.float64)
    mu, sigma = F.calculate_inception_stats(lambda b: b, feats, max_batch_size=50, device='cpu', feature_dim=32)
    gathered = [torch.zeros(500, dtype=torch.int64) for _ in range(world)]
    pad = torch.full((500,), -1, dtype=torch.int64); pad[:len(mine)] = mine
    dist.all_gather(gathered, pad)
    if rank == 0:
        q.put((torch.cat(gathered).tolist(), mu, sigma))
    dist.barrier()
    dist.destroy_process_group()


def test_two_gloo_ranks_shard_and_allreduce_moments():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_shard, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    allseeds, mu, sigma = q.get(timeout=120)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert sorted(s for s in allseeds if s >= 0) == list(range(1000, 1777))
    g = torch.Generator().manual_seed(0)
    feats = torch.randn(777, 32, generator=g, dtype=torch.float64).numpy()
    assert np.allclose(mu, feats.mean(0), atol=1e-12)
    assert np.allclose(sigma, np.cov(feats, rowvar=False), atol=1e-10)
    assert abs(F.calculate_fid_from_inception_stats(mu, sigma, mu, sigma)) < 1e-6


# ---------------------------------------------------------------------------------------------------- coefficient compilers
def _lin_from_oracle(order, px0, scale, ts, tn):
    """Coefficients of (x, m0, m1, m2) implied by the oracle's tensor code, extracted by probing with unit tensors."""
    def run(x, ms):
        X = torch.full((1, 1, 1, 1), x, dtype=torch.float64)
        M = [torch.full((1, 1, 1, 1), v, dtype=torch.float64) for v in ms]
        T = [torch.tensor(v, dtype=torch.float64) for v in ts]
        return float(solvers_ref.dpmpp_step(X, M, T, torch.tensor(tn, dtype=torch.float64), order, predict_x0=px0, scale=scale,
                                            scaled_form=(scale != 1)))
    base = [0.0, 0.0, 0.0]
    cx = run(1.0, base)
    cm = []
    for j in range(order):
        ms = list(base); ms[2 - j] = 1.0          # ms is oldest..newest; m_j newest-first
        cm.append(run(0.0, ms))
    return cx, cm


@pytest.mark.parametrize('order', [1, 2, 3])
@pytest.mark.parametrize('px0', [True, False])
@pytest.mark.parametrize('scale', [1, 0.93])
def test_dpmpp_coefficient_compiler(order, px0, scale):
    ts, tn = [9.6, 3.3, 1.15], 0.4
    cx, cm = U.dpmpp_coeffs(ts[-order:] if order < 3 else ts, tn, order, px0, scale)
    rx, rm = _lin_from_oracle(order, px0, scale, ts, tn)
    assert math.isclose(cx, rx, rel_tol=1e-12)
    assert np.allclose(cm, rm, rtol=1e-10, atol=1e-14)


def test_ipndm_v_coefficients_reduce_to_fixed_step_ab_on_uniform_grid():
    # orders 1-3 are the classical variable-step AB weights; the reference's order-4 expression (solvers.py:470-476)
    # does NOT reduce to AB4 on a uniform grid (57/24 instead of 55/24 ...) -- it is reproduced as is and pinned by the
    # golden trajectories (tests/test_hip_samplers.py, case 'ipndmv4').
    ts = [5.0, 4.0, 3.0, 2.0, 1.0]
    for order in (1, 2, 3):
        i = order - 1
        assert np.allclose(U.ipndm_v_coeffs(order, ts, i), U.ipndm_coeffs(order, ts[i + 1] - ts[i]), rtol=1e-12)


def test_unipc_coefficients_against_reference_formulas():
    """order-1 corrector and order-2 predictor use the reference's hard-coded 0.5 (solver_utils.py:233-243)."""
    th, tn = [3.0, 1.2], 0.5
    k1 = U.unipc_coeffs(th[-1:], tn, 1, predict_x0=True, variant='bh2', use_corrector=True)
    h = -math.log(tn) + math.log(th[-1]); hh = -h
    assert math.isclose(k1['cx'], tn / th[-1])
    assert math.isclose(k1['pred'][0], -math.expm1(hh))
    assert math.isclose(k1['corr'][-1], -math.expm1(hh) * 0.5)
    assert math.isclose(k1['corr'][0], -math.expm1(hh) + math.expm1(hh) * 0.5)
    k2 = U.unipc_coeffs(th, tn, 2, predict_x0=False, variant='bh1', use_corrector=False)
    assert k2['corr'] is None and len(k2['pred']) == 2 and k2['cx'] == 1.0


def test_get_schedule_errors_and_types():
    with pytest.raises(ValueError, match='Got wrong schedule type'):
        U.get_schedule(5, 0.002, 80., schedule_type='nope')
    t = U.get_schedule(6, 0.002, 80.)
    assert t.dtype == torch.float32 and t.shape == (6,) and float(t[0]) == pytest.approx(79.99998474, rel=1e-7)
    assert torch.all(t[:-1] > t[1:])


def test_gits_dynamic_programme_matches_reference_golden():
    from diff_sampler_amd import gits_utils
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'gits.npz'))
    for ns, coeff in [(4, 1.0), (6, 1.15), (8, 0.85)]:
        assert gits_utils.dp(z['dp_cost'], ns, 12, coeff) == list(z[f'dp_{ns}_{coeff}'])


def test_ldm_spec_matches_stable_diffusion_v15_inventory():
    """The SD-1.5 layer list: parameter count and FLOPs of SURVEY.md section 8d (859.5 M parameters, 803.27 GFLOP per forward),
    block counts of openaimodel.py's construction loop, and the CFG schedule end points."""
    import numpy as np
    import diff_sampler_amd.ldm_arch as la
    spec = la.ldm_unet_spec(**la.NAMED_LDM_CONFIGS['sd15'])
    n_params = sum(int(np.prod(shape)) for _, shape, _ in la.ldm_param_table(spec))
    assert n_params == 859_520_964
    assert abs(la.ldm_flops_per_image(spec) / 1e9 - 803.27) < 0.01
    names = [b.name for b in spec.blocks]
    assert names.count('middle_block') == 1
    assert sum(n.startswith('input_blocks') for n in names) == 12 and sum(n.startswith('output_blocks') for n in names) == 12
    kinds = [l.kind for b in spec.blocks for l in b.layers]
    assert kinds.count('st') == 16 and kinds.count('res') == 22 and kinds.count('down') == 3 and kinds.count('up') == 3
    keys = [k for k, _, _ in la.ldm_param_table(spec)]
    assert len(keys) == len(set(keys)) == 686
    ac = la.alphas_cumprod(spec)
    assert ac.shape == (1000,) and abs(float(((1 - ac[-1]) / ac[-1]).sqrt()) - 14.6146) < 1e-3
    with pytest.raises(NotImplementedError):
        la.ldm_unet_spec(use_spatial_transformer=False)


def test_png_sink_writes_reference_tree(tmp_path):
    """Background PNG sink: the reference's output tree <outdir>/<seed - seed % 1000:06d>/<seed:06d>.png (sample.py:313-316),
    pixel-exact, and failures surface on drain()."""
    import numpy as np
    import PIL.Image
    from diff_sampler_amd.sample import PngSink
    rng = np.random.default_rng(0)
    arr = rng.integers(0, 256, size=(70, 8, 8, 3), dtype=np.uint8)
    seeds = list(range(990, 1060))
    sink = PngSink(workers=3)
    sink.submit(arr, seeds, str(tmp_path), subdirs=True, chunk=16)
    sink.close()
    for i, s in enumerate(seeds):
        p = tmp_path / f'{s - s % 1000:06d}' / f'{s:06d}.png'
        assert np.array_equal(np.asarray(PIL.Image.open(p)), arr[i])
    (tmp_path / 'file_not_dir').write_text('x')
    bad = PngSink(workers=1)
    bad.submit(arr[:1], [1], str(tmp_path / 'file_not_dir' / 'y'), subdirs=False)      # makedirs under a regular file
    with pytest.raises(OSError):
        bad.close()


@pytest.mark.parametrize('predict_x0', [True, False])
@pytest.mark.parametrize('variant', ['bh1', 'bh2'])
@pytest.mark.parametrize('order', [1, 2, 3])
def test_unipc_coefficient_compiler_matches_oracle_update(order, variant, predict_x0):
    """solver_utils.unipc_coeffs (host, fp64) expands UniPC's predictor / corrector into plain linear combinations of
    (x, buffered model outputs, model output at the predicted point); applying them must reproduce the oracle's
    tensor-level restatement of unipc_update (solver_utils.py:174-287) on random tensors."""
    from diff_sampler_amd import solver_utils
    from oracle import solvers_ref
    g = torch.Generator().manual_seed(order * 10 + (variant == 'bh2') * 3 + predict_x0)
    times = [torch.tensor(v) for v in (9.0, 4.5, 2.0)][-order:]
    t_next = torch.tensor(0.8)
    x = torch.randn(2, 3, 4, 4, generator=g)
    models = [torch.randn(2, 3, 4, 4, generator=g) for _ in range(order)]
    model_t = torch.randn(2, 3, 4, 4, generator=g)
    # oracle: `evaluate` returns the model output at the predicted point (x0 form) or a denoised D with (x_t - D) / t = model_t
    seen = {}

    def evaluate(x_t, t):
        seen['x_pred'] = x_t.clone()
        return model_t if predict_x0 else x_t - t * model_t
    x_ref, m_ref = solvers_ref._unipc_update(x, models, times, t_next, order, variant, predict_x0, evaluate, True)
    cf = solver_utils.unipc_coeffs([float(t) for t in times], float(t_next), order, predict_x0=predict_x0, variant=variant, use_corrector=True)
    newest_first = models[::-1]
    x_pred = cf['cx'] * x + sum(c * m for c, m in zip(cf['pred'], newest_first))
    x_corr = cf['cx'] * x + sum(c * m for c, m in zip(cf['corr'][:-1], newest_first)) + cf['corr'][-1] * model_t
    assert torch.allclose(x_pred, seen['x_pred'], rtol=1e-5, atol=1e-5)
    assert torch.allclose(x_corr, x_ref, rtol=1e-5, atol=1e-5)
    assert torch.allclose(m_ref, model_t, rtol=1e-5, atol=1e-5)


def test_cfg_schedule_host_maps_match_reference_probes():
    """ldm_engine.CFGSchedule (sigma, sigma_inv, end points, the 'discrete' schedule of get_schedule) against the values the
    real reference CFGPrecond produced (tests/golden/ldm_sd15.npz) -- host math only, no GPU."""
    import numpy as np
    import diff_sampler_amd.ldm_arch as la
    from diff_sampler_amd import solver_utils
    from diff_sampler_amd.ldm_engine import CFGSchedule
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'ldm_sd15.npz'))
    sch = CFGSchedule(la.ldm_unet_spec(**la.NAMED_LDM_CONFIGS['sd15']))
    assert abs(sch.sigma_min - float(z['sigma_min'])) < 1e-7 and abs(sch.sigma_max - float(z['sigma_max'])) < 1e-5
    assert np.allclose(sch.sigma_inv(torch.from_numpy(z['probe_sigma'])).numpy(), z['probe_sigma_inv'], rtol=2e-6, atol=1e-7)
    assert np.allclose(sch.sigma(torch.from_numpy(z['probe_t'])).numpy(), z['probe_sigma_of_t'], rtol=2e-6, atol=1e-7)
    ts = solver_utils.get_schedule(6, sch.sigma_min, sch.sigma_max, device='cpu', schedule_type='discrete', schedule_rho=1, net=sch)
    assert np.allclose(ts.numpy(), z['sched_discrete_6'], rtol=5e-6)


def test_solver_utils_exports_every_public_name_of_the_reference_module():
    """`import solver_utils` drop-in: each public function of diff-solvers-main/solver_utils.py (+ the GITS / AMED variants'
    extra arguments) exists with the same leading parameters.  The reference signatures are written out here so that the
    test also runs where /root/reference is absent."""
    import inspect
    from diff_sampler_amd import solver_utils as su
    want = {
        'get_schedule': ['num_steps', 'sigma_min', 'sigma_max', 'device', 'schedule_type', 'schedule_rho', 'net'],
        'expand_dims': ['v', 'dims'],
        'dynamic_thresholding_fn': ['x0'],
        'dpm_pp_update': ['x', 'model_prev_list', 't_prev_list', 't', 'order', 'predict_x0'],
        'dpm_solver_first_update': ['x', 's', 't', 'model_s', 'predict_x0'],
        'multistep_dpm_solver_second_update': ['x', 'model_prev_list', 't_prev_list', 't', 'predict_x0'],
        'multistep_dpm_solver_third_update': ['x', 'model_prev_list', 't_prev_list', 't', 'predict_x0'],
        'unipc_update': ['x', 'model_prev_list', 't_prev_list', 't', 'order', 'x_t', 'variant', 'predict_x0', 'net', 'class_labels',
                         'use_corrector'],
        'edm2t': ['edm_steps', 'epsilon_s', 'sigma_min', 'sigma_max'],
        'cal_poly': ['prev_t', 'j', 'taus'],
        't2alpha_fn': ['beta_0', 'beta_1', 't'],
        'cal_intergrand': ['beta_0', 'beta_1', 'taus'],
        'get_deis_coeff_list': ['t_steps', 'max_order', 'N', 'deis_mode'],
    }
    for name, params in want.items():
        got = list(inspect.signature(getattr(su, name)).parameters)
        assert got[:len(params)] == params, (name, got)
    ref = '/root/reference/diff-solvers-main/solver_utils.py'
    if os.path.exists(ref):
        import ast
        names = [n.name for n in ast.parse(open(ref).read()).body if isinstance(n, ast.FunctionDef)]
        assert set(names) <= set(want), set(names) - set(want)


def test_deis_helper_functions_closed_form():
    import math
    from diff_sampler_amd import solver_utils as su
    tau = torch.linspace(0.9, 0.2, 7, dtype=torch.float64)
    b0, b1 = 0.1, 19.9
    alpha = su.t2alpha_fn(b0, b1, tau)
    assert torch.allclose(alpha, torch.exp(-0.5 * tau ** 2 * (b1 - b0) - tau * b0))
    # numerical derivative of log(alpha)
    eps = 1e-6
    dl = (su.t2alpha_fn(b0, b1, tau + eps).log() - su.t2alpha_fn(b0, b1, tau - eps).log()) / (2 * eps)
    assert torch.allclose(su.cal_intergrand(b0, b1, tau), -0.5 * dl / torch.sqrt(alpha * (1 - alpha)), rtol=1e-6)
    nodes = torch.tensor([0.9, 0.7, 0.5], dtype=torch.float64)
    for j in range(3):
        p = su.cal_poly(nodes, j, nodes)
        assert torch.allclose(p, torch.eye(3, dtype=torch.float64)[j])


def test_png_single_channel_and_grid_like_make_grid(tmp_path):
    """Single-channel images are saved in mode 'L' (sample.py:314 of the reference); the grid keeps a partial last row padded
    with zeros like torchvision.make_grid(images, nrow=int(sqrt(B)), padding=0)."""
    import PIL.Image
    from diff_sampler_amd.sample import PngSink, save_grid
    arr = (np.arange(2 * 8 * 8).reshape(2, 8, 8, 1) % 255).astype(np.uint8)
    PngSink._write(arr, [3, 4], str(tmp_path), subdirs=False)
    im = PIL.Image.open(tmp_path / '000003.png')
    assert im.mode == 'L' and np.array_equal(np.asarray(im), arr[0, :, :, 0])
    x = torch.rand(7, 3, 4, 4) * 2 - 1                         # 7 images: 2 per row -> 4 rows, last one half empty
    save_grid(x, str(tmp_path / 'g'))
    g = np.asarray(PIL.Image.open(tmp_path / 'g' / 'grid.png'))
    assert g.shape == (16, 8, 3)
    want = (torch.clamp(x / 2 + 0.5, 0, 1) * 255 + 0.5).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).numpy()
    assert np.array_equal(g[12:16, 0:4], want[6]) and not g[12:16, 4:8].any()
    assert np.array_equal(g[4:8, 4:8], want[3])


def test_amed_predictor_loading_rules(tmp_path):
    """amed-solver-main/sample.py:148-185: experiment-number lookup picks the newest snapshot; settings come from the predictor."""
    from diff_sampler_amd import sample
    from oracle import cases
    exp = tmp_path / 'exps' / '00012-cifar10-4-5-amed-heun-1-uni1.0-afs'
    exp.mkdir(parents=True)
    for idx in (3, 20, 100):
        (exp / f'network-snapshot-{idx:06d}.pkl').write_bytes(b'')
    assert sample.find_predictor('12', str(tmp_path / 'exps')).endswith('network-snapshot-000100.pkl')
    assert sample.find_predictor('a/b.pkl', str(tmp_path / 'exps')) == 'a/b.pkl'
    os.makedirs(tmp_path / 'exps' / '00013-empty')
    with pytest.raises(FileNotFoundError):                   # a matched experiment directory without snapshots: a clear error
        sample.find_predictor('13', str(tmp_path / 'exps'))
    pp = cases.amed_predictor_params(43, 0.01, 0)
    path = str(tmp_path / 'p.pt')
    torch.save(dict(state_dict=pp, dataset_name='cifar10', num_steps=4, sampler_stu='amed', schedule_type='time_uniform', schedule_rho=1,
                    afs=True, scale_dir=0.01, scale_time=0), path)
    p = sample.load_predictor(path, 'cpu')
    assert (p.dataset_name, p.num_steps, p.sampler_stu, p.afs, p.schedule_type) == ('cifar10', 4, 'amed', True, 'time_uniform')
    r = sample.load_predictor('random:5', 'cpu', random_init=True, dataset_name='ffhq', solver='dpmpp', num_steps=5, max_order=2,
                              scale_dir=0.02, scale_time=0.1)
    assert r.sampler_stu == 'dpmpp' and r.max_order == 2 and r.w['fc_scale_time.weight'] is not None
    with pytest.raises(ValueError):
        sample.load_predictor('random:5', 'cpu', random_init=False)
    assert sample.compute_nfe('ipndm', 7, True, False, 'cifar10', dp=True) == 6      # GITS: AFS inserts a free step
    assert sample.compute_nfe('heun', 7, True, False, 'cifar10', dp=True) == 13
    assert sample.compute_nfe('heun', 7, True, False, 'cifar10') == 11


def _toy_detector(device):
    """module:factory detector for the FID CLI tests: 8 fixed random projections of the mean-pooled image (deterministic)."""
    g = torch.Generator().manual_seed(0)
    w = torch.randn(3 * 4 * 4, 8, generator=g).to(device)

    def f(images):
        x = torch.nn.functional.adaptive_avg_pool2d(images.to(torch.float32) / 255.0, 4).reshape(images.shape[0], -1)
        return x @ w
    return f


def test_fid_cli_ref_then_calc_matches_numpy(tmp_path):
    """`fid.py ref` + `fid.py calc` end to end on a folder of PNGs (detector injected as module:factory): the moments equal a
    numpy fp64 computation, the subset rule is the reference's (dataset.py:44-48), and FID(x, x) = 0."""
    import PIL.Image
    from click.testing import CliRunner
    rng = np.random.RandomState(3)
    d = tmp_path / 'imgs' / '000000'
    d.mkdir(parents=True)
    imgs = rng.randint(0, 256, size=(23, 8, 8, 3), dtype=np.uint8)
    for i, a in enumerate(imgs):
        PIL.Image.fromarray(a, 'RGB').save(d / f'{i:06d}.png')
    det = 'tests.test_host_logic:_toy_detector'
    r = CliRunner().invoke(F.main, ['ref', '--data', str(tmp_path / 'imgs'), '--dest', str(tmp_path / 'ref.npz'), '--batch', '5',
                                    '--detector', det, '--device', 'cpu'])
    assert r.exit_code == 0, r.output
    z = np.load(tmp_path / 'ref.npz')
    feats = _toy_detector('cpu')(torch.from_numpy(imgs).permute(0, 3, 1, 2)).double().numpy()
    assert np.allclose(z['mu'], feats.mean(0), rtol=1e-9, atol=1e-9)     # moments are accumulated in fp64 from fp32 features
    assert np.allclose(z['sigma'], np.cov(feats, rowvar=False), rtol=1e-6, atol=1e-9)
    r = CliRunner().invoke(F.main, ['calc', '--images', str(tmp_path / 'imgs'), '--ref', str(tmp_path / 'ref.npz'), '--batch', '7',
                                    '--detector', det, '--device', 'cpu'])
    assert r.exit_code == 0, r.output
    assert abs(float(r.output.strip().splitlines()[-1])) < 1e-6
    # --num selects the reference's subset: shuffle with RandomState(seed), first k, sorted
    ds = F.ImageFolder(str(tmp_path / 'imgs'), max_size=10, random_seed=5)
    idx = np.arange(23); np.random.RandomState(5).shuffle(idx)
    assert list(ds.idx) == sorted(idx[:10].tolist()) and len(ds) == 10
    r = CliRunner().invoke(F.main, ['calc', '--images', str(tmp_path / 'imgs'), '--ref', str(tmp_path / 'ref.npz'), '--num', '40',
                                    '--detector', det, '--device', 'cpu'])
    assert r.exit_code != 0            # fewer images than --num


def test_gits_warmup_conditioning_by_model_source():
    """gits-main/gits_utils.py:86-102: integer class indices for 'adm', text conditions for 'ldm' on ms_coco (reference-style net: its own
    get_learned_conditioning, unconditional first; engine net: seeded N(0,1) CLIP-shaped states), one-hot rows otherwise."""
    import types
    from diff_sampler_amd import gits_utils
    dev = torch.device('cpu')
    edm = types.SimpleNamespace(label_dim=10)
    cl, c, uc = gits_utils._warmup_conditioning(edm, dev, 5, 'edm', 'cifar10', {}, None)
    assert cl.shape == (5, 10) and torch.equal(cl.sum(1), torch.ones(5)) and c is None and uc is None
    cl, c, uc = gits_utils._warmup_conditioning(edm, dev, 5, 'adm', 'imagenet64', {}, None)
    assert cl.dtype == torch.int64 and cl.shape == (5,) and int(cl.max()) < 10 and c is None
    assert gits_utils._warmup_conditioning(types.SimpleNamespace(label_dim=0), dev, 5, 'edm', 'cifar10', {}, None) == (None, None, None)
    calls = []
    model = types.SimpleNamespace(get_learned_conditioning=lambda p: (calls.append(list(p)), torch.zeros(len(p), 77, 8))[1])
    ref_net = types.SimpleNamespace(label_dim=True, model=model)
    cl, c, uc = gits_utils._warmup_conditioning(ref_net, dev, 3, 'ldm', 'ms_coco', dict(prompt='a cat', guidance_rate=7.5), None)
    assert cl is None and c.shape == (3, 77, 8) and uc.shape == (3, 77, 8) and calls == [[''] * 3, ['a cat'] * 3]
    calls.clear()
    cl, c, uc = gits_utils._warmup_conditioning(ref_net, dev, 2, 'ldm', 'ms_coco', dict(prompt=None, guidance_rate=1.0), ['x', 'y', 'z'])
    assert uc is None and len(calls) == 1 and set(calls[0]) <= {'x', 'y', 'z'} and len(calls[0]) == 2
    eng_net = types.SimpleNamespace(label_dim=True, spec=types.SimpleNamespace(context_dim=16))
    cl, c, uc = gits_utils._warmup_conditioning(eng_net, dev, 4, 'ldm', 'ms_coco', dict(prompt=None, guidance_rate=7.5), None)
    assert c.shape == (4, 77, 16) and uc.shape == (4, 77, 16) and torch.equal(uc[0], uc[3])


def test_halo_swizzle_of_the_fp16_convolution_is_bank_conflict_free_on_every_image_width():
    """The LDS layout rule of csrc/conv3x3_f16dma.hip (chunk slot = chunk ^ swizzle(halo pixel)) against the bank model of ds_read_b128
    (tools/probes/halo_bank_conflicts.py): the column-based swizzle the kernel uses costs no extra LDS cycle on 8-, 16-, 32- and 64-column images;
    the swizzle on the pixel index it replaced cost one extra cycle per lane group on 16-column images and two on 8-column images -- the
    ratios the SQ counters showed (profiles/r4_*_fp16_sq_counters.json, docs/HISTORY.md E.21).  The kernel's formula is mirrored by
    swizzle_by_column; the kernel tests (GPU) check that both sides of the involution agree."""
    sys.path.insert(0, os.path.join(ROOT, 'tools', 'probes'))
    import halo_bank_conflicts as hb
    for W in (8, 16, 32, 64):
        assert hb.conflicts(W, hb.swizzle_by_column) == (0, 144), W
    assert [hb.conflicts(W, hb.swizzle_by_index)[0] for W in (64, 32, 16, 8)] == [0, 0, 144, 288]


def test_split_k_ranges_of_the_fp16_convolution_partition_the_slabs():
    """The split-K ranges of csrc/conv3x3_f16dma.hip (`cut`: slab boundaries nearest to the equal-TAP cuts; a 3x3 slab weighs nine taps, an
    appended 1x1 slab one) mirrored on the host: for every layer shape (3x3 slabs, 1x1 slabs) and every split count the launcher can choose
    (conv3x3_f16dma_splits: at most 16, at least 18 taps per split) the ranges are non-empty, contiguous and cover every slab exactly once --
    an empty range would send a workgroup into the K loop with nothing staged.  (The GPU tests run six such shapes; this runs all of them.)"""
    def ranges(nchunks, nextra, S):
        kt_all = nchunks * 9 + nextra

        def cut(i):
            t = (kt_all * i) // S
            return (t + 4) // 9 if t <= nchunks * 9 else nchunks + (t - nchunks * 9)
        return [(cut(sp), cut(sp + 1) if sp + 1 < S else nchunks + nextra) for sp in range(S)]

    checked = 0
    for nchunks in range(0, 48):
        for nextra in range(0, 100):
            kt_all = nchunks * 9 + nextra
            for S in range(2, 17):
                if S > kt_all // 18:
                    continue
                r = ranges(nchunks, nextra, S)
                assert r[0][0] == 0 and r[-1][1] == nchunks + nextra and all(a < b for a, b in r), (nchunks, nextra, S, r)
                assert all(r[i][1] == r[i + 1][0] for i in range(S - 1)), (nchunks, nextra, S, r)
                taps = [(min(b, nchunks) - min(a, nchunks)) * 9 + (max(b, nchunks) - max(a, nchunks)) for a, b in r]
                assert sum(taps) == kt_all and max(taps) <= kt_all // S + 9 + 1, (nchunks, nextra, S, taps)     # balanced to within one 3x3 slab
                checked += 1
    assert checked > 20000
