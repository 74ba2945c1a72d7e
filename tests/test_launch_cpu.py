"""bench.py --gpus N really runs N ranks and cannot mis-report (CPU: gloo ranks through the `--stub` self-test, which
replaces the kernel work by a sleep and keeps the launcher, the world-size checks, the per-rank timing gather and the FID
moment all-reduce)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from diff_sampler_amd import launch  # noqa: E402

BENCH = os.path.join(ROOT, 'bench.py')


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['OMP_NUM_THREADS'] = '1'
    return env


def test_launch_command_is_torchrun_on_loopback():
    cmd = launch.launch_command(4, 'bench.py', ['--gpus', '4', '--steps', '2'], port=12345, python='python')
    assert cmd == ['python', '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=4', '--master-addr', '127.0.0.1',
                   '--master-port', '12345', 'bench.py', '--gpus', '4', '--steps', '2']


def test_resolve_rules():
    assert launch.resolve(1, 'x.py', [], env={}) == (0, 1, 0)
    assert launch.resolve(2, 'x.py', [], env=dict(WORLD_SIZE='2', RANK='1', LOCAL_RANK='1')) == (1, 2, 1)
    with pytest.raises(launch.LaunchError):
        launch.resolve(8, 'x.py', [], env=dict(WORLD_SIZE='1', RANK='0'))          # --gpus 8 under a 1-rank launcher
    with pytest.raises(launch.LaunchError):
        launch.resolve(1, 'x.py', [], env=dict(WORLD_SIZE='2', RANK='0'))
    with pytest.raises(launch.LaunchError):
        launch.resolve(0, 'x.py', [], env={})
    seen = {}

    def spawn(cmd, env=None):
        seen['cmd'], seen['env'] = cmd, env
        return 7
    with pytest.raises(SystemExit) as ex:                                          # un-launched parent: starts the ranks itself
        launch.resolve(2, 'x.py', ['--gpus', '2'], env={}, spawn=spawn)
    assert ex.value.code == 7
    assert '--nproc-per-node=2' in seen['cmd'] and seen['cmd'][-3:] == ['x.py', '--gpus', '2']
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_bench_gpus2_self_launch_runs_two_ranks():
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--stub', '--steps', '2', '--warmup', '1'], env=_clean_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    line = _line(r.stdout)
    assert line['n_gpus'] == 2 and line['stub'] is True and line['steps'] == 2
    mg = line['multi_gpu']
    assert len(mg['per_rank_ms_per_step']) == 2 and mg['per_rank_ms_per_step_min'] <= mg['per_rank_ms_per_step_max']
    assert mg['per_rank_ms_per_step'][1] > mg['per_rank_ms_per_step'][0]            # the stub makes rank 1 slower: skew is visible
    assert line['ms_per_step'] >= mg['per_rank_ms_per_step_max'] * 0.99             # max over ranks, barrier-bracketed
    assert mg['fid_moment_allreduce']['sigma_32MiB_ms'] > 0
    assert line['config']['images_per_step'] == 2 * 256


def test_bench_driver_form_and_mismatch():
    port = launch.free_port()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), BENCH, '--gpus', '2', '--stub', '--steps', '1', '--warmup', '0']
    r = subprocess.run(cmd, env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert _line(r.stdout)['n_gpus'] == 2
    env = _clean_env()
    env.update(WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, BENCH, '--gpus', '8', '--stub'], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and 'WORLD_SIZE=1' in r.stderr and not r.stdout.strip()


def test_bench_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    r = subprocess.run([sys.executable, BENCH, '--gpus', '1', '--steps', '1'], env=_clean_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and 'needs GPU' in r.stderr


def test_bench_gpus8_self_launch_runs_eight_ranks():
    """The command the driver's scaling run ends in -- `bench.py --gpus 8` -- on 8 gloo ranks: self-launch, communicator head count,
    per-rank skew fields, the FID moment all-reduce, whole-job images per step.  (The stub sleeps 2 ms x (1 + rank) per step.)"""
    r = subprocess.run([sys.executable, BENCH, '--gpus', '8', '--stub', '--steps', '2', '--warmup', '1'], env=_clean_env(),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _line(r.stdout)
    assert line['n_gpus'] == 8 and line['stub'] is True and line['scaling'] == 'weak'
    mg = line['multi_gpu']
    assert len(mg['per_rank_ms_per_step']) == 8 and mg['communicator']['world'] == 8, mg
    assert mg['per_rank_ms_per_step'][7] > mg['per_rank_ms_per_step'][0]
    assert line['ms_per_step'] >= mg['per_rank_ms_per_step_max'] * 0.99
    assert mg['fid_moment_allreduce']['sigma_32MiB_ms'] > 0 and mg['fid_moment_allreduce']['sigma_busbw_GBs'] > 0
    assert line['config']['images_per_step'] == 8 * 256
    assert line['other_configs'] is None and line['latency'] is None          # side measurements belong to the N = 1 default run only
    _assert_roofline_object(line)


def _assert_roofline_object(line):
    """The N > 1 line carries rank 0's `roofline` object (a SCALE record is self-contained): under --stub through the same report path
    over a synthetic record, with every contract key present."""
    roof = line['roofline']
    assert isinstance(roof, dict) and roof.get('stub') is True
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel'):
        assert k in roof, k
    assert roof['bound'] == 'mfma' and roof['unit'] == 'TFLOP/s' and abs(roof['frac'] - roof['achieved'] / roof['peak']) < 1e-3
    assert line['cpu_baseline'] is None                                       # rank 0 at N = 1 only


@pytest.mark.parametrize('config,batch', [('imagenet64', 64), ('sd15', 16)])
def test_bench_gpus8_fp16_configs_3_and_5(config, batch):
    """BASELINE configs 3 (ImageNet-64, 8 GPUs) and 5 (SD-1.5, 8 GPUs) are the 8-GPU ones: the driver-form command line of their fp16 benches
    -- `bench.py --gpus 8 --config <c> --dtype fp16` -- on 8 gloo ranks: default batch and solver of the configuration, whole-job images per
    step, per-rank value min / max next to the barrier skew."""
    r = subprocess.run([sys.executable, BENCH, '--gpus', '8', '--stub', '--steps', '2', '--warmup', '1', '--config', config, '--dtype', 'fp16'],
                       env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _line(r.stdout)
    assert line['n_gpus'] == 8 and line['stub'] is True and line['scaling'] == 'weak'
    assert line['config']['images_per_step'] == 8 * batch and f'batch {batch}/GPU' in line['config']['workload']
    assert ('ImageNet-64' if config == 'imagenet64' else 'Stable Diffusion') in line['config']['workload']
    assert line['dtype'].startswith('fp16 operands')
    mg = line['multi_gpu']
    assert len(mg['per_rank_value']) == 8 and mg['per_rank_value_min'] <= mg['per_rank_value_max']
    assert mg['per_rank_value'][0] > mg['per_rank_value'][7]                    # the stub makes rank 7 the slowest
    assert abs(mg['barrier_skew_ms_per_step'] - (mg['per_rank_ms_per_step_max'] - mg['per_rank_ms_per_step_min'])) < 1e-2
    # whole-job value = all ranks' images over the slowest rank's (barrier-bracketed) time: never above 8 x the slowest rank's own rate
    assert line['value'] <= 8 * mg['per_rank_value_min'] * 1.01
    _assert_roofline_object(line)


def test_pmc_traffic_is_tied_to_the_kernel_source_it_was_measured_on(tmp_path, monkeypatch):
    """`roofline.traffic` comes from a committed rocprofv3 PMC summary.  The summary carries the session it was collected in and the hash
    of every kernel translation unit at that time (tools/rocprof_summary.py); bench.py reports the bytes with that provenance while the
    hash of the dominant kernel's translation unit is unchanged and null -- with the reason -- once the kernel source has moved on."""
    sys.path.insert(0, ROOT)
    import bench
    from diff_sampler_amd import build
    name = bench.PMC_KEYS[2565]
    row = lambda f, w, n: {'FETCH_SIZE_launches': n, 'FETCH_SIZE_KiB_total': f * n, 'FETCH_SIZE_KiB_avg_per_launch': f, 'WRITE_SIZE_launches': n,
                           'WRITE_SIZE_KiB_total': w * n, 'WRITE_SIZE_KiB_avg_per_launch': w}       # a row as tools/rocprof_summary.py writes it
    kern = {name: row(200000.0, 160000.0, 1260)}
    f = tmp_path / 'pmc.json'
    monkeypatch.setattr(bench, 'PMC_FILE', str(f))
    f.write_text(json.dumps({'meta': {'session': 'gpurun_out/x', 'kernel_source_sha256': build.source_hashes()}, 'kernels': kern}))
    byts, src = bench.pmc_traffic(2565)
    assert byts == round(1024 * (2 * 200000.0 + 160000.0)) and src['session'] == 'gpurun_out/x' and src['kernel_source'] == 'conv3x3_halo.hip'
    stale = dict(build.source_hashes(), **{'conv3x3_halo.hip': '0' * 64})
    f.write_text(json.dumps({'meta': {'session': 'gpurun_out/x', 'kernel_source_sha256': stale}, 'kernels': kern}))
    assert bench.pmc_traffic(2565) is None and 'another build' in bench.PMC_NOTE['why']
    f.write_text(json.dumps({'kernels': kern}))                               # a summary without provenance (round 3's) is not trusted
    assert bench.pmc_traffic(2565) is None
    # kernel CLASSES of the fp16 engines (other_configs): all instantiations of the class, averaged over their launches
    kern16 = {'void igemm::(anonymous namespace)::conv3x3_f16dma_kernel<64, 3, true, false>(igemm::KParams)': row(1000.0, 400.0, 30),
              'void igemm::(anonymous namespace)::conv3x3_f16dma_kernel<8, 2, false, false>(igemm::KParams)': row(100.0, 40.0, 10),
              'void igemm::(anonymous namespace)::conv3x3_f16dma_kernel<32, 3, true, true>(igemm::KParams)': row(7.0, 3.0, 5),      # NORM instantiation: class 2572
              'norm_act_kernel(ds_norm_args, int, int, int)': row(50.0, 50.0, 7)}
    f16 = tmp_path / 'pmc16.json'
    monkeypatch.setitem(bench.PMC_FILES, ('imagenet64', 'fp16'), str(f16))
    f16.write_text(json.dumps({'meta': {'session': 'gpurun_out/y', 'kernel_source_sha256': build.source_hashes()}, 'kernels': kern16}))
    byts, src = bench.pmc_traffic(2566, ('imagenet64', 'fp16'))
    assert byts == round(1024 * (2 * (1000.0 * 30 + 100.0 * 10) + (400.0 * 30 + 40.0 * 10)) / 40) and src['instantiations'] == 2
    assert bench.pmc_traffic(2572, ('imagenet64', 'fp16'))[0] == round(1024 * 17.0)
    assert bench.pmc_traffic('norm_act', ('imagenet64', 'fp16'))[0] == round(1024 * 150.0)
    assert bench.pmc_traffic(2567, ('imagenet64', 'fp16')) is None            # no row of that class in the pass
    # the hash is of the CODE: comments and whitespace do not count, a changed token does
    a = build._code_only('int f(int x) {  // doubles\n  return 2 * x; /* really */ }\n')
    assert a == build._code_only('int f(int x) {\n\n return 2 * x; }') and a != build._code_only('int f(int x) { return 3 * x; }')


def test_committed_pmc_summary_names_the_dominant_kernel():
    """The committed summary bench.py reads must hold the kernel the headline attributes its dominant share to (a kernel whose template
    arguments change would silently lose the field), with HBM bytes per launch a plausible multiple of the algorithmic bytes of the
    headline's 3x3 layers (about 503 MB per launch at B = 256)."""
    import re
    sys.path.insert(0, ROOT)
    import bench
    if not os.path.exists(bench.PMC_FILE):
        pytest.skip('no PMC summary committed for this round yet')
    z = json.load(open(bench.PMC_FILE))
    k = z['kernels'][bench.PMC_KEYS[2565]]
    t = 1024.0 * (2.0 * k['FETCH_SIZE_KiB_avg_per_launch'] + k['WRITE_SIZE_KiB_avg_per_launch'])
    assert 0.9 * 503e6 < t < 1.5 * 503e6, t
    assert 'kernel_source_sha256' in z['meta'] and z['meta']['session']
    # ... and the name is the one the in-tree source instantiates as the default 256 x 256 kernel
    m = re.match(r'void igemm::conv3x3_halo_kernel<4, true, 2, (\d+), 4>', bench.PMC_KEYS[2565])
    assert m, bench.PMC_KEYS[2565]
    src = open(os.path.join(ROOT, 'diff_sampler_amd', 'csrc', 'conv3x3_halo.hip')).read()
    lean, ntepi = (int(re.search(r'constexpr int VAR_LEAN = (\d+)', src).group(1)), int(re.search(r'VAR_NTEPI = (\d+)', src).group(1)))
    assert int(m.group(1)) == (lean | ntepi) and 'launch_one<4, true, 2, VAR_TILE_OPTS, 4>' in src


def test_bench_defaults_are_the_workloads_the_kept_lines_are_quoted_on():
    """`python bench.py --config X` without --batch / --solver must be the workload of profiles/r3_bench_X_*_line.json (the driver and a
    reader re-run it that way): CIFAR-10 DPM-Solver++(2M) B = 256 (BASELINE config 2, the headline), FFHQ-64 B = 128, ImageNet-64 iPNDM-4
    B = 64 (config 3), SD-1.5 B = 16 (config 5); explicit flags win."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    want = {'cifar10': (256, 'dpmpp'), 'ffhq': (128, 'dpmpp'), 'imagenet64': (64, 'ipndm'), 'sd15': (16, 'dpmpp')}
    for cfg, (b, solver) in want.items():
        a = bench.parse(['--config', cfg])
        assert (a.batch, a.solver) == (b, solver), (cfg, a.batch, a.solver)
    a = bench.parse(['--config', 'imagenet64', '--batch', '8', '--solver', 'euler'])
    assert (a.batch, a.solver) == (8, 'euler')
    assert bench.parse([]).config == 'cifar10' and bench.parse([]).gpus == 1                 # the driver's default call: the headline on one GPU
    for name, b in (('r3_bench_line.json', 256), ('r3_bench_imagenet64_fp16_line.json', 64), ('r3_bench_sd15_fp16_line.json', 16),
                    ('r3_bench_ffhq_fp32_line.json', 128)):
        line = [x for x in open(os.path.join(ROOT, 'profiles', name)).read().splitlines() if x.startswith('{')][-1]
        assert f'batch {b}/GPU' in json.loads(line)['config']['workload'], name
    # round 4: ONE default run carries the headline and, under the same clock, the other configurations (`other_configs`) + `latency`
    line = json.loads([x for x in open(os.path.join(ROOT, 'profiles', 'r5_bench_line.json')).read().splitlines() if x.startswith('{')][-1])
    assert 'batch 256/GPU' in line['config']['workload'] and line['dtype'] == 'fp32'
    got = [(o['config']['workload'], o['dtype']) for o in line['other_configs']]
    for (cfg, dt, b), (wl, odt) in zip(bench.OTHER_CONFIGS, got):
        assert f'batch {b}/GPU' in wl and odt == dt and bench.WORKLOAD_NAMES[cfg].split(' (')[0] in wl, (cfg, wl)
    assert all('error' not in o and o['roofline']['frac'] > 0 and 0 < o['application']['frac'] < 1 for o in line['other_configs'])
    assert set(line['latency']) == {'cifar10_fp32_B8_nfe10_ms', 'sd15_fp16_B1_nfe10_ms'}
    # round 5: every line's dominant kernel carries its HBM traffic (PMC, with provenance), the fp16 lines an HBM-bound entry for the norm pass
    assert line['roofline']['traffic'] > 0 and line['roofline']['traffic_source']['file'] == 'profiles/r5_bench_pmc_hbm.json'
    for o in line['other_configs']:
        assert o['roofline']['traffic'] > 0 and o['roofline']['traffic_source']['file'].startswith('profiles/r5_pmc_hbm_'), o['config']
        if o['dtype'] == 'fp16':
            n = o['roofline_norm_act']
            assert n['bound'] == 'hbm' and 0.2 < n['frac'] < 1 and n['traffic'] > 0.9 * n['algorithmic_bytes_per_launch'], n


def test_committed_tile_table_matches_the_kernel_sources_and_covers_the_benchmarked_plans():
    """diff_sampler_amd/data/tile_table.json (package data; plan.Builder._autotune): measured on THESE kernel sources (else it would be ignored and plan builds would fall
    back to on-box timing races), and it holds every eligible fp16-activation launch of the two benchmarked fp16 plans -- built here on the
    CPU, where nothing is measured -- so that `bench.py` builds them without a single measurement launch."""
    sys.path.insert(0, ROOT)
    import diff_sampler_amd.arch as arch
    import diff_sampler_amd.ldm_arch as la
    from diff_sampler_amd import _lib, plan as plan_mod
    from diff_sampler_amd.engine import UNetEngine
    from diff_sampler_amd.ldm_engine import LDMUNetEngine
    entries = plan_mod.load_tile_table(force=True)
    assert plan_mod._TABLE['why'] is None, plan_mod._TABLE['why']
    assert len(entries) >= 300
    lib = _lib.load()
    spec = arch.edm_precond_spec(**dict(arch.NAMED_CONFIGS['imagenet64']))
    plans = [UNetEngine(spec, arch.init_params(spec, seed=0), device='cpu', use_fp16=True).plan(64, 64)]
    lspec = la.ldm_unet_spec(**dict(la.NAMED_LDM_CONFIGS['sd15']))
    plans.append(LDMUNetEngine(lspec, la.init_ldm_params(lspec, seed=0), device='cpu', use_fp16=True).plan(32, 1, 77))
    missing = n = 0
    for P in plans:
        for op in P.ops:
            if op.fn is lib.ds_conv2d_nhwc and op.keep[0].in_f16 and plan_mod._tile_neutral(op.keep[0]):
                a = op.keep[0]
                n += 1
                missing += str(plan_mod.Builder._tune_key(a, a.stride or 1)) not in entries
    assert n > 150 and missing == 0, (n, missing)


def test_sample_cli_under_two_gloo_ranks_writes_every_seed_exactly_once(tmp_path):
    """`python -m diff_sampler_amd.sample` as the reference launches it (torchrun, one process per device; sample.py:164-169, :268,
    torch_utils/distributed.py:14-31) under 2 gloo ranks on the CPU, in the CLI's launcher self-test mode (--stub: the real seed
    sharding, rank-0-first model barrier, per-batch barriers, PNG sink and output tree; no kernels, every image is a flat colour that
    encodes its seed).  The union of the PNGs must be the seed list, each seed written exactly once, by the rank the reference's
    partition assigns it to, in the reference's directory layout."""
    import numpy as np
    import PIL.Image
    from diff_sampler_amd import sample
    out = tmp_path / 'out'
    seeds = '0-20,1003,2500-2504'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(launch.free_port()), '-m', 'diff_sampler_amd.sample', '--stub', 'true', '--dataset_name', 'cifar10',
           '--solver', 'ipndm', '--num_steps', '6', '--batch', '4', '--seeds', seeds, '--outdir', str(out)]
    env = _clean_env()
    env['PYTHONPATH'] = ROOT + os.pathsep + env.get('PYTHONPATH', '')
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.count('Done.') == 1                              # only rank 0 talks (dist.print0)
    want = sample.parse_int_list(seeds)
    found = {}
    for d, _, files in os.walk(out):
        for f in files:
            assert f.endswith('.png')
            sd = int(f[:-4])
            assert sd not in found, f'seed {sd} written twice'
            assert os.path.basename(d) == f'{sd - sd % 1000:06d}'     # samples/.../<seed - seed % 1000>/<seed>.png (sample.py:313-316)
            found[sd] = os.path.join(d, f)
    assert sorted(found) == want
    for sd, path in found.items():
        img = np.asarray(PIL.Image.open(path))
        assert img.shape == (8, 8, 3) and (img == sd % 251).all()
    # both ranks had work, by the reference's partition
    parts = [sorted(int(s) for b in sample.shard_seeds(want, 4, r_, 2) for s in b) for r_ in range(2)]
    assert parts[0] and parts[1] and sorted(parts[0] + parts[1]) == want and not set(parts[0]) & set(parts[1])
