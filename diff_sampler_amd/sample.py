"""Batched generation CLI -- the reference's ``sample.py`` surface on the HIP engine.

Same click options, defaults, seed sharding, per-seed RNG, NFE bookkeeping and output tree as
diff-solvers-main/sample.py:125-320.  One CLI serves the three variants of the reference:
  * plain (diff-solvers-main): ``--solver ... --num_steps ...``;
  * GITS (gits-main/sample.py:135-170, :206-246): ``--dp True --metric/--coeff/--num_warmup/--solver_tea/--num_steps_tea``
    search the schedule on the device, then sample with ``teacher_t[dp_list]``;
  * AMED (amed-solver-main/sample.py:120-135, :160-185): ``--predictor_path`` loads the AMED predictor and EVERY solver
    setting (sampler_stu, num_steps, afs, max_order, predict_x0, lower_order_final, schedule_type/rho, guidance, dataset_name)
    is read back from it, exactly as the reference does; ``--predictor_path random:<seed> --random_init True`` builds a seeded
    predictor from the CLI's own options instead (there are no trained predictors in this environment).
Differences, all host-side:
  * the network object is an ``engine.EDMDenoiser`` (built from the unpickled EDM network when a pickle is given, or
    from the stated architecture with ``--random_init`` -- no pretrained weights exist in this environment);
  * uint8 conversion runs in a HIP kernel and PNG encoding is off the critical path only in the sense that it stays
    where the reference has it (one file per seed, ``<outdir>/<seed-seed%1000:06d>/<seed:06d>.png``);
  * ``torch.distributed`` is initialised here (env://, backend nccl = RCCL) instead of ``torch_utils.distributed``.
"""
from __future__ import annotations

import ast
import os
import pickle
import re
from typing import List

import torch

try:
    import click
except ImportError:                        # pragma: no cover
    click = None


# ------------------------------------------------------------------------------------------------------------------
class StackedRandomGenerator:
    """One generator per sample, seeded ``seed % 2**32`` (sample.py:22-36): a batch is reproducible per image
    regardless of how seeds are grouped into batches or sharded over ranks.

    The reference constructs B ``torch.Generator`` objects and issues B ``randn`` launches per batch.  Here float32 ``randn`` /
    ``randint`` on the GPU evaluate the same Philox4x32-10 streams for the whole batch in ONE launch (``ds_philox_randn`` /
    ``ds_philox_randint``: bit-identical to the per-generator calls, tests/test_hip_rng.py) and only track the generators' common
    Philox offset; anything else (CPU, other dtypes, ranges >= 2**32) goes through real per-seed generators as the reference does,
    fast-forwarded to the same offset."""

    def __init__(self, device, seeds):
        self.device = torch.device(device)
        self.seeds = [int(seed) % (1 << 32) for seed in seeds]
        self.offset = 0                      # Philox offset every generator of the stack has reached
        self._seeds_dev = None
        self._generators = None

    @property
    def generators(self):
        if self._generators is None:
            self._generators = [torch.Generator(self.device).manual_seed(s) for s in self.seeds]
            if self.offset and self.device.type == 'cuda':
                for g in self._generators:
                    g.set_offset(self.offset)
        return self._generators

    _fast_ok = {}            # device index -> did ds_philox_randn reproduce torch's own generators on this torch build / device?

    @classmethod
    def fast_path_verified(cls, device) -> bool:
        """The batched Philox kernel's Box-Muller is pinned to the device math of ONE torch / ROCm build (csrc/rng.hip); on another build
        latents could differ from the reference's per-generator ``randn`` by an ulp and seeds would silently stop reproducing.  So the
        first use on a device draws a small sample both ways -- at offset 0 and at a non-zero offset -- and the fast path is used only
        if they are bit-identical; otherwise real per-seed generators are (the reference's own path), with a one-line notice."""
        device = torch.device(device)
        key = device.index if device.index is not None else torch.cuda.current_device()
        ok = cls._fast_ok.get(key)
        if ok is None:
            from . import ops
            ok = True
            try:
                seeds, n = [3, 4000000007 % (1 << 32)], 1500
                sd = torch.tensor(seeds, dtype=torch.int64, device=device)
                for off in (0, 24):
                    got = torch.empty(len(seeds), n, dtype=torch.float32, device=device)
                    ops.philox_randn(sd, off, got, n)
                    for i, sv in enumerate(seeds):
                        g = torch.Generator(device).manual_seed(sv)
                        g.set_offset(off)
                        ok = ok and torch.equal(got[i], torch.randn(n, generator=g, device=device))
            except Exception as e:                                       # a missing library is reported by the samplers, not here
                print(f'StackedRandomGenerator: batched Philox check failed ({e}); using per-seed generators')
                ok = False
            if not ok:
                print('StackedRandomGenerator: ds_philox_randn does not reproduce torch.randn bit-for-bit on this torch build / device; '
                      'using per-seed torch generators (slower, reference-identical)')
            cls._fast_ok[key] = ok
        return ok

    def _fast(self, device, dtype):
        return (self._generators is None and self.device.type == 'cuda' and torch.device(device if device is not None else self.device).type == 'cuda'
                and dtype in (None, torch.float32) and self.fast_path_verified(self.device))

    def _dev_seeds(self):
        if self._seeds_dev is None:
            self._seeds_dev = torch.tensor(self.seeds, dtype=torch.int64, device=self.device)
        return self._seeds_dev

    def randn(self, size, **kwargs):
        assert size[0] == len(self.seeds)
        if self._fast(kwargs.get('device'), kwargs.get('dtype')) and kwargs.get('layout', torch.strided) == torch.strided:
            from . import ops
            n = 1
            for d in size[1:]:
                n *= int(d)
            out = torch.empty([len(self.seeds)] + [int(d) for d in size[1:]], dtype=torch.float32, device=self.device)
            self.offset += ops.philox_randn(self._dev_seeds(), self.offset, out, n)
            return out
        return torch.stack([torch.randn(size[1:], generator=g, **kwargs) for g in self.generators])

    def randn_like(self, input):
        return self.randn(input.shape, dtype=input.dtype, layout=input.layout, device=input.device)

    def randint(self, *args, size, **kwargs):
        assert size[0] == len(self.seeds)
        if (len(args) == 1 and len(size) == 1 and int(args[0]) < (1 << 32) and self._fast(kwargs.get('device'), None)
                and kwargs.get('dtype') in (None, torch.int64)):
            from . import ops
            out = torch.empty(len(self.seeds), dtype=torch.int32, device=self.device)
            self.offset += ops.philox_randint(self._dev_seeds(), self.offset, int(args[0]), out)
            return out.to(torch.int64)
        return torch.stack([torch.randint(*args, size=size[1:], generator=g, **kwargs) for g in self.generators])


def parse_int_list(s):
    """'1,2,5-10' -> [1, 2, 5, 6, 7, 8, 9, 10] (sample.py:42-52)."""
    if isinstance(s, list):
        return s
    out: List[int] = []
    for part in s.split(','):
        m = re.match(r'^(\d+)-(\d+)$', part)
        out.extend(range(int(m.group(1)), int(m.group(2)) + 1) if m else [int(part)])
    return out


def shard_seeds(seeds, max_batch_size: int, rank: int, world: int):
    """The reference partition (sample.py:167-169): ``ceil(len/(B*W))*W`` near-equal batches, rank r takes r::W."""
    num_batches = ((len(seeds) - 1) // (max_batch_size * world) + 1) * world
    all_batches = torch.as_tensor(seeds).tensor_split(num_batches)
    return all_batches[rank::world]


def compute_nfe(solver, num_steps, afs, denoise_to_zero, dataset_name, dp=False, guidance_rate=None):
    """sample.py:211-219; with a searched GITS schedule (``dp``) AFS INSERTS a step into the schedule instead of replacing
    one (gits-main/sample.py:231-241): free for the 1-NFE solvers, one extra evaluation for dpm / heun.  Classifier-free guidance
    doubles the count for ms_coco -- unconditionally in diff-solvers-main (:218), only for guidance rates outside {0, 1} in gits-main
    (:240-242; pass ``guidance_rate`` to get that rule)."""
    if solver in ('dpm', 'heun'):
        nfe = 2 * (num_steps - 1)
        if afs:
            nfe = nfe + 1 if dp else nfe - 1
    else:
        nfe = num_steps - 1
        if afs:
            nfe = nfe if dp else nfe - 1
    if denoise_to_zero:
        nfe += 1
    if dataset_name in ['ms_coco'] and (guidance_rate is None or guidance_rate not in (0., 1.)):
        nfe = 2 * nfe
    return nfe


AMED_SOLVER_FNS = dict(amed='amed_sampler', euler='euler_sampler', dpm='dpm_2_sampler', ipndm='ipndm_sampler', dpmpp='dpm_pp_sampler')


def find_predictor(predictor_path, exps_dir='./exps'):
    """amed-solver-main/sample.py:148-163: a value that does not end in 'pkl' is an experiment number -- the newest snapshot of
    ``exps/<5-digit number>-*/`` is used."""
    if predictor_path.endswith(('pkl', '.pt')) or not os.path.isdir(exps_dir):
        return predictor_path
    want = '0' * (5 - len(predictor_path)) + predictor_path
    for name in os.listdir(exps_dir):
        if name.split('-')[0] == want:
            best, best_idx = None, -1
            for ck in (f for f in os.listdir(os.path.join(exps_dir, name)) if f.endswith('pkl')):
                idx = int(ck.split('-')[-1].split('.')[0])
                if idx > best_idx:
                    best, best_idx = ck, idx
            if best is None:
                raise FileNotFoundError(f'experiment directory {os.path.join(exps_dir, name)!r} holds no predictor snapshot (*.pkl)')
            return os.path.join(exps_dir, name, best)
    return predictor_path


def load_predictor(predictor_path, device, random_init=False, **cli):
    """-> solvers_amed.AMEDPredictor.
      *.pkl          the reference's snapshot: ``pickle.load(f)['model']`` (needs training.networks importable), through
                     ``AMEDPredictor.from_module`` (weights by state_dict key, settings by attribute);
      *.pt           ``torch.save(dict(state_dict=..., **settings))`` -- the same content without pickled classes;
      random:<seed>  (with --random_init) seeded weights of the reference architecture (training/networks.py:107-117: 8x8 map,
                     64->128->4 encoder, 20->1 heads) with the settings taken from the CLI options."""
    from .solvers_amed import AMEDPredictor
    if predictor_path.startswith('random:'):
        if not random_init:
            raise ValueError("--predictor_path random:<seed> needs --random_init True (it is not a trained predictor)")
        seed = int(predictor_path.split(':', 1)[1])
        scale_dir = cli.get('scale_dir', 0.01) or 0
        scale_time = cli.get('scale_time', 0) or 0
        g = torch.Generator().manual_seed(seed)
        shapes = [('map_layer0.weight', (8, 8)), ('map_layer0.bias', (8,)), ('enc_layer0.weight', (128, 64)), ('enc_layer0.bias', (128,)),
                  ('enc_layer1.weight', (4, 128)), ('enc_layer1.bias', (4,)), ('fc_r.weight', (1, 20)), ('fc_r.bias', (1,))]
        if scale_dir:
            shapes += [('fc_scale_dir.weight', (1, 20)), ('fc_scale_dir.bias', (1,))]
        if scale_time:
            shapes += [('fc_scale_time.weight', (1, 20)), ('fc_scale_time.bias', (1,))]
        sd = {k: torch.randn(sh, generator=g) * (0.3 if k.endswith('weight') else 0.1) for k, sh in shapes}
        settings = dict(dataset_name=cli.get('dataset_name'), num_steps=cli.get('num_steps'), sampler_stu=cli.get('solver') or 'amed',
                        sampler_tea='heun', M=1, guidance_type=cli.get('guidance_type'), guidance_rate=cli.get('guidance_rate'),
                        schedule_type=cli.get('schedule_type', 'polynomial'), schedule_rho=cli.get('schedule_rho', 7),
                        afs=bool(cli.get('afs', False)), scale_dir=scale_dir, scale_time=scale_time, max_order=cli.get('max_order'),
                        predict_x0=cli.get('predict_x0', True), lower_order_final=cli.get('lower_order_final', True))
        return AMEDPredictor(sd, device=device, **settings)
    path = find_predictor(predictor_path)
    if path.endswith('.pt'):
        blob = torch.load(path, map_location='cpu')
        sd = blob.pop('state_dict')
        return AMEDPredictor(sd, device=device, **blob)
    with open(path, 'rb') as f:
        model = pickle.load(f)['model']
    return AMEDPredictor.from_module(model, device=device)


SOLVER_FNS = dict(euler='euler_sampler', heun='heun_sampler', dpm='dpm_2_sampler', ipndm='ipndm_sampler',
                  ipndm_v='ipndm_v_sampler', dpmpp='dpm_pp_sampler', unipc='unipc_sampler', deis='deis_sampler')


# ------------------------------------------------------------------------------------------------------------------
def create_model(dataset_name=None, model_path=None, random_init=False, device=None, seed=0, guidance_type=None, guidance_rate=None,
                 use_fp16=False):
    """EDM networks (cifar10 / ffhq / afhqv2 / imagenet64; sample.py:80-85) -> (net, 'edm'); Stable Diffusion v1.x latent
    U-Net under classifier-free guidance (ms_coco; sample.py:111-116) -> (net, 'ldm').

    use_fp16: the fp16-operand kernels (fp32 accumulation and storage).  The reference leaves `--use_fp16` unwired
    (sample.py:188-189): an EDM checkpoint's own `use_fp16` attribute decides (ImageNet-64 ADM: True) and the LDM sampler always runs
    under autocast (sample.py:296).  Here True selects that arithmetic; False (default) is exact fp32 for random-init / LDM nets and
    "follow the checkpoint" for a loaded EDM pickle."""
    from . import arch
    from .engine import EDMDenoiser
    if dataset_name == 'ms_coco':
        from . import ldm_arch
        from .ldm_engine import CFGDenoiser
        assert guidance_type == 'cfg', 'ms_coco samples with classifier-free guidance (sample.py:112)'
        spec = ldm_arch.ldm_unet_spec(**ldm_arch.NAMED_LDM_CONFIGS['sd15'])
        if random_init or model_path is None:
            params = ldm_arch.init_ldm_params(spec, seed=seed)
        else:                                   # SD checkpoint: the U-Net lives under 'model.diffusion_model.' (ddpm.py:1399)
            sd = torch.load(model_path, map_location='cpu')
            sd = sd.get('state_dict', sd)
            pre = 'model.diffusion_model.'
            params = {k[len(pre):]: v.float() for k, v in sd.items() if k.startswith(pre)}
        return CFGDenoiser(spec, params, device, guidance_rate=(7.5 if guidance_rate is None else guidance_rate), use_fp16=bool(use_fp16)), 'ldm'
    if dataset_name not in arch.NAMED_CONFIGS:
        raise ValueError(f'dataset {dataset_name!r}: only the EDM networks are in scope of the HIP engine '
                         f'({sorted(k for k in arch.NAMED_CONFIGS if not k.startswith("tiny"))} and ms_coco); CM / ADM-classifier-guided / LSUN-LDM models run on the reference')
    if random_init or model_path is None:
        net = EDMDenoiser.from_config(dataset_name, seed=seed, device=device, use_fp16=bool(use_fp16))
    else:
        with open(model_path, 'rb') as f:       # needs the reference's torch_utils/dnnlib importable for unpickling
            ref = pickle.load(f)['ema']
        net = EDMDenoiser.from_reference_module(ref, device=device, use_fp16=(True if use_fp16 else None))
    net.sigma_min, net.sigma_max = 0.002, 80.0
    return net, 'edm'


class PngSink:
    """Background PNG writer: the reference encodes every image with PIL in the sampling thread (sample.py:312-316), so the
    GPU idles while a batch is written.  Here the uint8 batch is handed to a small thread pool (PIL releases the GIL while
    compressing) and the next batch samples meanwhile; ``drain()`` waits for everything and re-raises the first failure."""

    def __init__(self, workers=4):
        from concurrent.futures import ThreadPoolExecutor
        self.pool = ThreadPoolExecutor(max_workers=workers)
        self.pending = []

    @staticmethod
    def _write(arr, seeds, outdir, subdirs):
        import PIL.Image
        for seed, img in zip(seeds, arr):
            seed = int(seed)
            d = os.path.join(outdir, f'{seed - seed % 1000:06d}') if subdirs else outdir
            os.makedirs(d, exist_ok=True)
            if img.shape[-1] == 1:                      # sample.py:314-315 of the reference: single channel -> mode 'L'
                PIL.Image.fromarray(img[:, :, 0], 'L').save(os.path.join(d, f'{seed:06d}.png'))
            else:
                PIL.Image.fromarray(img, 'RGB').save(os.path.join(d, f'{seed:06d}.png'))

    def submit(self, arr, seeds, outdir, subdirs=True, chunk=64):
        seeds = [int(s) for s in seeds]
        for i in range(0, len(seeds), chunk):
            self.pending.append(self.pool.submit(self._write, arr[i:i + chunk], seeds[i:i + chunk], outdir, subdirs))

    def drain(self):
        pending, self.pending = self.pending, []
        for f in pending:
            f.result()

    def close(self):
        self.drain()
        self.pool.shutdown()


def save_images(images: torch.Tensor, batch_seeds, outdir, subdirs=True, sink: 'PngSink' = None):
    """uint8 NHWC conversion on the GPU (sample.py:311), then one PNG per seed (sample.py:312-316) -- written by ``sink`` in
    the background when one is given."""
    from . import ops
    B, C, H, W = images.shape
    u8 = torch.empty(B, H, W, C, dtype=torch.uint8, device=images.device)
    ops.quantize_u8_nhwc(images.contiguous(), u8, B, C, H, W)
    arr = u8.cpu().numpy()
    if sink is not None:
        sink.submit(arr, batch_seeds, outdir, subdirs)
    else:
        PngSink._write(arr, [int(s) for s in batch_seeds], outdir, subdirs)


def save_grid(images: torch.Tensor, outdir):
    """``make_grid(images, nrows, padding=0)`` + ``save_image`` of the reference (sample.py:305-309): ``int(sqrt(B))`` images per
    row, ``ceil(B / per_row)`` rows, a partial last row is padded with zeros (black), quantisation ``x*255 + 0.5`` clamped."""
    import PIL.Image
    x = torch.clamp(images / 2 + 0.5, 0, 1)
    B, C, H, W = x.shape
    per_row = min(max(1, int(B ** 0.5)), B)
    n_rows = -(-B // per_row)
    if n_rows * per_row > B:
        x = torch.cat([x, torch.zeros(n_rows * per_row - B, C, H, W, dtype=x.dtype, device=x.device)], dim=0)
    if C == 1:
        x = x.expand(-1, 3, -1, -1)                     # make_grid replicates single-channel images to 3 channels
    rows = [torch.cat(list(x[r * per_row:(r + 1) * per_row]), dim=2) for r in range(n_rows)]
    grid = (torch.cat(rows, dim=1) * 255 + 0.5).clamp(0, 255).to(torch.uint8).permute(1, 2, 0).cpu().numpy()
    os.makedirs(outdir, exist_ok=True)
    PIL.Image.fromarray(grid, 'RGB').save(os.path.join(outdir, 'grid.png'))


def _dist():
    import torch.distributed as dist
    if 'RANK' in os.environ and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend='nccl' if torch.cuda.is_available() else 'gloo', init_method='env://')
    if dist.is_initialized():
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


def run(dataset_name=None, max_batch_size=64, seeds='0-63', grid=False, outdir=None, subdirs=True, t_steps=None, model_path=None,
        random_init=False, device=None, predictor_path=None, **solver_kwargs):
    """Body of the CLI, importable (the tests call it directly)."""
    from . import solvers, solver_utils
    seeds = parse_int_list(seeds)
    dist, rank, world = _dist()
    # Launcher self-test (tests/test_launch_cpu.py, hidden --stub): gloo ranks on the CPU, NO kernels and no arithmetic -- every
    # "image" is a flat colour that encodes its seed, written through the real sharding / per-batch barriers / PNG sink / output tree.
    # It exists so that the N > 1 host path is exercised where there is no GPU; it is not a CPU mode of the sampler.
    stub = bool(solver_kwargs.pop('stub', False))
    if stub:
        device = torch.device('cpu')
    else:
        device = torch.device('cuda', int(os.environ.get('LOCAL_RANK', 0))) if device is None else torch.device(device)
        torch.cuda.set_device(device)
    rank_batches = shard_seeds(seeds, max_batch_size, rank, world)
    log = print if rank == 0 else (lambda *a, **k: None)

    if dist is not None and rank != 0:
        dist.barrier()                                              # rank 0 goes first (sample.py:183-193)
    predictor = None
    if predictor_path is not None:
        # AMED (amed-solver-main/sample.py:146-185): the predictor decides the dataset and every solver setting
        log(f'Loading AMED predictor from "{predictor_path}"...')
        predictor = load_predictor(predictor_path, device, random_init=random_init, dataset_name=dataset_name, **solver_kwargs)
        prompt = solver_kwargs.get('prompt')
        solver_kwargs = {k: v for k, v in solver_kwargs.items() if v is not None}
        solver_kwargs.update(AMED_predictor=predictor, solver=predictor.sampler_stu, num_steps=predictor.num_steps,
                             guidance_type=predictor.guidance_type, guidance_rate=predictor.guidance_rate, afs=predictor.afs,
                             denoise_to_zero=False, max_order=predictor.max_order, predict_x0=predictor.predict_x0,
                             lower_order_final=predictor.lower_order_final, schedule_type=predictor.schedule_type,
                             schedule_rho=predictor.schedule_rho, prompt=prompt)
        solver_kwargs['dataset_name'] = dataset_name = predictor.dataset_name
        if solver_kwargs['solver'] not in AMED_SOLVER_FNS:
            raise ValueError(f"AMED predictor was trained for sampler {solver_kwargs['solver']!r}; supported: {sorted(AMED_SOLVER_FNS)}")
        solver_kwargs.pop('dp', None)
        t_steps = None
    if dataset_name is None:
        raise ValueError('--dataset_name is required (or --predictor_path, which carries it)')
    if stub:
        import types
        net = types.SimpleNamespace(img_channels=3, img_resolution=8, label_dim=0, sigma_min=0.002, sigma_max=80.)
        solver_kwargs['model_source'] = 'stub'
    else:
        net, solver_kwargs['model_source'] = create_model(dataset_name, model_path, random_init, device,
                                                          guidance_type=solver_kwargs.get('guidance_type'),
                                                          guidance_rate=solver_kwargs.get('guidance_rate'),
                                                          use_fp16=solver_kwargs.get('use_fp16', False))
    ldm = solver_kwargs['model_source'] == 'ldm'
    cond_table = None
    if ldm and solver_kwargs.get('condition_path'):
        # text-encoder states computed elsewhere (CLIP is not on the sampling path): {'c': [N, L, 768] indexed by seed, 'uc': [1, L, 768]}
        cond_table = torch.load(solver_kwargs['condition_path'], map_location='cpu')
    if dist is not None and rank == 0:
        dist.barrier()

    solver_kwargs.setdefault('num_steps', 6)
    for k, v in dict(afs=False, denoise_to_zero=False, schedule_type='polynomial', schedule_rho=7, predict_x0=True,
                     lower_order_final=True, variant='bh2', deis_mode='tab', max_order=None, prompt=None,
                     guidance_type=None, guidance_rate=None, return_inters=False).items():
        solver_kwargs.setdefault(k, v)
    solver_kwargs['sigma_min'], solver_kwargs['sigma_max'] = net.sigma_min, net.sigma_max
    dp = bool(solver_kwargs.get('dp'))
    gits_cli = 'dp' in solver_kwargs         # the option only exists in gits-main/sample.py: its NFE rule for guided sampling applies (:240-242)
    if predictor is not None:
        # amed-solver-main/sample.py:196-199: two evaluations per step (the first one free under AFS); the samplers build
        # their own schedule from the predictor's schedule_type / rho
        solver = solver_kwargs['solver']
        nfe = 2 * (solver_kwargs['num_steps'] - 1) - 1 if solver_kwargs['afs'] else 2 * (solver_kwargs['num_steps'] - 1)
        nfe = 2 * nfe if dataset_name in ['ms_coco'] else nfe
        solver_kwargs['nfe'] = nfe
        from . import solvers_amed
        sampler_fn = getattr(solvers_amed, AMED_SOLVER_FNS[solver])
        log('Solver settings:', {k: v for k, v in solver_kwargs.items() if k != 'AMED_predictor' and v is not None})
    else:
        if t_steps is None and dp:
            # GITS (gits-main/sample.py:206-219): search the schedule on the device, then sample with teacher_t[dp_list];
            # the reference's defaults: teacher ipndm on 21 points
            from . import gits_utils
            for k, v in dict(metric='dev', coeff=1.15, num_warmup=256, solver_tea='ipndm', num_steps_tea=21).items():
                if solver_kwargs.get(k) is None:
                    solver_kwargs[k] = v
            dp_list = gits_utils.get_dp_list(net, device, dataset_name=dataset_name, max_batch_size=max_batch_size, **solver_kwargs)
            t_steps = solver_utils.get_schedule(solver_kwargs['num_steps_tea'], net.sigma_min, net.sigma_max, device=device,
                                                schedule_type=solver_kwargs['schedule_type'], schedule_rho=solver_kwargs['schedule_rho'],
                                                net=net, dp_list=dp_list)
            log('Selected dp_list:', dp_list)
            log('Selected time schedule: ', [round(float(v), 4) for v in t_steps])
        elif t_steps is None:
            t_steps = solver_utils.get_schedule(solver_kwargs['num_steps'], net.sigma_min, net.sigma_max, device=device,
                                                schedule_type=solver_kwargs['schedule_type'], schedule_rho=solver_kwargs['schedule_rho'],
                                                net=net)
        else:
            if dp:
                log('t_steps is specified, ignored DP')             # gits-main/sample.py:220-221
            t_list = ast.literal_eval(t_steps) if isinstance(t_steps, str) else list(t_steps)
            t_steps = torch.tensor(t_list, device=device)           # all-int lists give an int64 tensor, as in the reference
            solver_kwargs['num_steps'] = t_steps.shape[0]
            solver_kwargs['sigma_max'], solver_kwargs['sigma_min'] = t_list[0], t_list[-1]
            solver_kwargs['schedule_type'] = solver_kwargs['schedule_rho'] = None
            solver_kwargs['dp'] = dp = False
            log('Pre-specified t_steps:', t_list)
        solver_kwargs['t_steps'] = t_steps
        solver = solver_kwargs['solver']
        nfe = compute_nfe(solver, solver_kwargs['num_steps'], solver_kwargs['afs'], solver_kwargs['denoise_to_zero'], dataset_name, dp=dp,
                          guidance_rate=((7.5 if solver_kwargs['guidance_rate'] is None else solver_kwargs['guidance_rate']) if gits_cli else None))
        solver_kwargs['nfe'] = nfe
        sampler_fn = getattr(solvers, SOLVER_FNS[solver])
        if solver == 'deis':
            solver_kwargs['coeff_list'] = solver_utils.get_deis_coeff_list(t_steps, solver_kwargs['max_order'],
                                                                           deis_mode=solver_kwargs['deis_mode'])
    if outdir is None:
        outdir = os.path.join(f'./samples/grids/{dataset_name}' if grid else f'./samples/{dataset_name}', f'{solver}_nfe{nfe}')
    log(f'Generating {len(seeds)} images to "{outdir}"...')

    n_done = 0
    sink = PngSink()
    for batch_seeds in rank_batches:
        if dist is not None:
            dist.barrier()                                          # per-batch barrier, as the reference (sample.py:268)
        B = len(batch_seeds)
        if B == 0:
            continue
        if stub:
            import numpy as np
            arr = np.stack([np.full((8, 8, 3), int(sd) % 251, dtype=np.uint8) for sd in batch_seeds])
            sink.submit(arr, batch_seeds, outdir, subdirs)
            n_done += B
            continue
        rnd = StackedRandomGenerator(device, batch_seeds)
        latents = rnd.randn([B, net.img_channels, net.img_resolution, net.img_resolution], device=device)
        class_labels = None
        if net.label_dim and not ldm:
            class_labels = torch.eye(net.label_dim, device=device)[rnd.randint(net.label_dim, size=[B], device=device)]
        with torch.no_grad():
            if ldm:
                if cond_table is not None:
                    c = cond_table['c'][torch.as_tensor(batch_seeds) % cond_table['c'].shape[0]].to(device)
                    uc = cond_table['uc'].to(device).expand(B, -1, -1)
                else:       # no text encoder here: seeded N(0,1) states of the CLIP shape (BASELINE config 5)
                    c = rnd.randn([B, 77, net.spec.context_dim], device=device)
                    uc = torch.randn(1, 77, net.spec.context_dim, generator=torch.Generator().manual_seed(0)).to(device).expand(B, -1, -1)
                if solver_kwargs['guidance_rate'] == 1.0:
                    uc = None
                images = sampler_fn(net, latents, condition=c, unconditional_condition=uc, **solver_kwargs)
            else:
                images = sampler_fn(net, latents, class_labels=class_labels, **solver_kwargs)
        if solver_kwargs.get('return_inters'):
            images = images[-1]
        if ldm:
            # latents [B, 4, 64, 64]: decoding them is the VAE's job (net.model.decode_first_stage, sample.py:303), not on this path
            import numpy as np
            for seed, z in zip(batch_seeds, images.cpu().numpy()):
                seed = int(seed)
                d = os.path.join(outdir, f'{seed - seed % 1000:06d}') if subdirs else outdir
                os.makedirs(d, exist_ok=True)
                np.save(os.path.join(d, f'{seed:06d}.npy'), z)
        elif grid:
            save_grid(images, outdir)
        else:
            save_images(images, batch_seeds, outdir, subdirs, sink=sink)
        n_done += B
    sink.close()                                                    # every PNG is on disk before the final barrier
    if dist is not None:
        dist.barrier()
    log('Done.')
    return outdir, n_done


if click is not None:
    @click.command()
    @click.option('--dataset_name', help='Name of the dataset (AMED: read from the predictor)', metavar='STR', type=str, default=None)
    @click.option('--predictor_path', help='AMED: path (.pkl / .pt), experiment number, or random:<seed> with --random_init', metavar='DIR', type=str, default=None)
    @click.option('--scale_dir', help='AMED random:<seed> predictor: scale_dir', type=float, default=None)
    @click.option('--scale_time', help='AMED random:<seed> predictor: scale_time', type=float, default=None)
    @click.option('--model_path', help='Network filepath', metavar='PATH|URL', type=str)
    @click.option('--batch', 'max_batch_size', help='Maximum batch size', metavar='INT', type=click.IntRange(min=1), default=64, show_default=True)
    @click.option('--seeds', help='Random seeds (e.g. 1,2,5-10)', metavar='LIST', type=parse_int_list, default='0-63', show_default=True)
    @click.option('--prompt', help='Prompt for Stable Diffusion sampling', metavar='STR', type=str)
    @click.option('--solver', help='Name of the solver', metavar='many solvers', type=click.Choice(list(SOLVER_FNS) + ['amed']))
    @click.option('--num_steps', help='Number of sampling steps', metavar='INT', type=click.IntRange(min=1), default=6, show_default=True)
    @click.option('--afs', help='Whether to use AFS', metavar='BOOL', type=bool, default=False, show_default=True)
    @click.option('--guidance_type', help='Guidance type', type=click.Choice(['cg', 'cfg', 'uncond', None]), default=None, show_default=True)
    @click.option('--guidance_rate', help='Guidance rate', type=float)
    @click.option('--denoise_to_zero', help='Whether to denoise from the last time step to 0', type=bool, default=False)
    @click.option('--return_inters', help='Whether to save intermediate outputs', metavar='BOOL', type=bool, default=False)
    @click.option('--use_fp16', help='Whether to use mixed precision', metavar='BOOL', type=bool, default=False)
    @click.option('--max_order', help='Max order for solvers', metavar='INT', type=click.IntRange(min=1))
    @click.option('--predict_x0', help='Whether to use data prediction mode', metavar='BOOL', type=bool, default=True)
    @click.option('--lower_order_final', help='Whether to lower the order at final stages', metavar='BOOL', type=bool, default=True)
    @click.option('--variant', help='Type of UniPC solver', metavar='STR', type=click.Choice(['bh1', 'bh2']), default='bh2')
    @click.option('--deis_mode', help='Type of DEIS solver', metavar='STR', type=click.Choice(['tab', 'rhoab']), default='tab')
    @click.option('--sigma_min', help='Lowest noise level', metavar='FLOAT', type=click.FloatRange(min=0, min_open=True), default=0.002)
    @click.option('--sigma_max', help='Highest noise level', metavar='FLOAT', type=click.FloatRange(min=0, min_open=True), default=80.)
    @click.option('--schedule_type', help='Time discretization schedule', metavar='STR', type=click.Choice(['polynomial', 'logsnr', 'time_uniform', 'discrete']), default='polynomial', show_default=True)
    @click.option('--schedule_rho', help='Time step exponent', metavar='FLOAT', type=click.FloatRange(min=0, min_open=True), default=7, show_default=True)
    @click.option('--t_steps', help='Pre-specified time schedule', metavar='STR', type=str, default=None)
    @click.option('--outdir', help='Where to save the output images', metavar='DIR', type=str)
    @click.option('--grid', help='Whether to make grid', type=bool, default=False)
    @click.option('--subdirs', help='Create subdirectory for every 1000 seeds', type=bool, default=True, is_flag=True)
    @click.option('--condition_path', help='ms_coco: torch file with precomputed text-encoder states {c: [N,L,768], uc: [1,L,768]}', type=str, default=None)
    @click.option('--random_init', help='Use a random-init network of the named architecture (no checkpoint)', type=bool, default=False)
    # GITS options (gits-main/sample.py:159-165)
    @click.option('--dp', help='Whether to search the time schedule with dynamic programming (GITS)', type=bool, default=False)
    @click.option('--metric', help='Metric of the GITS cost matrix', type=click.Choice(['l1', 'l2', 'dev']), default='dev')
    @click.option('--coeff', help='GITS coefficient', type=float, default=1.15)
    @click.option('--num_warmup', help='Number of warm-up trajectories', type=click.IntRange(min=1), default=256)
    @click.option('--solver_tea', help='Teacher solver', type=click.Choice(['euler', 'ipndm', 'ipndm_v', 'heun', 'dpm', 'dpmpp', 'deis']), default='ipndm', show_default=True)
    @click.option('--num_steps_tea', help='Number of timestamps for teacher', type=click.IntRange(min=1), default=21, show_default=True)
    @click.option('--stub', help='launcher self-test: gloo ranks on the CPU, no kernels (tests/test_launch_cpu.py)', type=bool, default=False, hidden=True)
    def main(**kw):
        run(**kw)

    if __name__ == '__main__':
        main()
