#!/usr/bin/env python
"""Headline benchmark: images/sec at NFE=10 on the EDM CIFAR-10 denoiser (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          # N > 1: starts the N ranks itself (one per GPU, RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W              # the driver's form; WORLD_SIZE must equal --gpus

`--gpus` is checked, never trusted: WORLD_SIZE != --gpus, fewer visible GPUs than ranks, or a communicator that counts a
different number of ranks all exit non-zero (diff_sampler_amd/launch.py); `n_gpus` in the line comes from an all-reduce.

One "step" = one full sampler call (DPM-Solver++(2M), 10 network evaluations, logSNR schedule -- BASELINE configs[1]
at the NFE the metric is quoted on) over a batch of synthetic N(0,1) latents already resident in HBM, on the
random-init (signal-carrying) CIFAR-10 SongUNet.  Ranks shard images, nothing is exchanged during sampling
(SURVEY.md section 8e), so scaling is weak: every rank runs ``--batch`` images per step.

Prints ONE JSON line on rank 0 with the driver's contract fields plus
  roofline         the dominant kernel (largest share of GPU time; the fp32-MFMA LDS-halo 3x3 convolution):
                   achieved = algorithmic FLOPs of its launches in one sampler step / their summed duration, measured
                   with HIP events on the launch stream in an instrumented replay of the timed workload;
                   peak = 157.3 TFLOP/s (MI355X fp32 matrix pipe, MI355X_MICROARCH.md);
                   traffic = HBM bytes per launch from the PMC pass committed under profiles/ (FETCH_SIZE doubled per
                   the guide's gfx950 correction + WRITE_SIZE), or null when that file is absent
  roofline_update  the headline solver's fused update kernel (ds_dpmpp_x0_step) against HBM 8 TB/s at the benchmark's batch
                   (latency-bound there: 3 MB per operand), and
  roofline_update_large_batch  the same kernel measured in-process at 16 384 images per launch, where it is bandwidth-bound
                   (roofline_update_large_batch_other_solver: the Adams-Bashforth family's ds_solver_update, iPNDM order 4)
  kernels          time share of every kernel class in the instrumented step
  launch_modes     the same sampler call eager vs replayed from one captured hipGraph, at B=8 and at the benchmark batch
  throughput_by_batch  (cifar10) the same call over SURVEY 8d's batch sweep {64, 256, 1024, 4096} (informational; 2 timed calls, 1 at 4096)
  other_configs    (N = 1, default run) the other benchmarked configurations / arithmetic modes, each timed in THIS process under the
                   driver's clock: ImageNet-64 fp16 B=64 iPNDM-4, SD-1.5 fp16 B=16, FFHQ-64 fp32 B=128, CIFAR-10 fp16x3 B=256 -- value,
                   value_min / value_max, ms_per_step (MEDIAN of 5 individually timed calls after 2 warm-ups), the dominant kernel's own roofline and the whole-application fraction of
                   the matrix peak of the datatype used (SURVEY 8d FLOPs per evaluation x evaluations / time / peak)
  latency          small-batch sampler calls (BASELINE config 1 is B = 8): CIFAR-10 B=8 NFE=10 and SD-1.5 fp16 B=1 NFE=10, ms per call
  cpu_baseline     the oracle (CPU restatement of the reference, ``oracle/``; the real reference when /root/reference is
                   importable) timed on this host's cores on a bounded sample of the same workload, at the best of a
                   thread-count sweep (cores = the thread count used)
  multi_gpu        (N > 1) per-rank ms/step min/max (barrier skew) and the timed FID moment all-reduce (16 KiB + 32 MiB
                   fp64 SUM, fid.py:74-75) with algbw / busbw against the 153 GB/s xGMI link.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3
PEAK_FP16_MFMA_TFLOPS = 2500.0      # dense fp16/bf16 MFMA peak (MI355X_MICROARCH.md; measured 2495)
# What the matrix pipes SUSTAIN on random operands (the part lowers its clock under data toggling; tools/probes/mfma32_data.hip,
# mfma16_pattern.hip; profiles/r3_probe_*): informational, `roofline.peak` stays the nominal figure
SUSTAINED_RANDOM_FP32_TFLOPS, SUSTAINED_RANDOM_FP16_TFLOPS = 145.5, 1720.0
PEAK_HBM_GBS = 8000.0
# algorithmic GFLOP per image and network evaluation (2 x MAC; FlopCounterMode on the reference modules)
GFLOP_PER_EVAL = {'cifar10': 42.38, 'ffhq': 83.73, 'afhqv2': 83.73, 'imagenet64': 219.33, 'sd15': 803.27}
OTHER_CONFIGS = (('imagenet64', 'fp16', 64), ('sd15', 'fp16', 16), ('ffhq', 'fp32', 128), ('cifar10', 'fp16x3', 256))
WORKLOAD_NAMES = {'cifar10': 'EDM CIFAR-10 32x32 SongUNet (55.7M params)', 'ffhq': 'EDM FFHQ-64 SongUNet (61.8M params)',
                  'afhqv2': 'EDM AFHQv2-64 SongUNet', 'imagenet64': 'EDM ImageNet-64 DhariwalUNet (295.9M params)',
                  'sd15': 'Stable Diffusion v1.5 64x64x4 latent U-Net (859.5M params)'}
SOLVER_NAMES = {'dpmpp': 'DPM-Solver++(2M) logSNR', 'euler': 'Euler', 'ipndm': 'iPNDM-4', 'heun': 'Heun'}
LDM_SOLVER_NAME = 'DPM-Solver++(2M) eps-prediction, discrete rho=1, CFG 7.5 (2 U-Net images per latent; new text conditions on every call: the context K / V projections are inside the timed region)'
PMC_FILE = os.path.join(ROOT, 'profiles', 'r6_bench_pmc_hbm.json')
PMC_NOTE = {}
KERNEL_NAMES = {0: 'igemm_f32_kernel<0> (generic gather conv / 1x1 / linear)', 128: 'conv3x3_halo_kernel<2> (128-pixel tiles)',
                1284: 'conv3x3_halo_kernel<2, WN=4, NT=1> (128-pixel tiles on 8 waves of 64 x 32)',
                256: 'conv3x3_halo_kernel<4> (256-pixel tiles)', 2561: 'gemm_dma8_kernel (256x128 tiles, LDS-DMA, 1x1 / linear)',
                2562: 'conv3x3_halo2_kernel<fp16 operands> (256-pixel tiles, v_mfma_f32_32x32x16_f16)',
                2563: 'conv3x3_halo2_kernel<split fp16 hi/lo operands, fp32-emulated> (256-pixel tiles, 3 x v_mfma_f32_32x32x16_f16 per product)',
                2564: 'gemm_f16_kernel (1x1 / Linear, fp16 operands, 256 x 128 tiles, v_mfma_f32_32x32x16_f16)',
                2565: 'conv3x3_halo_kernel<4, NT=4> (256-pixel x 256-channel tiles, 64 x 128 per wave)',
                2568: 'conv3x3_halo_kernel<4, NT=3> (256-pixel x 192-channel tiles, 64 x 96 per wave)',
                2566: 'conv3x3_f16dma_kernel (fp16 activations, both operands by LDS-DMA, 256-pixel x 64..256-channel tiles, v_mfma_f32_32x32x16_f16)',
                2572: 'conv3x3_f16dma_kernel<NORM> (fp16 activations by LDS-DMA, GroupNorm affine + SiLU applied to the LDS halo in place, v_mfma_f32_32x32x16_f16)',
                2569: 'conv3x3_f16dmah_kernel (fp16 activations, 32-channel half slabs, four waves on 128-pixel x 64..256-channel tiles, two workgroups per CU)',
                2567: 'gemm_f16dma_kernel (1x1 / Linear on fp16 activations, both operands by LDS-DMA, 256 x 64..256 tiles, v_mfma_f32_32x32x16_f16)',
                2570: 'conv3x3_thin_kernel (network head: 3x3 with <= 4 output channels, VALU, fused GroupNorm + SiLU)',
                2571: 'gemm_f16dma_kernel<GATHER> (3x3 stride-2 Downsample on fp16 activations: gathered A tile, both operands by LDS-DMA)',
                2573: 'gemv_rows_kernel (Linear on <= 4 rows: the embedding path, a wave per output column)'}
# roofline peaks by kernel: the split mode issues three fp16 MFMAs per algorithmic multiply-add, so its ceiling in ALGORITHMIC FLOPs is a third
KERNEL_PEAK = {2562: PEAK_FP16_MFMA_TFLOPS, 2563: PEAK_FP16_MFMA_TFLOPS / 3, 2564: PEAK_FP16_MFMA_TFLOPS, 2566: PEAK_FP16_MFMA_TFLOPS, 2572: PEAK_FP16_MFMA_TFLOPS, 2567: PEAK_FP16_MFMA_TFLOPS, 2569: PEAK_FP16_MFMA_TFLOPS,
               2571: PEAK_FP16_MFMA_TFLOPS}
PMC_KEYS = {0: 'void igemm::igemm_f32_kernel<0>(igemm::KParams)', 1284: 'void igemm::conv3x3_halo_kernel<2, true, 4, 0, 1>(igemm::KParams)', 128: 'void igemm::conv3x3_halo_kernel<2, true, 2, 0, 2>(igemm::KParams)',
            256: 'void igemm::conv3x3_halo_kernel<4, true, 2, 0, 2>(igemm::KParams)', 2561: 'igemm::gemm_dma8_kernel(igemm::KParams)',
            2562: None, 2563: None, 2564: None, 2566: None, 2572: None, 2567: None, 2568: None, 2569: None, 2570: None, 2571: None, 2573: None, 2565: 'void igemm::conv3x3_halo_kernel<4, true, 2, 160, 4>(igemm::KParams)'}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--batch', type=int, default=None, help='images per GPU per sampler call (default: 256 cifar10, 128 ffhq, 64 imagenet64, 16 sd15 -- the batches the kept lines in profiles/ are quoted on)')
    ap.add_argument('--nfe', type=int, default=10)
    ap.add_argument('--solver', default=None, choices=['dpmpp', 'euler', 'ipndm', 'heun'],
                    help='default: dpmpp (the headline, BASELINE config 2; also ffhq, sd15); ipndm for imagenet64 (BASELINE config 3: iPNDM order 4)')
    ap.add_argument('--config', default='cifar10')
    ap.add_argument('--dtype', default='fp32', choices=['fp32', 'fp16', 'fp16x3'],
                    help="fp16 = the reference's use_fp16 / autocast mode (configs 3 and 5): fp16 operands in the 3x3 convolutions, 1x1 / Linear layers and attention, fp16 activation storage, fp32 accumulation; fp16x3 = fp32 EMULATED in the 3x3 convolutions by split fp16 hi/lo operands (3 MFMA products, fp32 tolerances)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--graph', action='store_true', help='replay the sampler call from a captured hipGraph')
    ap.add_argument('--cpu-batch', type=int, default=8)
    ap.add_argument('--cpu-calls', type=int, default=2)
    ap.add_argument('--cpu-threads', default='sweep', help="'sweep' (8,16,32; best reported) or a thread count")
    ap.add_argument('--no-launch-modes', action='store_true', help='skip the eager-vs-hipGraph comparison')
    ap.add_argument('--no-batch-sweep', action='store_true', help='skip the throughput_by_batch sweep (SURVEY 8d config 2: 64 / 256 / 1024 / 4096 images per call)')
    ap.add_argument('--sweep-batches', type=lambda t: [int(v) for v in t.split(',') if v], default=[64, 1024, 4096],
                    help='batches of throughput_by_batch next to --batch (comma separated)')
    ap.add_argument('--no-other-configs', action='store_true', help='skip the other_configs / latency measurements of the default run')
    ap.add_argument('--stub', action='store_true', help=argparse.SUPPRESS)   # launcher self-test: gloo ranks on CPU, no kernels
    args = ap.parse_args(argv)
    if args.solver is None:
        args.solver = 'ipndm' if args.config == 'imagenet64' else 'dpmpp'
    if args.batch is None:
        args.batch = {'ffhq': 128, 'imagenet64': 64, 'sd15': 16}.get(args.config, 256)
    return args


_LDM_CALLS = [0]


def _next_conditions(ldm):
    """`ldm` = a (condition, unconditional_condition) tuple, or a LIST of such tuples used in rotation: every sampler call then sees text
    conditions different from the previous call's, so the denoiser's per-context cache (ldm_engine.CFGDenoiser: the 32 cross-attention K / V
    projections of a context are computed once per context tensor) misses on every call and the projections are INSIDE the timed region --
    what a generation service that receives new prompts per batch pays (sample.py:276-301 encodes the prompts of every batch)."""
    if isinstance(ldm, list):
        _LDM_CALLS[0] += 1
        return ldm[_LDM_CALLS[0] % len(ldm)]
    return ldm


def sampler_call(solvers, solver, net, latents, nfe, ldm=None):
    if ldm is not None:
        ldm = _next_conditions(ldm)
        # BASELINE config 5: DPM-Solver++(2M) noise prediction, discrete rho=1 schedule; every step is one CFG-doubled
        # evaluation = 2 NFE (sample.py:218), so NFE=10 is num_steps=6
        return solvers.dpm_pp_sampler(net, latents, condition=ldm[0], unconditional_condition=ldm[1], num_steps=nfe // 2 + 1,
                                      sigma_min=net.sigma_min, sigma_max=net.sigma_max, schedule_type='discrete', schedule_rho=1,
                                      max_order=2, predict_x0=False, lower_order_final=True)
    if solver == 'dpmpp':
        return solvers.dpm_pp_sampler(net, latents, num_steps=nfe + 1, sigma_min=0.002, sigma_max=80., schedule_type='logsnr',
                                      schedule_rho=7, max_order=2, predict_x0=True, lower_order_final=True)
    if solver == 'euler':
        return solvers.euler_sampler(net, latents, num_steps=nfe + 1, sigma_min=0.002, sigma_max=80.)
    if solver == 'ipndm':
        return solvers.ipndm_sampler(net, latents, num_steps=nfe + 1, sigma_min=0.002, sigma_max=80., max_order=4)
    if solver == 'heun':
        return solvers.heun_sampler(net, latents, num_steps=nfe // 2 + 1, sigma_min=0.002, sigma_max=80.)
    raise ValueError(solver)


def instrumented_pass(net, solvers, solver, latents, nfe, ldm=None):
    """Replay one sampler call with every launch bracketed by HIP events on the launch stream.
    Returns {kernel class: [time_ms, launches, algorithmic flops]}."""
    from diff_sampler_amd import _lib, engine, ops
    rec = {}
    plan_run = engine._Plan.run
    upd = ops.solver_update
    thr = ops.dynamic_threshold
    lib = _lib.load()

    def add(kind, ms, flops=0.0):
        r = rec.setdefault(kind, [0.0, 0, 0.0])
        r[0] += ms
        r[1] += 1
        r[2] += flops

    def classify(op):
        if op.fn is lib.ds_conv2d_nhwc:
            a = op.keep[0]
            kid = lib.ds_conv_kernel_id(C.byref(a))
            fl = 2.0 * a.n * a.h * a.w * a.cout * (a.taps * (a.c0 + a.c1) + a.ec0 + a.ec1)
            return ('conv', kid), fl
        if op.fn is lib.ds_gemm_nt_batched:
            g = op.keep[0]
            return 'igemm_f32_kernel<1> (batched GEMM)', 2.0 * g.m * g.n * g.k * g.batch * g.heads
        if op.fn is lib.ds_attention:
            t = op.keep[0]
            return 'flash_attn_kernel (fused attention)', 4.0 * t.batch * t.heads * t.sq * t.skv * t.d
        if op.fn is lib.ds_attention_f16:
            t = op.keep[0]
            return 'flash_attn_f16_kernel (fused attention, fp16 operands)', 4.0 * t.batch * t.heads * t.sq * t.skv * t.d
        if op.fn is lib.ds_layernorm_rows or op.fn is lib.ds_layernorm_rows_f16 or op.fn is lib.ds_layernorm_rows_f16io:
            return 'layernorm_rows_kernel', 0.0
        if op.fn is lib.ds_gn_stats:
            return 'gn_stats_kernel', 0.0
        if op.fn is lib.ds_gn_finalize:
            return 'gn_from_partials_kernel (GroupNorm statistics from the conv epilogues)', 0.0
        if op.fn is lib.ds_norm_act:
            a = op.keep[0]
            k = {1: 0.25, 2: 4.0}.get(a.resample, 1.0)           # output pixels per input pixel (2x2 box filter down / nearest x2 up)
            px = a.n * a.h * a.w
            byts = px * (a.c0 * (2 if a.in_f16 & 1 else 4) + a.c1 * (2 if a.in_f16 & 2 else 4))
            byts += px * k * (a.c0 + a.c1) * ((2 if a.out_f16 else 4) + (2 if a.raw_out else 0))
            if a.stats0:          # the pass computes its own statistics: + the producers' column sums, once per image (algorithmic; every workgroup of an image re-reads them)
                byts += (px // 64) * 2 * (a.c0 + a.c1) * 4
            return 'norm_act_kernel', float(byts)
        return 'other', 0.0

    # Every launch of the call is bracketed by its own event pair and NOTHING synchronises until the sampler call has been enqueued to its end:
    # the stream stays busy, so an event pair measures its kernel, not the host's time to enqueue it on an idle stream (until round 6 the update
    # launches were read back one by one -- each then started on an idle GPU and its "duration" was 12 - 40 us of host latency around a 10-us kernel)
    pending = []

    def timed_plan_run(self, stream):
        for op in self.ops:
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = op.fn(*op.args, stream)
            e1.record()
            if rc:
                _lib.check(rc, op.name)
            pending.append((classify(op), e0, e1))

    def timed_call(kind, fn):
        def w(*a, **k):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); r = fn(*a, **k); e1.record()
            pending.append(((kind, 0.0), e0, e1))
            return r
        return w

    cfg = ops.cfg_denoise
    x0s = ops.dpmpp_x0_step
    per_sample = int(latents[0].numel())
    ops.dpmpp_x0_step = timed_call(X0_KIND if lib.ds_dpmpp_x0_step_in_registers(per_sample) else X0_KIND_LDS, x0s)
    engine._Plan.run = timed_plan_run
    ops.solver_update = timed_call('solver_update_kernel', upd)
    ops.dynamic_threshold = timed_call('dynamic_threshold_kernel', thr)
    ops.cfg_denoise = timed_call('cfg_denoise_kernel', cfg)
    try:
        sampler_call(solvers, solver, net, latents, nfe, ldm)
        torch.cuda.synchronize()
        for (kind, fl), e0, e1 in pending:
            add(kind, e0.elapsed_time(e1), fl)
    finally:
        engine._Plan.run = plan_run
        ops.solver_update = upd
        ops.dynamic_threshold = thr
        ops.cfg_denoise = cfg
        ops.dpmpp_x0_step = x0s
    return rec


X0_KIND = 'dpmpp_x0_step_reg_kernel (D -> dynamic threshold -> multistep update in one launch, sample held in registers)'
X0_KIND_LDS = 'dpmpp_x0_step_kernel (D -> dynamic threshold -> multistep update in one launch, |D| staged in LDS)'


def update_roofline_large_batch(dev, batch=16384, kind='ipndm'):
    """The fused solver-update kernels where they are bandwidth-bound, HIP-event timed on `batch` CIFAR-10 images per launch.
    kind 'ipndm': one iPNDM order-4 step of ds_solver_update (x, F, 3 history tensors read; x' and d written = 7 passes of 12 288 B
    per image).  kind 'dpmpp': one DPM-Solver++(2M) data-prediction step of ds_dpmpp_x0_step -- the headline
    solver's update: x, F and m1 read, m0 = thresh(D) and x' written = 5 passes (the fused launch touches every operand once; SURVEY 8d
    prices the unfused sequence D pass + threshold + combination at 7)."""
    from diff_sampler_amd import ops
    C_, H = 3, 32
    x = torch.randn(batch, C_, H, H, device=dev)
    fr = torch.randn(batch, C_, H, H, device=dev)           # raw network output, channel-planar as the engine's output conv writes it
    hist = [torch.randn(batch, C_, H, H, device=dev) for _ in range(3)]
    xo, mo = torch.empty_like(x), torch.empty_like(x)
    if kind == 'dpmpp':
        a = ops.make_update_args(x, x, fr, batch, C_, H, H, xo, raw=True, f_ld=0, hist=hist[:1], hcoefs=[.5, .6, -.1, 0, 0, 2., 2., 0], m_out=mo,
                                 store_d=False)
        launch, passes, name = (lambda: ops.dpmpp_x0_step(a)), 5, 'dpmpp_x0_step_reg_kernel (DPM-Solver++(2M) data-prediction step: D, dynamic threshold, combination)'
    else:
        a = ops.make_update_args(x, x, fr, batch, C_, H, H, xo, raw=True, f_ld=0, hist=hist, hcoefs=[1, -.5, .1, .2, .3, 2., 2., 0], m_out=mo)
        launch, passes, name = (lambda: ops.solver_update(a)), 7, 'solver_update_fast_kernel (iPNDM order-4 step)'
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        launch()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    byts = passes * C_ * H * H * 4 * batch
    gbs = byts / (ms * 1e-3) / 1e9
    return dict(bound='hbm', kernel=name, achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit='GB/s',
                frac=round(gbs / PEAK_HBM_GBS, 4), traffic=None, images_per_launch=batch, avg_launch_ms=round(ms, 4),
                algorithmic_bytes_per_launch=byts, passes_per_image=passes,
                note='micro-benchmark inside bench.py at the batch where the update is bandwidth-bound (every operand is touched once)')


KERNEL_TU = {0: 'gemm_conv.hip', 128: 'conv3x3_halo.hip', 1284: 'conv3x3_halo.hip', 256: 'conv3x3_halo.hip', 2565: 'conv3x3_halo.hip',
             2568: 'conv3x3_halo.hip', 2561: 'gemm_dma8.hip', 2570: 'conv3x3_thin.hip', 2563: 'conv3x3_halo2.hip', 2564: 'gemm_f16.hip',
             2566: 'conv3x3_f16dma.hip', 2572: 'conv3x3_f16dma.hip', 2567: 'gemm_f16dma.hip', 2571: 'gemm_f16dma.hip', 2573: 'gemm_conv.hip', 'norm_act': 'norm_act.hip'}
# kernel CLASSES of the fp16 engines cover several template instantiations (tile shapes): their PMC rows are matched by pattern and averaged
# over all launches of the class, exactly as the HIP-event average of the class is taken
PMC_PATTERNS = {2566: r'conv3x3_f16dma_kernel<\d+, \d+, (?:true|false), false>', 2572: r'conv3x3_f16dma_kernel<\d+, \d+, (?:true|false), true>', 2567: r'gemm_f16dma_kernel<\d+, \d+, (?:true|false), false>', 2571: r'gemm_f16dma_kernel<\d+, \d+, (?:true|false), true>',
                2563: r'conv3x3_halo2_kernel<\d+, 2>', 2564: r'gemm_f16_kernel', 'norm_act': r'norm_act(?:16)?_kernel'}
# one PMC summary per benchmarked workload (tools/gpu_session.sh pmc:<bench.py args>): (config, dtype) -> file under profiles/
PMC_FILES = {('cifar10', 'fp32'): PMC_FILE,
             ('imagenet64', 'fp16'): os.path.join(ROOT, 'profiles', 'r6_pmc_hbm_imagenet64_fp16.json'),
             ('sd15', 'fp16'): os.path.join(ROOT, 'profiles', 'r6_pmc_hbm_sd15_fp16.json'),
             ('ffhq', 'fp32'): os.path.join(ROOT, 'profiles', 'r6_pmc_hbm_ffhq_fp32.json'),
             ('cifar10', 'fp16x3'): os.path.join(ROOT, 'profiles', 'r6_pmc_hbm_cifar10_fp16x3.json')}


def pmc_traffic(kid, workload=('cifar10', 'fp32')):
    """(HBM bytes per launch, provenance) of a kernel (class) from the committed rocprofv3 PMC summary of `workload` (profiles/), or None.
    The summary records the session it was collected in and the hash of every kernel translation unit at that time
    (tools/rocprof_summary.py); a kernel whose source changed since then reports null instead of a stale number (PMC_NOTE['why'] says so).
    Bytes = FETCH_SIZE x 2 (the guide's gfx950 correction) + WRITE_SIZE, averaged over the launches of the class."""
    import re
    path = PMC_FILES.get(tuple(workload))
    if tuple(workload) == ('cifar10', 'fp32'):
        path = PMC_FILE                                    # (module attribute: the tests point it elsewhere)
    try:
        from diff_sampler_amd import build
        z = json.load(open(path))
        tu = KERNEL_TU[kid]
        then, now = z['meta']['kernel_source_sha256'][tu], build.source_sha256(tu)
        if then != now:
            PMC_NOTE['why'] = f'{os.path.relpath(path, ROOT)} was collected on another build of {tu} ({then[:12]} != {now[:12]}): traffic withheld'
            return None
        if kid in PMC_PATTERNS:
            rows = [v for k, v in z['kernels'].items() if re.search(PMC_PATTERNS[kid], k)]
        else:
            rows = [z['kernels'][PMC_KEYS[kid]]]
        launches = sum(r['FETCH_SIZE_launches'] for r in rows)
        if not rows or launches <= 0 or launches != sum(r['WRITE_SIZE_launches'] for r in rows):
            PMC_NOTE['why'] = f'{os.path.relpath(path, ROOT)} holds no consistent rows for kernel class {kid}'
            return None
        byts = 1024.0 * (2.0 * sum(r['FETCH_SIZE_KiB_total'] for r in rows) + sum(r['WRITE_SIZE_KiB_total'] for r in rows)) / launches
        return round(byts), dict(file=os.path.relpath(path, ROOT), session=z['meta']['session'], kernel_source=tu, kernel_source_sha256=now,
                                 launches_in_pmc_pass=launches, instantiations=len(rows))
    except Exception as e:
        PMC_NOTE['why'] = f'no usable PMC summary ({type(e).__name__}: {e})'
        return None


def attach_traffic(roof, kid, workload):
    t = pmc_traffic(kid, workload)
    roof['traffic'] = t[0] if t else None
    roof['traffic_unit'] = 'HBM bytes per launch (rocprofv3 PMC: FETCH_SIZE x 2 + WRITE_SIZE)'
    roof['traffic_source'] = t[1] if t else PMC_NOTE.get('why')
    return roof


def norm_act_roofline(rec, workload):
    """HBM roofline of the GroupNorm-apply + SiLU pass of the fp16 engines (norm_act_kernel): algorithmic bytes (every source element read once,
    every output -- and raw copy -- element written once, at the width it is stored in) over the HIP-event time of its launches."""
    if 'norm_act_kernel' not in rec or rec['norm_act_kernel'][2] <= 0:
        return None
    ms, launches, byts = rec['norm_act_kernel']
    gbs = byts / (ms * 1e-3) / 1e9
    r = dict(bound='hbm', kernel='norm_act_kernel (GroupNorm apply + SiLU [+ resample / concatenation] -> fp16 rows)', achieved=round(gbs, 1), peak=PEAK_HBM_GBS,
             unit='GB/s', frac=round(gbs / PEAK_HBM_GBS, 4), launches_per_step=launches, avg_launch_ms=round(ms / launches, 4),
             algorithmic_bytes_per_launch=round(byts / launches))
    return attach_traffic(r, 'norm_act', workload)


def cpu_baseline_ldm(args, nfe):
    """Oracle CFG denoiser + DPM-Solver++(2M) on the host cores, bounded sample (1 latent, CFG-doubled)."""
    from oracle import solvers_ref
    from oracle.ldm_net import OracleCFG
    import diff_sampler_amd.ldm_arch as la
    kw = dict(la.NAMED_LDM_CONFIGS[args.config])
    spec = la.ldm_unet_spec(**kw)
    net = OracleCFG(la.init_ldm_params(spec, seed=0), kw, la.alphas_cumprod(spec), guidance_rate=7.5)
    g = torch.Generator().manual_seed(0)
    lat = torch.randn(1, kw['in_channels'], kw['img_resolution'], kw['img_resolution'], generator=g)
    c, uc = torch.randn(1, 77, kw['context_dim'], generator=g), torch.randn(1, 77, kw['context_dim'], generator=g)
    t_min, t_max = net.sigma_inv(torch.tensor(net.sigma_min)), net.sigma_inv(torch.tensor(net.sigma_max))
    n = nfe // 2 + 1
    ts = net.sigma(t_max + torch.arange(n) / (n - 1) * (t_min - t_max))
    default_threads = torch.get_num_threads()
    torch.set_num_threads(_thread_candidates('16' if args.cpu_threads == 'sweep' else args.cpu_threads)[0])   # see _thread_candidates
    with torch.no_grad():
        t0 = time.time()
        solvers_ref.sample('dpm_pp', net, lat, ts, condition=c, unconditional_condition=uc, max_order=2, predict_x0=False,
                           lower_order_final=True, num_steps=n)
        dt = time.time() - t0
    used = torch.get_num_threads()
    torch.set_num_threads(default_threads)
    return dict(value=round(1 / dt, 4), unit='images/sec', cores=used, kind='port', host_cores=os.cpu_count(),
                sample=f'1 sampler call x batch 1 (2 U-Net images per evaluation), NFE={nfe}, same net/solver ({dt:.1f} s of CPU work)')


def _thread_candidates(spec):
    """Thread counts of the CPU-baseline sweep.  Never "all cores" of a big host: measured on the 256-core GPU box, an 8-image
    batch ran at 3.0 / 5.3 / 2.9 / 1.4 images/s on 8 / 16 / 32 / 64 threads and at 0.008 images/s (1000 s per call) on 256."""
    n = os.cpu_count() or 1
    if spec != 'sweep':
        return [max(1, min(int(spec), n))]
    return sorted({t for t in (8, 16, 32) if t <= n} or {n})


def _reference_sampler(args, nfe):
    """The REAL reference (diff-solvers-main) when it is importable here (build container); None on the GPU box."""
    ref = '/root/reference/diff-solvers-main'
    if not os.path.isdir(ref):
        return None
    try:
        sys.path.insert(0, ref)
        import solvers as ref_solvers
        from models.networks_edm import EDMPrecond
        import diff_sampler_amd.arch as arch
        kw = dict(arch.NAMED_CONFIGS[args.config])
        net = EDMPrecond(**kw).eval()
        net.load_state_dict(arch.init_params(arch.edm_precond_spec(**kw), seed=0), strict=False)
        fn = {'dpmpp': ref_solvers.dpm_pp_sampler, 'euler': ref_solvers.euler_sampler, 'ipndm': ref_solvers.ipndm_sampler,
              'heun': ref_solvers.heun_sampler}[args.solver]
        kws = dict(num_steps=nfe + 1, sigma_min=0.002, sigma_max=80.)
        if args.solver == 'dpmpp':
            kws.update(schedule_type='logsnr', max_order=2, predict_x0=True, lower_order_final=True)
        elif args.solver == 'ipndm':
            kws.update(max_order=4)
        elif args.solver == 'heun':
            kws.update(num_steps=nfe // 2 + 1)
        return lambda lat: fn(net, lat, **kws)
    except Exception:
        return None
    finally:
        if sys.path and sys.path[0] == ref:
            sys.path.pop(0)


def cpu_baseline(args, nfe):
    """Same net, same sampler on the host cores, bounded sample.  kind 'reference' = the reference's own code (only where
    /root/reference exists), else 'port' = the oracle restatement.  One call per thread count of the sweep, the best is
    reported with the thread count that produced it (128 threads on an 8-image batch is oversubscription)."""
    import diff_sampler_amd.arch as arch
    kw = dict(arch.NAMED_CONFIGS[args.config])
    spec = arch.edm_precond_spec(**kw)
    g = torch.Generator().manual_seed(0)
    lat = torch.randn(args.cpu_batch, 3, spec.img_resolution, spec.img_resolution, generator=g)
    call, kind = _reference_sampler(args, nfe), 'reference'
    if call is None:
        from oracle import solvers_ref
        from oracle.edm_net import OracleNet
        net = OracleNet(arch.init_params(spec, seed=0), kw)
        ts = solvers_ref.schedule(nfe + 1, 0.002, 80., kind='logsnr' if args.solver == 'dpmpp' else 'polynomial', rho=7)
        name = {'dpmpp': 'dpm_pp', 'euler': 'euler', 'ipndm': 'ipndm', 'heun': 'heun'}[args.solver]
        kws = dict(max_order=2, predict_x0=True, lower_order_final=True, num_steps=nfe + 1) if args.solver == 'dpmpp' else \
            (dict(max_order=4) if args.solver == 'ipndm' else {})
        call, kind = (lambda l: solvers_ref.sample(name, net, l, ts, **kws)), 'port'
    default_threads = torch.get_num_threads()
    sweep, total = {}, 0.0
    with torch.no_grad():
        for th in _thread_candidates(args.cpu_threads):
            torch.set_num_threads(th)
            dts = []
            for _ in range(max(1, args.cpu_calls)):   # first call of a thread count also pays its pool start-up: keep the best
                t0 = time.time()
                call(lat)
                dts.append(time.time() - t0)
            total += sum(dts)
            sweep[th] = round(args.cpu_batch / min(dts), 3)
            if total > 45.0:                      # bounded sample: stop sweeping once the budget is spent
                break
    torch.set_num_threads(default_threads)
    best = max(sweep, key=sweep.get)
    return dict(value=sweep[best], unit='images/sec', cores=best, kind=kind, host_cores=os.cpu_count(),
                threads_sweep={str(k): v for k, v in sweep.items()},
                sample=f'{max(1, args.cpu_calls)} sampler calls x batch {args.cpu_batch} per thread count (best call kept), NFE={nfe}, same net/solver ({total:.1f} s of CPU work in all)')


def launch_modes(args, solvers, net_factory, spec, dev):
    """Eager vs hipGraph replay of the SAME sampler call (north_star: a hipGraph-captured step), ms per call."""
    from diff_sampler_amd.graph import GraphedSampler

    class _Rec:        # records the sampler function and kwargs sampler_call() would use
        def __getattr__(self, name):
            return lambda net_, lat_, **kw: (getattr(solvers, name), kw)

    out = {}
    for B in sorted({8, args.batch}):
        net = net_factory()
        lat = torch.randn(B, spec.in_channels, spec.img_resolution, spec.img_resolution, device=dev)
        fn, kw = sampler_call(_Rec(), args.solver, net, lat, args.nfe)
        graphed = GraphedSampler(fn, net, tuple(lat.shape), device=dev, **kw)
        res = {}
        for mode, step in (('eager_ms', lambda: sampler_call(solvers, args.solver, net, lat, args.nfe)),
                           ('graph_ms', lambda: graphed(lat, clone=False))):
            step(); torch.cuda.synchronize()
            n = 2 if B > 64 else 5
            t0 = time.perf_counter()
            for _ in range(n):
                step()
            torch.cuda.synchronize()
            res[mode] = round((time.perf_counter() - t0) / n * 1e3, 3)
        out[f'batch_{B}'] = res
        del graphed, net
    return out


def fid_allreduce_timing(dist, dev, world, reps=3):
    """The only data collective of the scope (fid.py:74-75): SUM all-reduce of mu (16 KiB) and sigma (32 MiB), fp64."""
    from diff_sampler_amd.fid import MomentAccumulator
    acc = MomentAccumulator(2048, dev)
    acc.mu.normal_(); acc.sigma.normal_()
    acc.all_reduce()                                    # warm-up (communicator set-up, buffer registration)
    mu_s = sg_s = 0.0
    for _ in range(reps):
        acc.all_reduce()
        mu_s += acc.last_timing['mu_s']; sg_s += acc.last_timing['sigma_s']
    mu_s /= reps; sg_s /= reps
    byts = acc.sigma.numel() * 8
    alg = byts / sg_s / 1e9
    return dict(mu_16KiB_ms=round(mu_s * 1e3, 4), sigma_32MiB_ms=round(sg_s * 1e3, 4), sigma_algbw_GBs=round(alg, 2),
                sigma_busbw_GBs=round(alg * 2 * (world - 1) / world, 2), xgmi_link_GBs=153.0, dtype='f64', op='SUM')


def build_net(config, dtype, dev):
    """(net factory, is latent-diffusion) for a named configuration in an arithmetic mode."""
    import diff_sampler_amd.ldm_arch as ldm_arch
    if config in ldm_arch.NAMED_LDM_CONFIGS:
        from diff_sampler_amd.ldm_engine import CFGDenoiser
        return (lambda: CFGDenoiser.from_config(config, seed=0, device=dev, guidance_rate=7.5, use_fp16=(dtype == 'fp16'))), True
    from diff_sampler_amd.engine import EDMDenoiser
    return (lambda: EDMDenoiser.from_config(config, seed=0, device=dev, use_fp16=(dtype == 'fp16'), split_fp16=(dtype == 'fp16x3'))), False


def kernel_report(rec):
    """(kernels table, dominant conv/GEMM kernel's roofline dict) from an instrumented pass."""
    total_ms = sum(v[0] for v in rec.values())
    kernels = {(KERNEL_NAMES[k[1]] if isinstance(k, tuple) else k): dict(ms=round(v[0], 2), launches=v[1], share=round(v[0] / total_ms, 4))
               for k, v in sorted(rec.items(), key=lambda kv: -kv[1][0])}
    convs = {k: v for k, v in rec.items() if isinstance(k, tuple)}
    dom, (ms, launches, fl) = max(convs.items(), key=lambda kv: kv[1][0])
    ach = fl / (ms * 1e-3) / 1e12
    peak = KERNEL_PEAK.get(dom[1], PEAK_FP32_MFMA_TFLOPS)
    roof = dict(bound='mfma', kernel=KERNEL_NAMES[dom[1]], achieved=round(ach, 2), peak=peak, unit='TFLOP/s', frac=round(ach / peak, 4),
                launches_per_step=launches, avg_launch_ms=round(ms / launches, 4), gflop_per_launch=round(fl / launches / 1e9, 2),
                share_of_gpu_time=round(ms / total_ms, 4),
                frac_of_rate_sustained_on_random_operands=round(ach / (SUSTAINED_RANDOM_FP32_TFLOPS if peak == PEAK_FP32_MFMA_TFLOPS else SUSTAINED_RANDOM_FP16_TFLOPS * peak / PEAK_FP16_MFMA_TFLOPS), 4))
    return kernels, roof, dom[1], total_ms


def application_fraction(config, dtype, images_per_sec, nfe):
    """Whole-application fraction of the matrix peak of the datatype used: SURVEY 8d's algorithmic FLOPs per image and evaluation x the
    evaluations of one image (NFE; a latent-diffusion latent takes NFE U-Net images: NFE/2 steps x 2 CFG halves) / time / peak.
    fp16x3 issues three fp16 MFMA products per fp32 product: its ceiling in algorithmic FLOPs is a third of the fp16 peak."""
    peak = {'fp32': PEAK_FP32_MFMA_TFLOPS, 'fp16': PEAK_FP16_MFMA_TFLOPS, 'fp16x3': PEAK_FP16_MFMA_TFLOPS / 3}[dtype]
    tf = images_per_sec * GFLOP_PER_EVAL[config] * nfe / 1e3
    return dict(achieved=round(tf, 1), peak=round(peak, 1), unit='TFLOP/s', frac=round(tf / peak, 4),
                gflop_per_image=round(GFLOP_PER_EVAL[config] * nfe, 1))


def measure_config(config, dtype, batch, nfe, dev, calls=5, warmup=2, latency_batch=None):
    """One of `other_configs`: build the net, `warmup` untimed + `calls` individually timed sampler calls (value = the MEDIAN call; min / max
    next to it: box-to-box and call-to-call spread on the fp16 lines is a few per cent), one instrumented call for the kernel shares."""
    from diff_sampler_amd import solvers
    t_build = time.perf_counter()
    factory, is_ldm = build_net(config, dtype, dev)
    net = factory()
    spec = net.spec
    solver = 'ipndm' if config == 'imagenet64' else 'dpmpp'
    g = torch.Generator(device='cpu').manual_seed(4321)

    def inputs(b):
        lat = torch.randn(b, spec.in_channels, spec.img_resolution, spec.img_resolution, generator=g).to(dev)
        # two condition tuples used in rotation: fresh text conditions on every call (_next_conditions)
        ldm = [(torch.randn(b, 77, spec.context_dim, generator=g).to(dev), torch.randn(b, 77, spec.context_dim, generator=g).to(dev))
               for _ in range(2)] if is_ldm else None
        return lat, ldm

    def timed(lat, ldm, n, w=1):
        """Seconds of each of n sampler calls (each bracketed by a device synchronisation) after w untimed ones."""
        for _ in range(w):
            sampler_call(solvers, solver, net, lat, nfe, ldm)
        torch.cuda.synchronize()
        dts = []
        for _ in range(n):
            t0 = time.perf_counter()
            out = sampler_call(solvers, solver, net, lat, nfe, ldm)
            torch.cuda.synchronize()
            dts.append(time.perf_counter() - t0)
        assert torch.isfinite(out).all()
        return sorted(dts)

    lat, ldm = inputs(batch)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build
    dts = timed(lat, ldm, calls, warmup)
    dt = dts[len(dts) // 2]                                   # median call
    rec = instrumented_pass(net, solvers, solver, lat, nfe, ldm)
    kernels, roof, dom_id, _ = kernel_report(rec)
    attach_traffic(roof, dom_id, (config, dtype))
    roof_norm = norm_act_roofline(rec, (config, dtype)) if dtype == 'fp16' else None
    top = dict(list(kernels.items())[:5])
    res = dict(config=dict(workload='%s, %s NFE=%d, batch %d/GPU' % (WORKLOAD_NAMES.get(config, config), LDM_SOLVER_NAME if is_ldm else SOLVER_NAMES[solver], nfe, batch)),
               dtype=dtype, value=round(batch / dt, 2), value_min=round(batch / dts[-1], 2), value_max=round(batch / dts[0], 2),
               value_stat=f'median of {calls} individually timed sampler calls after {warmup} warm-up calls', unit='images/sec',
               ms_per_step=round(dt * 1e3, 2), steps=calls, warmup=warmup,
               roofline=roof, roofline_norm_act=roof_norm, application=application_fraction(config, dtype, batch / dt, nfe), kernels_top5=top,
               setup_s=round(t_build, 1))
    lat_ms = None
    if latency_batch is not None:
        l2, ldm2 = inputs(latency_batch)
        d2 = timed(l2, ldm2, 5, 1)
        lat_ms = round(d2[len(d2) // 2] * 1e3, 2)
    del net
    torch.cuda.empty_cache()
    return res, lat_ms


def main(argv=None):
    from diff_sampler_amd import launch
    args = parse(argv)
    try:
        rank, world, local = launch.resolve(args.gpus, os.path.abspath(__file__), sys.argv[1:] if argv is None else list(argv))
    except launch.LaunchError as e:
        print(f'bench.py: {e}', file=sys.stderr)
        raise SystemExit(2)
    stub = args.stub
    if not stub:
        if not torch.cuda.is_available() or torch.cuda.device_count() <= local:
            print(f'bench.py: rank {rank} needs GPU {local}, {torch.cuda.device_count() if torch.cuda.is_available() else 0} visible',
                  file=sys.stderr)
            raise SystemExit(2)
        torch.cuda.set_device(local)
    dev = torch.device('cpu') if stub else torch.device('cuda', local)
    dist, n_comm = None, 1
    if world > 1:
        try:
            dist, n_comm = launch.init_group(rank, world, local, 'gloo' if stub else 'nccl', dev)
        except launch.LaunchError as e:
            print(f'bench.py: {e}', file=sys.stderr)
            raise SystemExit(2)
    sync = (lambda: None) if stub else torch.cuda.synchronize

    B = args.batch
    g = torch.Generator(device='cpu').manual_seed(1234 + rank)
    ldm = None
    cpu = None
    if stub:
        solvers = net = spec = latents = None
        run_step = lambda: time.sleep(0.002 * (1 + rank))
    else:
        from diff_sampler_amd import solvers
        net_factory, is_ldm = build_net(args.config, args.dtype, dev)
        net = net_factory()
        spec = net.spec
        if is_ldm:
            ldm = [(torch.randn(B, 77, spec.context_dim, generator=g).to(dev), torch.randn(B, 77, spec.context_dim, generator=g).to(dev))
                   for _ in range(2)]                                # rotated: fresh text conditions on every call (_next_conditions)
            args.solver = 'dpmpp'
        latents = torch.randn(B, spec.in_channels, spec.img_resolution, spec.img_resolution, generator=g).to(dev)

        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline_ldm(args, args.nfe) if ldm is not None else cpu_baseline(args, args.nfe)

        run_step = lambda: sampler_call(solvers, args.solver, net, latents, args.nfe, ldm)
        if args.graph:
            assert ldm is None, '--graph covers the EDM nets'
            from diff_sampler_amd.graph import GraphedSampler

            class _Rec:        # records the sampler function and kwargs sampler_call() would use
                def __getattr__(self, name):
                    return lambda net_, lat_, **kw: (getattr(solvers, name), kw)
            fn, kw = sampler_call(_Rec(), args.solver, net, latents, args.nfe)
            graphed = GraphedSampler(fn, net, tuple(latents.shape), device=dev, **kw)
            run_step = lambda: graphed(latents, clone=False)
    for _ in range(args.warmup):
        run_step()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    out = None
    for _ in range(args.steps):
        out = run_step()
    sync()
    dt_local = time.perf_counter() - t0                  # this rank's own K steps (before waiting for the others)
    if dist is not None:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    multi = None
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        per = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(per, torch.tensor([dt_local], dtype=torch.float64, device=dev))
        per_ms = [float(t.item()) / args.steps * 1e3 for t in per]
        per_val = [B / (v * 1e-3) for v in per_ms]          # images / s of each rank's own K steps (before it waits for the others)
        multi = dict(per_rank_ms_per_step_min=round(min(per_ms), 3), per_rank_ms_per_step_max=round(max(per_ms), 3),
                     per_rank_ms_per_step=[round(v, 3) for v in per_ms],
                     per_rank_value_min=round(min(per_val), 2), per_rank_value_max=round(max(per_val), 2), per_rank_value=[round(v, 2) for v in per_val],
                     per_rank_value_unit='images/sec of one rank over its own timed steps; `value` = all ranks\' images / the slowest rank\'s barrier-bracketed time',
                     barrier_skew_ms_per_step=round(max(per_ms) - min(per_ms), 3), backend='gloo' if stub else 'nccl (RCCL)', communicator=dict(launch.COMM_INFO),
                     fid_moment_allreduce=fid_allreduce_timing(dist, dev, world))
    if not stub:
        assert torch.isfinite(out).all()

    roof = roof_u = kernels = roof_norm = None
    if rank == 0 and stub:
        # launcher self-test: no kernels ran, but the N > 1 line must carry rank 0's roofline object like a real line does (a SCALE record is
        # then self-contained): the SAME report path over a synthetic one-kernel record, marked as such
        kernels, roof, dom_id, _ = kernel_report({('conv', 2565): [1.0, 1, 1.0e9]})
        attach_traffic(roof, dom_id, (args.config, args.dtype))
        roof['stub'] = True
    if rank == 0 and not stub:
        rec = instrumented_pass(net, solvers, args.solver, latents, args.nfe, ldm)
        kernels, roof, dom_id, total_ms = kernel_report(rec)
        if args.dtype == 'fp16':
            roof_norm = norm_act_roofline(rec, (args.config, args.dtype))
        default_batch = {'ffhq': 128, 'imagenet64': 64, 'sd15': 16}.get(args.config, 256)      # the PMC passes are of the default workloads
        if B == default_batch and args.nfe == 10:
            attach_traffic(roof, dom_id, (args.config, args.dtype))
        else:
            roof['traffic'], roof['traffic_source'] = None, 'no PMC pass for this batch / NFE'
        per = spec.in_channels * spec.img_resolution ** 2 * 4
        if X0_KIND in rec and ldm is None:
            # the headline solver's update: ONE ds_dpmpp_x0_step launch per evaluation.  Algorithmic bytes of the fused launch: x, F read;
            # m0, x' written; + one history tensor on the 2M steps (order 1 on the first and, with lower_order_final, the last step)
            ums, ul, _ = rec[X0_KIND]
            n2m = max(0, ul - 2)
            byts = (4 * (ul - n2m) + 5 * n2m) * per * B
            gbs = byts / (ums * 1e-3) / 1e9
            roof_u = dict(bound='hbm', kernel=X0_KIND, achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit='GB/s',
                          frac=round(gbs / PEAK_HBM_GBS, 4), traffic=None, launches_per_step=ul, avg_launch_ms=round(ums / ul, 4),
                          algorithmic_bytes_per_step=byts,
                          note='latency-bound at this batch (3 MB per operand, one workgroup per image); roofline_update_large_batch has the same kernel at 16 384 images per launch')
        elif ldm is None and args.solver != 'dpmpp' and 'solver_update_kernel' not in rec:
            # the linear solvers' update runs INSIDE the network head (ds_conv_args.update, csrc/conv3x3_thin.hip): there is no update launch
            # to time -- its bytes (x, history in; x', history entry out) move in the head's epilogue while F is still in a register
            passes = {'euler': 3, 'ipndm': 7, 'heun': 3.5}[args.solver]
            roof_u = dict(bound='hbm', kernel='conv3x3_thin_kernel<., UPD> epilogue (the solver update rides in the network head: no launch of its own)',
                          fused=True, achieved=None, peak=PEAK_HBM_GBS, unit='GB/s', frac=None, traffic=None, launches_per_step=0,
                          algorithmic_bytes_per_step=int((passes - 1) * per * B * args.nfe),
                          note='F is never re-read; see roofline_update_large_batch for the stand-alone kernel where it is bandwidth-bound')
        elif 'solver_update_kernel' in rec and ldm is None:
            ums, ul, _ = rec['solver_update_kernel']
            passes = {'dpmpp': 7, 'euler': 3, 'ipndm': 7, 'heun': 3.5}[args.solver]
            byts = passes * per * B * args.nfe
            gbs = byts / (ums * 1e-3) / 1e9
            roof_u = dict(bound='hbm', kernel='solver_update_kernel', achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit='GB/s',
                          frac=round(gbs / PEAK_HBM_GBS, 4), traffic=None, launches_per_step=ul, avg_launch_ms=round(ums / ul, 4),
                          note='latency-bound at this batch (3 MB per operand); see roofline_update_large_batch')

    roof_ul = roof_ul_other = modes = by_batch = None
    if rank == 0 and ldm is None and not stub:
        roof_ul = update_roofline_large_batch(dev, kind=('dpmpp' if args.solver == 'dpmpp' else 'ipndm'))
        roof_ul_other = update_roofline_large_batch(dev, kind=('ipndm' if args.solver == 'dpmpp' else 'dpmpp'))
        if not args.no_launch_modes and world == 1:
            modes = launch_modes(args, solvers, net_factory, spec, dev)
        if not args.no_batch_sweep and world == 1 and args.config == 'cifar10' and not args.graph:
            # the same sampler call over SURVEY 8d's batch sweep of this configuration {64, 256, 1024, 4096}: per-launch fixed costs are
            # amortised over more rounds of tiles.  Informational; `value` stays the batch named in config.workload.  Two timed calls after
            # one warm-up call (4 096 images: one network evaluation as warm-up and ONE timed call -- a call takes ~13 s there); the plans
            # of a swept batch (4 096 images: ~180 GB of workspaces) are released before the next one is built.
            by_batch = {str(B): dict(value=round(B * args.steps / dt_local, 2), steps=args.steps, warmup=args.warmup)}
            for b2 in args.sweep_batches:
                if b2 == B:
                    continue
                try:
                    lat2 = torch.randn(b2, spec.in_channels, spec.img_resolution, spec.img_resolution, device=dev)
                    n2 = 1 if b2 > 2048 else 2
                    if b2 > 2048:
                        net(lat2, 1.0); sync()                      # plan build + one evaluation
                    else:
                        sampler_call(solvers, args.solver, net, lat2, args.nfe); sync()
                    t1 = time.perf_counter()
                    for _ in range(n2):
                        o2 = sampler_call(solvers, args.solver, net, lat2, args.nfe)
                    sync()
                    by_batch[str(b2)] = dict(value=round(n2 * b2 / (time.perf_counter() - t1), 2), steps=n2, warmup='1 call' if b2 <= 2048 else '1 network evaluation',
                                             finite=bool(torch.isfinite(o2).all()))
                    del lat2, o2
                except Exception as e:                              # (e.g. out of device memory on a smaller part): the headline line survives
                    by_batch[str(b2)] = dict(error=f'{type(e).__name__}: {str(e)[:200]}')
                net._last = None
                for k_ in [k_ for k_ in net.engine._plans if k_[0] == b2]:
                    del net.engine._plans[k_]
                torch.cuda.empty_cache()
            by_batch = dict(sorted(by_batch.items(), key=lambda kv: int(kv[0])))
    others = latency = None
    if (rank == 0 and world == 1 and not stub and not args.no_other_configs and not args.graph
            and (args.config, args.dtype, B, args.solver, args.nfe) == ('cifar10', 'fp32', 256, 'dpmpp', 10)):
        # the default run: the other benchmarked configurations / modes and the small-batch latencies, under the same clock
        latency = {}
        l8 = torch.randn(8, spec.in_channels, spec.img_resolution, spec.img_resolution, device=dev)
        sampler_call(solvers, args.solver, net, l8, args.nfe); sync()
        t1 = time.perf_counter()
        for _ in range(5):
            sampler_call(solvers, args.solver, net, l8, args.nfe)
        sync()
        latency['cifar10_fp32_B8_nfe10_ms'] = round((time.perf_counter() - t1) / 5 * 1e3, 2)
        del net, latents, l8, out
        torch.cuda.empty_cache()
        others = []
        for cfg_, dt_, b_ in OTHER_CONFIGS:
            try:
                res, lat_ms = measure_config(cfg_, dt_, b_, args.nfe, dev, latency_batch=(1 if cfg_ == 'sd15' else None))
                others.append(res)
                if lat_ms is not None:
                    latency['sd15_fp16_B1_nfe10_ms'] = lat_ms
            except Exception as e:                                   # a failing side measurement must not take the headline line with it
                others.append(dict(config=dict(workload=f'{cfg_} {dt_} batch {b_}'), error=f'{type(e).__name__}: {e}'))
    if rank == 0:
        workload_name = WORKLOAD_NAMES.get(args.config, args.config)
        total_images = B * world * args.steps
        line = {
            'metric': 'images/sec (whole node) at NFE=%d, %s' % (args.nfe, 'EDM CIFAR-10' if args.config == 'cifar10' else workload_name.split(' (')[0]),
            'value': round(total_images / dt, 2), 'unit': 'images/sec', 'n_gpus': n_comm, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 2), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': {'fp32': 'fp32', 'fp16': 'fp16 operands in the 3x3 convolutions, 1x1 / Linear layers and attention, activations between layers stored in fp16 (residual stream included, as the reference does in this mode); fp32 accumulate, fp32 softmax / norm arithmetic',
                      'fp16x3': 'fp16x3 (fp32-emulated: split fp16 hi/lo operands, 3 MFMA products, fp32 accumulate) in the 3x3 convolutions, rest fp32'}[args.dtype],
            'data': 'synthetic N(0,1) latents, random-init (signal-carrying) weights',
            'config': {'workload': '%s, %s NFE=%d, batch %d/GPU' %
                       (workload_name, LDM_SOLVER_NAME if ldm is not None else SOLVER_NAMES[args.solver], args.nfe, B),
                       'images_per_step': B * world, 'sharding': 'independent image batches per rank, no collective',
                       'launch': 'hipGraph replay' if args.graph else 'eager'},
            'roofline': roof, 'roofline_norm_act': roof_norm, 'roofline_update': roof_u, 'roofline_update_large_batch': roof_ul, 'roofline_update_large_batch_other_solver': roof_ul_other, 'kernels': kernels,
            'launch_modes': modes, 'throughput_by_batch': by_batch, 'cpu_baseline': cpu, 'multi_gpu': multi,
            'application': (application_fraction(args.config, args.dtype, total_images / dt / world, args.nfe) if not stub and args.config in GFLOP_PER_EVAL else None),
            'other_configs': others, 'latency': latency,
        }
        if stub:
            line['stub'] = True
            line['data'] = 'launcher self-test: no kernels ran, value is meaningless'
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
