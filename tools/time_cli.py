"""End-to-end speed of the CLI body (seeded RNG + sampler + uint8 quantisation + background PNG sink) vs the bare sampler."""
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sampler_amd import sample  # noqa: E402

n, batch = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, int(sys.argv[2]) if len(sys.argv) > 2 else 512
with tempfile.TemporaryDirectory() as d:
    kw = dict(max_batch_size=batch, outdir=d, solver='dpmpp', max_order=2, num_steps=11, schedule_type='logsnr', random_init=True)
    sample.run('cifar10', seeds=f'0-{batch - 1}', **kw)          # warm-up: plan, allocator
    torch.cuda.synchronize()
    phases = {}

    def timed(name, fn):
        def w(*a, **k):
            torch.cuda.synchronize(); t = time.perf_counter()
            r = fn(*a, **k)
            torch.cuda.synchronize(); phases[name] = phases.get(name, 0.0) + time.perf_counter() - t
            return r
        return w
    sample.create_model = timed('create_model', sample.create_model)
    sample.save_images = timed('save_images (quantise + D2H + submit)', sample.save_images)
    sample.StackedRandomGenerator.randn = timed('seeded randn', sample.StackedRandomGenerator.randn)
    from diff_sampler_amd import solvers
    solvers.dpm_pp_sampler = timed('sampler', solvers.dpm_pp_sampler)
    sample.PngSink.close = timed('png drain', sample.PngSink.close)
    t0 = time.perf_counter()
    out, done = sample.run('cifar10', seeds=f'0-{n - 1}', **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    files = sum(len(f) for _, _, f in os.walk(out))
print({k: round(v, 3) for k, v in phases.items()})
print(f'sample.run: {done} images (NFE=10, batch {batch}) in {dt:.2f} s = {done / dt:.1f} images/s end to end, {files} PNG files')
