#!/usr/bin/env python
"""Regenerate diff_sampler_amd/data/tile_table.json -- the persisted tile-shape table of the fp16-activation kernels (diff_sampler_amd/plan.py, AUTOTUNE).

    python tools/make_tile_table.py [out.json]          # on a GPU box; default out = gpurun_out/tile_table.json

Builds the plans of every benchmarked / tested fp16 configuration at its batch with the persisted table IGNORED (DS_TILE_TABLE=off), so every
eligible layer shape is measured in this process (cold operands, the library's own choice timed first and last, a candidate must win by 3 %),
and writes what was measured together with the hashes of the two kernel translation units.  Copy the file to diff_sampler_amd/data/tile_table.json and
commit it: plan builds then look shapes up instead of racing timers, so two runs of one tree choose identical tiles."""
import os
import sys
import time

os.environ['DS_TILE_TABLE'] = 'off'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from diff_sampler_amd import plan as plan_mod  # noqa: E402

# (config, images per network evaluation, embedding rows): bench.py's fp16 lines and latency legs, the parity tests' batches
EDM = [('imagenet64', [(64, 64), (4, 4), (1, 1)]), ('ffhq', [(128, 1)]), ('cifar10', [(256, 1), (64, 1), (8, 1)])]
LDM = [('sd15', [(32, 1), (2, 1), (4, 1), (2, 2)])]


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'tile_table.json')
    assert torch.cuda.is_available(), 'the table is measured on the GPU'
    from diff_sampler_amd.engine import EDMDenoiser
    from diff_sampler_amd.ldm_engine import CFGDenoiser
    t0 = time.time()
    for cfg, batches in EDM:
        net = EDMDenoiser.from_config(cfg, seed=0, use_fp16=True)
        for n, er in batches:
            net.engine.plan(n, er)
            print(f'{cfg} {n}: {len(plan_mod._MEASURED)} shapes measured so far ({time.time() - t0:.0f} s)', flush=True)
        del net
        torch.cuda.empty_cache()
    for cfg, batches in LDM:
        net = CFGDenoiser.from_config(cfg, seed=0, guidance_rate=7.5, use_fp16=True)
        for n, er in batches:
            net.engine.plan(n, er, 77)
            print(f'{cfg} {n}: {len(plan_mod._MEASURED)} shapes measured so far ({time.time() - t0:.0f} s)', flush=True)
        del net
        torch.cuda.empty_cache()
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    n = plan_mod.save_tile_table(out, session=os.environ.get('DS_SESSION', ''))
    changed = sum(1 for v in plan_mod._MEASURED.values() if (v[0], v[1]) != (0, 0))
    print(f'wrote {out}: {n} shapes, {changed} of them away from the cost model', flush=True)


if __name__ == '__main__':
    main()
