#!/bin/bash
# A/B of engine.fuse_norm16 (GroupNorm apply + SiLU inside conv3x3_f16dma's LDS halo vs the ds_norm_act pass) on the two fp16 sampler lines,
# alternating in one gpurun session; then the per-kernel-class table of both plans (bench.py's instrumented pass).
Q="--steps 3 --warmup 1 --no-cpu-baseline --no-launch-modes --no-batch-sweep --no-other-configs"
for rep in 1 2; do
  for f in 0 1; do
    for cfg in imagenet64 sd15; do
      DS_FUSE_NORM16=$f python bench.py --config $cfg --dtype fp16 $Q 2>/dev/null | python -c "
import sys,json
z=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=z['kernels']
print('fuse_norm16=$f $cfg', z['value'], 'img/s', z['ms_per_step'], 'ms |', ' | '.join('%s %.1f ms x%d' % (n.split(' (')[0][:34], v['ms'], v['launches']) for n, v in list(k.items())[:7]))"
    done
  done
done
