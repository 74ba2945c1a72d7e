#!/bin/bash
# Round-2 session 21: kernel suite after the tile-option defaults; A/B of the plain / default (lean + non-temporal) / late-DMA 256 x 256
# kernels; 8x8 layers with the coefficient planes in LDS vs global (variant bit 9); whole-network A/B.
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s21; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_kernels.py -q -m gpu > $O/pytest_kernels.txt 2>&1
tail -5 $O/pytest_kernels.txt
timeout 200 python tools/bench_conv.py --batch 256 --norm --only 0 1 2 4 5 --rounds 5 --variants 256 0 64 > $O/conv_ab.txt 2>&1
timeout 200 python tools/bench_conv.py --batch 256 --norm --only 6 7 --rounds 7 --iters 20 --variants 512 0 >> $O/conv_ab.txt 2>&1
cat $O/conv_ab.txt | cut -c1-600
for v in 256 0 64 512; do
  DS_CONV_VARIANT=$v timeout 120 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-launch-modes > $O/bench_v$v.json 2> $O/bench_v$v.err
  echo "variant $v: $(tail -1 $O/bench_v$v.json | cut -c1-160)"
done
true
