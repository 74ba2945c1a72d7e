#!/bin/bash
# Round-2 session 25: conv_tile refactor + persistent phase-shifted schedule of the 256 x 256 tile (variant bit 6): parity, A/B.
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s25; mkdir -p $O
timeout 400 python -m pytest tests/test_hip_kernels.py -q -m gpu > $O/pytest_kernels.txt 2>&1
tail -5 $O/pytest_kernels.txt
timeout 200 python tools/bench_conv.py --batch 256 --norm --ws --only 0 1 2 --rounds 5 --variants 0 64 > $O/conv_ab.txt 2>&1
timeout 200 python tools/bench_conv.py --batch 1024 --norm --ws --only 0 4 --rounds 3 --iters 5 --variants 0 64 >> $O/conv_ab.txt 2>&1
cat $O/conv_ab.txt | cut -c1-400
for v in 0 64; do
  DS_CONV_VARIANT=$v timeout 120 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-launch-modes --no-batch-sweep > $O/bench_v$v.json 2> $O/bench_v$v.err
  echo "variant $v: $(tail -1 $O/bench_v$v.json | cut -c1-160)"
done
true
