#!/bin/bash
# Round-2 session 29: 128 x 128 tiles on eight half-size waves (conv3x3_halo_kernel<2, true, 4, 0, 1>) for layers with at most one tile per
# CU: parity (kernel + denoiser suites), A/B on the 8x8 layers (variant 2048 = the four-wave kernel), whole-network A/B.
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s29; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_kernels.py tests/test_hip_denoiser.py -q -m gpu > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
timeout 200 python tools/bench_conv.py --batch 256 --norm --ws --only 6 7 --rounds 7 --iters 20 --variants 2048 0 > $O/conv_ab.txt 2>&1
timeout 200 python tools/bench_conv.py --batch 64 --norm --ws --only 4 5 6 7 --rounds 7 --iters 20 --variants 2048 0 >> $O/conv_ab.txt 2>&1
grep "^\[" $O/conv_ab.txt | cut -c1-300
for v in 2048 0; do
  DS_CONV_VARIANT=$v timeout 120 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-launch-modes --no-batch-sweep > $O/bench_v$v.json 2> $O/bench_v$v.err
  echo "variant $v: $(tail -1 $O/bench_v$v.json | cut -c1-160)"
done
true
