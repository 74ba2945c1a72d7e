#!/bin/bash
# Round-2 session 20: options of the 256 x 256 conv tile (lean DMA addressing, non-temporal epilogue, staggered first round) --
# parity, interleaved A/B on the main shapes, timing ablations (-DDS_CONV_ABLATIONS build), whole-network A/B.
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s20; mkdir -p $O
timeout 300 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "wide_n" > $O/pytest_wide.txt 2>&1
tail -3 $O/pytest_wide.txt
timeout 200 python tools/bench_conv.py --batch 256 --norm --only 0 1 2 4 5 --rounds 5 \
    --variants 0 32 128 160 544 1056 2080 65552 65564 65540 65556 > $O/conv_ab.txt 2>&1
cat $O/conv_ab.txt | cut -c1-1200
for v in 0 32 160 1056; do
  DS_CONV_VARIANT=$v timeout 120 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-launch-modes > $O/bench_v$v.json 2> $O/bench_v$v.err
  echo "variant $v: $(tail -1 $O/bench_v$v.json | cut -c1-160)"
done
true
