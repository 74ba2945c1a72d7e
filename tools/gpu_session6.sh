#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s6; mkdir -p $O
timeout 200 python -m pytest tests/test_hip_rng.py -q -m gpu > $O/pytest_rng.txt 2>&1
timeout 300 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "split or f16 or halo2" > $O/pytest_kernels.txt 2>&1
timeout 120 python tools/bench_conv.py --batch 256 --only 0 1 3 4 6 --iters 5 --norm --split > $O/conv_split.txt 2>&1
timeout 120 python tools/bench_conv.py --batch 256 --only 0 1 3 --iters 5 --norm --split --extra > $O/conv_split_extra.txt 2>&1
timeout 600 python -m pytest tests/test_hip_split.py -x -q -m gpu > $O/pytest_split.txt 2>&1
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-launch-modes --dtype fp16x3 > $O/bench_fp16x3.json 2> $O/bench_fp16x3.err
for f in $O/*.txt; do echo "== $f"; tail -6 $f; done
true
