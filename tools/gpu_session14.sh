#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s14; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -m gpu > $O/pytest_kernels.txt 2>&1
tail -4 $O/pytest_kernels.txt
timeout 200 python tools/bench_conv.py --batch 256 --only 0 1 3 4 6 --norm --iters 10 > $O/conv.txt 2>&1
grep -v amdgpu.ids $O/conv.txt | tail -8
timeout 600 python bench.py --no-cpu-baseline --no-launch-modes > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | cut -c1-300; tail -2 $O/bench.err
timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_hip_kernels.py > $O/pytest_rest.txt 2>&1
tail -4 $O/pytest_rest.txt
true
