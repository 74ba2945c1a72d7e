#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s15; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "f16 or split or halo2" > $O/pytest_kernels.txt 2>&1
tail -4 $O/pytest_kernels.txt
timeout 120 python tools/bench_conv.py --batch 256 --only 0 1 3 4 --iters 5 --norm --f16 > $O/conv_f16.txt 2>&1
timeout 120 python tools/bench_conv.py --batch 256 --only 0 1 3 4 --iters 5 --norm --split > $O/conv_split.txt 2>&1
grep -v amdgpu.ids $O/conv_f16.txt | tail -6; grep -v amdgpu.ids $O/conv_split.txt | tail -6
timeout 600 python -m pytest tests/test_hip_fp16.py tests/test_hip_split.py -x -q -m gpu > $O/pytest_modes.txt 2>&1
tail -4 $O/pytest_modes.txt
timeout 300 python bench.py --config imagenet64 --batch 64 --solver ipndm --dtype fp16 --steps 2 --warmup 1 --no-cpu-baseline --no-launch-modes > $O/bench_in64_f16.json 2> $O/bench_in64_f16.err
tail -1 $O/bench_in64_f16.json | cut -c1-200
timeout 300 python bench.py --dtype fp16x3 --steps 2 --warmup 1 --no-cpu-baseline --no-launch-modes > $O/bench_fp16x3.json 2> $O/bench_fp16x3.err
tail -1 $O/bench_fp16x3.json | cut -c1-200
true
