#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s16; mkdir -p $O
DS_CONV_VARIANT=6 timeout 600 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "conv or halo" > $O/pytest_kernels_v6.txt 2>&1
tail -4 $O/pytest_kernels_v6.txt
DS_CONV=256 timeout 300 python tools/bench_conv.py --batch 256 --only 0 1 3 4 --norm --variants 0 6 --rounds 5 --iters 5 > $O/ab_v6.txt 2>&1
grep "^\[" $O/ab_v6.txt
DS_CONV_VARIANT=6 timeout 600 python -m pytest tests/test_hip_full_goldens.py tests/test_hip_denoiser.py -x -q -m gpu > $O/pytest_nets_v6.txt 2>&1
tail -4 $O/pytest_nets_v6.txt
DS_CONV_VARIANT=6 timeout 600 python bench.py --no-cpu-baseline --no-launch-modes > $O/bench_v6.json 2> $O/bench_v6.err
tail -1 $O/bench_v6.json | cut -c1-300; tail -2 $O/bench_v6.err
true
