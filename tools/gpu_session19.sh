#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s19; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "wide_n or conv or halo" > $O/pytest_wide.txt 2>&1
tail -4 $O/pytest_wide.txt
timeout 900 python -m pytest tests/test_hip_full_goldens.py tests/test_hip_ldm.py tests/test_hip_vs_torch_rocm.py -x -q -m gpu > $O/pytest_nets.txt 2>&1
tail -4 $O/pytest_nets.txt
timeout 400 python bench.py --config sd15 --batch 16 --no-cpu-baseline --no-launch-modes > $O/bench_sd15_fp32.json 2> $O/bench_sd15_fp32.err
tail -1 $O/bench_sd15_fp32.json | cut -c1-220
timeout 300 python bench.py --no-cpu-baseline --no-launch-modes > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | cut -c1-220
true
