#!/bin/bash
# GPU session 2: second-generation halo kernel -- correctness (routing asserted), A/B against the first kernel, end-to-end bench.
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s2; mkdir -p $O
timeout 300 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "halo2" > $O/pytest_halo2.txt 2>&1
timeout 200 python tools/bench_conv.py --batch 256 --only 0 1 3 4 --variants 0 3 --rounds 5 --iters 5 --norm > $O/ab_norm.txt 2>&1
timeout 200 python tools/bench_conv.py --batch 256 --only 0 1 3 4 --variants 0 3 --rounds 5 --iters 5 --norm --extra > $O/ab_norm_extra.txt 2>&1
DS_CONV_VARIANT=3 timeout 600 python -m pytest tests/test_hip_kernels.py tests/test_hip_denoiser.py tests/test_hip_full_goldens.py tests/test_hip_samplers.py -x -q -m gpu > $O/pytest_variant3.txt 2>&1
DS_CONV_VARIANT=3 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-launch-modes > $O/bench_v3.json 2> $O/bench_v3.err
timeout 300 python -m pytest tests/test_hip_solver_utils.py tests/test_hip_sample_cli.py -x -q -m gpu > $O/pytest_new.txt 2>&1
tail -3 $O/*.txt
