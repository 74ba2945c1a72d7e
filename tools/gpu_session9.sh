#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s9; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_kernels.py -q -m gpu -k "attention" > $O/pytest_attn16.txt 2>&1
tail -25 $O/pytest_attn16.txt
timeout 300 python tools/bench_attn.py --images 16 > $O/bench_attn.txt 2>&1
cat $O/bench_attn.txt
timeout 600 python -m pytest tests/test_hip_fp16.py -q -m gpu > $O/pytest_fp16.txt 2>&1
tail -15 $O/pytest_fp16.txt
timeout 600 python bench.py --config sd15 --batch 16 --dtype fp16 --steps 2 --warmup 1 --no-cpu-baseline --no-launch-modes > $O/bench_sd15_f16.json 2> $O/bench_sd15_f16.err
tail -2 $O/bench_sd15_f16.json; tail -3 $O/bench_sd15_f16.err
timeout 600 python bench.py --config imagenet64 --batch 64 --solver ipndm --dtype fp16 --steps 2 --warmup 1 --no-cpu-baseline --no-launch-modes > $O/bench_in64_f16.json 2> $O/bench_in64_f16.err
tail -2 $O/bench_in64_f16.json; tail -3 $O/bench_in64_f16.err
true
