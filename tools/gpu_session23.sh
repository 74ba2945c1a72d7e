#!/bin/bash
# Round-2 session 23: kernel + denoiser + FID suites after the PIPE routing of single-wave-per-SIMD layers and the runtime non-temporal
# epilogue flag; A/B: variant 2048 = no PIPE routing, 1024 = non-temporal epilogues (outputs >= 32 MiB) in every igemm-family kernel.
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s23; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_kernels.py tests/test_hip_denoiser.py tests/test_hip_fid.py -q -m gpu > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
for v in 0 2048 1024; do
  DS_CONV_VARIANT=$v timeout 120 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-launch-modes > $O/bench_v$v.json 2> $O/bench_v$v.err
  echo "cifar10 variant $v: $(tail -1 $O/bench_v$v.json | cut -c1-160)"
done
timeout 200 python bench.py --batch 1024 --steps 3 --warmup 1 --no-cpu-baseline --no-launch-modes > $O/bench_b1024.json 2> $O/bench_b1024.err
echo "cifar10 B=1024: $(tail -1 $O/bench_b1024.json | cut -c1-160)"
for v in 0 1024; do
  DS_CONV_VARIANT=$v timeout 200 python bench.py --config imagenet64 --batch 64 --solver ipndm --dtype fp16 --steps 4 --warmup 1 --no-cpu-baseline --no-launch-modes > $O/bench_in64_fp16_v$v.json 2> $O/bench_in64_fp16_v$v.err
  echo "imagenet64 fp16 variant $v: $(tail -1 $O/bench_in64_fp16_v$v.json | cut -c1-160)"
done
for v in 0 1024; do
  DS_CONV_VARIANT=$v timeout 300 python bench.py --config sd15 --batch 16 --dtype fp16 --steps 2 --warmup 1 --no-cpu-baseline --no-launch-modes > $O/bench_sd15_fp16_v$v.json 2> $O/bench_sd15_fp16_v$v.err
  echo "sd15 fp16 variant $v: $(tail -1 $O/bench_sd15_fp16_v$v.json | cut -c1-160)"
done
true
