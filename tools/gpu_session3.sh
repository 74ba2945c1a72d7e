#!/bin/bash
# GPU session 3: fp16-operand convolution (kernel parity, engine parity, throughput), batched Philox RNG, counters of the v2 fp32 kernel.
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s3; mkdir -p $O
timeout 300 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "f16 or halo2" > $O/pytest_kernels.txt 2>&1
timeout 200 python -m pytest tests/test_hip_rng.py -q -m gpu > $O/pytest_rng.txt 2>&1
timeout 120 python tools/bench_conv.py --batch 256 --only 0 1 3 4 6 --iters 5 --norm --f16 > $O/conv_f16.txt 2>&1
timeout 120 python tools/bench_conv.py --batch 256 --only 0 1 3 --iters 5 --norm --f16 --extra > $O/conv_f16_extra.txt 2>&1
timeout 600 python -m pytest tests/test_hip_fp16.py -x -q -m gpu > $O/pytest_fp16.txt 2>&1
timeout 200 python bench.py --config imagenet64 --batch 64 --solver ipndm --steps 2 --warmup 1 --no-cpu-baseline --no-launch-modes > $O/bench_in64_fp32.json 2> $O/bench_in64_fp32.err
timeout 200 python bench.py --config imagenet64 --batch 64 --solver ipndm --steps 2 --warmup 1 --no-cpu-baseline --no-launch-modes --dtype fp16 > $O/bench_in64_fp16.json 2> $O/bench_in64_fp16.err
timeout 200 python bench.py --config sd15 --batch 16 --steps 2 --warmup 1 --no-cpu-baseline --dtype fp16 > $O/bench_sd15_fp16.json 2> $O/bench_sd15_fp16.err
timeout 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $O/pmcA_v3 -- python tools/bench_conv.py --batch 256 --only 0 --variants 3 --rounds 1 --iters 2 --norm > $O/pmcA_v3.log 2>&1
python tools/rocprof_summary.py counters $O/pmc_v3.json $(find $O/pmcA_v3 -name "*.db") > $O/pmc_summary.txt 2>&1
find $O -name "*.db" -delete
for f in $O/*.txt; do echo "== $f"; tail -4 $f; done
true
