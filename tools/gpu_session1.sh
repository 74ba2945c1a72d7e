#!/bin/bash
# GPU session 1 (round 2): validate the host-side changes, A/B the pipelined halo kernel + ablations, counters.
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s1; mkdir -p $O
python tools/bench_conv.py --batch 256 --only 0 1 3 4 6 --variants 0 1 2 4 8 16 28 5 17 29 --rounds 5 --iters 5 --norm > $O/ab_norm.txt 2>&1
python tools/bench_conv.py --batch 256 --only 0 --variants 0 1 --rounds 5 --iters 5 > $O/ab_nonorm.txt 2>&1
DS_CONV_VARIANT=1 timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_denoiser.py tests/test_hip_full_goldens.py -x -q -m gpu > $O/pytest_variant1.txt 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1
python bench.py --steps 3 --warmup 1 > $O/bench_v0.json 2> $O/bench_v0.err
DS_CONV_VARIANT=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-launch-modes > $O/bench_v1.json 2> $O/bench_v1.err
for V in 0 1; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $O/pmcA_v$V -- python tools/bench_conv.py --batch 256 --only 0 --variants $V --rounds 1 --iters 2 --norm > $O/pmcA_v$V.log 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_LDS -d $O/pmcB_v$V -- python tools/bench_conv.py --batch 256 --only 0 --variants $V --rounds 1 --iters 2 --norm > $O/pmcB_v$V.log 2>&1
done
for V in 0 1; do python tools/rocprof_summary.py counters $O/pmc_v$V.json $(find $O/pmcA_v$V $O/pmcB_v$V -name "*.db"); done > $O/pmc_summary.txt 2>&1
find $O -name "*.db" -size +8M -delete
