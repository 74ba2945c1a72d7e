#!/bin/bash
# Round-2 session 28: SQ counters of the final 256 x 256 conv kernel (variant 0) and of its plain predecessor (variant 256) on the main shape
# (counters in their own runs, no tracing domains).
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s28; mkdir -p $O
for V in 0 256; do
  timeout 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $O/pmcA_v$V -- python tools/bench_conv.py --batch 256 --only 0 --variants $V --rounds 1 --iters 2 --norm > $O/pmcA_v$V.log 2>&1
  timeout 150 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_LDS -d $O/pmcB_v$V -- python tools/bench_conv.py --batch 256 --only 0 --variants $V --rounds 1 --iters 2 --norm > $O/pmcB_v$V.log 2>&1
  python tools/rocprof_summary.py counters $O/pmc_v$V.json $(find $O/pmcA_v$V $O/pmcB_v$V -name "*.db") > $O/pmc_summary_v$V.txt 2>&1
done
find $O -name "*.db" -delete
head -c 1500 $O/pmc_v0.json
true
