#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s18; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "wide_n or halo2 or halo" > $O/pytest_wide.txt 2>&1
tail -5 $O/pytest_wide.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | cut -c1-300; tail -2 $O/bench.err
true
