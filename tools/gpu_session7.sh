#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s7; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_split.py -q -m gpu > $O/pytest_split.txt 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1
tail -4 $O/pytest_split.txt $O/pytest_all.txt
true
