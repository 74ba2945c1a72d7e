#!/bin/bash
# Round-2 session 36: every solver family on the full-size CIFAR-10 net against the real reference's outputs (new golden).
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s36; mkdir -p $O
timeout 50 python -m pytest tests/test_hip_full_goldens.py -q -m gpu -k every_solver > $O/pytest.txt 2>&1
tail -6 $O/pytest.txt
true
