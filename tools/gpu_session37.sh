#!/bin/bash
# Round-2 session 37: GITS schedule search on the full-size CIFAR-10 net against the real reference's dp_list (new golden).
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s37; mkdir -p $O
timeout 40 python -m pytest tests/test_hip_gits.py -q -m gpu -k full_size > $O/pytest.txt 2>&1
tail -6 $O/pytest.txt
true
