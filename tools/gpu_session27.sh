#!/bin/bash
# Round-2 session 27: the new benchmark-batch golden test + the FID moment tests on the final tree.
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s27; mkdir -p $O
timeout 400 python -m pytest tests/test_hip_full_goldens.py tests/test_hip_fid.py -q -m gpu > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
true
