#!/bin/bash
# Round-2 session 31: configs 3 and 4 at full size against the REAL reference's own sampler trajectories (new goldens).
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s31; mkdir -p $O
timeout 150 python -m pytest tests/test_hip_full_goldens.py -q -m gpu -k "config3 or config4" > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
true
