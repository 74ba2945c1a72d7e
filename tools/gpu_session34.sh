#!/bin/bash
# Round-2 session 34: EDMDenoiser.block_output (the tensor persistence_hook hands to the AMED forward hooks) against the oracle's tap.
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s34; mkdir -p $O
timeout 70 python -m pytest tests/test_hip_amed.py -q -m gpu -k block_output > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
true
