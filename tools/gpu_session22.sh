#!/bin/bash
# Round-2 session 22: dual-stream half-batch probe (tools/probe_dual_stream.py).
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s22; mkdir -p $O
timeout 300 python tools/probe_dual_stream.py --half 128 256 > $O/dual_stream.txt 2>&1
cat $O/dual_stream.txt | tail -20
true
