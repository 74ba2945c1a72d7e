#!/bin/bash
# Round-2 session 33: the default bench line on the closing tree.
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s33; mkdir -p $O
timeout 125 python bench.py > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | cut -c1-300
true
