#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s13; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1
tail -6 $O/pytest_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
tail -3 $O/smoke.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | cut -c1-400; tail -3 $O/bench.err
true
