#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s4; mkdir -p $O
timeout 100 python tools/probe_rng.py > $O/probe_rng.json 2> $O/probe_rng.err
timeout 400 python -m pytest tests/test_hip_samplers.py tests/test_hip_graph.py tests/test_hip_full_goldens.py tests/test_hip_amed.py -x -q -m gpu > $O/pytest_samplers.txt 2>&1
timeout 200 python bench.py --steps 3 --warmup 1 --cpu-threads 16 > $O/bench.json 2> $O/bench.err
tail -3 $O/pytest_samplers.txt
true
