#!/bin/bash
# Round-2 session 35: smoke() + the AMED / sampler suites after the import clean-up of engine.py / ops.py / solvers_amed.py.
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s35; mkdir -p $O
timeout 30 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 45 python -m pytest tests/test_hip_amed.py tests/test_hip_samplers.py -q -m gpu -x > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
true
