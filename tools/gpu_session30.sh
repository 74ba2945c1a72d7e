#!/bin/bash
# Round-2 session 30 (closing): rocprofv3 kernel stats + bench line of the final tree (half-size-wave tiles on), then the whole GPU suite.
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s30; mkdir -p $O
timeout 150 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --no-cpu-baseline --no-launch-modes --no-batch-sweep > $O/bench_under_rocprof.json 2> $O/prof.err
python tools/rocprof_summary.py stats $(find $O/prof -name "*.db" | head -1) $O/kernel_stats.txt > /dev/null 2>> $O/prof.err
find $O -name "*.db" -delete
head -8 $O/kernel_stats.txt
tail -1 $O/bench_under_rocprof.json | cut -c1-200
timeout 340 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1
tail -3 $O/pytest_all.txt
true
