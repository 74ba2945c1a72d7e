#!/bin/bash
# Round-2 final session (second half of the round): full GPU suite, headline bench line, rocprofv3 kernel stats + HBM counters of the
# headline, bench lines of the other fp32 configurations that run the 256 x 256 conv tile.
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s24; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1
tail -4 $O/pytest_all.txt
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | cut -c1-260
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --no-cpu-baseline --no-launch-modes --no-batch-sweep > $O/bench_under_rocprof.json 2> $O/prof.err
python tools/rocprof_summary.py stats $(find $O/prof -name "*.db" | head -1) $O/kernel_stats.txt > /dev/null 2>> $O/prof.err
head -8 $O/kernel_stats.txt
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-launch-modes --no-batch-sweep > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-launch-modes --no-batch-sweep > $O/pmc_write.log 2>&1
python tools/rocprof_summary.py pmc $(find $O/pmc_fetch -name "*.db" | head -1) $(find $O/pmc_write -name "*.db" | head -1) $O/pmc_hbm.json > $O/pmc_summary.txt 2>&1
head -5 $O/pmc_summary.txt | cut -c1-300
find $O -name "*.db" -delete
timeout 200 python bench.py --dtype fp16x3 --no-cpu-baseline --no-launch-modes --no-batch-sweep > $O/bench_fp16x3.json 2> $O/bench_fp16x3.err
timeout 200 python bench.py --config ffhq --batch 128 --no-cpu-baseline --no-launch-modes > $O/bench_ffhq_fp32.json 2> $O/bench_ffhq_fp32.err
timeout 200 python bench.py --config imagenet64 --batch 64 --solver ipndm --no-cpu-baseline --no-launch-modes > $O/bench_in64_fp32.json 2> $O/bench_in64_fp32.err
for f in $O/bench_*.json; do echo "== $f"; tail -1 $f | cut -c1-200; done
true
