#!/bin/bash
# Final round-2 session: full GPU suite, bench lines for every configuration / dtype, rocprofv3 kernel stats and HBM counters of the headline.
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s17; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1
tail -4 $O/pytest_all.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | cut -c1-260
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --no-cpu-baseline --no-launch-modes > $O/bench_under_rocprof.json 2> $O/prof.err
python tools/rocprof_summary.py stats $(find $O/prof -name "*.db" | head -1) $O/kernel_stats.txt > /dev/null 2>> $O/prof.err
head -12 $O/kernel_stats.txt
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-launch-modes > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-launch-modes > $O/pmc_write.log 2>&1
python tools/rocprof_summary.py pmc $(find $O/pmc_fetch -name "*.db" | head -1) $(find $O/pmc_write -name "*.db" | head -1) $O/pmc_hbm.json > $O/pmc_summary.txt 2>&1
head -5 $O/pmc_summary.txt | cut -c1-300
find $O -name "*.db" -delete
timeout 300 python bench.py --dtype fp16x3 --no-cpu-baseline --no-launch-modes > $O/bench_fp16x3.json 2> $O/bench_fp16x3.err
timeout 300 python bench.py --config imagenet64 --batch 64 --solver ipndm --no-cpu-baseline --no-launch-modes > $O/bench_in64_fp32.json 2> $O/bench_in64_fp32.err
timeout 300 python bench.py --config imagenet64 --batch 64 --solver ipndm --dtype fp16 --no-cpu-baseline --no-launch-modes > $O/bench_in64_fp16.json 2> $O/bench_in64_fp16.err
timeout 300 python bench.py --config ffhq --batch 128 --no-cpu-baseline --no-launch-modes > $O/bench_ffhq_fp32.json 2> $O/bench_ffhq_fp32.err
timeout 400 python bench.py --config sd15 --batch 16 --no-cpu-baseline --no-launch-modes > $O/bench_sd15_fp32.json 2> $O/bench_sd15_fp32.err
timeout 300 python bench.py --config sd15 --batch 16 --dtype fp16 --no-cpu-baseline --no-launch-modes > $O/bench_sd15_fp16.json 2> $O/bench_sd15_fp16.err
for f in $O/bench_*.json; do echo "== $f"; tail -1 $f | cut -c1-200; done
true
