#!/bin/bash
# One GPU session = one `gpurun` call.  Usage (on the GPU box, from the repo root):
#     bash tools/gpu_session.sh <tag> <step> [<step> ...]
# Everything a step writes goes to gpurun_out/<tag>/ (merged back by gpurun); summaries worth keeping are copied to profiles/ by hand.
# Steps (each bounded by its own timeout so that a hang cannot become a gpurun strike):
#     suite                     the whole GPU test suite                          -> pytest_all.txt
#     tests:<pytest args>       e.g. "tests:tests/test_hip_fp16.py -k edm"        -> pytest_<n>.txt
#     smoke                     __graft_entry__.smoke()                           -> smoke.txt
#     bench[:<bench.py args>]   one bench line (args may name --config/--dtype)   -> bench_<n>.json
#     prof[:<bench.py args>]    rocprofv3 --kernel-trace --stats of bench.py      -> kernel_stats_<n>.txt + the line under rocprof
#     pmc[:<bench.py args>]     separate FETCH_SIZE / WRITE_SIZE passes           -> pmc_hbm_<n>.json
#     sq[:<bench.py args>]      SQ counter passes (MFMA busy, VALU / LDS instr.)  -> sq_counters_<n>.json
#     run:<command>             any command (output -> run_<n>.txt)
set -x
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
QUICK="--no-cpu-baseline --no-launch-modes --no-batch-sweep --no-other-configs"
n=0
for step in "$@"; do
  n=$((n+1)); kind=${step%%:*}; arg=""; [ "$kind" != "$step" ] && arg=${step#*:}
  case $kind in
    suite) timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1; tail -4 $O/pytest_all.txt ;;
    tests) eval "timeout 900 python -m pytest $arg -q -m gpu" > $O/pytest_$n.txt 2>&1; tail -15 $O/pytest_$n.txt ;;   # eval: -k 'a or b' stays one argument
    smoke) timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt ;;
    bench) timeout 600 python bench.py $arg > $O/bench_$n.json 2> $O/bench_$n.err; tail -1 $O/bench_$n.json | cut -c1-300; tail -2 $O/bench_$n.err ;;
    prof)  timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_$n -o bench -- python bench.py $QUICK $arg > $O/bench_under_rocprof_$n.json 2> $O/prof_$n.err
           python tools/rocprof_summary.py stats $(find $O/prof_$n -name "*.db" | head -1) $O/kernel_stats_$n.txt > /dev/null 2>> $O/prof_$n.err
           head -12 $O/kernel_stats_$n.txt; tail -1 $O/bench_under_rocprof_$n.json | cut -c1-200 ;;
    pmc)   for c in FETCH_SIZE WRITE_SIZE; do
             timeout 400 rocprofv3 --pmc $c -d $O/pmc_${c}_$n -o bench -- python bench.py --steps 1 --warmup 1 $QUICK $arg > $O/pmc_${c}_$n.log 2>&1
           done
           python tools/rocprof_summary.py pmc $(find $O/pmc_FETCH_SIZE_$n -name "*.db" | head -1) $(find $O/pmc_WRITE_SIZE_$n -name "*.db" | head -1) $O/pmc_hbm_$n.json > $O/pmc_summary_$n.txt 2>&1
           head -5 $O/pmc_summary_$n.txt | cut -c1-300 ;;
    sq)    i=0; dbs=""
           for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
             i=$((i+1))
             timeout 400 rocprofv3 --pmc $c -d $O/sq_${n}_$i -o bench -- python bench.py --steps 1 --warmup 1 $QUICK $arg > $O/sq_${n}_$i.log 2>&1
             dbs="$dbs $(find $O/sq_${n}_$i -name "*.db" | head -1)"
           done
           python tools/rocprof_summary.py counters $O/sq_counters_$n.json $dbs > $O/sq_summary_$n.txt 2>&1; head -6 $O/sq_summary_$n.txt | cut -c1-400 ;;
    run)   timeout 900 bash -c "$arg" > $O/run_$n.txt 2>&1; tail -25 $O/run_$n.txt ;;
    *) echo "unknown step $step" ;;
  esac
done
find $O -name "*.db" -delete
true
