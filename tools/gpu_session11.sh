#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s11; mkdir -p $O
DS_CONV=256 timeout 200 python tools/bench_conv.py --batch 256 --only 0 1 3 4 --norm --variants 0 2 4 8 16 28 --rounds 3 --iters 5 > $O/tile256.txt 2>&1
DS_CONV=128 timeout 200 python tools/bench_conv.py --batch 256 --only 0 1 3 4 --norm --variants 0 2 4 8 16 28 --rounds 3 --iters 5 > $O/tile128.txt 2>&1
grep -v amdgpu.ids $O/tile256.txt | tail -30; grep -v amdgpu.ids $O/tile128.txt | tail -30
true
