#!/bin/bash
# Round-2 session 26: is the 256 x 256 tile's epilogue an HBM burst (cost grows with the number of CUs that reach it together) or a per-CU
# cost?  One round of tiles on 64 / 128 / 256 CUs (B = 16 / 32 / 64 images of 32 x 32, forced tile shape, variant 6), then 2 and 4 rounds.
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s26; mkdir -p $O
for b in 8 16 32 64; do
  DS_CONV=256 timeout 100 python tools/bench_conv.py --batch $b --norm --only 0 --rounds 7 --iters 20 --variants 6 >> $O/tiles_vs_time.txt 2>&1
done
grep "^\[" $O/tiles_vs_time.txt | cut -c1-200
true
