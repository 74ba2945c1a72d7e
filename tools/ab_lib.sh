#!/bin/bash
# A/B of two builds of libdsamd.so in one gpurun session (alternating): tools/ab_lib.sh <alt .so> -- layers alone, then the two fp16 sampler lines.
# Used for the halo swizzle of 8- / 16-column images (docs/HISTORY.md E.21): gpurun_alt/libdsamd_oldswz.so = the previous swizzle.
ALT=$1
for rep in 1 2; do
  for lib in "" "$ALT"; do
    echo "LIB=$lib"
    DS_LIB_PATH=$lib python tools/bench_conv.py --shapes imagenet64 --batch 64 --f16 --dma16 --f16io --ws --only 4 5 6 2>/dev/null
    DS_LIB_PATH=$lib python tools/bench_conv.py --shapes sd15 --batch 32 --f16 --dma16 --f16io --ws --only 4 5 6 7 2>/dev/null
  done
done
for lib in "" "$ALT" "" "$ALT"; do
  echo "LIB=$lib"
  DS_LIB_PATH=$lib python bench.py --config imagenet64 --dtype fp16 --steps 3 --warmup 1 --no-cpu-baseline --no-launch-modes --no-batch-sweep --no-other-configs 2>/dev/null | python -c "import sys,json; z=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(z['metric'], z['value'], z['ms_per_step'])"
done
for lib in "" "$ALT"; do
  echo "LIB=$lib"
  DS_LIB_PATH=$lib python bench.py --config sd15 --dtype fp16 --steps 3 --warmup 1 --no-cpu-baseline --no-launch-modes --no-batch-sweep --no-other-configs 2>/dev/null | python -c "import sys,json; z=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(z['metric'], z['value'], z['ms_per_step'])"
done
