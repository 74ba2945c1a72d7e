#!/bin/bash
# Phase timelines of the fp16 convolution and GEMM on the layer shapes of the two fp16 lines (profiles/r6_gemm_timeline.txt (3), (9)); needs the
# diagnostics build: python -c "from diff_sampler_amd import build; build.build_libs(('timeline',))"
export DS_LIB_PATH=diff_sampler_amd/csrc/libdsamd_timeline.so
python tools/timeline_gemm.py --conv 64 64 192 192 --res --stats
python tools/timeline_gemm.py --conv 64 64 192 192 --stats --silu
python tools/timeline_gemm.py --conv 64 32 384 384 --res --stats
python tools/timeline_gemm.py --conv 64 16 576 576 --res --stats
python tools/timeline_gemm.py --conv 32 64 320 320 --res --stats
python tools/timeline_gemm.py --m 131072 --k 320 --n 960
python tools/timeline_gemm.py --m 131072 --k 320 --n 2560 --geglu
python tools/timeline_gemm.py --m 131072 --k 1280 --n 320 --res
