export DS_LIB_PATH=diff_sampler_amd/csrc/libdsamd_timeline.so
python tools/timeline_gemm.py --conv 64 64 192 192 --res --stats
python tools/timeline_gemm.py --conv 64 64 192 192 --stats --silu
python tools/timeline_gemm.py --conv 64 32 384 384 --res --stats
python tools/timeline_gemm.py --conv 64 16 576 576 --res --stats
python tools/timeline_gemm.py --conv 32 64 320 320 --res --stats
