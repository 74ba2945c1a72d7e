# session r10g: HBM traffic of single GEMM launches (separate --pmc passes)
O=gpurun_out/r10g
for shp in "13 --geglu" "1" "2"; do
  tag=$(echo $shp | tr -d ' -')
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    ct=$(echo $c | cut -d' ' -f1)
    timeout 300 rocprofv3 --pmc $c -d $O/p_${tag}_$ct -o x -- python tools/bench_conv.py --shapes sd15gemm --batch 32 --f16 --dma16 --f16io --only $shp --iters 3 > $O/log_${tag}_$ct.txt 2>&1
  done
  python tools/rocprof_summary.py counters $O/c_$tag.json $(find $O -name "*.db" -path "*p_${tag}_*") > /dev/null 2>&1
  python - <<PY
import json
d=json.load(open('$O/c_$tag.json'))
for k,v in d.items():
    if 'gemm_f16dma' in k: print('$tag', k[:60], {a:round(b/1024,1) if 'SIZE' in a else b for a,b in v.items()})
PY
done
find $O -name "*.db" -delete
