#!/bin/bash
OUT=gpurun_out/${1:-r02z}; mkdir -p $OUT
for R in 1 2; do for V in 1024 768 640 896; do
  GFICF_HIP_LIB=$PWD/tools/lab/libgficf_sl$V.so timeout -k 10 200 python bench.py --no-knn --no-cpu-baseline > $OUT/b.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("$OUT/b.json")); g=d["gficf"]
print("SL_THREADS $V: gficf ms %.4f frac %.4f scale_ms %.4f  (jaccard ms/ds %.4f)"%(g["ms_per_pass"], g["roofline"]["frac"], g["roofline"]["scale_kernel_ms"], d["ms_per_data_set"]))
PY
done; done | tee $OUT/gfab.txt
