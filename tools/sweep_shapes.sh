#!/bin/bash
# Edge-kernel figures over the shapes the callers use (clustcells() default k = 15; the BASELINE configs).  Usage: bash tools/sweep_shapes.sh <tag>
OUT=gpurun_out/${1:-shapes}; mkdir -p $OUT
for S in "10000 30" "54000 30" "100000 15" "100000 30" "100000 50" "100000 100" "1000000 15" "1000000 30"; do
  set -- $S
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --cells-per-gpu $1 --k $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
e=$1*$2
print('N=%8d k=%3d  row %3d B  ingest %7.1f us  edges %8.1f us in-run (%8.1f back to back)  %5.1f G edges/s kernel-only  frac %.3f  whole data set %5.1f G edges/s' % ($1, $2, r['row_bytes'], r['ingest_kernel_ms']*1e3, r['kernel_ms']*1e3, r['kernel_ms_back_to_back']*1e3, e/r['kernel_ms']/1e6, r['frac'], d['value']/1e9))"
done | tee $OUT/shapes.txt
