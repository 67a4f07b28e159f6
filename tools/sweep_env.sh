#!/bin/bash
# Usage: bash tools/sweep_env.sh VAR v1 v2 ...   -> bench (jaccard only) per value
VAR=$1; shift
for v in "$@"; do
  export $VAR=$v
  timeout 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-gficf --no-knn 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$VAR=$v', 'ms/step %.4f kernel_ms %.4f standalone %.4f ingest_ms %.4f ok=%s'%(d['ms_per_step'], r['kernel_ms'], r['kernel_ms_standalone'], r['ingest_kernel_ms'], d.get('checked_vs_oracle')))"
done
