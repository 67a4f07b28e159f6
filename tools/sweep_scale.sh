#!/bin/bash
for v in "$@"; do
  export GFICF_SCALE_VARIANT=$v
  timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); g=d['gficf']; r=g['roofline']
print('variant=$v', 'ms/pass %.4f scale_ms %.4f count_ms %.4f'%(g['ms_per_pass'], r['scale_kernel_ms'], r['count_kernel_ms']))"
done
