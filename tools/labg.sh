#!/bin/bash
for v in "" nogenegather; do
  if [ -n "$v" ]; then export GFICF_HIP_LIB=$PWD/gficf_amd/labg_$v.so; else unset GFICF_HIP_LIB; fi
  timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); g=d['gficf']; r=g['roofline']
print('variant=${v:-product}', 'ms/pass %.4f scale_ms %.4f count_ms %.4f'%(g['ms_per_pass'], r['scale_kernel_ms'], r['count_kernel_ms']))"
done
