#!/bin/bash
for v in "" noprobe linprobe nogather nostore; do
  if [ -n "$v" ]; then export GFICF_HIP_LIB=$PWD/gficf_amd/lab_$v.so; else unset GFICF_HIP_LIB; fi
  timeout 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-gficf 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('variant=${v:-product}', 'ms/step %.4f kernel_ms %.4f'%(d['ms_per_step'], r['kernel_ms']))"
done
