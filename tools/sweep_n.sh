#!/bin/bash
# Usage: bash tools/sweep_n.sh k n1 n2 ...  -> bench (jaccard only) per cell count
K=$1; shift
for n in "$@"; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-gficf --cells-per-gpu $n --k $K 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
e=$n*$K
print('N=$n k=$K', 'ms/step %.4f kernel_ms %.4f (%.2f ns/kedge; %.1f Gedges/s kernel-only) ingest_ms %.4f frac %.3f ok=%s'%(d['ms_per_step'], r['kernel_ms'], r['kernel_ms']*1e6/e*1e3/1e3, e/r['kernel_ms']/1e6, r['ingest_kernel_ms'], r['frac'], d.get('checked_vs_oracle')))"
done
