#!/bin/bash
for b in 3 4 5 6; do for f in "" "--no-pipeline"; do
export GFICF_JACCARD_BLOCKS_PER_CU=$b
timeout 300 python bench.py --no-gficf --no-cpu-baseline $f 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('blocks/CU=$b $f', '%.4g edges/s'%d['value'], 'ms/step %.4f'%d['ms_per_step'], 'kernel %.4f'%d['roofline']['kernel_ms'], d['checked_vs_oracle'])"
done; done
