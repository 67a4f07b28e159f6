#!/bin/bash
# round 5, GPU call 4: N > 1 rehearsals of the restructured bench (2 and 5 ranks on the one GPU), the whole GPU suite, the
# phenograph order A/B, the one-buffer A/B, configs 4 / 5 lines with the spatial-ids figure
TAG=${1:-r05d}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for A in 2 5; do
  S=$SECONDS
  timeout -k 10 420 python bench.py --gpus $A --rehearse-one-gpu --steps 5 --no-gficf > $OUT/rehearsal_gpus$A.jsonl 2> $OUT/rehearsal_gpus$A.err; echo "rehearsal $A rc=$? in $((SECONDS-S)) s"
  tail -2 $OUT/rehearsal_gpus$A.err | cut -c1-300; wc -l $OUT/rehearsal_gpus$A.jsonl; tail -1 $OUT/rehearsal_gpus$A.jsonl | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d.get(k) for k in ('legs_done','leg_seconds','skipped_legs','wall_s','checked_vs_oracle')})"
done
timeout -k 10 120 python tools/one_buffer_ab.py > $OUT/one_buffer_ab.txt 2>&1; echo "one buffer rc=$?"; cat $OUT/one_buffer_ab.txt
timeout -k 10 300 python tools/phenograph_order_ab.py 400000 10 30 > $OUT/phenograph_order_ab.txt 2>&1; echo "phenograph ab rc=$?"; cat $OUT/phenograph_order_ab.txt
CONFIGS_TO_TRACE="c4 c5" bash tools/trace_configs.sh $TAG > $OUT/trace_configs.txt 2>&1; echo "trace configs rc=$?"; tail -12 $OUT/trace_configs.txt | cut -c1-220
timeout -k 10 1000 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.log
find $OUT -name "*.db" -delete; find $OUT -size +3M -delete
