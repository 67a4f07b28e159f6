#!/bin/bash
# round 5, GPU call 3: the restructured bench.py (default line, legs, budget), configs 1-3 with enough steps + traces, N > 1 rehearsals
# (2 and 5 ranks on the one GPU: 5 ranks + the peer child = the box's limit of 6 processes on the card), bits-kernel lab variants
TAG=${1:-r05c}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-300 $OUT/bench.json; tail -3 $OUT/bench.err
CONFIGS_TO_TRACE="c1 c2 c3" bash tools/trace_configs.sh $TAG > $OUT/trace_configs.txt 2>&1; echo "trace configs rc=$?"; tail -20 $OUT/trace_configs.txt | cut -c1-200
for A in 2 5; do
  /usr/bin/time -f "rehearsal $A ranks: %e s wall" timeout -k 10 420 python bench.py --gpus $A --rehearse-one-gpu --steps 5 --no-gficf > $OUT/rehearsal_gpus$A.jsonl 2> $OUT/rehearsal_gpus$A.err; echo "rehearsal $A rc=$?"
  tail -2 $OUT/rehearsal_gpus$A.err; wc -l $OUT/rehearsal_gpus$A.jsonl; tail -1 $OUT/rehearsal_gpus$A.jsonl | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d.get(k) for k in ('legs_done','leg_seconds','skipped_legs','wall_s','checked_vs_oracle')})"
done
bash tools/bits_ab.sh $TAG > $OUT/bits_ab.txt 2>&1; cat $OUT/bits_ab.txt
timeout -k 10 600 python -m pytest tests/test_dist_gpu.py tests/test_multi_gpu.py tests/test_glue_run.py -q -m gpu -x > $OUT/pytest_part.log 2>&1; echo "pytest part rc=$?"; tail -3 $OUT/pytest_part.log
find $OUT -name "*.db" -delete; find $OUT -size +3M -delete
