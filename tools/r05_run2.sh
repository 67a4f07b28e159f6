#!/bin/bash
# round 5, GPU call 2: the one-launch form (tests, in-process A/B), k > 256 tests again, configs 1 / 2 lines
TAG=${1:-r05b}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_jaccard_gpu.py -x -q -m gpu -k "one_launch or beyond_256 or thousands or unsupported or pipelined" > $OUT/pytest_new.log 2>&1; echo "new tests rc=$?"; tail -5 $OUT/pytest_new.log
timeout -k 10 300 python tools/direct_ab.py > $OUT/direct_ab.txt 2>&1; echo "direct ab rc=$?"; cat $OUT/direct_ab.txt
for C in c1 c2 c3; do
  timeout -k 10 200 python bench.py --config $C --no-gficf --no-knn --no-live-traffic > $OUT/bench_$C.json 2> $OUT/bench_$C.err; echo "bench $C rc=$?"; cut -c1-300 $OUT/bench_$C.json
done
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
