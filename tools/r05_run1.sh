#!/bin/bash
# round 5, GPU call 1: the k > 256 path, the whole GPU suite, baseline lines of configs 1 / 2 before the one-call step
TAG=${1:-r05a}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_jaccard_gpu.py -x -q -m gpu -k "beyond_256 or thousands or truncat or option" > $OUT/pytest_bigk.log 2>&1; echo "bigk rc=$?"; tail -5 $OUT/pytest_bigk.log
timeout -k 10 900 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
for C in c1 c2; do
  timeout -k 10 200 python bench.py --config $C --no-gficf --no-knn --no-live-traffic > $OUT/bench_$C.json 2> $OUT/bench_$C.err; echo "bench $C rc=$?"; cut -c1-300 $OUT/bench_$C.json
done
