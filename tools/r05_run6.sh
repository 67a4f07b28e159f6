#!/bin/bash
# round 5, GPU call 6: where does the sorted-row path beat the general kernel (56 < k <= 256)?  + the new sorted-path test
TAG=${1:-r05f}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python -u tools/sorted_vs_general.py > $OUT/sorted_vs_general.txt 2> $OUT/sorted_vs_general.err; echo "ab rc=$?"; cat $OUT/sorted_vs_general.txt
timeout -k 10 200 python -m pytest tests/test_jaccard_gpu.py -q -m gpu -x -k "local_ids_with_empty" > $OUT/pytest_new.log 2>&1; echo "new test rc=$?"; tail -3 $OUT/pytest_new.log
