#!/bin/bash
OUT=gpurun_out/${1:-r02o}; mkdir -p $OUT
for V in "" "GFICF_JACCARD_NO_PIPE=1" "GFICF_JACCARD_COMPACT=0" "GFICF_JACCARD_BLOCKS_PER_CU=3" "GFICF_JACCARD_BLOCKS_PER_CU=8"; do
  echo "== $V"; env $V timeout -k 10 100 ./tools/lab/gather_lab 100000 30 1 2>&1 | grep -E "back to back"
done > $OUT/pipe.txt 2>&1
cat $OUT/pipe.txt
timeout -k 10 600 python -m pytest tests/test_jaccard_gpu.py tests/test_multi_gpu.py tests/test_adjacency_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
