#!/bin/bash
OUT=gpurun_out/${1:-r02g}; mkdir -p $OUT
for V in "" "GFICF_LAB_INGEST_NODUP=1" "GFICF_JACCARD_COMPACT=0"; do
  echo "== $V"; env $V timeout -k 10 100 ./tools/lab/gather_lab 100000 30 1 2>&1 | grep -E "back to back"
done > $OUT/ingest_variants.txt 2>&1
cat $OUT/ingest_variants.txt
timeout -k 10 600 python -m pytest tests/test_jaccard_gpu.py tests/test_multi_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
