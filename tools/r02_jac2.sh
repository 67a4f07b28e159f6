#!/bin/bash
# product kernel timing at several occupancies (compact and wide rows), then the Jaccard parity tests
TAG=${1:-r02c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for B in 3 6 8; do echo "== blocks per CU $B"; GFICF_JACCARD_BLOCKS_PER_CU=$B timeout -k 10 120 ./tools/lab/gather_lab 100000 30 1 2>&1 | grep -E "ids|PRODUCT"; done > $OUT/product_occ.txt 2>&1
echo "== compact off, 6 per CU" >> $OUT/product_occ.txt; GFICF_JACCARD_COMPACT=0 timeout -k 10 120 ./tools/lab/gather_lab 100000 30 1 2>&1 | grep -E "ids|PRODUCT" >> $OUT/product_occ.txt
cat $OUT/product_occ.txt
timeout -k 10 900 python -m pytest tests/test_jaccard_gpu.py -x -q -m gpu > $OUT/pytest_jac.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_jac.log
