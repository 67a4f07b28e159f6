#!/bin/bash
# Round-2 Jaccard check: parity tests, then the product kernel at several occupancies (tools/lab/gather_lab).
TAG=${1:-r02b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests/test_jaccard_gpu.py tests/test_adjacency_gpu.py -x -q -m gpu > $OUT/pytest_jac.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_jac.log
for B in 3 4 6 8 12; do echo "== blocks per CU $B"; GFICF_JACCARD_BLOCKS_PER_CU=$B timeout -k 10 120 ./tools/lab/gather_lab 100000 30 1 2>&1 | grep -E "ids|PRODUCT"; done > $OUT/product_occ.txt 2>&1
echo "== compact off"; GFICF_JACCARD_COMPACT=0 timeout -k 10 120 ./tools/lab/gather_lab 100000 30 1 2>&1 | grep -E "ids|PRODUCT" >> $OUT/product_occ.txt
cat $OUT/product_occ.txt
