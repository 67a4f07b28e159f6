#!/bin/bash
# Quick GPU check of the Jaccard path: product kernel timings on the three id models, then the parity tests that touch it.
# Usage: bash tools/quick.sh <tag>
TAG=${1:-q}
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 120 ./tools/lab/gather_lab 100000 30 1 2>&1 | grep -E "ids|PRODUCT" > $OUT/product.txt; cat $OUT/product.txt
timeout -k 10 1000 python -m pytest tests/test_jaccard_gpu.py tests/test_multi_gpu.py tests/test_dist_gpu.py tests/test_adjacency_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest.log
