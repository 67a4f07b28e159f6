#!/bin/bash
OUT=gpurun_out/${1:-r02s}; mkdir -p $OUT
for P in "" "GFICF_JACCARD_NO_PIPE=1"; do for B in 4 6 8 10 12; do
  echo "== $P blocks=$B"; env $P GFICF_JACCARD_BLOCKS_PER_CU=$B timeout -k 10 100 ./tools/lab/gather_lab 100000 30 1 2>&1 | grep -E "back to back"
done; done > $OUT/sweep.txt 2>&1
cat $OUT/sweep.txt
