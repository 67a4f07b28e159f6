#!/bin/bash
OUT=gpurun_out/${1:-r02lab2}; mkdir -p $OUT
timeout -k 10 200 ./tools/lab/gather_lab 100000 30 0 > $OUT/gather.txt 2>&1; echo "gather rc=$?"
for V in "" st0 st1 st3; do
  echo "== store aux variant '$V' ('' = product, aux 2)"
  if [ -z "$V" ]; then timeout -k 10 100 ./tools/lab/gather_lab 100000 30 1 2>&1 | grep -E "back to back"
  else LD_LIBRARY_PATH=$PWD/tools/lab/abl$V:$LD_LIBRARY_PATH timeout -k 10 100 ./tools/lab/gather_lab 100000 30 1 2>&1 | grep -E "back to back"; fi
done > $OUT/store_aux.txt 2>&1
grep -E "ids|PLAIN|folded|grid  2048" $OUT/gather.txt; cat $OUT/store_aux.txt
