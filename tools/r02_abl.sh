#!/bin/bash
OUT=gpurun_out/${1:-r02p}; mkdir -p $OUT
for V in "" abl2 abl3 abl4; do
  echo "== variant '$V' (abl2: no LDS probes; abl3: no hash work at all, fold + group sum; abl4: fold only)"
  if [ -z "$V" ]; then timeout -k 10 100 ./tools/lab/gather_lab 100000 30 1 2>&1 | grep -E "back to back"
  else LD_LIBRARY_PATH=$PWD/tools/lab/$V:$LD_LIBRARY_PATH timeout -k 10 100 ./tools/lab/gather_lab 100000 30 1 2>&1 | grep -E "back to back"; fi
done > $OUT/ablation.txt 2>&1
cat $OUT/ablation.txt
