#!/bin/bash
# Config-4 edge kernel: the bit-set kernel on dual rows in several builds (cells in flight per wave x waves per workgroup:
# tools/lab/abl_bits/libgficf_hip_d<D>w<W>.so, built with -DGFICF_BITS_DEPTH / -DGFICF_BITS_WAVES) against the general kernel on
# plain compact rows (GFICF_JACCARD_DUAL=0), each as a `bench.py --config c4` line.  Usage (through gpurun): bash tools/bits_ab.sh <tag>
TAG=${1:-r04}; OUT=gpurun_out/$TAG; mkdir -p $OUT
IDS=permuted
run() { # name, env...
  name=$1_$IDS; shift
  env "$@" timeout -k 10 200 python bench.py --config c4 --ids $IDS --no-cpu-baseline --no-live-traffic > $OUT/bench_c4_$name.json 2>> $OUT/bits_ab.err
  python3 -c "
import json; d=json.load(open('$OUT/bench_c4_$name.json')); r=d['roofline']
print('%-26s value %.2f G edges/s  edges %.1f us in-run, %.1f back to back  ingest %.1f us  frac %.4f  row %d B  checked %s' % ('$name', d['value']/1e9, r['kernel_ms']*1e3, r['kernel_ms_back_to_back']*1e3, r['ingest_kernel_ms']*1e3, r['frac'], r['row_bytes'], d['checked_vs_oracle']))"
}
for IDS in permuted spatial; do
run general GFICF_JACCARD_DUAL=0
run bits_default GFICF_JACCARD_DUAL=1
for f in tools/lab/abl_bits/libgficf_hip_*.so; do
  [ -e "$f" ] || continue
  n=$(basename $f .so); n=${n#libgficf_hip_}
  run bits_$n GFICF_JACCARD_DUAL=1 GFICF_HIP_LIB=$PWD/$f
done
run general_again GFICF_JACCARD_DUAL=0
done
