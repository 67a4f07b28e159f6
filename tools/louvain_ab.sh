#!/bin/bash
# A/B of the small move kernel's compile-time choices on ONE box, variants interleaved (tools/lab/build_variant.sh lvXY).  Usage: bash tools/louvain_ab.sh N k n_start
for round in 1 2; do
  for v in 00 10 01 11; do
    echo "variant copy_outer=${v:0:1} ballot_insert=${v:1:1} round $round"
    GFICF_HIP_LIB=$GRAFT_REPO_ROOT/tools/lab/abllv$v/libgficf_hip.so timeout -k 10 200 python tools/louvain_time.py "$@" 7 2>&1 | grep louvain_device
  done
done
