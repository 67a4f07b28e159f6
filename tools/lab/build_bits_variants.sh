#!/bin/bash
# Builds of libgficf_hip.so with other (cells in flight per wave) x (waves per workgroup) of the bit-set edge kernel, for
# tools/bits_ab.sh: tools/lab/abl_bits/libgficf_hip_d<D>w<W>.so.  Usage: bash tools/lab/build_bits_variants.sh "3 3" "2 4" "2 1" ...
set -e
cd "$(dirname "$0")/../../gficf_amd/csrc"
make -s
mkdir -p ../../tools/lab/abl_bits
for V in "$@"; do
  set -- $V; D=$1; W=$2; X=${3:-}     # X: extra macro of a lab variant (PREFETCH, WHATIF_HALF_SET — the latter gives WRONG counts: timing only)
  N=d${D}w${W}${X:+_$X}
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -I../../include -DGFICF_BITS_DEPTH=$D -DGFICF_BITS_WAVES=$W ${X:+-DGFICF_BITS_$X} -c jaccard.hip -o /tmp/jaccard_$N.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/lab/abl_bits/libgficf_hip_$N.so ctx.o /tmp/jaccard_$N.o halo.o gficf_csc.o knn.o adjacency.o transpose.o louvain.o phenograph.o multi.o
done
