#!/bin/bash
# [VARIANT_SRC=gficf_csc] build_variant.sh NAME "-DFLAG=..." : libgficf_hip.so with one source (default jaccard.hip) compiled under extra flags, into tools/lab/ablNAME/
# (git-ignored; used with LD_LIBRARY_PATH in front of gather_lab / bench to A/B a compile-time choice on one box)
set -e
cd "$(dirname "$0")/../.."
NAME=$1; shift
SRCF=${VARIANT_SRC:-jaccard}
D=tools/lab/abl$NAME; mkdir -p $D
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -Iinclude "$@" -c gficf_amd/csrc/$SRCF.hip -o $D/$SRCF.o
OBJS=$(ls gficf_amd/csrc/*.o | grep -v /$SRCF.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libgficf_hip.so $D/$SRCF.o $OBJS
echo built $D/libgficf_hip.so
