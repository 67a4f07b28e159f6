#!/bin/bash
# the adjacency build three ways on ONE box, interleaved: tools/lab/abladj{old,packed,rows}/libgficf_hip.so (built by hand from the three sources)
for round in 1 2; do for v in old packed rows; do echo "variant $v round $round"; GFICF_HIP_LIB=$GRAFT_REPO_ROOT/tools/lab/abladj$v/libgficf_hip.so timeout -k 10 200 python tools/adjacency_time.py 54000 30 100000 50 1000000 30 2>&1 | grep "N="; done; done
