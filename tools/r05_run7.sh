#!/bin/bash
# round 5, GPU call 7: the compact return inside gficf_jaccard_host, A/B + the host-entry tests (Python and the `.Call` glue)
TAG=${1:-r05g}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python -u tools/host_compact_ab.py > $OUT/host_compact_ab.txt 2> $OUT/host_compact_ab.err; echo "ab rc=$?"; cat $OUT/host_compact_ab.txt; tail -3 $OUT/host_compact_ab.err
timeout -k 10 900 python -m pytest tests/test_jaccard_gpu.py tests/test_glue_run.py tests/test_multi_gpu.py -q -m gpu -x > $OUT/pytest_host.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_host.log
