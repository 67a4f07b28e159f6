#!/bin/bash
# One kept run of the round: the default `python bench.py` line AND, in this same gpurun call, the rocprofv3 kernel trace of the same command —
# any kept run can then be the page's detailed record (tools/make_results.py takes the one nearest the medians).  Usage (through gpurun):
#   bash tools/run_record.sh <tag> <name>      ->  gpurun_out/<tag>/runs/<name>.json, <name>_traced.json, <name>_kernel_stats.csv
TAG=${1:-r06}; NAME=${2:-call}
OUT=gpurun_out/$TAG/runs
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 400 python bench.py > $OUT/$NAME.json 2> $OUT/$NAME.err; echo "bench rc=$?"; cut -c1-300 $OUT/$NAME.json
(cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace_$NAME -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-live-traffic > $GRAFT_REPO_ROOT/$OUT/${NAME}_traced.json 2> $GRAFT_REPO_ROOT/$OUT/${NAME}_trace.log); echo "trace rc=$?"
cp $(find $OUT/trace_$NAME -name "bench_kernel_stats.csv" | head -1) $OUT/${NAME}_kernel_stats.csv && head -4 $OUT/${NAME}_kernel_stats.csv | cut -c1-160
rm -rf $OUT/trace_$NAME; find $OUT -name "*.db" -delete
