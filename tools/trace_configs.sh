#!/bin/bash
# Plain rocprofv3 kernel-trace stats (no counters) of the config-4 and config-5 bench invocations.
# Usage (through gpurun): bash tools/trace_configs.sh <tag>   -> gpurun_out/<tag>/bench_<c>_kernel_stats.csv + the bench lines
TAG=${1:-r04}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for C in ${CONFIGS_TO_TRACE:-c4 c5}; do
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace_$C -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --config $C --no-cpu-baseline --no-live-traffic > $GRAFT_REPO_ROOT/$OUT/bench_${C}.json 2> $GRAFT_REPO_ROOT/$OUT/trace_$C.log) || { echo "trace $C failed"; tail -5 $OUT/trace_$C.log; exit 1; }
  f=$(find $OUT/trace_$C -name "bench_kernel_stats.csv" | head -1)
  cp "$f" $OUT/bench_${C}_kernel_stats.csv && head -6 $OUT/bench_${C}_kernel_stats.csv
done
find $OUT -name "*.db" -delete; find $OUT -size +3M -delete
