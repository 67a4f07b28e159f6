#!/bin/bash
# The config-4 and config-5 bench lines (plain runs) and plain rocprofv3 kernel-trace stats (no counters) of the same invocations.
# Usage (through gpurun): bash tools/trace_configs.sh <tag>   -> gpurun_out/<tag>/bench_<c>_kernel_stats.csv + the bench lines
TAG=${1:-r04}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for C in ${CONFIGS_TO_TRACE:-c4 c5}; do
  # configs 1-3 are ONE data set of a few microseconds per step: K = 20 steps would time the fences, not the steps (round 4's 23 / 27.6 us
  # "steps" were mostly that) — their lines take K = 2000 steps (a 10-60 ms timed region)
  case $C in c1|c2|c3) ST="--steps 2000 --warmup 100";; *) ST="";; esac
  # the TRACED run takes the `value` leg only: the spatial-ids leg of configs 4 / 5 launches the same kernel on other ids (354 us against
  # 697 at config 5) and would be averaged into the same row of the trace's statistics
  TR="--legs value"
  # the line from a PLAIN run (under the profiler's kernel trace the same command runs 3-6 % slower), the kernel stats from the traced one
  timeout -k 10 300 python3 bench.py --config $C $ST --no-cpu-baseline --no-live-traffic > $OUT/bench_${C}.json 2> $OUT/bench_$C.err || { echo "bench $C failed"; tail -5 $OUT/bench_$C.err; exit 1; }
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace_$C -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --config $C $ST $TR --no-cpu-baseline --no-live-traffic > $GRAFT_REPO_ROOT/$OUT/bench_${C}_traced.json 2> $GRAFT_REPO_ROOT/$OUT/trace_$C.log) || { echo "trace $C failed"; tail -5 $OUT/trace_$C.log; exit 1; }
  f=$(find $OUT/trace_$C -name "bench_kernel_stats.csv" | head -1)
  cp "$f" $OUT/bench_${C}_kernel_stats.csv && head -6 $OUT/bench_${C}_kernel_stats.csv
done
find $OUT -name "*.db" -delete; find $OUT -size +3M -delete
