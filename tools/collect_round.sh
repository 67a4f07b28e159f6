#!/bin/bash
# Copies what the round's gpurun calls left under gpurun_out/<tag>/ into profiles/ under the names tools/make_results.py reads
# (profiles/<tag>_<file>; kept runs to profiles/<tag>_runs/).  Traces' raw directories, logs of the profiler and .err files stay behind.
TAG=${1:-r06}; SRC=gpurun_out/$TAG
mkdir -p profiles/${TAG}_runs
for f in $SRC/*.json $SRC/*.jsonl $SRC/*.csv $SRC/*.txt $SRC/*.md $SRC/pytest_gpu.log $SRC/smoke.log; do
  [ -f "$f" ] || continue
  b=$(basename $f)
  case $b in trace_configs.txt|*_trace.log) continue;; esac
  cp $f profiles/${TAG}_$b
done
for f in $SRC/runs/*.json $SRC/runs/*_kernel_stats.csv; do [ -f "$f" ] && cp $f profiles/${TAG}_runs/; done
ls profiles | grep -c "^${TAG}_"
