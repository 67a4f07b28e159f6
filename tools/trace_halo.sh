#!/bin/bash
# Kernel durations (rocprofv3 kernel trace, no counters) of the halo form's stages for one rank of a P-rank job.
# Usage (through gpurun): bash tools/trace_halo.sh <tag> [P]
TAG=${1:-r04}; P=${2:-8}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace_halo -o halo -- python3 $GRAFT_REPO_ROOT/tools/halo_stage_times.py $P > $GRAFT_REPO_ROOT/$OUT/halo_traced.jsonl 2> $GRAFT_REPO_ROOT/$OUT/trace_halo.log) || { tail -5 $OUT/trace_halo.log; exit 1; }
f=$(find $OUT/trace_halo -name "halo_kernel_stats.csv" | head -1)
cp "$f" $OUT/halo_kernel_stats_P$P.csv
python3 - <<PY
import csv
for r in csv.DictReader(open("$OUT/halo_kernel_stats_P$P.csv")):
    n = r["Name"]
    if "k_halo" in n or "k_ingest" in n or "k_jaccard" in n or "k_pack" in n or "k_unpack" in n:
        print("%-60s calls %5s avg %8.2f us min %8.2f" % (n.split("(")[0][-60:], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
find $OUT -name "*.db" -delete; find $OUT -size +3M -delete
