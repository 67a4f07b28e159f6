#!/bin/bash
# Per-kernel averages of t(gficf).  Usage: bash tools/transpose_prof.sh <tag> [G N]
TAG=${1:-tp}; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace -o t -- python3 $GRAFT_REPO_ROOT/tools/transpose_bench.py "$@" > $GRAFT_REPO_ROOT/$OUT/trace.log 2>&1)
grep "ms/transpose" $OUT/trace.log
python - <<PY
import csv, re
for r in csv.DictReader(open("$OUT/trace/t_kernel_stats.csv")):
    m = re.search(r"(k_(tr|scan)[a-z_0-9]*)", r["Name"])
    if m: print("%-24s calls %4s avg %9.1f us  min %8.1f max %9.1f" % (m.group(1), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
find $OUT -name "*.db" -delete
