#!/bin/bash
# Per-kernel totals of the device Louvain at the shapes of tools/louvain_time.py.  Usage: bash tools/louvain_trace.sh <tag> [N k n_start reps]
TAG=${1:-lt}; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace -o t -- python3 $GRAFT_REPO_ROOT/tools/louvain_time.py "$@" > $GRAFT_REPO_ROOT/$OUT/trace.log 2>&1)
grep -E "louvain_device|phenograph" $OUT/trace.log
python3 - <<PY
import csv, re, glob
f = glob.glob("$OUT/trace/**/t_kernel_stats.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((float(r["TotalDurationNs"]) / 1e3, r["Name"][:70], int(r["Calls"]), float(r["AverageNs"]) / 1e3))
for t, n, c, a in sorted(rows, reverse=True)[:28]: print("%-72s calls %6d total %10.1f us avg %8.1f us" % (n, c, t, a))
print("sum %.1f us over %d launches" % (sum(r[0] for r in rows), sum(r[2] for r in rows)))
PY
find $OUT -name "*.db" -delete
