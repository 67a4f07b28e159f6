#!/bin/bash
# Per-kernel totals of the device Louvain.  Usage: bash tools/louvain_prof.sh <tag> N k [data] [resolution] [n_iter]
TAG=${1:-lp}; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
export LAB_REF=0
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace -o t -- python3 $GRAFT_REPO_ROOT/tools/louvain_lab.py "$@" > $GRAFT_REPO_ROOT/$OUT/trace.log 2>&1)
grep -E "graph:|device:" $OUT/trace.log
python - <<PY
import csv, re
rows = []
for r in csv.DictReader(open("$OUT/trace/t_kernel_stats.csv")):
    m = re.search(r"(k_lv_[a-z_0-9]*|k_scan[a-z_0-9]*|radix[a-z_0-9_]*|onesweep[a-z_0-9_]*)", r["Name"])
    if m: rows.append((float(r["TotalDurationNs"]) / 1e3, m.group(1), int(r["Calls"]), float(r["AverageNs"]) / 1e3))
for t, n, c, a in sorted(rows, reverse=True)[:16]: print("%-28s calls %5d total %9.1f us avg %8.1f us" % (n, c, t, a))
print("sum %.1f us" % sum(r[0] for r in rows))
PY
find $OUT -name "*.db" -delete
