#!/bin/bash
# per-kernel averages of the GF-ICF pass for the product library and for lab variants (tools/lab/abl<V>/libgficf_hip.so)
OUT=gpurun_out/${1:-r02lab4}; shift; mkdir -p $OUT; export TMPDIR=/tmp
for V in "" "$@"; do
  echo "== variant '$V'"
  if [ -n "$V" ]; then export GFICF_HIP_LIB=$GRAFT_REPO_ROOT/tools/lab/abl$V/libgficf_hip.so; else unset GFICF_HIP_LIB; fi
  (cd /tmp && PROF_JACCARD=0 PROF_REPS=8 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace$V -o t -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py > $GRAFT_REPO_ROOT/$OUT/trace$V.log 2>&1) || { echo "rocprof failed"; tail -5 $OUT/trace$V.log; exit 1; }
  python - <<PY
import csv, re
for r in csv.DictReader(open("$OUT/trace$V/t_kernel_stats.csv")):
    m = re.search(r"(?<![A-Za-z0-9_])(k_[a-z_0-9]+(<[^>]*>)?)", r["Name"])
    if m and float(r["AverageNs"]) > 1500: print("  %-40s calls %4s avg %9.1f us  min %8.1f max %9.1f" % (m.group(1), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
find $OUT -name "*.db" -delete
