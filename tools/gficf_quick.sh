#!/bin/bash
# GF-ICF only: parity tests + pass time + per-kernel averages.  Usage: bash tools/gficf_quick.sh <tag>
TAG=${1:-gq}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_gficf_gpu.py -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-knn > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
g=json.load(open("$OUT/bench.json"))["gficf"]; print("gficf %.4g cells/s  ms/pass %.4f  frac %.3f" % (g["value"], g["ms_per_pass"], g["roofline"]["frac"]))
PY
(cd /tmp && PROF_JACCARD=0 PROF_REPS=5 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace -o t -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py > $GRAFT_REPO_ROOT/$OUT/trace.log 2>&1)
python - <<PY
import csv, re
for r in csv.DictReader(open("$OUT/trace/t_kernel_stats.csv")):
    m = re.search(r"(k_[a-z_0-9]+(<[^>]*>)?)", r["Name"])
    if m: print("%-40s calls %4s avg %9.1f us  min %8.1f max %9.1f" % (m.group(1), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
find $OUT -name "*.db" -delete
