#!/bin/bash
# (needs tools/lab/jaccard_cell_order.patch applied: the ordered walk is not in the product, profiles/r03_cell_order.txt)
# kernel trace of the ordered walk at 1 M x 30 (scenario B: 2 label rounds, D: 1), per-kernel medians
OUT=gpurun_out/${1:-order_trace}; mkdir -p $OUT; export TMPDIR=/tmp
for S in ${2:-B D}; do
  export SCEN=$S
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/$S -o t -- python3 $GRAFT_REPO_ROOT/tools/lab/order_pmc_driver.py > $GRAFT_REPO_ROOT/$OUT/$S.log 2>&1) || echo "scen $S failed: $(tail -2 $OUT/$S.log)"
done
python - <<PY
import csv, glob, collections, re
for S in "$2".split() or ["B", "D"]:
    dur = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/t_kernel_trace.csv" % S):
        for row in csv.DictReader(open(f)):
            n = re.sub(r"\(.*", "", re.sub(r"^void |\(anonymous namespace\)::", "", row["Kernel_Name"]))[:70]
            dur[n].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    print("scenario", S)
    for kname in sorted(dur):
        d = sorted(dur[kname]); print("  %-72s median %8.1f us n=%d" % (kname, d[len(d)//2], len(d)))
PY
