#!/bin/bash
# rocprofv3 kernel trace of tools/prof_driver.py; prints per-kernel duration stats (us).
TAG=${1:-tr}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/t -o tr -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py > $GRAFT_REPO_ROOT/$OUT/t.log 2>&1) || tail -3 $OUT/t.log
python - <<PY
import csv, collections, re
d = collections.defaultdict(list)
rows = list(csv.DictReader(open("$OUT/t/tr_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev_end = None
gaps = []
for r in rows:
    m = re.search(r"(k_[a-z_0-9]+)", r["Kernel_Name"])
    name = m.group(1) if m else r["Kernel_Name"][:40]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    d[name].append((e - s) / 1e3)
    if prev_end is not None and m: gaps.append((name, (s - prev_end) / 1e3))
    prev_end = e
for k, v in d.items():
    if k.startswith("k_"):
        v = sorted(v); print("%-22s n=%3d min=%.2f med=%.2f max=%.2f us" % (k, len(v), v[0], v[len(v)//2], v[-1]))
g = collections.defaultdict(list)
for n, x in gaps: g[n].append(x)
for k, v in g.items():
    v = sorted(v); print("gap before %-18s med=%.2f us" % (k, v[len(v)//2]))
PY
find $OUT -name "*.db" -delete
