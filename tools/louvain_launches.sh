#!/bin/bash
# The launches of one gficf_louvain_device call in order, with their durations (rocprofv3 kernel trace of the LAST call of tools/louvain_time.py N k n_start 3).
# Usage (through gpurun): bash tools/louvain_launches.sh [N k n_start]
N=${1:-54000}; K=${2:-30}; S=${3:-10}
OUT=gpurun_out/lvla; mkdir -p $OUT; export TMPDIR=/tmp
(cd /tmp && LT_NO_PHENOGRAPH=1 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace -o t -- python3 $GRAFT_REPO_ROOT/tools/louvain_time.py $N $K $S 3 > $GRAFT_REPO_ROOT/$OUT/trace.log 2>&1)
grep -E "louvain_device" $OUT/trace.log
python3 - <<PY
import csv, glob, re
f = glob.glob("$OUT/trace/**/t_kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
starts = [i for i, r in enumerate(rows) if "k_lv_fix" in r[2]]
call = rows[starts[-1]:]
def short(n):
    m = re.search(r"(k_[a-z_0-9]+)(<[^>]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:24]
line = []
tot = {}
for s, e, n in call:
    k = short(n); d = (e - s) / 1e3
    tot.setdefault(k, []).append(d)
    if "move" in k: line.append("%s %.0f" % (k.replace("k_lv_move_", ""), d))
print("move launches in order (us):", ", ".join(line))
for k, v in sorted(tot.items(), key=lambda kv: -sum(kv[1])):
    print("%-28s %4d launches  total %8.1f us  median %6.1f  min %6.1f  max %7.1f" % (k, len(v), sum(v), sorted(v)[len(v) // 2], min(v), max(v)))
PY
rm -rf $OUT/trace
