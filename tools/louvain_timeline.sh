#!/bin/bash
# Where one gficf_louvain_device call spends its wall time: from the rocprofv3 kernel trace (start / end stamps), the busy time and the gaps between
# the kernels of the LAST call of tools/louvain_time.py N k n_start 3 (phenograph lines off).  Usage (through gpurun): bash tools/louvain_timeline.sh [N k n_start]
N=${1:-54000}; K=${2:-30}; S=${3:-10}
OUT=gpurun_out/lvtl; mkdir -p $OUT; export TMPDIR=/tmp
(cd /tmp && LT_NO_PHENOGRAPH=1 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace -o t -- python3 $GRAFT_REPO_ROOT/tools/louvain_time.py $N $K $S 3 > $GRAFT_REPO_ROOT/$OUT/trace.log 2>&1)
grep -E "louvain_device" $OUT/trace.log
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/t_kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# the calls: split at k_lv_fix (first kernel of a call)
starts = [i for i, r in enumerate(rows) if "k_lv_fix" in r[2]]
i0 = starts[-1]
call = [r for r in rows[i0:] if "k_lv" in r[2] or "scan" in r[2] or "rocprim" in r[2] or "fill" in r[2].lower() or "copy" in r[2].lower()]
t0, t1 = call[0][0], max(r[1] for r in call)
busy = 0; cur_end = t0; gaps = []
for s, e, n in call:
    if s > cur_end: gaps.append((s - cur_end, n)); 
    if e > cur_end: busy += e - max(s, cur_end); cur_end = e
print("last call: %d launches, span %.2f ms, device busy %.2f ms, idle in %d gaps %.2f ms" % (len(call), (t1 - t0) / 1e6, busy / 1e6, len(gaps), sum(g for g, _ in gaps) / 1e6))
big = sorted(gaps, reverse=True)[:12]
print("largest gaps (us, before kernel):", [(round(g / 1e3, 1), n.split("(")[0][-28:]) for g, n in big])
import collections
by = collections.Counter()
for g, n in gaps: by[n.split("(")[0].split("::")[-1][:30]] += g
print("idle by the kernel that follows (us):", [(k, round(v / 1e3)) for k, v in by.most_common(8)])
PY
rm -rf $OUT/trace
