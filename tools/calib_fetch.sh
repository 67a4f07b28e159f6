#!/bin/bash
# FETCH_SIZE / TCC_EA0_RDREQ against known bytes (tools/lab/fetch_calib): counter passes -> gpurun_out/<tag>/calib_*.csv + summary
# Usage (through gpurun): bash tools/calib_fetch.sh <tag>
TAG=${1:-calib}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
(cd /tmp && rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(TCC_EA0?_RD[A-Z0-9_]*|TCC_BUBBLE[A-Z0-9_]*|FETCH_SIZE|TCC_REQ[A-Z0-9_]*|TCC_READ[A-Z0-9_]*|TCC_MISS[A-Z0-9_]*|TCC_HIT[A-Z0-9_]*|TCP_TCC_READ[A-Z0-9_]*)\b" | sort -u | tr '\n' ' ') > $OUT/avail.txt
echo "available: $(cat $OUT/avail.txt)"
P=0
for CTRS in "FETCH_SIZE TCC_EA0_RDREQ_sum" "TCC_EA0_RDREQ_32B_sum TCC_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_DRAM_sum"; do
  P=$((P+1))
  (cd /tmp && timeout 200 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/c$P -o calib -- $GRAFT_REPO_ROOT/tools/lab/fetch_calib 2048 3 > $GRAFT_REPO_ROOT/$OUT/c$P.log 2>&1) || echo "pass $P ($CTRS) failed: $(tail -1 $OUT/c$P.log | cut -c1-200)"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/c*/calib_counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
T = 2048 * (1 << 20)
for k in sorted(acc):
    known = T + (T // 64 * 4 if "gather<64" in k else T // 128 * 4 if "gather<128" in k else 0)
    print(k, "known bytes per launch (table + index) = %d" % known)
    for c, v in sorted(acc[k].items()):
        m = sum(v) / len(v)
        extra = ""
        if c == "FETCH_SIZE": extra = "  x1024 = %.4g B = %.3f of known" % (m * 1024, m * 1024 / known)
        if c.startswith("TCC_EA0_RDREQ_sum"): extra = "  x64 B = %.4g B = %.3f of known" % (m * 64, m * 64 / known)
        if c.startswith("TCC_EA0_RDREQ_32B"): extra = "  x32 B = %.4g B" % (m * 32)
        if c.startswith("TCC_EA0_RDREQ_64B"): extra = "  x64 B = %.4g B = %.3f of known" % (m * 64, m * 64 / known)
        if c.startswith("TCC_EA0_RDREQ_128B"): extra = "  x128 B = %.4g B = %.3f of known" % (m * 128, m * 128 / known)
        print("   %-28s n=%d mean=%.6g%s" % (c, len(v), m, extra))
PY
find $OUT -name "*.db" -delete
