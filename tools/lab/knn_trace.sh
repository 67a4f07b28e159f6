mkdir -p gpurun_out/knn22; export TMPDIR=/tmp
(cd /tmp && KNN_DATA=blobs timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/knn22/t -o k -- python3 $GRAFT_REPO_ROOT/tools/knn_bench.py 100000 50 31 manhattan > $GRAFT_REPO_ROOT/gpurun_out/knn22/log.txt 2>&1)
python - <<PY
import csv, re
rows=[]
for r in csv.DictReader(open("gpurun_out/knn22/t/k_kernel_stats.csv")):
    n=r["Name"]; m=re.search(r"(k_knn_[a-z_]+(<[^>]*>)?|onesweep\w*|radix\w*)", n)
    rows.append((float(r["TotalDurationNs"]), int(r["Calls"]), (m.group(1) if m else n[:60])))
for t,c,n in sorted(rows, reverse=True)[:16]: print("%-50s calls %4d avg %9.1f us" % (n, c, t/c/1e3))
PY
find gpurun_out/knn22 -name "*.db" -delete
