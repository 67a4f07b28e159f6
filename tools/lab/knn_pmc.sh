mkdir -p gpurun_out/knn15; export TMPDIR=/tmp
for K in 31 1; do
(cd /tmp && PROF_K=$K PROF_REPS=1 timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/knn15/k$K -o pmc -- python3 $GRAFT_REPO_ROOT/tools/prof_knn.py > $GRAFT_REPO_ROOT/gpurun_out/knn15/k$K.log 2>&1)
python - <<PY
import csv,collections
acc=collections.defaultdict(list)
for row in csv.DictReader(open("gpurun_out/knn15/k$K/pmc_counter_collection.csv")):
    if "knn_tiles" in row["Kernel_Name"]: acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("k=$K", {k: "%.4g"%(sum(v)/len(v)) for k,v in acc.items()})
PY
done
find gpurun_out/knn15 -name "*.db" -delete
