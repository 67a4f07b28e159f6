#!/bin/bash
# (needs tools/lab/jaccard_cell_order.patch applied: the ordered walk is not in the product, profiles/r03_cell_order.txt)
OUT=gpurun_out/${1:-order_pmc}; mkdir -p $OUT; export TMPDIR=/tmp
for S in A B C; do
  export SCEN=$S
  P=0
  for CTRS in "" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_VALU"; do
    P=$((P+1))
    if [ -z "$CTRS" ]; then PM=""; else PM="--pmc $CTRS"; fi
    (cd /tmp && timeout 300 rocprofv3 $PM --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/$S.p$P -o pmc -- python3 $GRAFT_REPO_ROOT/tools/lab/order_pmc_driver.py > $GRAFT_REPO_ROOT/$OUT/$S.p$P.log 2>&1) || echo "scen $S pass $P failed: $(tail -2 $OUT/$S.p$P.log)"
  done
done
python - <<PY
import csv, glob, collections, re
for S in "ABC":
    acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s.p*/pmc_counter_collection.csv" % S):
        for row in csv.DictReader(open(f)):
            m = re.search(r"(?<![A-Za-z0-9_])(k_[a-z_0-9]+)", row["Kernel_Name"])
            if m: acc[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for f in glob.glob("$OUT/%s.p1/pmc_kernel_trace.csv" % S):
        for row in csv.DictReader(open(f)):
            m = re.search(r"(?<![A-Za-z0-9_])(k_[a-z_0-9]+)", row["Kernel_Name"])
            if m: dur[m.group(1)].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    print("scenario", S, {"A": "rows scattered, plain walk", "B": "rows scattered, label walk (2 hops)", "C": "rows placed by locality, plain walk"}[S])
    for kname in sorted(dur):
        d = sorted(dur[kname]); print("  %-28s median %.1f us n=%d  " % (kname, d[len(d)//2], len(d)) + "  ".join("%s=%.4g" % (c, sum(v)/len(v)) for c, v in sorted(acc[kname].items())))
PY
find $OUT -name "*.db" -delete
