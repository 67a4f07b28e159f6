#!/bin/bash
# rocprofv3 counter passes over tools/prof_driver.py; prints per-kernel means for kernels matching $2.
# Usage: bash tools/pmc.sh <tag> <kernel-regex> [PROF_* env assignments...]
TAG=${1:-pmc}; PAT=${2:-k_jaccard_edges}; shift; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
P=0
for CTRS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
            "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
            "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES_EQ_64 SQ_THREAD_CYCLES_VALU SQ_INSTS_FLAT" \
            "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum TCC_EA0_WRREQ_64B_sum" \
            "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_DRAM_sum" \
            "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" ; do
  P=$((P+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/p$P -o pmc -- python3 $GRAFT_REPO_ROOT/tools/${PROF_DRIVER:-prof_driver.py} > $GRAFT_REPO_ROOT/$OUT/p$P.log 2>&1) || echo "pass $P failed: $(tail -2 $OUT/p$P.log)"
done
python - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/pmc_counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        m = re.search(r"(?<![A-Za-z0-9_])(k_[a-z_0-9]+)", row["Kernel_Name"])
        if m and re.search("$PAT", m.group(1)):
            acc[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
for f in glob.glob("$OUT/p*/pmc_kernel_trace.csv"):
    for row in csv.DictReader(open(f)):
        m = re.search(r"(?<![A-Za-z0-9_])(k_[a-z_0-9]+)", row["Kernel_Name"])
        if m and re.search("$PAT", m.group(1)):
            dur[m.group(1)].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
for k in acc:
    d = sorted(dur[k])
    print(k, "median_us=%.1f n=%d" % (d[len(d)//2], len(d)))
    print("  " + "  ".join("%s=%.4g" % (c, sum(v)/len(v)) for c, v in sorted(acc[k].items())))
PY
find $OUT -name "*.db" -delete
