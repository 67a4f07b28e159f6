#!/bin/bash
# rocprofv3 counter passes over one batched Louvain call (tools/louvain_time.py N k n_start 1); per-kernel means over the level-0 launches
# (the persistent grid of 2048 workgroups) of the kernels matching $2.  Usage: bash tools/louvain_pmc.sh <tag> <kernel-regex> N k n_start
TAG=${1:-lvpmc}; PAT=${2:-k_lv_move_small}; shift; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
P=0
for CTRS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
            "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
            "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES_EQ_64 SQ_THREAD_CYCLES_VALU SQ_INSTS_FLAT" \
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" \
            "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
            "TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TA_BUSY_avr TA_TA_BUSY_sum" ; do
  P=$((P+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/p$P -o pmc -- python3 $GRAFT_REPO_ROOT/tools/louvain_time.py "$@" 1 > $GRAFT_REPO_ROOT/$OUT/p$P.log 2>&1) || echo "pass $P failed: $(tail -2 $OUT/p$P.log)"
done
python3 - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/pmc_counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"(k_lv_[a-z_0-9]+(<[a-z0-9, ]+>)?)", row["Kernel_Name"])
        if m and re.search("$PAT", m.group(1)) and int(row["Grid_Size"]) >= 1024 * 128:
            acc[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(k, "launches", len(next(iter(acc[k].values()))))
    print("  " + "  ".join("%s=%.4g" % (c, sum(v)/len(v)) for c, v in sorted(acc[k].items())))
PY
find $OUT -name "*.db" -delete
