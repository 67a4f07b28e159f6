#!/bin/bash
# One GPU-box round: tests, bench, rocprofv3 kernel trace of the bench command, PMC passes.
# Usage (through gpurun): bash tools/gpu_round.sh <tag> [tests]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ "$2" == "tests" ]; then
  timeout 1200 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
fi
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/trace.log 2>&1); echo "trace rc=$?"
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -r head -20
P=0
for CTRS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
            "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
            "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum" ; do
  P=$((P+1))
  (cd /tmp && PROF_REPS=3 timeout 300 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc$P -o pmc -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py > $GRAFT_REPO_ROOT/$OUT/pmc$P.log 2>&1); echo "pmc$P rc=$?"
done
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1; cat $OUT/pmc_summary.txt
# keep only the small csv files
find $OUT -name "*.db" -delete; find $OUT -size +4M -delete
