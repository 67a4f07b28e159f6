#!/bin/bash
# One GPU-box round: GPU tests, smoke, bench, rocprofv3 kernel trace of the bench command, PMC passes, config 4 / 5 lines,
# N > 1 rehearsals on the one GPU (plain `python bench.py --gpus N`: the script starts its own ranks).
# Usage (through gpurun): bash tools/gpu_round.sh <tag>
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json | cut -c1-600
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace -o bench -- python3 $GRAFT_REPO_ROOT/bench.py > $GRAFT_REPO_ROOT/$OUT/trace.log 2>&1); echo "trace rc=$?"
head -14 $OUT/trace/bench_kernel_stats.csv
for C in c4 c5; do timeout -k 10 200 python bench.py --config $C --no-cpu-baseline > $OUT/bench_$C.json 2>> $OUT/bench.err; echo "bench $C rc=$?"; done
for A in "--gpus 2 --ids spatial" "--gpus 3 --ids spatial" "--gpus 2 --ids permuted"; do
  timeout -k 10 300 python bench.py $A --rehearse-one-gpu --no-extras --steps 5 > "$OUT/rehearsal_$(echo $A | tr -d ' -').json" 2>> $OUT/bench.err; echo "rehearsal $A rc=$?"
done
bash tools/pmc_round.sh $TAG 2>&1 | tail -14
find $OUT -name "*.db" -delete; find $OUT -size +3M -delete
