#!/bin/bash
# One GPU-box round in two calls (a gpurun call is limited to 20 minutes).  Usage (through gpurun): bash tools/gpu_round.sh <tag> <part>
#   part 1: GPU tests, smoke, bench (+ rocprofv3 kernel trace of the same command), config 4 / 5 lines with their kernel traces,
#           stage times of the sharded step, the config-4 kernel A/B
#   part 2: N > 1 rehearsals on the one GPU (plain `python bench.py --gpus N`: the script starts its own ranks), PMC passes, fuzz
TAG=${1:-r04}; PART=${2:-1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ "$PART" = "1" ]; then
timeout -k 10 500 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-400 $OUT/bench.json
(cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-live-traffic > $GRAFT_REPO_ROOT/$OUT/trace.log 2>&1); echo "trace rc=$?"
cp $(find $OUT/trace -name "bench_kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv && head -8 $OUT/bench_kernel_stats.csv | cut -c1-200
bash tools/trace_configs.sh $TAG > $OUT/trace_configs.txt 2>&1; echo "trace configs rc=$?"
timeout -k 10 200 python tools/halo_stage_times.py 1 2 4 8 > $OUT/halo_stage_times.jsonl 2>> $OUT/bench.err; echo "stage times rc=$?"
python tools/project_scaling.py $OUT/halo_stage_times.jsonl > $OUT/scaling_projection.md; tail -20 $OUT/scaling_projection.md
bash tools/bits_ab.sh $TAG > $OUT/bits_ab_final.txt 2>&1; cat $OUT/bits_ab_final.txt
else
for A in "--gpus 2" "--gpus 3"; do
  timeout -k 10 420 python bench.py $A --rehearse-one-gpu --steps 5 --no-gficf > "$OUT/rehearsal_$(echo $A | tr -d ' -').json" 2>> $OUT/rehearsal.err; echo "rehearsal $A rc=$?"
done
bash tools/pmc_round.sh $TAG 2>&1 | tail -14
timeout -k 10 200 python tools/fuzz_gpu.py 120 > $OUT/fuzz.txt 2>&1; echo "fuzz rc=$?"; tail -3 $OUT/fuzz.txt
fi
find $OUT -name "*.db" -delete; find $OUT -size +3M -delete
