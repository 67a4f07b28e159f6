#!/bin/bash
# round 5, GPU call 5: a third plain `python bench.py` line of the round, configs 4 / 5 line + trace again (the traced run with the
# `value` leg only), the GF-ICF pass over the five shapes with one prepared call per pass, kernel trace of the GF-ICF pass at configs 1 / 2
TAG=${1:-r05e}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-200 $OUT/bench.json
CONFIGS_TO_TRACE="c4 c5" bash tools/trace_configs.sh $TAG > $OUT/trace_configs.txt 2>&1; echo "trace configs rc=$?"; grep -c . $OUT/trace_configs.txt
timeout -k 10 500 python tools/sweep_gficf.py $OUT/gficf_shapes.txt > $OUT/sweep_gficf.log 2>&1; echo "sweep rc=$?"; cut -c1-250 $OUT/gficf_shapes.txt
for C in 1 2; do
  export GFICF_SWEEP_CONFIGS=$C
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace_gficf_c$C -o sweep -- python3 $GRAFT_REPO_ROOT/tools/sweep_gficf.py > $GRAFT_REPO_ROOT/$OUT/sweep_traced_c$C.log 2>&1); echo "gficf trace c$C rc=$?"
  cp $(find $OUT/trace_gficf_c$C -name "sweep_kernel_stats.csv" | head -1) $OUT/gficf_c${C}_kernel_stats.csv && head -8 $OUT/gficf_c${C}_kernel_stats.csv | cut -c1-160
done
unset GFICF_SWEEP_CONFIGS
find $OUT -name "*.db" -delete; find $OUT -size +3M -delete
timeout -k 10 400 python -u tools/bigk_time.py > $OUT/bigk_time.txt 2> $OUT/bigk_time.err; echo "bigk rc=$?"; cat $OUT/bigk_time.txt | cut -c1-220; tail -3 $OUT/bigk_time.err
timeout -k 10 500 python -m pytest tests/test_dist_gpu.py -q -m gpu -x -k "budget or killed or plain" > $OUT/pytest_bench_tests.log 2>&1; echo "bench tests rc=$?"; tail -3 $OUT/pytest_bench_tests.log
