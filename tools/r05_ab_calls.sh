#!/bin/bash
# The A/B and verification calls of round 5 other than the two record calls (tools/gpu_round.sh), kept as the evidence chain of the
# profiles/r05_*.txt files: each function is what ONE `gpurun -- bash tools/r05_ab_calls.sh <n> <tag>` call ran.
N=${1:?call number 1..7}; TAG=${2:-r05_ab$N}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp

call1() {
  # round 5, GPU call 1: the k > 256 path, the whole GPU suite, baseline lines of configs 1 / 2 before the one-call step
  timeout -k 10 600 python -m pytest tests/test_jaccard_gpu.py -x -q -m gpu -k "beyond_256 or thousands or truncat or option" > $OUT/pytest_bigk.log 2>&1; echo "bigk rc=$?"; tail -5 $OUT/pytest_bigk.log
  timeout -k 10 900 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
  for C in c1 c2; do
    timeout -k 10 200 python bench.py --config $C --no-gficf --no-knn --no-live-traffic > $OUT/bench_$C.json 2> $OUT/bench_$C.err; echo "bench $C rc=$?"; cut -c1-300 $OUT/bench_$C.json
  done
}

call2() {
  # round 5, GPU call 2: the one-launch form (tests, in-process A/B), k > 256 tests again, configs 1 / 2 lines
  timeout -k 10 600 python -m pytest tests/test_jaccard_gpu.py -x -q -m gpu -k "one_launch or beyond_256 or thousands or unsupported or pipelined" > $OUT/pytest_new.log 2>&1; echo "new tests rc=$?"; tail -5 $OUT/pytest_new.log
  timeout -k 10 300 python tools/direct_ab.py > $OUT/direct_ab.txt 2>&1; echo "direct ab rc=$?"; cat $OUT/direct_ab.txt
  for C in c1 c2 c3; do
    timeout -k 10 200 python bench.py --config $C --no-gficf --no-knn --no-live-traffic > $OUT/bench_$C.json 2> $OUT/bench_$C.err; echo "bench $C rc=$?"; cut -c1-300 $OUT/bench_$C.json
  done
  timeout -k 10 900 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
}

call3() {
  # round 5, GPU call 3: the restructured bench.py (default line, legs, budget), configs 1-3 with enough steps + traces, N > 1 rehearsals
  # (2 and 5 ranks on the one GPU: 5 ranks + the peer child = the box's limit of 6 processes on the card), bits-kernel lab variants
  timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-300 $OUT/bench.json; tail -3 $OUT/bench.err
  CONFIGS_TO_TRACE="c1 c2 c3" bash tools/trace_configs.sh $TAG > $OUT/trace_configs.txt 2>&1; echo "trace configs rc=$?"; tail -20 $OUT/trace_configs.txt | cut -c1-200
  for A in 2 5; do
    /usr/bin/time -f "rehearsal $A ranks: %e s wall" timeout -k 10 420 python bench.py --gpus $A --rehearse-one-gpu --steps 5 --no-gficf > $OUT/rehearsal_gpus$A.jsonl 2> $OUT/rehearsal_gpus$A.err; echo "rehearsal $A rc=$?"
    tail -2 $OUT/rehearsal_gpus$A.err; wc -l $OUT/rehearsal_gpus$A.jsonl; tail -1 $OUT/rehearsal_gpus$A.jsonl | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d.get(k) for k in ('legs_done','leg_seconds','skipped_legs','wall_s','checked_vs_oracle')})"
  done
  bash tools/bits_ab.sh $TAG > $OUT/bits_ab.txt 2>&1; cat $OUT/bits_ab.txt
  timeout -k 10 600 python -m pytest tests/test_dist_gpu.py tests/test_multi_gpu.py tests/test_glue_run.py -q -m gpu -x > $OUT/pytest_part.log 2>&1; echo "pytest part rc=$?"; tail -3 $OUT/pytest_part.log
  find $OUT -name "*.db" -delete; find $OUT -size +3M -delete
}

call4() {
  # round 5, GPU call 4: N > 1 rehearsals of the restructured bench (2 and 5 ranks on the one GPU), the whole GPU suite, the
  # phenograph order A/B, the one-buffer A/B, configs 4 / 5 lines with the spatial-ids figure
  for A in 2 5; do
    S=$SECONDS
    timeout -k 10 420 python bench.py --gpus $A --rehearse-one-gpu --steps 5 --no-gficf > $OUT/rehearsal_gpus$A.jsonl 2> $OUT/rehearsal_gpus$A.err; echo "rehearsal $A rc=$? in $((SECONDS-S)) s"
    tail -2 $OUT/rehearsal_gpus$A.err | cut -c1-300; wc -l $OUT/rehearsal_gpus$A.jsonl; tail -1 $OUT/rehearsal_gpus$A.jsonl | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d.get(k) for k in ('legs_done','leg_seconds','skipped_legs','wall_s','checked_vs_oracle')})"
  done
  timeout -k 10 120 python tools/one_buffer_ab.py > $OUT/one_buffer_ab.txt 2>&1; echo "one buffer rc=$?"; cat $OUT/one_buffer_ab.txt
  timeout -k 10 300 python tools/phenograph_order_ab.py 400000 10 30 > $OUT/phenograph_order_ab.txt 2>&1; echo "phenograph ab rc=$?"; cat $OUT/phenograph_order_ab.txt
  CONFIGS_TO_TRACE="c4 c5" bash tools/trace_configs.sh $TAG > $OUT/trace_configs.txt 2>&1; echo "trace configs rc=$?"; tail -12 $OUT/trace_configs.txt | cut -c1-220
  timeout -k 10 1000 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.log
  find $OUT -name "*.db" -delete; find $OUT -size +3M -delete
}

call5() {
  # round 5, GPU call 5: a third plain `python bench.py` line of the round, configs 4 / 5 line + trace again (the traced run with the
  # `value` leg only), the GF-ICF pass over the five shapes with one prepared call per pass, kernel trace of the GF-ICF pass at configs 1 / 2
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
}

call6() {
  # round 5, GPU call 6: where does the sorted-row path beat the general kernel (56 < k <= 256)?  + the new sorted-path test
  timeout -k 10 900 python -u tools/sorted_vs_general.py > $OUT/sorted_vs_general.txt 2> $OUT/sorted_vs_general.err; echo "ab rc=$?"; cat $OUT/sorted_vs_general.txt
  timeout -k 10 200 python -m pytest tests/test_jaccard_gpu.py -q -m gpu -x -k "local_ids_with_empty" > $OUT/pytest_new.log 2>&1; echo "new test rc=$?"; tail -3 $OUT/pytest_new.log
}

call7() {
  # round 5, GPU call 7: the compact return inside gficf_jaccard_host, A/B + the host-entry tests (Python and the `.Call` glue)
  timeout -k 10 600 python -u tools/host_compact_ab.py > $OUT/host_compact_ab.txt 2> $OUT/host_compact_ab.err; echo "ab rc=$?"; cat $OUT/host_compact_ab.txt; tail -3 $OUT/host_compact_ab.err
  timeout -k 10 900 python -m pytest tests/test_jaccard_gpu.py tests/test_glue_run.py tests/test_multi_gpu.py -q -m gpu -x > $OUT/pytest_host.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_host.log
}

call$N
find $OUT -name "*.db" -delete; find $OUT -size +3M -delete
