#!/bin/bash
# One GPU-box round in two calls (a gpurun call is limited to 20 minutes).  Usage (through gpurun): bash tools/gpu_round.sh <tag> <part>
#   part 1: GPU tests, smoke, the bench line AND the rocprofv3 kernel trace of the same command in this same call (tools/make_results.py
#           checks one against the other), the lines + traces of configs 1-5, stage times of the sharded step + the projection
#   part 2: N > 1 rehearsals on the one GPU (plain `python bench.py --gpus N`: the script starts its own ranks; 3 ranks at full size,
#           5 ranks at 5 k cells per rank — 5 ranks + the peer child are the 6 processes a box allows on its card), PMC passes, fuzz,
#           kernel trace of the k > 256 path
#   part 3: the "next" rows' own records: Louvain + fused call times, the fused call's stages, adjacency build (windowed and real kNN graph),
#           the search's two forms on the config-3 stand-in, three more fuzz seeds
TAG=${1:-r05}; PART=${2:-1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ "$PART" = "1" ]; then
timeout -k 10 500 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-400 $OUT/bench.json
(cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-live-traffic > $GRAFT_REPO_ROOT/$OUT/bench_traced.json 2> $GRAFT_REPO_ROOT/$OUT/trace.log); echo "trace rc=$?"
cp $(find $OUT/trace -name "bench_kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv && head -8 $OUT/bench_kernel_stats.csv | cut -c1-200
CONFIGS_TO_TRACE="c1 c2 c3 c4 c5" bash tools/trace_configs.sh $TAG > $OUT/trace_configs.txt 2>&1; echo "trace configs rc=$?"
timeout -k 10 200 python tools/halo_stage_times.py 1 2 4 8 > $OUT/halo_stage_times.jsonl 2>> $OUT/bench.err; echo "stage times rc=$?"
python tools/project_scaling.py $OUT/halo_stage_times.jsonl > $OUT/scaling_projection.md; tail -20 $OUT/scaling_projection.md
elif [ "$PART" = "3" ]; then
timeout -k 10 300 python tools/louvain_time.py 2>&1 | grep -v amdgpu.ids > $OUT/louvain_time.txt; echo "louvain_time rc=$?"; tail -6 $OUT/louvain_time.txt | cut -c1-200
(timeout -k 10 200 python tools/phenograph_stages.py && timeout -k 10 200 python tools/phenograph_stages.py 100000 50 1 && timeout -k 10 200 python tools/phenograph_stages.py 400000 30 10) 2>&1 | grep -v amdgpu.ids > $OUT/phenograph_stages.txt; echo "stages rc=$?"
(timeout -k 10 200 python tools/adjacency_time.py 54000 30 100000 50 1000000 30 && ADJ_GRAPH=blobs timeout -k 10 300 python tools/adjacency_time.py 54000 30 100000 50 100000 100) 2>&1 | grep -v amdgpu.ids > $OUT/adjacency.txt; echo "adjacency rc=$?"; cat $OUT/adjacency.txt | cut -c1-200
timeout -k 10 200 python tools/knn_prune_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/knn_prune_probe.txt; echo "probe rc=$?"; cat $OUT/knn_prune_probe.txt
for SEED in 11 12 13; do timeout -k 10 150 python tools/fuzz_gpu.py 100 $SEED > $OUT/fuzz_seed$SEED.txt 2>&1; echo "fuzz seed $SEED rc=$?"; tail -2 $OUT/fuzz_seed$SEED.txt | cut -c1-300; done
else
S=$SECONDS
timeout -k 10 420 python bench.py --gpus 3 --rehearse-one-gpu --steps 5 --no-gficf > $OUT/rehearsal_gpus3.jsonl 2> $OUT/rehearsal_gpus3.err; echo "rehearsal 3 ranks (full size) rc=$? in $((SECONDS-S)) s"
S=$SECONDS
# (five processes time-slicing ONE GPU over gloo: a data set takes ~1.5 s there, so this one runs 5 k cells per rank and 2 steps — and, with a
# budget of 200 s, shows the budget doing its work: the legs it has no time for are skipped and named)
timeout -k 10 420 python bench.py --gpus 5 --rehearse-one-gpu --steps 2 --warmup 1 --no-gficf --cells-per-gpu 5000 --budget-s 200 > $OUT/rehearsal_gpus5.jsonl 2> $OUT/rehearsal_gpus5.err; echo "rehearsal 5 ranks (5 k cells per rank, budget 200 s) rc=$? in $((SECONDS-S)) s"
for A in 3 5; do wc -l $OUT/rehearsal_gpus$A.jsonl; tail -1 $OUT/rehearsal_gpus$A.jsonl | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d.get(k) for k in ('legs_done','leg_seconds','skipped_legs','wall_s','checked_vs_oracle')})"; done
bash tools/pmc_round.sh $TAG 2>&1 | tail -14
timeout -k 10 200 python tools/fuzz_gpu.py 120 > $OUT/fuzz.txt 2>&1; echo "fuzz rc=$?"; tail -3 $OUT/fuzz.txt
timeout -k 10 300 python tools/config3_pipeline.py 2>&1 | grep -v amdgpu.ids > $OUT/config3_pipeline.txt; echo "config 3 pipeline rc=$?"; tail -3 $OUT/config3_pipeline.txt | cut -c1-300
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace_bigk -o bigk -- python3 $GRAFT_REPO_ROOT/tools/bigk_time.py > $GRAFT_REPO_ROOT/$OUT/bigk_time_traced.txt 2> $GRAFT_REPO_ROOT/$OUT/trace_bigk.log); echo "bigk trace rc=$?"
cp $(find $OUT/trace_bigk -name "bigk_kernel_stats.csv" | head -1) $OUT/bigk_kernel_stats.csv && head -6 $OUT/bigk_kernel_stats.csv | cut -c1-180
fi
find $OUT -name "*.db" -delete; find $OUT -size +3M -delete
