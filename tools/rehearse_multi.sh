#!/bin/bash
# N > 1 paths on a one-GPU box: the 2-rank GPU tests (gloo), and a rehearsal of bench.py's N = 2 / 3 code path.
OUT=gpurun_out/${1:-r02w}; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_dist_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
for ARGS in "--gpus 2" "--gpus 2 --config c4 --exchange halo --ids spatial" "--gpus 3 --config c4 --no-extras"; do
  N=$(echo $ARGS | awk '{print $2}')
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29533 bench.py $ARGS --steps 3 --warmup 1 --rehearse-one-gpu --no-cpu-baseline > $OUT/bench_n.json 2> $OUT/bench_n.err
  echo "bench [$ARGS] rc=$?"; tail -2 $OUT/bench_n.err | cut -c1-300
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_n.json").read().strip().splitlines()[-1])
    print({k:d[k] for k in ("n_gpus","scaling","value","ms_per_step")}, d.get("exchange"), "gficf" in d and d["gficf"].get("value"))
except Exception as ex: print("parse failed", ex)
PY
done
