#!/bin/bash
# Quick GPU check: Jaccard parity tests + a short bench.  Usage: bash tools/quick.sh <tag> [pytest -k expr]
TAG=${1:-q}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests -x -q -m gpu ${2:+-k "$2"} > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print("value %.4g edges/s  ms/step %.4f"%(d["value"], d["ms_per_step"]), d["roofline"], d.get("checked_vs_oracle"))
g=d.get("gficf")
if g: print("gficf %.4g cells/s ms %.4f"%(g["value"], g["ms_per_pass"]), g["roofline"])
PY
