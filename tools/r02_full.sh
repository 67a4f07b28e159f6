#!/bin/bash
# Round-2 full GPU check: all GPU tests, smoke, bench, product kernel timings.
TAG=${1:-r02e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 120 ./tools/lab/gather_lab 100000 30 1 2>&1 | grep -E "ids|PRODUCT" > $OUT/product.txt; cat $OUT/product.txt
timeout -k 10 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err; python - <<PY
import json
try:
    d=json.load(open("$OUT/bench.json"))
    print("value %.4g edges/s  ms/step %.4f region_ms %.3f ms/ds %.4f"%(d["value"], d["ms_per_step"], d["timed_region_ms"], d["ms_per_data_set"]), d["roofline"], d.get("checked_vs_oracle"))
    for k in ("host_abi","host_abi_counts","stress_uniform_ids"): print(k, d.get(k))
    g=d.get("gficf")
    if g: print("gficf %.4g cells/s ms %.4f"%(g["value"], g["ms_per_pass"]), g["roofline"], g.get("checked_vs_oracle"))
except Exception as ex: print("bench parse failed", ex)
PY
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest_gpu.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
