#!/bin/bash
OUT=gpurun_out/${1:-r02k}; mkdir -p $OUT
timeout -k 10 300 python bench.py --no-knn --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$OUT/bench.json")); g=d["gficf"]
print("jaccard value %.4g ms/ds %.4f"%(d["value"], d["ms_per_data_set"]))
print("gficf %.4g cells/s ms %.4f"%(g["value"], g["ms_per_pass"]), g["roofline"])
PY
timeout -k 10 600 python -m pytest tests/test_gficf_gpu.py tests/test_multi_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
