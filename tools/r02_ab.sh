#!/bin/bash
OUT=gpurun_out/${1:-r02t}; mkdir -p $OUT
for R in 1 2 3; do for P in "X=1" "GFICF_JACCARD_NO_PIPE=1"; do
  env $P timeout -k 10 200 python bench.py --no-extras > $OUT/b.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("$OUT/b.json")); r=d["roofline"]
print("$P", "value %.4g  ms/ds %.4f  kernel_ms %.5f b2b %.5f ingest %.5f frac %.4f"%(d["value"], d["ms_per_data_set"], r["kernel_ms"], r["kernel_ms_back_to_back"], r["ingest_kernel_ms"], r["frac"]))
PY
done; done | tee $OUT/ab.txt
