#!/bin/bash
# (needs tools/lab/jaccard_cell_order.patch applied: the ordered walk is not in the product, profiles/r03_cell_order.txt)
# config 5 (1 M x 30) and the 8 x 100 k all-gather shape with and without the locality order of the cells
OUT=gpurun_out/${1:-order_ab}; mkdir -p $OUT
run() { # name env...
  local name=$1; shift
  env "$@" python bench.py --config c5 --ids permuted --no-extras --no-cpu-baseline --steps 10 > $OUT/c5.$name.json 2> $OUT/c5.$name.err || { tail -5 $OUT/c5.$name.err; exit 1; }
}
run off GFICF_JACCARD_ORDER=0
run hops1 GFICF_JACCARD_ORDER_HOPS=1
run hops2 GFICF_JACCARD_ORDER_HOPS=2
env GFICF_JACCARD_ORDER_HOPS=1 python bench.py --config c5 --ids spatial --no-extras --no-cpu-baseline --steps 10 > $OUT/c5.spatial_hops1.json 2> $OUT/c5.spatial_hops1.err
env GFICF_JACCARD_ORDER=0 python bench.py --config c5 --ids spatial --no-extras --no-cpu-baseline --steps 10 > $OUT/c5.spatial_off.json 2> $OUT/c5.spatial_off.err
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/c5.*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
    print("%-28s step %.4f ms  value %.2f G edges/s  kernel %-22s frac %.4f  oracle %s" % (f.split("/")[-1], d["ms_per_step"], d["value"]/1e9, r.get("kernel"), r["frac"], d.get("checked_vs_oracle")))
PY
