#!/bin/bash
# edge kernel on scrambled and on ordered ids, three shapes, with an environment switch ($1, e.g. GFICF_JACCARD_XCD) set to 0 and to 1
OUT=gpurun_out/${2:-ids_ab}; mkdir -p $OUT
for sw in 0 1; do for c in north_star c4 c5; do for ids in permuted spatial; do
  export $1=$sw
  python bench.py --config $c --ids $ids --no-extras --no-cpu-baseline --steps 10 > $OUT/$c.$ids.$sw.json 2> $OUT/$c.$ids.$sw.err || exit 1
done; done; done
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
    print("%-40s step %.4f ms  kernel %-24s frac %.4f  oracle %s" % (f.split("/")[-1], d["ms_per_step"], r.get("kernel"), r["frac"], d.get("checked_vs_oracle")))
PY
