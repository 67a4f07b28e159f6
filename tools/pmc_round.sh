#!/bin/bash
# Counter passes of the round: the bench shapes (100 k x 30 + GF-ICF config 3), config 4 (100 k x 50) and config 5 (1 M x 30)
# -> gpurun_out/<tag>/pmc_traffic.json + per-shape summaries.   Usage (through gpurun): bash tools/pmc_round.sh <tag>
TAG=${1:-r03}; OUT=gpurun_out/$TAG; mkdir -p $OUT
bash tools/pmc.sh $TAG/pmc "k_" PROF_REPS=3 PROF_META=$GRAFT_REPO_ROOT/$OUT/gficf_nnz.txt > $OUT/pmc_summary.txt 2>&1
python3 tools/make_traffic.py $OUT/pmc 100000 30 $OUT/pmc_traffic.json --gficf-nnz $(cat $OUT/gficf_nnz.txt) > /dev/null
bash tools/pmc.sh $TAG/pmc_c4 "k_jaccard|k_ingest" PROF_REPS=3 PROF_N=100000 PROF_K=50 PROF_GFICF=0 > $OUT/pmc_summary_c4.txt 2>&1
python3 tools/make_traffic.py $OUT/pmc_c4 100000 50 $OUT/pmc_traffic.json --merge $OUT/pmc_traffic.json > /dev/null
bash tools/pmc.sh $TAG/pmc_c5 "k_jaccard|k_ingest" PROF_REPS=2 PROF_N=1000000 PROF_K=30 PROF_GFICF=0 > $OUT/pmc_summary_c5.txt 2>&1
python3 tools/make_traffic.py $OUT/pmc_c5 1000000 30 $OUT/pmc_traffic.json --merge $OUT/pmc_traffic.json > /dev/null
python3 -c "
import json; d=json.load(open('$OUT/pmc_traffic.json'))
for k,v in sorted(d.items()):
    if isinstance(v,dict): print('%-42s read %8.1f MB (%s, corr %s)  write %8.1f MB  total %8.1f MB' % (k, v['read_bytes']/1e6, v['read_bytes_from'], v['fetch_correction'], v['WRITE_SIZE_KiB']*1024/1e6, v['hbm_bytes_per_launch']/1e6))
"
find $OUT -name "*.db" -delete; find $OUT -size +3M -delete
