#!/bin/bash
# PMC passes + kernel trace of the Jaccard kernels (prof_driver: ingest + edges at 100 k x 30), then the rest of the GPU tests
TAG=${1:-r02f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
PROF_GFICF=0 PROF_REPS=5 bash tools/pmc.sh $TAG/pmc "k_ingest|k_jaccard" PROF_GFICF=0 PROF_REPS=5 > $OUT/pmc_summary.txt 2>&1; tail -12 $OUT/pmc_summary.txt | cut -c1-2500
timeout -k 10 900 python -m pytest tests/test_gficf_gpu.py tests/test_dist_gpu.py tests/test_knn_gpu.py tests/test_louvain_gpu.py -x -q -m gpu > $OUT/pytest_rest.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest_rest.log
