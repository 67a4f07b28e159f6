mkdir -p gpurun_out/knn4
for S in 1 2 3 4; do echo "SPLIT=$S"; GFICF_KNN_SPLIT=$S timeout -k 10 120 python tools/knn_bench.py 100000 50 31 manhattan 100000 50 1 manhattan 100000 2 31 manhattan 100000 2 1 manhattan; done > gpurun_out/knn4/split.txt 2>&1
cat gpurun_out/knn4/split.txt
