#!/bin/bash
# Kernel trace of the adjacency build at one size.  Usage (through gpurun): bash tools/adjacency_trace.sh [N k]  ->  gpurun_out/adj_trace.txt
N=${1:-100000}; K=${2:-50}
export TMPDIR=/tmp
mkdir -p gpurun_out
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/adj_trace -o adj -- python3 $GRAFT_REPO_ROOT/tools/adjacency_time.py $N $K > $GRAFT_REPO_ROOT/gpurun_out/adj_trace.log 2>&1)
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/adj_trace/**/adj_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:24]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
rm -rf gpurun_out/adj_trace
