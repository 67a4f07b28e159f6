#!/bin/bash
# Kernel trace (rocprofv3 --kernel-trace --stats) of one python tool.  Usage (through gpurun): bash tools/trace_py.sh <name> tools/<script>.py [args]
#   -> gpurun_out/<name>_stats.txt (the 24 largest kernels), gpurun_out/<name>.log (the tool's own output)
NAME=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out
SCRIPT=$GRAFT_REPO_ROOT/$1; shift
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tr_$NAME -o t -- python3 $SCRIPT "$@" > $GRAFT_REPO_ROOT/gpurun_out/$NAME.log 2>&1)
python3 - <<PY > gpurun_out/${NAME}_stats.txt
import csv, glob
f = glob.glob("gpurun_out/tr_$NAME/**/t_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:24]:
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
rm -rf gpurun_out/tr_$NAME
grep -v amdgpu gpurun_out/$NAME.log | tail -12; cat gpurun_out/${NAME}_stats.txt
