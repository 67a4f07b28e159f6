#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output: per kernel, mean counter value per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "?")
            short = name.split("(")[0].split("::")[-1][:60]
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for kname in sorted(acc):
    print(kname)
    for c in sorted(acc[kname]):
        v = acc[kname][c]
        print(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
