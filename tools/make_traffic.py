#!/usr/bin/env python3
"""rocprofv3 --pmc csv output -> per-kernel HBM traffic per launch (profiles/pmc_traffic.json).

Correction (MI355X_MICROARCH.md §HBM): on gfx950 FETCH_SIZE = TCC_EA0_RDREQ x 64 B while the read
requests are 128 B wide, so FETCH_SIZE is doubled.  Calibrated on this code's own access patterns
against known byte counts (23 k x 54 k matrix, nnz = 59 809 258): k_gene_count reads exactly
12 B/nnz = 717.7 MB (16 B vector loads) and reports FETCH_SIZE = 351 MiB-units -> x2 = 719 MB;
k_cell_kept_count reads 4 B/nnz = 239 MB (4 B-per-lane loads) and reports 120 -> x2 = 246 MB;
k_ingest_reg reads 12.0 MB (4 B-per-lane loads) and reports 5.96 -> x2 = 12.2 MB.  WRITE_SIZE is
taken as is (k_ingest_reg writes 12.8 MB, reports 12.8).  Both counters are in KiB.  Usage: make_traffic.py <pmc dir> <N> <k> <out.json>"""
import collections
import csv
import glob
import json
import os
import re
import sys

root, N, k, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "p*", "pmc_counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        m = re.search(r"(?<![A-Za-z0-9_])(k_[a-z_0-9]+)", row["Kernel_Name"])
        if m:
            acc[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
def mean(v):
    """Mean over the launches that did real work: a launch whose counter is below 5 % of the kernel's largest is an empty
    one (a variant that returns at once because the other variant handles the input) and would dilute the per-launch figure."""
    top = max(v)
    real = [x for x in v if x >= 0.05 * top] if top > 0 else v
    return sum(real) / len(real)


res = {}
for kern, c in acc.items():
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    f, w = mean(c["FETCH_SIZE"]), mean(c["WRITE_SIZE"])
    fcorr = 2.0
    key = f"{kern[2:]}_N{N}_k{k}" if kern.startswith("k_jaccard") or kern.startswith("k_ingest") else kern[2:]
    res[key] = {"kernel": kern, "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "fetch_correction": fcorr,
                "hbm_bytes_per_launch": int((fcorr * f + w) * 1024)}
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
