"""Per-kernel durations and the idle time in front of each kernel from a rocprofv3 kernel trace CSV.
Usage: python tools/lab/trace_gaps.py <kernel_trace.csv> [last_n_rows]"""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
if len(sys.argv) > 2: rows = rows[-int(sys.argv[2]):]
short = lambda n: re.sub(r"\(.*", "", re.sub(r"^void |\(anonymous namespace\)::", "", n))[:48]
agg = collections.OrderedDict()
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = short(r["Kernel_Name"])
    a = agg.setdefault(k, [0, 0.0, 0.0])
    a[0] += 1; a[1] += (e - s) / 1e3
    if prev_end is not None: a[2] += max(0, s - prev_end) / 1e3
    prev_end = e
for k, (n, d, g) in agg.items():
    print("%-50s n=%4d  dur %8.2f us  idle in front %7.2f us" % (k, n, d / n, g / n))
