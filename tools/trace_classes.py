#!/usr/bin/env python3
"""Per launch-shape kernel durations from a rocprofv3 --kernel-trace CSV (graph replays: duration includes the dispatch gap).
Usage: python tools/trace_classes.py <kernel_trace.csv> [skip_first_n]"""
import collections
import csv
import statistics as st
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""),
                     r["Grid_Size_X"], r["Workgroup_Size_X"], r["LDS_Block_Size"]))
rows.sort()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
seg = rows[skip:]
d = collections.defaultdict(list)
for s, e, n, g, w, l in seg:
    d[(n[:28], g, w, l)].append(e - s)
tot = sum(sum(v) for v in d.values())
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if len(v) > 20:
        print("%-30s grid %8s wg %5s lds %6s  n %6d  med %7.2f us  share %5.1f%%" % (*k, len(v), st.median(v) / 1e3, 100 * sum(v) / tot))
