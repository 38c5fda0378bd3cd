"""Per-kernel calls and average duration of a rocprofv3 --stats kernel_stats.csv: kstats.py <file> [rows]"""
import csv, sys
for i, r in enumerate(csv.DictReader(open(sys.argv[1]))):
    if i >= (int(sys.argv[2]) if len(sys.argv) > 2 else 16):
        break
    n = r["Name"].split("(")[0].replace("void ", "").replace("lgh::", "")[:44]
    print("%-46s calls %6s  avg %8.2f us  %6.2f %%" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
