"""Per-kernel means of rocprofv3 --pmc counters: pmc_table.py <counter_collection.csv> [name filter]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("lgh::", "").replace("void ", "")[:40] + " grid " + r.get("Grid_Size", "?")
    if flt and flt not in k:
        continue
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k)
    for c, v in sorted(d.items()):
        print(f"    {c:34s} n={len(v):4d} mean={sum(v)/len(v):14.1f}")
