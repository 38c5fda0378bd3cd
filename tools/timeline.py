"""Per-kernel timeline of one decode step from a rocprofv3 --kernel-trace CSV: start, duration, queue, and how much of each
launch overlaps the previous one (flag-ordered two-stream graphs).  usage: timeline.py <kernel_trace.csv> [token_index_from_end]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
anchor = sys.argv[3] if len(sys.argv) > 3 else "embed_kernel"     # the kernel a step starts with ("embed_batch_kernel": a prompt pass)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
a = starts[-back - 1]
b = starts[-back] if back > 0 else len(rows) - 1
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
tot = {}
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].split("(")[0].replace("lgh::", "").replace("void ", "")[:44]
    print(f"{s/1e3:9.2f} us  +{(e-s)/1e3:7.2f}  gap {(s-prev_end)/1e3:7.2f}  q{r.get('Queue_Id','?'):>3}  {name}")
    prev_end = max(prev_end, e)
    k = name
    tot.setdefault(k, [0, 0.0])
    tot[k][0] += 1
    tot[k][1] += (e - s) / 1e3
print(f"token span {(int(rows[b]['Start_Timestamp'])-t0)/1e3:.1f} us")
for k, (n, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:44s} x{n:4d}  {t:9.1f} us  avg {t/n:7.2f}")
