"""Summary of a rocprofv3 --hip-trace --kernel-trace --memory-copy-trace run of the in-library pipeline: per token, the GPU
timeline (kernels and copies in start order with the idle gaps between them) and the host API calls' durations."""
import csv
import glob
import os
import sys

d = sys.argv[1]
def load(pat):
    fs = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return list(csv.DictReader(open(fs[0]))) if fs else []
k = load("*kernel_trace.csv")
m = load("*memory_copy_trace.csv")
h = load("*hip_api_trace.csv")
ev = []
for r in k:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0].replace("lgh::", "").replace("void ", "")[:40], r.get("Queue_Id", "?")))
for r in m:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", r.get("Name", "copy")), "-"))
ev.sort()
# the last tokens: find embed kernels
idx = [i for i, e in enumerate(ev) if "embed_kernel" in e[2]]
print(f"{len(k)} kernel records, {len(m)} copy records, {len(h)} API records; {len(idx)} tokens")
if len(idx) >= 4:
    a, b = idx[-3], idx[-2]
    t0 = ev[a][0]
    prev = t0
    print(f"\n## one token on the GPU ({(ev[b][0] - t0) / 1e3:.1f} us embed to embed): events with a gap > 1.5 us before them, and every copy")
    busy = 0
    for s, e, name, q in ev[a:b]:
        gap = (s - prev) / 1e3
        busy += (e - s)
        if gap > 1.5 or name.startswith("C "):
            print(f"  +{(s - t0) / 1e3:9.2f} us  gap {gap:7.2f}  dur {(e - s) / 1e3:7.2f}  q{q}  {name}")
        prev = max(prev, e)
    print(f"  busy {busy / 1e3:.1f} us of {(ev[b][0] - t0) / 1e3:.1f}")
# API durations per call name during the last 8 tokens
if h:
    tend = max(int(r["End_Timestamp"]) for r in h)
    tbeg = ev[idx[-9]][0] if len(idx) >= 10 else 0
    tot = {}
    for r in h:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s < tbeg:
            continue
        n = r.get("Function", r.get("Name", "?"))
        t = tot.setdefault(n, [0, 0])
        t[0] += 1
        t[1] += e - s
    ntok = 8
    print(f"\n## host API calls over the last {ntok} tokens (count per token, mean us)")
    for n, (c, ns) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print(f"  {n:40s} {c / ntok:6.1f} per token   {ns / c / 1e3:8.2f} us each   {ns / ntok / 1e3:8.1f} us per token")
