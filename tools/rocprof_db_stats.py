#!/usr/bin/env python3
"""Per-kernel totals out of a rocprofv3 (rocpd) results database: tools/rocprof_db_stats.py <results.db> [top_n]."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 16
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
q = (f"select s.kernel_name, count(*), sum(d.end-d.start)/1e3, avg(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id "
     "group by s.kernel_name order by 3 desc")
rows = list(db.execute(q))
tot = sum(r[2] for r in rows)
print(f"total kernel time {tot / 1e3:.2f} ms over {sum(r[1] for r in rows)} launches")
for r in rows[:top]:
    print(f"{r[0][:100]:100s} n={r[1]:6d} total={r[2] / 1e3:9.2f} ms avg={r[3]:8.2f} us {100 * r[2] / tot:5.1f}%")
