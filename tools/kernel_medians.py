#!/usr/bin/env python3
"""Per-kernel median durations (and the timeline of one late pass) from a rocprofv3 --kernel-trace database.
usage: kernel_medians.py <dir with *_results.db> [substring of the kernel that starts a pass]"""
import collections
import glob
import sqlite3
import statistics
import sys

db = glob.glob(sys.argv[1].rstrip("/") + "/*.db")[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
m = collections.defaultdict(list)
for n, a, b in rows:
    m[n.split("(")[0][:64]].append((b - a) / 1e3)
for n, v in sorted(m.items(), key=lambda x: -sum(x[1]))[:14]:
    print(f"{n:66s} n={len(v):4d} med={statistics.median(v):8.1f} us")
if len(sys.argv) > 2:
    idx = [i for i, r in enumerate(rows) if sys.argv[2] in r[0]]
    if len(idx) >= 3:
        i0 = idx[-3]
        t0 = rows[i0][1]
        for n, a, b in rows[max(0, i0 - 2):i0 + 9]:
            print(f"  {n[9:44]:36s} start {(a - t0) / 1e3:8.1f} dur {(b - a) / 1e3:6.1f}")
