#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 (--kernel-trace --stats) rocpd SQLite result, the same table
`--stats` prints: python profiles/summarize_rocpd.py gpurun_out/prof/r1_results.db > profiles/rNN_*.txt"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, "
                      "max(end-start)/1e3 from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print(f"# {sys.argv[1]}: {sum(r[1] for r in rows)} dispatches, {tot:.2f} ms of kernel time")
print(f"{'kernel':<100} {'calls':>6} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>10} {'pct':>6}")
for r in rows:
    print(f"{r[0][:100]:<100} {r[1]:>6} {r[2]:>10.2f} {r[3]:>10.1f} {r[4]:>9.1f} {r[5]:>10.1f} {100*r[2]/tot:>6.1f}")
