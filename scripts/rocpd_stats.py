#!/usr/bin/env python3
"""Dump the per-kernel summary (the `--stats` view) out of a rocprofv3 rocpd sqlite file."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
print(f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'pct':>6}  kernel")
for name, calls, tot, avg, pct in rows:
    print(f"{calls:7d} {tot:12.1f} {avg:10.3f} {pct:6.2f}  {name}")
