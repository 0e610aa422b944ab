#!/usr/bin/env python
"""Print the per-kernel table of one rocprofv3 --kernel-trace --stats database (rocpd .db), durations in ms."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
print(f"{'kernel':44s} {'calls':>6s} {'total_ms':>10s} {'avg_ms':>10s} {'pct':>7s}")
for name, calls, tot, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    n = name.split("(")[0].replace("void ", "").replace("afq::", "")
    print(f"{n:44s} {calls:6d} {tot / 1e3:10.3f} {avg / 1e3:10.4f} {pct:7.2f}")
