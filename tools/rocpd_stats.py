#!/usr/bin/env python3
"""Kernel statistics (the table `rocprofv3 --kernel-trace --stats` prints) from the rocpd SQLite file it writes.
usage: rocpd_stats.py RESULTS.db [OUT.csv]"""
import sqlite3, sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                       "from kernels group by name order by 3 desc"))
total = float(sum(r[2] for r in rows))
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
out.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"\n')
for name, calls, tot, avg, mn, mx in rows:
    out.write(f'"{name}",{calls},{tot},{avg:.3f},{100.0 * tot / total:.2f},{mn},{mx}\n')
