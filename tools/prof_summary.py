"""Summarise a rocprofv3 --kernel-trace --stats result (.db, rocpd sqlite) as text:
per-kernel calls / total / average duration.  Usage: prof_summary.py <results.db> > profiles/<name>.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                   "from kernels group by %s order by 3 desc" % (name_col, name_col)).fetchall()
tot = sum(r[2] for r in rows) or 1
print("%-72s %7s %14s %14s %14s %14s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"))
for n, c, s, a, mn, mx in rows[:40]:
    print("%-72s %7d %14d %14.0f %14d %14d %6.2f%%" % (n[:72], c, s, a, mn, mx, 100.0 * s / tot))
