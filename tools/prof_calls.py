"""List the last N kernel calls of a rocprofv3 --kernel-trace result (.db) in launch order with their durations.
Usage: prof_calls.py <results.db> [N]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = cur.execute("select %s, start, end from kernels order by start" % name_col).fetchall()[-n:]
t0 = rows[0][1]
for name, s, e in rows:
    print("%9.1f us  +%8.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, name[:90]))
print("span %.1f us, busy %.1f us" % ((rows[-1][2] - t0) / 1e3, sum(e - s for _, s, e in rows) / 1e3))
