"""Export the per-kernel summary of a rocprofv3 --kernel-trace --stats run (its rocpd SQLite
database) as CSV:  python profiles/export_stats.py <results.db> <out.csv>
The view's durations are microseconds."""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.execute("select * from top_kernels")
cols = [c[0] for c in cur.description]
rows = cur.fetchall()
with open(sys.argv[2], "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(cols)
    w.writerows(rows)
total = sum(r[2] for r in rows)
lib = sum(r[2] for r in rows if "plc_" in r[0])
print(f"{len(rows)} kernels, total {total / 1e3:.2f} ms, libplconvlstm {lib / 1e3:.2f} ms")
print(" calls   total ms   avg us  kernel")
for r in rows[:40]:
    print(f"{r[1]:6d} {r[2] / 1e3:10.2f} {r[3]:8.2f}  {r[0][:120]}")
