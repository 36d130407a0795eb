"""Timeline of the last train step in a rocprofv3 --kernel-trace database (rocpd SQLite):
    python3 devtools/step_timeline.py <results.db> [kernel-name substring that ends a step, default plc_clip_adam_kernel]
One line per dispatch: start and end (us from the step's first dispatch), duration, queue, kernel name."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
end_name = sys.argv[2] if len(sys.argv) > 2 else "plc_clip_adam_kernel"
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("kernels")] or [t for t in tabs if "kernel_dispatch" in t]
cur = db.execute(f"select * from {kd[0]} limit 1")
cols = [c[0] for c in cur.description]
print(kd[0], cols, file=sys.stderr)
rows = db.execute(f"select start, end, queue_id, stream_id, name from {kd[0]} order by start").fetchall()
ends = [i for i, r in enumerate(rows) if end_name in r[4]]
step = rows[ends[-2] + 1: ends[-1] + 1]
t0 = step[0][0]
for s, e, q, st, n in step:
    print(f"{(s - t0) / 1e3:10.1f} {(e - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f}  q{q} s{st}  {n[:70]}")
