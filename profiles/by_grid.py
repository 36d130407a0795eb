"""Launches of the library's kernels split by grid size, from the rocpd database of a rocprofv3 --kernel-trace
run:  python profiles/by_grid.py <results.db> <out.csv>
The per-name summary (export_stats.py) averages launches of one kernel over different operand sizes: the paired
recurrence launches of a train step (a half or the whole batch) with the stand-alone launches bench.py makes to
time its roofline kernels, cell-1 with cell-2 launches of the single pointwise pass, first-step with later-step
launches.  One row per (kernel, grid, workgroup size) keeps them apart."""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute(
    "select name, grid_x, workgroup_x, count(*), avg(duration) / 1000.0, min(duration) / 1000.0, max(duration) / 1000.0 "
    "from kernels where name like '%plc_%' group by name, grid_x, workgroup_x order by name, grid_x").fetchall()
with open(sys.argv[2], "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "grid_size_x", "workgroup_size_x", "launches", "avg_us", "min_us", "max_us"])
    for name, grid, wg, n, avg, lo, hi in rows:
        w.writerow([name, grid, wg, n, round(avg, 3), round(lo, 3), round(hi, 3)])
print(f"{len(rows)} (kernel, grid) rows")
