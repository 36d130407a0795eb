"""Split the launches of bench.py's two roofline kernels by grid size, from the rocpd database of a
rocprofv3 --kernel-trace run:  python profiles/by_grid.py <results.db> <out.csv>
The per-name summary averages the pointwise backward over its cell-1 and cell-2 launches; the
bench line's roofline object times the cell-2 launch only."""
import csv
import sqlite3
import sys

WANTED = ("plc_lstm_pointwise_bwd_kernel", "plc_conv_mfma_kernel<2, 4, 4, 3, 0, 0, 16, 1>")
NOTES = {
    ("plc_lstm_pointwise_bwd_kernel", 2): "cell 1 (2 channel blocks/pixel)",
    ("plc_lstm_pointwise_bwd_kernel", 4): "cell 2 (4 channel blocks/pixel: the launch bench.py times for roofline)",
}

db = sqlite3.connect(sys.argv[1])
rows = db.execute(
    "select name, grid_x, workgroup_x, count(*), avg(duration) / 1000.0 from kernels "
    "group by name, grid_x, workgroup_x order by name, grid_x").fetchall()
picked = [r for r in rows if any(w in r[0] for w in WANTED)]
smallest = {}
for name, grid, wg, n, avg in picked:
    smallest[name] = min(smallest.get(name, grid), grid)
with open(sys.argv[2], "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "grid_size_x", "workgroup_size_x", "launches", "avg_us", "note"])
    for name, grid, wg, n, avg in picked:
        note = ""
        if "pointwise" in name:
            note = NOTES[("plc_lstm_pointwise_bwd_kernel", 2 if grid == smallest[name] else 4)]
        w.writerow([name, grid, wg, n, round(avg, 3), note])
        print(name[:70], grid, wg, n, round(avg, 3), note)
