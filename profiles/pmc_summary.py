"""Per-kernel averages of a rocprofv3 --pmc --output-format csv run:
    python3 profiles/pmc_summary.py <output dir> [kernel-name substring]
Prints one line per (kernel, counter): dispatches, mean, min, max of the counter value."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
needle = sys.argv[2] if len(sys.argv) > 2 else ""
files = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
if not files:
    sys.exit(f"no *counter_collection.csv under {root}")
vals = collections.defaultdict(list)
for path in files:
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            if needle in name:
                vals[(name, row["Counter_Name"])].append(float(row["Counter_Value"]))
for (name, counter), v in sorted(vals.items()):
    print(f"{counter:28s} n={len(v):4d} mean={sum(v) / len(v):16.1f} min={min(v):16.1f} max={max(v):16.1f}  {name[:70]}")
