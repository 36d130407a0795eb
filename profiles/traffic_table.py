"""Join the FETCH_SIZE / WRITE_SIZE passes of profiles/pmc_train.sh with the kernel stats of
profiles/prof_train.sh:  python3 profiles/traffic_table.py <fetch dir> <write dir> <stats.csv> <out.csv>
One row per kernel name: launches averaged as in the stats file; HBM bytes = 2 x FETCH_SIZE +
WRITE_SIZE (KiB; gfx950 counts 64 B per 128-byte read request of 16 B/lane coalesced loads)."""
import collections
import csv
import glob
import os
import sys


def means(root):
    acc = collections.defaultdict(list)
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch, write = means(sys.argv[1]), means(sys.argv[2])
rows = []
with open(sys.argv[3], newline="") as fh:
    for r in csv.DictReader(fh):
        name = r["name"]
        if name in fetch and name in write and "plc_" in name:
            mb = (2 * fetch[name] + write[name]) * 1024 / 1e6
            us = float(r["average"])
            rows.append((name, int(r["total_calls"]), us, round(mb, 1), round(mb / us, 2)))
with open(sys.argv[4], "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "launches_in_trace", "avg_us", "hbm_MB_per_launch", "TB_per_s"])
    w.writerows(rows)
for r in rows:
    print(f"{r[2]:9.2f} us {r[3]:9.1f} MB {r[4]:6.2f} TB/s  {r[0][:90]}")
