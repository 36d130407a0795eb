"""HBM bytes per launch of one kernel from a FETCH_SIZE and a WRITE_SIZE pass (rocprofv3 --pmc, csv output):
    python3 profiles/kernel_traffic.py <fetch dir> <write dir> <kernel name substring> <images per launch> \
            <algorithmic bytes per launch> <out.json> <command that was traced>
Takes the launches of the kernel with the largest grid in the trace (the others are other operand sizes), averages
each counter over them and records  2 x FETCH_SIZE + WRITE_SIZE  KiB: on gfx950 FETCH_SIZE counts 64 B per 128-byte
request of 16 B/lane coalesced loads, WRITE_SIZE is exact.  The record is merged into <out.json> under
by_images[<images per launch>], so one file holds the launch sizes of several bench modes."""
import csv
import glob
import json
import os
import sys


def rows(root, needle):
    out = []
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            out += [r for r in csv.DictReader(fh) if needle in r["Kernel_Name"]]
    if not out:
        raise SystemExit(f"no dispatch of {needle!r} under {root}")
    big = max(int(r["Grid_Size"]) for r in out)
    sel = [r for r in out if int(r["Grid_Size"]) == big]
    return [float(r["Counter_Value"]) for r in sel], sel[0]["Kernel_Name"], big


fetch_dir, write_dir, needle, images, alg, out_path, command = sys.argv[1:8]
f, name, grid = rows(fetch_dir, needle)
w, _, _ = rows(write_dir, needle)
fetch_kb, write_kb = sum(f) / len(f), sum(w) / len(w)
rec = {"kernel": name[:120], "grid_size": grid, "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
       "launches_fetch": len(f), "launches_write": len(w),
       "hbm_bytes_per_launch": (2 * fetch_kb + write_kb) * 1024,
       "algorithmic_bytes_per_launch": int(alg), "command": command}
doc = json.load(open(out_path)) if os.path.exists(out_path) else {
    "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request for 16 B/lane coalesced reads -> doubled; WRITE_SIZE exact",
    "by_images": {}}
doc["by_images"][str(int(images))] = rec
json.dump(doc, open(out_path, "w"), indent=1)
print(f"{needle} x{images} images: {rec['hbm_bytes_per_launch'] / 1e6:.1f} MB measured, {int(alg) / 1e6:.1f} MB algorithmic, "
      f"{len(f)} launches")
