"""HBM bytes of ONE whole train step from the FETCH_SIZE / WRITE_SIZE passes of profiles/pmc_train.sh:
    python3 profiles/step_traffic.py <fetch dir> <write dir> <out.json> [per-GPU batch of the traced run, default 4]
The passes trace every dispatch in order; a train step ends with the optimizer's plc_clip_adam_kernel, so the
dispatches between two of them are one step (forward, loss, backward, clip, Adam; the library's kernels and
the few torch kernels between them).  The LAST complete step of the trace is taken.  Bytes = 2 x FETCH_SIZE +
WRITE_SIZE in KiB (gfx950 counts 64 B per 128-byte read request of 16 B/lane coalesced loads)."""
import collections
import csv
import glob
import json
import os
import sys


def dispatches(root, counter):
    rows = []
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == counter:
                    rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    return rows


def last_step(rows):
    ends = [i for i, r in enumerate(rows) if r[1].startswith("plc_clip_adam_kernel")]
    if len(ends) < 2:
        raise SystemExit("fewer than two train steps in the trace")
    return rows[ends[-2] + 1: ends[-1] + 1]


fetch = last_step(dispatches(sys.argv[1], "FETCH_SIZE"))
write = last_step(dispatches(sys.argv[2], "WRITE_SIZE"))
if [r[1] for r in fetch] != [r[1] for r in write]:
    raise SystemExit("the two passes traced different kernel sequences")
by_kernel = collections.OrderedDict()
total = 0.0
for (_, name, f), (_, _, w) in zip(fetch, write):
    b = (2 * f + w) * 1024
    total += b
    key = name.split("(")[0][:80]
    e = by_kernel.setdefault(key, {"launches": 0, "bytes": 0.0})
    e["launches"] += 1
    e["bytes"] += b
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 4
out = {"what": f"HBM bytes of one c3 train step (B={batch}, T=10, 256x256), PMC: 2 x FETCH_SIZE + WRITE_SIZE over every dispatch of the step",
       "dispatches": len(fetch), "hbm_bytes_per_step": total,
       "by_kernel": {k: {"launches": v["launches"], "MB": round(v["bytes"] / 1e6, 1)}
                     for k, v in sorted(by_kernel.items(), key=lambda kv: -kv[1]["bytes"])}}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(f"{len(fetch)} dispatches, {total / 1e9:.3f} GB per train step")
for k, v in list(out["by_kernel"].items())[:12]:
    print(f"{v['MB']:9.1f} MB  x{v['launches']:3d}  {k}")
