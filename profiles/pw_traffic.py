"""HBM traffic of the cell-2 launches of the pointwise LSTM backward from the FETCH_SIZE / WRITE_SIZE
passes of profiles/pmc_train.sh (the per-name summary mixes its cell-1 and cell-2 launches):
    python3 profiles/pw_traffic.py <fetch dir> <write dir> <out.json>"""
import csv
import glob
import json
import os
import sys


def rows(root):
    out = []
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            out += [r for r in csv.DictReader(fh) if "plc_lstm_pointwise_bwd_kernel" in r["Kernel_Name"]]
    big = max(int(r["Grid_Size"]) for r in out)
    return [float(r["Counter_Value"]) for r in out if int(r["Grid_Size"]) == big]


f, w = rows(sys.argv[1]), rows(sys.argv[2])
fetch_kb, write_kb = sum(f) / len(f), sum(w) / len(w)
d = {"kernel": "plc_lstm_pointwise_bwd_kernel(LstmBwdArgs), cell-2 launches (4 channel blocks/pixel), dc carried as bf16",
     "workload": "c3 train step: B=4, 256x256 per launch",
     "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb, "launches_fetch": len(f), "launches_write": len(w),
     "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request for 16 B/lane coalesced reads -> doubled; WRITE_SIZE exact",
     "hbm_bytes_per_launch": (2 * fetch_kb + write_kb) * 1024,
     "algorithmic_bytes_per_launch": 224 * 4 * 4 * 256 * 256,
     "note": "the trace also holds bench.py's own timing launches of the same kernel (two dh inputs, dc present); "
             "the step's last time step has no dc / second dh input and its first no c_prev, so the mean sits slightly below the algorithmic figure",
     "command": "bash profiles/pmc_train.sh fetch write  (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline)"}
json.dump(d, open(sys.argv[3], "w"), indent=1)
print(json.dumps(d, indent=1))
